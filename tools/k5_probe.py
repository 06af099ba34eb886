#!/usr/bin/env python
"""K5 probe (tools only): correctness of bflow_corr_build_split against an fp64 GEMM on the GPU for a list of shapes (shared and
per-target references, ragged N, every supported D) and its duration at the BASELINE sizes.
    python tools/k5_probe.py [--reps 30] [--big]        (BFLOW_CORR_TILE_KERNEL=1 selects the 256x128 tile kernel for A/B)
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bflow_amd import hip  # noqa: E402


def check(B, D, N, T, shared, dev, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    f1 = torch.randn(((1 if shared else T) * B, D, N), generator=g).to(dev)
    f2 = torch.randn((T * B, D, N), generator=g).to(dev)
    f2[0, :, 0] *= 3e-3
    f2[0, :, min(1, N - 1)] *= 300.0
    p1, p2 = hip.split_pack(f1), hip.split_pack(f2)
    out = torch.full((T, B, N, N), float("nan"), device=dev)
    hip.corr_build_split(p1, p2, out, T, B, N, shared_f1=shared)
    a = f1.double().view(-1, B, D, N)
    a = a.expand(T, B, D, N) if shared else a
    b = f2.double().view(T, B, D, N)
    ref = a.transpose(2, 3) @ b / np.sqrt(D)
    mag = a.abs().transpose(2, 3) @ b.abs() / np.sqrt(D)
    err = float(((out.double() - ref).abs() / mag).max())
    ok = bool(torch.isfinite(out).all()) and err < 1.2e-6
    print(f"  B={B} D={D} N={N} T={T} shared={shared}: max err / sum|a||b| = {err:.2e}  {'ok' if ok else 'FAIL'}", flush=True)
    return ok


def timeit(fn, reps):
    for _ in range(3):
        fn()
    evs = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    ts = [a.elapsed_time(b) for a, b in evs]
    return float(np.median(ts)), float(np.min(ts))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--time-only", action="store_true", help="skip the correctness checks (ablation builds)")
    ap.add_argument("--stamps", action="store_true", help="needs the STREAM_STAMPS build (tools/k5_ablate.sh stamps:-DSTREAM_STAMPS)")
    ap.add_argument("--f16", action="store_true", help="also time the fp16 tiled volume (bflow_corr_build_f16_tiled)")
    ap.add_argument("--big", action="store_true", help="also time C3 (B=8, T=5) and C5 (N=16384, T=6)")
    ap.add_argument("--stamp-mode", default="split", help="arithmetic of the stamped launch: split | split8 | f16/w (tiled volume)")
    ap.add_argument("--stamp-shape", default="C2", help="C2 | C5")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    print("kernel:", "tile (256x128)" if os.environ.get("BFLOW_CORR_TILE_KERNEL") else "stream", flush=True)
    ok = True
    for (B, D, N, T, shared) in [] if args.time_only else [(1, 256, 4800, 4, True), (1, 256, 4800, 2, False), (2, 64, 99, 3, True), (2, 64, 99, 3, False),
                                 (1, 128, 1320, 1, True), (3, 256, 300, 5, True), (1, 256, 31, 1, True), (2, 256, 1000, 2, False),
                                 (1, 256, 2304, 5, True), (1, 96, 500, 2, True)]:
        ok &= check(B, D, N, T, shared, dev)
    # repeatability under load: the same launch 20 times must give bit-identical volumes (a vmcnt / barrier race would not)
    g = torch.Generator(device="cpu").manual_seed(5)
    B, D, N, T = 1, 256, 4800, 4
    f1 = torch.randn((B, D, N), generator=g).to(dev)
    f2 = torch.randn((T * B, D, N), generator=g).to(dev)
    p1, p2 = hip.split_pack(f1), hip.split_pack(f2)
    vol = torch.empty((T, B, N, N), device=dev)
    hip.corr_build_split(p1, p2, vol, T, B, N, shared_f1=True)
    first = vol.clone()
    same = True
    for _ in range(0 if args.time_only else 20):
        vol.fill_(float("nan"))
        hip.corr_build_split(p1, p2, vol, T, B, N, shared_f1=True)
        same &= bool(torch.equal(vol, first))
    print("  20 repeated launches bit-identical:", same, flush=True)
    ok &= same
    ms, mn = timeit(lambda: hip.corr_build_split(p1, p2, vol, T, B, N, shared_f1=True), args.reps)
    by = 4.0 * ((1 + T) * B * D * N + T * B * N * N)
    print(f"C2 (T=4 B=1 N=4800): median {ms*1e3:.1f} us (min {mn*1e3:.1f})  {by/ms/1e6:.0f} GB/s algorithmic = {by/ms/1e6/8000:.3f} of 8 TB/s", flush=True)
    if args.f16:
        for name, B, T, hh, ww, shared in (("C2", 1, 4, 60, 80, True), ("C4 shard", 8, 4, 60, 80, True), ("C5", 1, 6, 128, 128, False)):
            N = hh * ww
            f1 = torch.randn(((1 if shared else T) * B, D, N), generator=g).to(dev)
            f2 = torch.randn((T * B, D, N), generator=g).to(dev)
            p1, p2 = hip.split_pack(f1), hip.split_pack(f2)
            for dt, label in ((torch.float32, "split -> fp32 tiled"), (torch.float16, "fp16  -> fp16 tiled")):
                vol = torch.empty((T, B, N, hip.tiled_plane_size(hh, ww)), dtype=dt, device=dev)
                ms, mn = timeit(lambda: hip.corr_build_split(p1, p2, vol, T, B, N, shared_f1=shared, tiled_hw=(hh, ww)), max(args.reps // 3, 5))
                eb = vol.element_size()
                by = 2.0 * (eb // 2) * (((1 if shared else T) + T) * B * D * N) + eb * T * B * N * N   # fp16: hi planes only
                print(f"{name} {label}: median {ms*1e3:.1f} us (min {mn*1e3:.1f})  {by/1e6:.0f} MB algorithmic -> {by/ms/1e6:.0f} GB/s = "
                      f"{by/ms/1e6/8000:.3f} of 8 TB/s", flush=True)
                del vol
            del f1, f2, p1, p2
    if args.big:
        for name, B, T, N, shared in (("C4 shard (B=8 T=4)", 8, 4, 4800, True), ("C3 (B=8 T=4+1 M-to-N)", 8, 5, 4800, False),
                                      ("C1 (N=2304 T=5)", 1, 5, 2304, True), ("C5 (N=16384 T=6 M-to-N)", 1, 6, 16384, False)):
            f1 = torch.randn(((1 if shared else T) * B, D, N), generator=g).to(dev)
            f2 = torch.randn((T * B, D, N), generator=g).to(dev)
            p1, p2 = hip.split_pack(f1), hip.split_pack(f2)
            vol = torch.empty((T, B, N, N), device=dev)
            ms, mn = timeit(lambda: hip.corr_build_split(p1, p2, vol, T, B, N, shared_f1=shared), max(args.reps // 3, 5))
            by = 4.0 * (((1 if shared else T) + T) * B * D * N + T * B * N * N)
            fl = 2.0 * T * B * D * N * N
            print(f"{name}: median {ms*1e3:.1f} us  {by/ms/1e6:.0f} GB/s = {by/ms/1e6/8000:.3f} of 8 TB/s, {fl/ms/1e9:.0f} TFLOP/s-equiv", flush=True)
            del vol, f1, f2, p1, p2
    if args.stamps:
        import ctypes
        st = torch.zeros((256, 64), dtype=torch.int64, device=dev)
        hip.lib().bflow_k5_set_stamp_buffer(ctypes.c_void_p(st.data_ptr()))
        B, D, hh, ww, T, shared = (1, 256, 60, 80, 4, True) if args.stamp_shape == "C2" else (1, 256, 128, 128, 6, False)
        N = hh * ww
        f1 = torch.randn(((1 if shared else T) * B, D, N), generator=g).to(dev)
        f2 = torch.randn((T * B, D, N), generator=g).to(dev)
        p1, p2 = hip.split_pack(f1), hip.split_pack(f2)
        ar = {"split": hip.ARITH_SPLIT, "split8": hip.ARITH_SPLIT8, "f16/w": hip.ARITH_F16}[args.stamp_mode]
        vol = torch.empty((T, B, N, hip.tiled_plane_size(hh, ww)), device=dev)
        x8 = (hip.split_to_x8(p1), hip.split_to_x8(p2))
        for _ in range(3):
            hip.corr_build_tiled(p1, p2, vol, T, B, N, shared_f1=shared, tiled_hw=(hh, ww), arithmetic=ar, x8=x8)
        torch.cuda.synchronize()
        print(f"stamped launch: {args.stamp_shape} {args.stamp_mode} (tiled volume)")
        s = st.cpu().numpy()
        print("cycle stamps = s_memtime (shader clock); real-time stamps = s_memrealtime (100 MHz)")
        for wg in (0, 1, 8, 9, 100, 255):
            r = s[wg]
            its = [int(x) for x in r[:61] if x]
            d = np.diff(its)
            print(f"wg {wg}: prologue {d[0]} steps(2 chunks each) {[int(x) for x in d[1:]]} total cycles {int(r[61] - r[0])}")
        rt0 = s[:, 62].min()
        st, en = (s[:, 62] - rt0) / 100.0, (s[:, 63] - rt0) / 100.0       # us
        cyc = (s[:, 61] - s[:, 0]).astype(np.float64)
        live = s[:, 61] > 0
        print(f"workgroups that ran: {int(live.sum())}; start us min/max {st[live].min():.2f}/{st[live].max():.2f}; "
              f"end us min/median/max {en[live].min():.1f}/{np.median(en[live]):.1f}/{en[live].max():.1f}")
        print(f"per-workgroup cycles min/median/max {cyc[live].min():.0f}/{np.median(cyc[live]):.0f}/{cyc[live].max():.0f}; "
              f"clock = cycles / real time: {np.median(cyc[live] / (en[live] - st[live])):.0f} MHz")
        for x in range(8):
            m = live & (np.arange(256) % 8 == x)
            print(f"  xcd {x}: end us median {np.median(en[m]):.1f} max {en[m].max():.1f}, cycles median {np.median(cyc[m]):.0f}")
        # ---- cycle budget of the launch (round 6): every workgroup's stamps are prologue | step | step | ... | end
        floor = {"split8": 2 * 1024 * 2, "split": 2 * 1536 * 2, "f16/w": 2 * 512 * 2}[args.stamp_mode]     # matrix-pipe cycles per step and SIMD: 2 chunks x 2 waves per SIMD
        pro, stp, nst, tail = [], [], [], []
        for wg in range(256):
            r = s[wg]
            its = [int(x) for x in r[:61] if x]
            if len(its) < 3 or not live[wg]:
                continue
            d = np.diff(its)
            pro.append(d[0]); stp.extend(d[1:].tolist()); nst.append(len(d) - 1); tail.append(int(r[61]) - its[-1])
        pro, stp, nst, tail = np.array(pro), np.array(stp), np.array(nst), np.array(tail)
        ghz = float(np.median(cyc[live] / (en[live] - st[live]))) / 1e3
        smed = float(np.median(stp))
        slow = float((stp > 1.2 * smed).mean())
        print(f"BUDGET {args.stamp_shape} {args.stamp_mode}: clock {ghz:.3f} GHz; prologue median {np.median(pro):.0f} cycles ({np.median(pro)/ghz/1e3:.1f} us); "
              f"step (2 chunks) median {smed:.0f} cycles = {floor/smed:.2f} of them matrix-pipe busy (floor {floor}); {100*slow:.1f} % of the steps take > 1.2 x the median "
              f"(panel hand-over: mean extra {float((stp[stp > 1.2*smed] - smed).mean()) if slow > 0 else 0:.0f} cycles); steps per workgroup min/median/max "
              f"{nst.min()}/{int(np.median(nst))}/{nst.max()}; tail (last step stamp -> end: last stores) median {np.median(tail):.0f} cycles")
        worst = int(np.argmax(np.where(live, en, 0)))
        r = s[worst]; its = [int(x) for x in r[:61] if x]; d = np.diff(its)
        parts = {"start skew": st[worst] * ghz * 1e3, "prologue": float(d[0]), "steps x median": float((len(d) - 1) * smed),
                 "steps above the median (hand-overs)": float(d[1:].sum() - (len(d) - 1) * smed), "tail": float(int(r[61]) - its[-1])}
        tot = sum(parts.values())
        print(f"BUDGET critical workgroup {worst} (ends last, {en[worst]:.1f} us): " + "; ".join(f"{k} {v:.0f} cyc = {v/ghz/1e3:.1f} us" for k, v in parts.items()) +
              f"; sum {tot/ghz/1e3:.1f} us of {en[worst]:.1f} us")
        mean_steps = float(nst.mean())
        print(f"BUDGET balance: mean steps per workgroup {mean_steps:.2f} vs max {nst.max()}: a perfectly balanced panel assignment would end at "
              f"{(np.median(pro) + mean_steps * smed + np.median(tail))/ghz/1e3:.1f} us (stamps of wave 0 only; the launch itself adds dispatch + the write-back at its end)")
    print("PROBE", "OK" if ok else "FAILED")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
