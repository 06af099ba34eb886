#!/usr/bin/env python
"""bench.py -- headline benchmark of the RAFT-spline inference hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W            (N == 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W               (N > 1, one rank per GPU, RCCL)

Metric (BASELINE.json): frames/s of the whole job + ms per GRU iteration, raft-spline E_LU4_BD2 (events only),
DSEC 640x480, 12 iterations -- BASELINE configs[1] at N=1 (batch 1 per GPU); for N>1 every rank runs the same
per-GPU batch on its own shard of the global batch (weak scaling, no data-path collective) and the per-rank
EPE state is all-gathered once over RCCL after the timed region.

One "step" = one forward (voxel grid resident in HBM -> full-resolution Bezier flow) over the per-GPU batch, replayed
from a captured hipGraph.  Prints ONE JSON line on rank 0 (see DESIGN.md section "Measurement" for every field).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import bflow_amd  # noqa: E402
from bflow_amd import configs, dist as bdist, hip, synthetic  # noqa: E402
from bflow_amd.metrics import epe_masked  # noqa: E402
from bflow_amd.weights import deterministic_state_dict  # noqa: E402

H, W, ITERS, CFG = 480, 640, 12, "E_LU4_BD2"
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense fp32 matrix peak (v_mfma_f32_32x32x2_f32)
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)


def time_steps(fn, steps, barrier):
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    barrier()
    return time.perf_counter() - t0


def kernel_event_ms(launch, n):
    """Average duration (ms) of one launch, hipEvents recorded on the launch stream (torch's current stream)."""
    evs = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        launch()
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    return float(np.mean([a.elapsed_time(b) for a, b in evs]))


def usable_cores() -> int:
    """Cores this process may actually use: min(affinity mask, cgroup CPU quota).  The GPU box exposes 256 logical CPUs
    but caps the container at a 16-CPU quota; oversubscribing the quota makes the baseline ~100x slower."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except Exception:
            pass
    return n


def cpu_baseline(cfg, sd, vox_cpu, budget_s=25.0):
    """The CPU oracle (op-for-op restatement of the reference, pinned to it in the build container) on this host's cores.
    Bounded sample: 1 warm-up + up to 4 timed forwards of the same workload (one 640x480 frame, 12 iterations)."""
    from oracle import raft_spline_oracle as O   # checker / baseline only -- never on the product path
    cores = usable_cores()
    torch.set_num_threads(cores)
    with torch.inference_mode():
        O.forward(sd, cfg, vox_cpu, None, iters=ITERS, test_mode=True)
        times = []
        t_start = time.perf_counter()
        while len(times) < 4 and (time.perf_counter() - t_start) < budget_s:
            t0 = time.perf_counter()
            O.forward(sd, cfg, vox_cpu, None, iters=ITERS, test_mode=True)
            times.append(time.perf_counter() - t0)
    sec = float(np.mean(times))
    return {"value": round(1.0 / sec, 4), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{len(times)} forwards of 1 frame (640x480, {ITERS} iters) after 1 warm-up, torch CPU fp32, {cores} threads",
            "ms_per_frame": round(sec * 1e3, 1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=1, help="frames per GPU per step (BASELINE configs[1]: 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay (debug)")
    args = ap.parse_args()

    rank, world, local = bdist.init_from_env("nccl")
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    torch.backends.cudnn.benchmark = True   # MIOpen: pick the fastest algorithm per conv shape during warm-up

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    cfg = configs.model_config(CFG)
    model = bflow_amd.RAFTSpline(cfg).eval()
    sd = deterministic_state_dict(model, seed=0)
    model.load_state_dict(sd)
    model.to(dev)
    if not args.no_graph:
        model.enable_hipgraph()

    B = args.batch
    first = rank * B
    vox_np = synthetic.voxel_grid(B, 9, H, W, seed=1234, first_sample=first)
    vox = torch.from_numpy(vox_np).to(dev)

    def step():
        return model(voxel_grid=vox, iters=ITERS, test_mode=True)

    for _ in range(max(args.warmup, 1)):
        step()
    elapsed = time_steps(step, args.steps, barrier)
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    elapsed = float(t.item())
    frames = world * B * args.steps
    value = frames / elapsed

    # ---- the path's single exchange step: per-rank EPE state all-gathered over RCCL (outside the timed region)
    low, up = step()
    gt = torch.from_numpy(synthetic.gt_flow(B, H, W, seed=99, first_sample=first)).to(dev)
    e = epe_masked(up.get_flow_from_reference(1.0).contiguous(), gt)
    epe_mean, epe_sum, epe_cnt = bdist.reduce_epe(e.double(), torch.ones((), dtype=torch.float64, device=dev))

    out = None
    if rank == 0:
        out = {
            "metric": "frames/sec (whole node), raft-spline DSEC 640x480 12-iter",
            "value": round(value, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"raft-spline {CFG} events-only, DSEC-shaped voxel grid (9x{H}x{W}), batch {B}/GPU, "
                                   f"{ITERS} GRU iters (BASELINE configs[1]), random-init deterministic weights",
                       "batch_per_gpu": B, "global_batch": world * B, "iters": ITERS, "hipgraph": not args.no_graph},
            "epe_vs_synthetic_gt": round(float(epe_mean), 4), "epe_ranks_gathered": int(epe_cnt),
        }

    # ---- single-GPU extras: ms/GRU-iter, roofline of the dominant hand-written kernel, CPU baseline
    if world == 1:
        def step6():
            return model(voxel_grid=vox, iters=ITERS // 2, test_mode=True)
        for _ in range(3):
            step6()
        n_it = max(args.steps // 2, 5)
        t12 = time_steps(step, n_it, barrier) / n_it
        t6 = time_steps(step6, n_it, barrier) / n_it
        out["ms_per_gru_iter"] = round((t12 - t6) / (ITERS // 2) * 1e3, 4)
        out["ms_fixed_part"] = round((t12 - ITERS * (t12 - t6) / (ITERS // 2)) * 1e3, 4)

        # dominant hand-written kernel: K5 correlation build (fp32 MFMA), on the operands of this very workload
        with torch.no_grad():
            grids, _ = model.gen_voxel_grids(vox)
            fm = model.fnet_ev(torch.cat(grids, dim=0)).float()
        T, D, N = 4, fm.shape[1], fm.shape[2] * fm.shape[3]
        f1 = fm[:B].reshape(B, D, N).contiguous()
        f2 = fm[B:].reshape(T, B, D, N).contiguous()
        vol = torch.empty((T, B, N, N), device=dev)
        for _ in range(3):
            hip.corr_build_f32(f1, f2, vol)
        ms = kernel_event_ms(lambda: hip.corr_build_f32(f1, f2, vol), max(args.steps, 10))
        flops = 2.0 * T * B * D * N * N                       # 2*T*B*D*N^2 = 47.19 GFLOP per sample at C2
        bytes_alg = 4.0 * ((1 + T) * B * D * N + T * B * N * N)  # 393.2 MB per sample at C2
        tflops = flops / (ms * 1e-3) / 1e12
        out["roofline"] = {"kernel": "corr_build_f32_kernel", "bound": "mfma", "achieved": round(tflops, 2),
                           "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(tflops / PEAK_F32_MFMA_TFLOPS, 4),
                           "traffic": None, "avg_launch_ms": round(ms, 4), "flop_per_launch": flops,
                           "algorithmic_bytes_per_launch": bytes_alg,
                           "hbm_GBs_at_this_rate": round(bytes_alg / (ms * 1e-3) / 1e9, 1)}
        # secondary: the look-up gather (HBM-bound), 24.33 MB algorithmic per sample-iteration at C2
        from bflow_amd.corr import CorrBlockParallelMultiTarget, CorrComputation
        with torch.no_grad():
            blk = CorrBlockParallelMultiTarget(corr_computation_events=CorrComputation(fm[:B], fm[B:].view(T, B, D, *fm.shape[2:]),
                                                                                    cfg["correlation"]["ev"]["levels"]))
            params = low.get_params().clone()
            feat = blk.new_output()
            coef = model._coefficients()
            for _ in range(3):
                blk.lookup_bezier(params, coef, out=feat)
            ms_l = kernel_event_ms(lambda: blk.lookup_bezier(params, coef, out=feat), max(args.steps, 10))
        lbytes = 4.0 * B * N * blk.num_planes * (100 + 81)
        out["roofline_lookup"] = {"kernel": "corr_lookup_kernel<fused bezier>", "bound": "hbm",
                                  "achieved": round(lbytes / (ms_l * 1e-3) / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                  "frac": round(lbytes / (ms_l * 1e-3) / 1e9 / PEAK_HBM_GBS, 4), "traffic": None,
                                  "avg_launch_ms": round(ms_l, 4), "algorithmic_bytes_per_launch": lbytes}
        del blk, vol
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, sd, torch.from_numpy(vox_np[:1]))
            out["gpu_over_cpu"] = round(value / out["cpu_baseline"]["value"], 1)

    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
