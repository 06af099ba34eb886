#!/usr/bin/env python
"""bench.py -- headline benchmark of the RAFT-spline inference hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
        N == 1: runs in this process.
        N  > 1: if WORLD_SIZE is not set the script re-launches itself as N ranks (one per GPU, RCCL) through
                `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py ...`;
                launched that way by a driver it just runs as the rank it is.

Metric (BASELINE.json): frames/s of the whole job + ms per GRU iteration, raft-spline E_LU4_BD2 (events only), DSEC 640x480,
12 iterations.
  `value` is ONE workload at every N: BASELINE configs[1] -- batch 1 per GPU, one frame per rank and step (weak scaling; the same
          number is repeated as `c2_weak`).  A 1/2/4/8 series of `value` therefore never changes workload between two points.
  `c4_strong` (every N, next to `value`): BASELINE configs[3] -- GLOBAL batch 64 sharded over the N ranks (64/N frames per rank and
          step, processed in micro-batches of at most 8 = C4's per-GPU batch at N = 8): strong scaling, no data-path collective.
          A rank's micro-batches are independent: they run two at a time as parallel branches of one captured graph
          (bflow_amd/pipeline.py; same frames, bit-identical outputs).  `c2_two_in_flight` (N = 1 extras) is the same mechanism on two
          batch-1 frames: a frame-stream throughput, reported next to `value`, never as `value`.
  BFLOW_DIST_BACKEND (default nccl = RCCL) / BFLOW_DEVICE (default LOCAL_RANK) exist so that the N > 1 code path can be exercised on a
          one-GPU box (two gloo ranks on device 0: tests/test_multi_gpu.py); a driver never sets them.
After the timed regions the per-rank EPE state is all-gathered once over RCCL (the path's single exchange step).

One "step" = one forward (voxel grids resident in HBM -> full-resolution Bezier flow) over the rank's frames of that step,
replayed from a captured hipGraph.  Prints ONE JSON line on rank 0 (DESIGN.md section "Measurement" explains every field).
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

H, W, ITERS, CFG = 480, 640, 12, "E_LU4_BD2"
GLOBAL_BATCH, MICRO_BATCH = 64, 8          # BASELINE configs[3]
PEAK_SPLIT_TFLOPS = round(2500.0 / 3, 1)   # fp16 dense MFMA peak (~2.5 PFLOP/s) / 3 MFMA passes per fp32-class product
PEAK_HBM_GBS = 8000.0                      # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
# Shader clock a PURE stream of the split format's matrix instructions (6 x v_mfma_f32_32x32x16_f16 per 32-channel block, random data, nothing
# else) sustains on this part: 1.39-1.58 GHz (tools/micro/fp8_cross, profiles/r05_mfma_clock_fp8_cross.txt) -- the chip clocks to its power
# budget, so the 2.4-GHz peak is not reachable by ANY kernel that keeps the matrix pipes busy.  The encoder's persistent convolution runs at
# 1.31-1.38 GHz (batch 40) / 1.6-1.7 GHz (batch 5) with the matrix pipes 0.75 busy in cycles (profiles/r05_enc_stream_clock.txt).
MFMA_STREAM_SUSTAINED_GHZ = 1.5
PEAK_CLOCK_GHZ = 2.4
K5_SUSTAINED_GHZ = 1.7                     # shader clock of K5 under load (cycle stamps / wall clock per workgroup: profiles/r0x_k5_stamps_split8.txt, 1.66-1.8)
PMC_FILE = "r06_pmc.json"                  # rocprofv3 --pmc evidence of this round (tools/collect_profiles.sh)


def kernel_source_hash() -> str:
    """Hash of everything that determines the kernels' HBM traffic: the HIP sources and the ABI header.  profiles/<PMC_FILE>
    records it; bench.py only quotes the PMC traffic of kernels built from the SAME sources."""
    h = hashlib.sha256()
    cs = os.path.join(ROOT, "bflow_amd", "csrc")
    for f in sorted(os.listdir(cs)) + ["../../include/bflow_hip.h"]:
        p = os.path.join(cs, f)
        if os.path.isfile(p) and (f.endswith(".hip") or f.endswith(".h")):
            h.update(f.encode())
            h.update(open(p, "rb").read())
    return h.hexdigest()[:16]


def micro_batch_plan(global_batch: int, rank: int, world: int, graph: bool = True):
    """The frames rank `rank` of `world` runs per step of BASELINE configs[3]: its contiguous shard of the global batch cut into micro-batches of
    at most MICRO_BATCH frames, but at least two per rank when graphs are on (two forwards in flight fill the chip better than one: 8 frames
    at N = 8 as two concurrent micro-batches of 4 = 358 frames/s, one of 8 = 332; tools/two_frame_probe.py).
    Returns (micro, [(first_sample, n), ...])."""
    from bflow_amd import dist as bdist
    s0, s1 = bdist.shard_range(global_batch, rank, world)
    micro = min(MICRO_BATCH, max(1, (s1 - s0) // 2)) if graph else min(MICRO_BATCH, s1 - s0)
    n_micro = (s1 - s0) // micro
    # every frame of the shard must be run: frames/s counts s1 - s0 frames per step (a remainder micro-batch would be counted, not run)
    assert n_micro * micro == s1 - s0, f"the rank's {s1 - s0} frames do not split into micro-batches of {micro}: use a world size that divides {global_batch} into even shards"
    return micro, [(s0 + k * micro, micro) for k in range(n_micro)]


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay (debug)")
    ap.add_argument("--no-extras", action="store_true", help="skip the per-kernel rooflines and the secondary workloads")
    ap.add_argument("--frame-only", action="store_true", help="value, c4_strong, ms/GRU-iter only: no per-kernel rooflines, stage tables or other configs (tools/ab_bench.sh)")
    return ap.parse_args()


def relaunch(args) -> int:
    """--gpus N > 1 without a torchrun environment: become the launcher of N ranks on this node."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # RCCL needs dmabuf IPC on this driver stack
    # host-thread policy, stated (torchrun would otherwise set OMP_NUM_THREADS=1 silently): the ranks share the box's usable cores evenly;
    # a rank's host work is launch enqueueing on one thread, the rest only serves torch's CPU ops outside the timed loop
    env.setdefault("OMP_NUM_THREADS", str(max(1, usable_cores() // args.gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(relaunch(args))

    import numpy as np
    import torch

    import bflow_amd
    from bflow_amd import configs, dist as bdist, synthetic
    from bflow_amd.metrics import epe_masked
    from bflow_amd.weights import deterministic_state_dict

    backend = os.environ.get("BFLOW_DIST_BACKEND", "nccl")          # "nccl" IS RCCL on ROCm; "gloo" only for the one-GPU test of the N > 1 path
    rank, world, local = bdist.init_from_env(backend)
    if "BFLOW_DEVICE" in os.environ:                                # every rank on one device (the same test)
        local = int(os.environ["BFLOW_DEVICE"])
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with `python bench.py --gpus {args.gpus}` (self-launching) "
                         f"or `python -m torch.distributed.run --nnodes=1 --nproc-per-node {args.gpus} --master-addr 127.0.0.1 bench.py --gpus {args.gpus}`")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    host_threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or max(1, usable_cores() // max(world, 1))
    torch.set_num_threads(host_threads)

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    def time_steps(fn, steps):
        """EXACTLY `steps` steps bracketed by barrier + synchronize on both sides; max over ranks."""
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        barrier()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        if world > 1:
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        return float(t.item())

    def kernel_event_ms(launch, n):
        """(average duration in ms of one launch, average SHADER clock in GHz during the launches): `n` launches captured into one hipGraph
        between two bflow_shader_clock_stamp launches (s_memtime / s_memrealtime per CU), hipEvents recorded on the launch stream (torch's
        current stream) around its replay.  (An event pair around every single eager launch also times the HOST's enqueue latency -- 5-8 us
        of Python per launch, box dependent -- which is 40 % of the 13-us look-up and moved its fraction between 0.14 and 0.17.)  The
        clock is what tells a slow box from a regression: the part clocks to its power budget, 1.3-2.1 GHz under load."""
        from bflow_amd import hip as _hip
        for _ in range(3):
            launch()
        torch.cuda.synchronize()
        pairs = _hip.shader_clock_tables(2, dev)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            _hip.shader_clock_stamp(pairs, 0)
            for _ in range(n):
                launch()
            _hip.shader_clock_stamp(pairs, 1)
        g.replay()
        torch.cuda.synchronize()
        times, clocks = [], []
        for _ in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            g.replay()
            b.record()
            torch.cuda.synchronize()
            times.append(a.elapsed_time(b) / n)
            clocks.append(_hip.shader_clock_ghz(pairs))
        del g
        return float(np.mean(times)), float(np.mean(clocks))

    def clocked(fn, steps):
        """time_steps(fn, steps) with the average shader clock of the timed region (two stamps on the launch stream around it)."""
        from bflow_amd import hip as _hip
        pairs = _hip.shader_clock_tables(2, dev)
        barrier()
        torch.cuda.synchronize()
        _hip.shader_clock_stamp(pairs, 0)
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        _hip.shader_clock_stamp(pairs, 1)
        torch.cuda.synchronize()
        barrier()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        if world > 1:
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        return float(t.item()), round(_hip.shader_clock_ghz(pairs), 3)

    cfg = configs.model_config(CFG)
    model = bflow_amd.RAFTSpline(cfg).eval()
    sd = deterministic_state_dict(model, seed=0)
    model.load_state_dict(sd)
    model.to(dev)
    # NO opt-in here: `value` is measured on what the drop-in seam delivers by itself -- the call val.py makes (eval(), inference_mode,
    # test_mode=True; modules/raft_spline.py:57-58 under val.py:75) replays a captured hipGraph and returns private copies of its outputs
    # (bflow_amd/raft_spline.py `_use_graph`).  `value_eager` below is the same call with BFLOW_HIPGRAPH=0 semantics.
    if args.no_graph:
        model.enable_hipgraph(False)

    # ---- workload A: C2, batch 1 per GPU (weak scaling); sample index = rank
    vox1_np = synthetic.voxel_grid(1, 9, H, W, seed=1234, first_sample=rank)
    vox1 = torch.from_numpy(vox1_np).to(dev)

    def step_c2():
        with torch.inference_mode():
            return model(voxel_grid=vox1, iters=ITERS, test_mode=True)

    # ---- workload B: C4, global batch 64 -> contiguous shard of this rank, micro-batches of 8 (one captured graph, replayed per micro-batch)
    assert GLOBAL_BATCH % (world * MICRO_BATCH) == 0 or world * MICRO_BATCH > GLOBAL_BATCH, "global batch 64 must split into micro-batches of 8"
    s0, s1 = bdist.shard_range(GLOBAL_BATCH, rank, world)
    micro, plan = micro_batch_plan(GLOBAL_BATCH, rank, world, graph=not args.no_graph)
    n_micro = len(plan)
    vox8 = None

    def setup_c4():
        nonlocal vox8
        # the shard's micro-batches share one resident buffer set: every micro-batch replays the same graph on its own frames
        vox8 = [torch.from_numpy(synthetic.voxel_grid(n, 9, H, W, seed=1234, first_sample=first)).to(dev) for first, n in plan]

    # micro-batches are independent: two of them run as parallel branches of one captured graph (bflow_amd/pipeline.py) -- same frames,
    # same arithmetic (bit-identical outputs), the second forward fills the CUs the first one's GRU loop leaves idle
    pair = None
    if not args.no_graph and n_micro >= 2 and n_micro % 2 == 0:
        from bflow_amd.pipeline import ConcurrentRunner
        pair = ConcurrentRunner(model, ITERS, streams=2)

    def step_c4():
        out = None
        if pair is not None:
            for k in range(0, n_micro, 2):
                out = pair([vox8[k], vox8[k + 1]])
            return out
        with torch.inference_mode():
            for v in vox8:
                out = model(voxel_grid=v, iters=ITERS, test_mode=True)
        return out

    def measure(step, frames_per_step_per_rank, steps, warmup):
        for _ in range(max(warmup, 1)):
            step()
        el, ghz = clocked(step, steps)
        return {"value": round(world * frames_per_step_per_rank * steps / el, 3), "ms_per_step": round(el / steps * 1e3, 4), "steps": steps,
                "clock_ghz": ghz}

    # `value` = C2 weak at EVERY N (one workload for the whole 1/2/4/8 series); C4 strong is measured next to it
    res_c2 = measure(step_c2, 1, args.steps, args.warmup)
    res_c4 = None
    if world > 1 or not args.no_extras:
        setup_c4()
        res_c4 = measure(step_c4, s1 - s0, args.steps if world > 1 else max(2, min(args.steps, 6)), max(1, args.warmup if world > 1 else 1))
        vox8 = None

    # ---- the path's single exchange step: per-rank EPE state all-gathered over RCCL (outside the timed regions)
    low, up = step_c2()
    gt = torch.from_numpy(synthetic.gt_flow(1, H, W, seed=99, first_sample=rank)).to(dev)
    e = epe_masked(up.get_flow_from_reference(1.0).contiguous(), gt)
    epe_mean, epe_sum, epe_cnt = bdist.reduce_epe(e.double(), torch.ones((), dtype=torch.float64, device=dev))

    # LOCAL_RANK -> device of every rank (one process per GPU), gathered with the same 16-B-per-rank exchange
    rec = bdist.all_gather_records(torch.tensor([float(rank), float(local), float(torch.cuda.current_device())], dtype=torch.float64, device=dev))
    rank_devices = [[int(v) for v in row] for row in rec.cpu().tolist()]

    out = None
    if rank == 0:
        wl_c2 = (f"raft-spline {CFG} events-only, DSEC-shaped voxel grid (9x{H}x{W}), batch 1/GPU, {ITERS} GRU iters (BASELINE configs[1]), "
                 "random-init deterministic weights")
        wl_c4 = (f"raft-spline {CFG} events-only, DSEC-shaped voxel grids (9x{H}x{W}), GLOBAL batch {GLOBAL_BATCH} sharded over {world} GPU(s) "
                 f"({s1 - s0} frames per rank and step in micro-batches of {micro}" + (", two micro-batches in flight" if pair is not None else "") +
                 f"), {ITERS} GRU iters (BASELINE configs[3]), random-init deterministic weights")
        out = {
            "metric": "frames/sec (whole node), raft-spline DSEC 640x480 12-iter",
            "value": res_c2["value"], "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": res_c2["ms_per_step"], "clock_ghz": res_c2.get("clock_ghz"), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32 (split-fp16 pairs)",
            "arithmetic": "fp32 values carried as split fp16 pairs (hi + lo*2^-11) on the fp16 matrix cores, fp32 accumulation; correlation cross terms (hi*lo + lo*hi) on the fp8 matrix rate; parity 2e-5 px EPE vs the fp32 CPU reference; `value_split` = the same frame with the correlation on three fp16 passes (fp32 class everywhere)",
            "data": "synthetic",
            "config": {"workload": wl_c2, "global_batch": world, "frames_per_rank_per_step": 1, "iters": ITERS, "hipgraph": not args.no_graph,
                       "hipgraph_mode": "off (--no-graph)" if args.no_graph else "auto: the model's default for eval + inference_mode + test_mode=True forwards, no opt-in call (what val.py gets)",
                       "same_workload_at_every_n_gpus": True,
                       "input_handover": "every step copies its resident frame (device to device, 11 MB) into the captured graph's static input buffer, inside "
                                         "the timed region (measured: no difference to a zero-copy hand-over, 277-279 frames/s either way)"},
            "c2_weak": dict(res_c2, unit="frames/s", workload=wl_c2, scaling="weak"),
            "epe_vs_synthetic_gt": round(float(epe_mean), 4), "epe_ranks_gathered": int(epe_cnt),
            "dist_backend": backend if world > 1 else None,
            "rccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()) if (world > 1 and backend == "nccl") else None,
            "rank_devices": rank_devices, "host_threads_per_rank": host_threads,
        }
        if res_c4 is not None:
            out["c4_strong"] = dict(res_c4, unit="frames/s", workload=wl_c4, scaling="strong", frames_per_rank_per_step=s1 - s0, micro_batch=micro,
                                    micro_batches_per_rank=n_micro)

    # ---- extras outside the timed region: ms/GRU-iter (N=1), rooflines of the hand-written kernels (rank 0), CPU baseline (N=1)
    if world == 1 and not args.no_extras:
        def step6():
            with torch.inference_mode():
                return model(voxel_grid=vox1, iters=ITERS // 2, test_mode=True)
        for _ in range(3):
            step6()
        n_it = max(args.steps // 2, 5)
        # the difference of two ~4 ms timings: three interleaved rounds, the fastest of each (clock ramps only ever add time)
        t12 = min(time_steps(step_c2, n_it) / n_it for _ in range(1))
        t6 = time_steps(step6, n_it) / n_it
        for _ in range(2):
            t12 = min(t12, time_steps(step_c2, n_it) / n_it)
            t6 = min(t6, time_steps(step6, n_it) / n_it)
        # two batch-1 frames in flight (parallel branches of one graph, bflow_amd/pipeline.py): throughput of a frame STREAM; reported
        # next to `value`, never as `value` (which stays one frame per step)
        if not args.no_graph:
            from bflow_amd.pipeline import ConcurrentRunner
            vox1b = torch.from_numpy(synthetic.voxel_grid(1, 9, H, W, seed=1234, first_sample=rank + 1)).to(dev)
            two = ConcurrentRunner(model, ITERS, streams=2)
            for _ in range(3):
                two([vox1, vox1b])
            t2 = min(time_steps(lambda: two([vox1, vox1b]), n_it) / n_it for _ in range(2))
            out["c2_two_in_flight"] = {"value": round(2.0 / t2, 3), "unit": "frames/s", "ms_per_replay": round(t2 * 1e3, 4), "frames_per_replay": 2,
                                       "note": "two independent batch-1 forwards as parallel branches of one hipGraph; outputs bit-identical to the sequential forward"}
        out["ms_per_gru_iter"] = round((t12 - t6) / (ITERS // 2) * 1e3, 4)
        out["ms_fixed_part"] = round((t12 - ITERS * (t12 - t6) / (ITERS // 2)) * 1e3, 4)
        # the fp32-class number next to `value`: the same frame with the correlation volume on THREE fp16 MFMA passes ("split": 2^-22 per
        # product, the arithmetic of every convolution of the network) instead of hi*hi + fp8 cross terms ("split8", 2^-16 per product)
        if not args.no_graph:
            model.corr_precision = "split"
            r_split = measure(step_c2, 1, max(args.steps // 2, 5), 2)
            model.corr_precision = None
            out["value_split"] = dict(r_split, unit="frames/s", corr_precision="split",
                                      note="whole frame, correlation on three fp16 MFMA passes: fp32-class arithmetic everywhere (1.3e-5 px vs the fp32 oracle)")
            step_c2()        # back on the default graph
            # the same call with graph replay switched off (BFLOW_HIPGRAPH=0 / enable_hipgraph(False)): ~230 launches enqueued by the host
            model.enable_hipgraph(False)
            r_eager = measure(step_c2, 1, max(args.steps // 2, 5), 2)
            model.enable_hipgraph(None)
            out["value_eager"] = dict(r_eager, unit="frames/s", note="the same forward as eager launches (graph replay off): what the seam delivered "
                                      "before replay became its default; host-enqueue-bound")
            # and with the graph's static output buffers handed out as they are (enable_hipgraph(): aliasing documented in INTEGRATION.md)
            model.enable_hipgraph()
            r_static = measure(step_c2, 1, max(args.steps // 2, 5), 2)
            model.enable_hipgraph(None)
            out["value_static_outputs"] = dict(r_static, unit="frames/s", note="explicit enable_hipgraph(): no private copies of the two outputs")
            step_c2()
        # frame-level matrix roofline: algorithmic FLOPs as executed (context share of the gate convolutions hoisted, mask head once) over
        # the frame / the marginal iteration, against the split format's peak (fp16 dense / 3)
        from tools.roofline_kernels import frame_flops
        ff, fi, parts = frame_flops(model, 1, H, W, ITERS)
        out["roofline_frame"] = {"bound": "mfma", "flop_per_frame": ff, "achieved": round(ff / (res_c2["ms_per_step"] * 1e-3) / 1e12, 1), "peak": PEAK_SPLIT_TFLOPS,
                                 "unit": "TFLOP/s", "frac": round(ff / (res_c2["ms_per_step"] * 1e-3) / 1e12 / PEAK_SPLIT_TFLOPS, 4),
                                 "parts_gflop": {k: round(v / 1e9, 2) for k, v in parts.items()},
                                 "note": "whole frame incl. its HBM-bound and latency-bound stages; FLOPs after hoisting the context share of the gate "
                                         "convolutions and running the mask head once"}
        out["roofline_update_iter"] = {"bound": "mfma", "flop_per_iteration": fi, "achieved": round(fi / (out["ms_per_gru_iter"] * 1e-3) / 1e12, 1),
                                       "peak": PEAK_SPLIT_TFLOPS, "unit": "TFLOP/s", "frac": round(fi / (out["ms_per_gru_iter"] * 1e-3) / 1e12 / PEAK_SPLIT_TFLOPS, 4),
                                       "note": "one pass of raft.py:166-195 at batch 1: ten dependent launches on <= 240 workgroups each"}

    if rank == 0 and not args.no_extras and not args.frame_only:
        # rooflines of the hand-written kernels (tools/roofline_kernels.py), each timed with hipEvents on the launch stream on the operands
        # of the C2 workload (the timed region above is graph replays, inside which events cannot be recorded)
        from tools.roofline_kernels import build as roofline_kernels, build_big, build_c4_convs
        vox8 = None
        torch.cuda.empty_cache()
        src_hash = kernel_source_hash()
        try:
            pmc_doc = json.load(open(os.path.join(ROOT, "profiles", PMC_FILE)))
        except Exception:
            pmc_doc = {}
        pmc_ok = pmc_doc.get("kernel_source_hash") == src_hash
        pmc = pmc_doc.get("kernels", {}) if pmc_ok else {}
        out["kernel_source_hash"] = src_hash

        # the clock a PURE stream of the split format's matrix instructions sustains on THIS box, measured now (tools/micro/fp8_cross, built by
        # __graft_entry__.build(); the constant is the round-5 reading and only stands in when the binary is missing)
        mfma_clock = {"ghz": MFMA_STREAM_SUSTAINED_GHZ, "source": "constant: profiles/r05_mfma_clock_fp8_cross.txt (1.39-1.58 GHz); tools/micro/fp8_cross not built"}
        fx = os.path.join(ROOT, "tools", "micro", "fp8_cross")
        if os.path.isfile(fx) and os.access(fx, os.X_OK):
            try:
                txt = subprocess.run([fx], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=120).stdout
                cl = sorted(float(l.split("shader clock")[1].split("GHz")[0]) for l in txt.splitlines() if l.startswith("6 x fp16 MFMA per block") and "shader clock" in l)
                if cl:
                    mfma_clock = {"ghz": cl[len(cl) // 2], "all": cl, "source": "measured in this run: tools/micro/fp8_cross rate_kernel<0> (6 x v_mfma_f32_32x32x16_f16 "
                                  "per 32-channel block on every SIMD, random data; s_memtime / s_memrealtime per workgroup), median of 3"}
            except Exception as e:       # a tools binary: its failure must not take the bench line down
                mfma_clock["error"] = repr(e)[:200]
        out["mfma_stream_clock"] = mfma_clock

        def price(k):
            for _ in range(3):
                k["launch"]()
            ms, ghz = kernel_event_ms(k["launch"], max(args.steps, 10))
            if k["bound"] == "mfma":
                tf = k["flops"] / (ms * 1e-3) / 1e12
                r = {"kernel": k["name"], "bound": "mfma", "achieved": round(tf, 2), "peak": PEAK_SPLIT_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(tf / PEAK_SPLIT_TFLOPS, 4), "traffic": None, "avg_launch_ms": round(ms, 4), "flop_per_launch": k["flops"],
                     "algorithmic_bytes_per_launch": k["bytes"],
                     "note": "algorithmic (fp32-equivalent) FLOPs; the split scheme executes 3 fp16 MFMAs per product, so "
                             "peak = 2500 TFLOP/s fp16 dense / 3",
                     # second reading of the same number: against what the matrix pipes deliver at the clock the chip SUSTAINS under a pure
                     # matrix stream (power-limited); `frac` above stays against the 2.4-GHz peak
                     "power_model": {"mfma_stream_clock_ghz": mfma_clock["ghz"],
                                     "peak_at_that_clock": round(PEAK_SPLIT_TFLOPS * mfma_clock["ghz"] / PEAK_CLOCK_GHZ, 1),
                                     "frac_of_that": round(tf / (PEAK_SPLIT_TFLOPS * mfma_clock["ghz"] / PEAK_CLOCK_GHZ), 4),
                                     "source": mfma_clock["source"]}}
            else:
                gbs = k["bytes"] / (ms * 1e-3) / 1e9
                r = {"kernel": k["name"], "bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                     "frac": round(gbs / PEAK_HBM_GBS, 4), "traffic": None, "avg_launch_ms": round(ms, 4),
                     "algorithmic_bytes_per_launch": k["bytes"]}
                if k["flops"]:
                    r["flop_per_launch"] = k["flops"]
                    r["tflops_equivalent"] = round(k["flops"] / (ms * 1e-3) / 1e12, 1)
                    if k.get("mfma_peak"):     # the kernel's second roof: matrix cores (fp32-equivalent peak of its arithmetic)
                        r["mfma_peak_tflops_equivalent"] = round(k["mfma_peak"], 1)
                        r["frac_mfma"] = round(k["flops"] / (ms * 1e-3) / 1e12 / k["mfma_peak"], 4)
                if k.get("line_bytes"):        # the look-up: what a materialised fp32 volume allows at 128-B line granularity
                    r["line_bytes"] = round(k["line_bytes"])
                    r["frac_of_line_granular_cap"] = round(k["line_bytes"] / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)
                    r["line_note"] = ("line_bytes = the 128-B lines (4 x 8-element tiles) a 10 x 10 tap window touches on tiled fp32 planes + the 81 values written: the "
                                      "floor of ANY gather on a materialised volume; frac_of_line_granular_cap = line_bytes / time / 8 TB/s")
            r["clock_ghz"] = round(ghz, 3)      # average shader clock DURING these launches (bflow_shader_clock_stamp): measured, in-run
            if k["bound"] == "mfma":
                r["frac_at_measured_clock"] = round(r["achieved"] / (PEAK_SPLIT_TFLOPS * ghz / PEAK_CLOCK_GHZ), 4)
            if k.get("note"):
                r["note"] = (r.get("note", "") + "; " if r.get("note") else "") + k["note"]
            # HBM traffic per launch cannot be read from inside this process: it comes from the rocprofv3 --pmc passes of the SAME launches
            # (tools/collect_profiles.sh -> profiles/<PMC_FILE>), and only when that file was collected on kernels built from these sources
            if k["name"] in pmc:
                r["traffic"] = pmc[k["name"]]["traffic"]
                mu = (pmc[k["name"]].get("mfma") or {}).get("mfma_utilisation")
                if mu is not None:      # matrix-pipe busy cycles per cycle with waves of this kernel, normalised by a pure MFMA stream (same file)
                    r["mfma_utilisation"] = round(mu, 4)
                r["traffic_source"] = f"profiles/{PMC_FILE} (rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE; separate passes; same kernel sources)"
            elif pmc_doc and not pmc_ok:
                r["traffic_source"] = f"none: profiles/{PMC_FILE} was collected on different kernel sources (re-run tools/collect_profiles.sh)"
            out[k["key"]] = r
            k.clear()

        for k in roofline_kernels(model, vox1, cfg, low.get_params()):
            rider_bytes = k.get("rider_bytes")
            price(k)
            if rider_bytes is not None:      # the batch-1 product launch = look-up + im2col rider: reported inside roofline_lookup, never in its frac
                w = out.pop("_lookup_with_rider")
                out["roofline_lookup"]["product_launch_with_rider"] = {
                    "avg_launch_ms": w["avg_launch_ms"], "rider_bytes": rider_bytes,
                    "rider_extra_us": round((w["avg_launch_ms"] - out["roofline_lookup"]["avg_launch_ms"]) * 1e3, 2),
                    "note": "bflow_corr_lookup_im2col: the 7x7 windows of the Bezier parameters for convf1 written by the first workgroups of the look-up launch"}
        torch.cuda.empty_cache()
        for k in build_big(model, cfg, dev):
            price(k)
        torch.cuda.empty_cache()
        # the convolution kernels that lead the FULL trace, at the shape that sets whole-node throughput (C4 per-GPU shard, batch 8)
        for k in build_c4_convs(model, dev):
            price(k)
        torch.cuda.empty_cache()
        # K5 is bound by its STORE stream: the ceiling of a pure store stream, measured in this process on this box -- hipMemsetAsync of the
        # C2 volume (368.6 MB; torch's zero_() is that call) -- and, when the tools binary is there, K5's own store shape
        # (tools/micro/store_patterns: 256 persistent 8-wave workgroups, 32 rows x 1 KB per item, 4 B per lane, nt)
        volz = torch.empty((4, 4800, 4800), device=dev)
        ms_set, _ = kernel_event_ms(lambda: volz.zero_(), 10)
        ceil = {"hipMemsetAsync_gbs": round(volz.numel() * 4 / (ms_set * 1e-3) / 1e9, 1)}
        del volz
        sp = os.path.join(ROOT, "tools", "micro", "store_patterns")
        if os.path.isfile(sp) and os.access(sp, os.X_OK):
            try:
                txt = subprocess.run([sp], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=120).stdout
                for line in txt.splitlines():
                    if line.startswith("k5 order,   4 B/lane, nt"):
                        ceil["k5_shape_4B_nt_gbs"] = float(line.split("us")[1].split("GB/s")[0])
                    if line.startswith("lock order, 4 B/lane, nt"):
                        ceil["lockstep_4B_nt_gbs"] = float(line.split("us")[1].split("GB/s")[0])
            except Exception as e:       # a tools binary: its absence or failure must not take the bench line down
                ceil["store_patterns_error"] = repr(e)[:200]
        for key, units in (("roofline_corr_build", 2.0), ("roofline_corr_build_split", 3.0)):
            if key in out:
                r = out[key]
                best = max(v for k_, v in ceil.items() if k_.endswith("_gbs"))
                r["store_ceiling_gbs"] = best
                r["store_ceilings"] = ceil
                r["frac_of_store_ceiling"] = round(r["achieved"] / best, 4)
                # the bound this arithmetic can reach: the matrix pipes at the clock the chip sustains under this load (32 cycles per
                # v_mfma_f32_32x32x16_f16 per SIMD, 16 k-steps x `units` per 32x32x256 block, 1024 SIMDs) NEXT TO the store stream at the
                # measured ceiling of a pure store stream of K5's own shape; perfectly overlapped = the slower of the two
                blocks = r["flop_per_launch"] / (2.0 * 32 * 32 * 256)
                k5_ghz = r.get("clock_ghz") or K5_SUSTAINED_GHZ          # the clock measured during THESE launches
                t_mfma = blocks * 16 * units * 32 / 1024 / (k5_ghz * 1e9)
                k5_store = ceil.get("k5_shape_4B_nt_gbs", ceil["hipMemsetAsync_gbs"])
                t_store = r["algorithmic_bytes_per_launch"] / (k5_store * 1e9)
                r["model_cap"] = {"frac": round(r["algorithmic_bytes_per_launch"] / max(t_mfma, t_store) / 1e9 / PEAK_HBM_GBS, 4),
                                  "frac_if_additive": round(r["algorithmic_bytes_per_launch"] / (t_mfma + t_store) / 1e9 / PEAK_HBM_GBS, 4),
                                  "t_mfma_us": round(t_mfma * 1e6, 1), "t_store_us": round(t_store * 1e6, 1), "sustained_clock_ghz": round(k5_ghz, 3),
                                  "clock_source": "measured in this run (bflow_shader_clock_stamp around the timed launches)",
                                  "store_stream_gbs": k5_store,
                                  "note": "cap of this arithmetic = max(matrix pipes at the sustained clock, store stream of K5's shape); "
                                          "frac is measured against 8 TB/s, which no arithmetic with matrix work can reach here"}
        if world == 1:
            out["gpu_stage_ms"] = gpu_stage_ms(cfg, sd, vox1, dev)
            v8 = torch.from_numpy(synthetic.voxel_grid(MICRO_BATCH, 9, H, W, seed=1234)).to(dev)
            out["gpu_stage_ms_c4"] = gpu_stage_ms(cfg, sd, v8, dev)      # ONE micro-batch of 8 (C4's per-GPU batch at N = 8), one forward in flight
            out["gpu_stage_ms_c4"]["workload"] = "one forward at batch 8 (BASELINE configs[3]'s per-GPU batch), same in-graph stamps"
            del v8
            torch.cuda.empty_cache()
            out["voxel_kernels"] = voxel_kernels(dev)
            out["pipeline_from_events"] = pipeline_from_events(model, cfg, dev)
            torch.cuda.empty_cache()
            out.update(other_baseline_configs(dev, max(3, min(args.steps, 10))))
            torch.cuda.empty_cache()
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, sd, torch.from_numpy(vox1_np[:1]))
            out["gpu_over_cpu"] = round(out["c2_weak"]["value"] / out["cpu_baseline"]["value"], 1)

    if rank == 0:
        # the scalars a reader needs, LAST: a driver that keeps the tail of a long line keeps these
        def g(*path):
            d = out
            for k in path:
                if not isinstance(d, dict) or k not in d:
                    return None
                d = d[k]
            return d
        fr = {k: out[k].get("frac") for k in out if k.startswith("roofline") and isinstance(out[k], dict)}
        ck = {k: out[k].get("clock_ghz") for k in out if k.startswith("roofline") and isinstance(out[k], dict) and out[k].get("clock_ghz") is not None}
        out["summary"] = {"value": out["value"], "ms_per_step": out["ms_per_step"], "clock_ghz": out.get("clock_ghz"), "clock_ghz_per_roofline": ck, "ms_per_gru_iter": out.get("ms_per_gru_iter"),
                          "ms_fixed_part": out.get("ms_fixed_part"), "value_split": g("value_split", "value"),
                          "value_eager": g("value_eager", "value"), "value_static_outputs": g("value_static_outputs", "value"),
                          "c4_strong": g("c4_strong", "value"), "c2_two_in_flight": g("c2_two_in_flight", "value"),
                          "c4_rank_shape_at_n8": g("c4_rank_shape_at_n8", "value"), "c3_batch8": g("c3_batch8", "value"), "c5": g("c5", "value"),
                          "pipeline_from_events": g("pipeline_from_events", "value"),
                          "k1_float_xy_ms": g("voxel_kernels", "k1_float_xy", "ms"), "k1_float_xy_frac": g("voxel_kernels", "k1_float_xy", "frac"),
                          "k1_int_xy_frac": g("voxel_kernels", "k1_int_xy", "frac"), "k2_frac": g("voxel_kernels", "k2_norm", "frac"),
                          "frac": fr, "lookup_frac_of_line_cap": g("roofline_lookup", "frac_of_line_granular_cap"),
                          "lookup_c4_frac_of_line_cap": g("roofline_lookup_c4_shard", "frac_of_line_granular_cap"),
                          "k5_model_cap": g("roofline_corr_build", "model_cap", "frac"),
                          "roofline_frac_at_sustained_mfma_clock": g("roofline", "power_model", "frac_of_that"), "cpu_frames_s": g("cpu_baseline", "value"),
                          "gpu_stage_ms": {k: v for k, v in (out.get("gpu_stage_ms") or {}).items() if isinstance(v, (int, float))},
                          "gpu_stage_ms_c4": {k: v for k, v in (out.get("gpu_stage_ms_c4") or {}).items() if isinstance(v, (int, float))},
                          "epe_ranks_gathered": out.get("epe_ranks_gathered"), "rccl": out.get("rccl_version")}
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


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


def cpu_baseline(cfg, sd, vox_cpu, budget_s=32.0):
    """The CPU oracle (op-for-op restatement of the reference, pinned to it in the build container) on this host's cores, with the
    protocol of SURVEY 8(d): torch.set_num_threads(all usable cores) AND 1 thread (the reference's own data path pins torch to one
    thread, representations.py:5-6; val.py:5-9 exports OMP_NUM_THREADS=1), inference_mode, 2 warm-ups (utils/timers.py:64) + 5 timed
    forwards of the same frame (the 1-thread leg stops at `budget_s` seconds of timed work, >= 3 forwards), per-stage milliseconds under
    the reference's CudaTimer names (raft.py:116-186).  Bounded sample: one 640x480 frame, 12 iterations."""
    import numpy as np
    import torch
    from oracle import raft_spline_oracle as O   # checker / baseline only -- never on the product path
    cores = usable_cores()

    def leg(threads, warmups, forwards, budget):
        torch.set_num_threads(threads)
        stages, opened = {}, {}

        def hook(name, begin):
            t = time.perf_counter()
            if begin:
                opened[name] = t
            else:
                stages.setdefault(name, []).append(t - opened.pop(name))

        with torch.inference_mode():
            for _ in range(warmups):
                O.forward(sd, cfg, vox_cpu, None, iters=ITERS, test_mode=True)
            times = []
            t_start = time.perf_counter()
            while len(times) < forwards and (len(times) < 3 or time.perf_counter() - t_start < budget):
                t0 = time.perf_counter()
                O.forward(sd, cfg, vox_cpu, None, iters=ITERS, test_mode=True, stage_hook=hook)
                times.append(time.perf_counter() - t0)
        sec = float(np.mean(times))
        return {"value": round(1.0 / sec, 4), "unit": "frames/s", "threads": threads, "ms_per_frame": round(sec * 1e3, 1),
                "forwards": len(times), "warmups": warmups,
                "stage_ms": {k: round(float(np.mean(v)) * 1e3, 2) for k, v in stages.items()}}

    full = leg(cores, 2, 5, budget_s)
    one = leg(1, 1, 5, budget_s) if cores > 1 else full
    torch.set_num_threads(cores)
    return {"value": full["value"], "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{full['forwards']} forwards of 1 frame (640x480, {ITERS} iters) after {full['warmups']} warm-ups, torch CPU fp32, {cores} threads; "
                      f"1-thread leg: {one['forwards']} forwards after {one['warmups']} warm-up",
            "ms_per_frame": full["ms_per_frame"], "ms_per_gru_iter": full["stage_ms"].get("1 iter"),
            "stage_ms": full["stage_ms"], "single_thread": one,
            "stage_names": "the reference's CudaTimer hooks, models/raft_spline/raft.py:116-186 (per-iteration stages: mean per occurrence)"}


def gpu_stage_ms(cfg, sd, vox, dev):
    """Per-stage milliseconds of the HIP path under the reference's hook names, read from INSIDE the hipGraph replay `value` is measured
    on: one-thread bflow_clock_stamp launches (100 MHz device wall clock) at the stage boundaries of a captured forward
    (bflow_amd/timers.py StampTimer), mean of 5 replays after 3 warm-ups.  `cnet` runs on its own branch next to `fnet_ev`."""
    import numpy as np
    import bflow_amd
    from bflow_amd.timers import StampTimer
    m = bflow_amd.RAFTSpline(cfg).eval()
    m.load_state_dict(sd)
    m.to(dev)
    m.enable_hipgraph()
    st = StampTimer(dev)
    m._probe = st
    acc = {}
    for k in range(8):
        m(voxel_grid=vox, iters=ITERS, test_mode=True)
        if k >= 3:
            for name, v in st.stage_ms().items():
                acc.setdefault(name, []).append(v)
    out = {k: round(float(np.mean(v)), 4) for k, v in acc.items()}
    out["note"] = ("in-graph clock stamps (each a one-thread launch on the chain: ~1-2 us per boundary, 2 per iteration); 'get_flow (per iter)' is fused into "
                   "the look-up kernel; 'corr lookup (per iter)' includes the im2col rider of the same launch")
    return out


def voxel_kernels(dev):
    """K1 / K2 on synthetic DSEC-shaped events: 2 M events per 100 ms window into the 15-bin 480x640 grid; algorithmic
    bytes per SURVEY 8(d): 16 B per event + 8 (float xy) or 2 (int xy) fp32 read-modify-writes (8 B each) + the grid once; K2: four passes
    over the grid.  Graph-timed whole calls (K1 = four launches: count, scan, place, gather) with a resident workspace."""
    import numpy as np
    import torch
    from bflow_amd import hip, synthetic
    C, Hh, Ww, n_ev = 15, H, W, 2_000_000
    out = {}

    def graph_ms(fn, reps=10):
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(reps):
                fn()
        g.replay()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            g.replay()
            b.record()
            torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b) / reps)
        return best

    for tag, int_xy in (("float_xy", False), ("int_xy", True)):
        ev = synthetic.events(n_ev, Hh, Ww, 0, 100_000, seed=7, int_xy=int_xy)
        x, y, p, t = (torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in ev)
        grid = torch.empty((C, Hh, Ww), device=dev)
        ws = hip._voxel_workspace(n_ev, C, Hh, Ww, not int_xy, dev)
        ms = graph_ms(lambda: hip.voxel_grid(x, y, p, t, 0, 100_000, grid, ws))
        a = grid.clone()
        hip.voxel_grid(x, y, p, t, 0, 100_000, grid, ws)
        atom = 2 if int_xy else 8
        by = n_ev * (16 + atom * 8) + C * Hh * Ww * 4
        out["k1_" + tag] = {"ms": round(ms, 4), "events_per_s": round(n_ev / ms * 1e3), "algorithmic_gb_s": round(by / ms / 1e6, 1),
                            "frac": round(by / ms / 1e6 / PEAK_HBM_GBS, 4), "bound": "hbm", "events": n_ev, "grid": [C, Hh, Ww],
                            "bit_identical_run_to_run": bool(torch.equal(a, grid)), "workspace_mb": round(ws.numel() / 1e6, 1),
                            "kernel": "voxel_count / scan / place / gather (tile-binned, LDS fixed-point accumulation, no global atomics)"}
    g = torch.randn(9, Hh, Ww, device=dev) * (torch.rand(9, Hh, Ww, device=dev) < 0.3)
    wsn = hip.voxel_norm_workspace(dev)
    q = g.clone()
    ms = graph_ms(lambda: hip.voxel_norm(q, wsn))
    by = 4 * g.numel() * 4
    out["k2_norm"] = {"ms": round(ms, 4), "algorithmic_gb_s": round(by / ms / 1e6, 1), "frac": round(by / ms / 1e6 / PEAK_HBM_GBS, 4), "bound": "hbm",
                      "grid": [9, Hh, Ww], "note": "algorithmic bytes as in rounds 1-5 (4 passes over the 11-MB grid); round 6 executes 2 reads + 1 write in two launches"}
    return out


def pipeline_from_events(model, cfg, dev, steps=10):
    """Frames/s of the chain from RAW events (SURVEY 8(f-1) + a-1/a-2 + the forward): per frame two windows of ~2 M raw DSEC-style events
    (uint16 x / y, rectification-map gather inside K1) -> two 5-bin grids -> merge -> K2 -> the C2 forward (graph replay).  Events, map
    and weights resident in HBM; the assembly is eager launches (K1's grid size depends on the window's event count)."""
    import numpy as np
    import torch
    from bflow_amd.dsec import EventStream, TwoStepAssembler
    bins = cfg["num_bins"]["correlation"]
    rs = np.random.RandomState(21)
    n_frames = 16                                                   # consecutive 100-ms frames (3 warm-ups + `steps` timed ones below)
    span_us = 60_000 + 100_000 * n_frames
    n = 20 * span_us                                                # 20 M events/s: 2 M per 100-ms interval, 3 M per extended window
    ev = dict(x=rs.randint(0, W, n, dtype=np.int32).astype(np.uint16), y=rs.randint(0, H, n, dtype=np.int32).astype(np.uint16),
              p=rs.randint(0, 2, n, dtype=np.int32).astype(np.uint8), t=np.sort(rs.randint(1_000_000, 1_000_000 + span_us, n)).astype(np.int64))
    yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    rect = np.stack([xx * 1.01 - 3 + np.sin(yy / 40.0), yy * 0.99 + 2 + np.cos(xx / 50.0)], -1).astype(np.float32)
    ts = np.array([[1_030_000 + 100_000 * k, 1_130_000 + 100_000 * k] for k in range(n_frames)], dtype=np.int64)
    assert steps + 3 + 1 <= n_frames
    stream = EventStream(**ev, device=dev)
    asm = TwoStepAssembler(bins, H, W, rect, device=dev)

    from bflow_amd.pipeline import EventFrameGraph, EventFramePipeline
    pipe = EventFramePipeline(model, asm, ITERS)
    n_win = max(asm.window_descriptor(stream, int(a), int(b))[1] for a, b in ts)
    graph_pipe = EventFrameGraph(model, asm, stream, ITERS, max_events=n_win + n_win // 8)
    graph_full = EventFrameGraph(model, asm, stream, ITERS, max_events=n_win + n_win // 8, reuse_windows=False)
    graph_over = EventFrameGraph(model, asm, stream, ITERS, max_events=n_win + n_win // 8, overlap=True)
    cursor = {"k": 0}

    def frame_graph():
        # the STREAM: consecutive frames 1, 2, 3 ...; ONE replay per frame = its assembly (device-side windows; the previous window's grid is
        # the one the frame before built: one K1 per frame) + its forward
        cursor["k"] += 1
        with torch.inference_mode():
            return graph_pipe(ts, cursor["k"])

    def frame_graph_full():
        with torch.inference_mode():       # the same replay with both windows assembled every frame (any frame order)
            return graph_full(ts, 1)

    def frame_graph_overlap():
        with torch.inference_mode():       # ONE replay per frame: the forward of the previous submission | the assembly of this one
            return graph_over.submit(ts, 1)

    def frame():
        with torch.inference_mode():       # (the call val.py makes: the forward is a graph replay; the assembly of the NEXT frame is eager, on a second stream)
            return pipe(stream, ts, 1)

    def frame_serial():
        with torch.inference_mode():
            vox = asm.assemble(stream, ts, 1, check=False)
            return model(voxel_grid=vox[None], iters=ITERS, test_mode=True)

    def assemble_only():
        with torch.inference_mode():
            return asm.assemble(stream, ts, 1, check=False)

    def timed(fn, k):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / k

    t_graph = timed(frame_graph, steps)
    k1_sets = graph_pipe.k1_launch_sets
    graph_pipe.close()
    t_full = timed(frame_graph_full, steps)
    graph_full.close()
    t_over = timed(frame_graph_overlap, steps)
    graph_over.flush()
    graph_over.close()
    t_frame, t_ser, t_asm = timed(frame, steps), timed(frame_serial, steps), timed(assemble_only, steps)
    return {"value": round(1.0 / t_graph, 2), "unit": "frames/s", "ms_per_frame": round(t_graph * 1e3, 4), "ms_assembly": round(t_asm * 1e3, 4),
            "k1_launch_sets_per_frame": round(k1_sets / (steps + 3), 3),
            "graph_both_windows": {"value": round(1.0 / t_full, 2), "ms_per_frame": round(t_full * 1e3, 4),
                                   "note": "EventFrameGraph(reuse_windows=False): both windows assembled in every replay (2 x K1), as any frame order needs"},
            "graph_branch": {"value": round(1.0 / t_over, 2), "ms_per_frame": round(t_over * 1e3, 4),
                             "note": "EventFrameGraph(overlap=True): the assembly of frame k + 1 as a branch of frame k's graph, next to its GRU loop -- K1's "
                                     "chip-filling launches take the wave slots the loop's dependent launches wait for"},
            "two_streams_eager": {"value": round(1.0 / t_frame, 2), "ms_per_frame": round(t_frame * 1e3, 4),
                                  "note": "round 6 first form (EventFramePipeline): the assembly as eager launches on a second stream -- eager launches do not run "
                                          "next to a graph replay on this runtime"},
            "one_stream": {"value": round(1.0 / t_ser, 2), "ms_per_frame": round(t_ser * 1e3, 4),
                           "note": "assembly and forward of a frame one after the other on one stream (round 5's number)"},
            "events_per_window": int(n_win), "planned_max_events_per_window": int(graph_pipe.max_events), "steps": steps,
            "workload": "a stream of frames of one resident recording: raw events -> 2 x K1 (rectified, 5 bins, windows read from a device descriptor) -> merge + "
                        "K2 (one launch pair) -> C2 forward (12 iters); ONE hipGraph replay per frame = its assembly, then its forward (bflow_amd/pipeline.py "
                        "EventFrameGraph; outputs bit-identical to eager assemble-then-forward); the frames are CONSECUTIVE 100-ms steps of a 1.66-s recording, "
                        "so the previous window of a frame is the current window of the frame before and its grid is reused (one K1 per frame; the "
                        "reference caches the same per-window grids on disk, base.py:93-104); wall clock incl. the host's window search and descriptor copy; "
                        "the other entries assemble frame 1 over and over (both windows)"}


def other_baseline_configs(dev, steps):
    """Frames/s of the BASELINE configurations `value` is NOT quoted on (parity-tested at full size in tests/test_hip_parity.py): configs[2]
    (C3: events + images, batch 8), configs[4] (C5: 1024 x 1024, degree 10, 20 iterations, f16/w correlation) and the per-rank shape of
    configs[3] at N = 8 (8 frames as two micro-batches of 4 in flight).  Synthetic inputs, deterministic random-init weights, graph replay."""
    import torch
    import bflow_amd
    from bflow_amd import configs, synthetic
    from bflow_amd.pipeline import ConcurrentRunner
    from bflow_amd.weights import deterministic_state_dict
    out = {}

    def run(fn, frames, k):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k):
            fn()
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / k
        return {"value": round(frames / el, 2), "unit": "frames/s", "ms_per_step": round(el * 1e3, 3), "frames_per_step": frames, "steps": k}

    for key, idx in (("c3_batch8", 2), ("c5", 4)):
        e = configs.baseline_config(idx)
        cfg = e["model"]
        m = bflow_amd.RAFTSpline(cfg).eval()
        m.load_state_dict(deterministic_state_dict(m, seed=0))
        m.to(dev)
        m.enable_hipgraph()
        C = cfg["num_bins"]["context"] + cfg["num_bins"]["correlation"] - 1
        B, hh, ww = e["batch"], e["height"], e["width"]
        vox = torch.from_numpy(synthetic.voxel_grid(B, C, hh, ww, seed=7)).to(dev)
        imgs = None
        if cfg["use_boundary_images"]:
            a, b = synthetic.image_pair(B, hh, ww, seed=8)
            imgs = [torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)]
        r = run(lambda: m(voxel_grid=vox, images=imgs, iters=e["iters"], test_mode=True), B, steps if idx == 2 else max(3, steps // 2))
        r["workload"] = f"BASELINE configs[{idx}]: raft-spline {e['experiment']}, {C}x{hh}x{ww} voxel grid" + (" + 2 RGB images" if imgs else "") + \
                        f", batch {B}, {e['iters']} iters" + (f", correlation.precision = {cfg['correlation'].get('precision')}" if cfg["correlation"].get("precision") else "")
        out[key] = r
        del m, vox, imgs
        torch.cuda.empty_cache()
    cfg = configs.model_config(CFG)
    m = bflow_amd.RAFTSpline(cfg).eval()
    m.load_state_dict(deterministic_state_dict(m, seed=0))
    m.to(dev)
    m.enable_hipgraph()
    pair = ConcurrentRunner(m, ITERS, streams=2)
    v = [torch.from_numpy(synthetic.voxel_grid(4, 9, H, W, seed=1234, first_sample=4 * k)).to(dev) for k in range(2)]
    r = run(lambda: pair(v), 8, steps)
    r["workload"] = "what ONE rank runs per step of configs[3] at N = 8: its 8 frames as two micro-batches of 4, both in flight in one graph"
    out["c4_rank_shape_at_n8"] = r
    return out


if __name__ == "__main__":
    main()
