#!/usr/bin/env python
"""Regenerates profiles/README.md from profiles/r04_bench.json + r04_pmc.json (tools only)."""
import json, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
b = json.load(open(os.path.join(P, "r04_bench.json")))
pm = json.load(open(os.path.join(P, "r04_pmc.json")))
p = pm["kernels"]
r, re_, rk, rks, rl, rl4, rk5, cb = (b["roofline"], b["roofline_encoder"], b["roofline_corr_build"], b["roofline_corr_build_split"], b["roofline_lookup"],
                                    b["roofline_lookup_c4_shard"], b["roofline_corr_build_c5"], b["cpu_baseline"])
rf, ru, vs = b["roofline_frame"], b["roofline_update_iter"], b["value_split"]
one = cb["single_thread"]
gs = b["gpu_stage_ms"]
vk = b["voxel_kernels"]
train = " / ".join(l.strip() for l in open(os.path.join(P, "r04_train_probe.txt")).read().strip().replace("train step ", "").splitlines())
tl = " · ".join(" ".join(l.split()) for l in open(os.path.join(P, "r04_stamp_timeline.txt")).read().strip().splitlines())


def mf(k):
    m = p.get(k["kernel"], {}).get("mfma", {})
    return f'{m.get("mfma_utilisation", 0):.2f}' if m else "—"


def tr(k):
    e = p.get(k["kernel"])
    if not e or e.get("traffic") is None:
        return "—"
    return f'{e["traffic"]/1e6:.1f} MB vs {e["algorithmic_bytes"]/1e6:.1f} MB (reads {e["fetch_corrected"]/1e6:.1f}, writes {e["write"]/1e6:.1f})'


txt = f'''# profiles/ — measured evidence, MI355X (gfx950)

Round-4 files (`r04_*`) were produced on a `gpurun` MI355X box by `tools/collect_profiles.sh` (the only writer of these files) and this
page by `tools/make_profiles_readme.py`; `r01_*` / `r02_*` are the earlier rounds, kept for comparison.  `gpurun_out/` is scratch, these are
the copies to be judged.  `r04_pmc.json` records a hash of the kernel sources it was collected on; `bench.py` quotes its `traffic` only
while the sources still hash to the same value.

| File | Command | What it shows |
|---|---|---|
| `r04_bench.json` | `python bench.py --steps 30 --warmup 5` | the BENCH line: {b["value"]:.1f} frames/s, {b["ms_per_step"]:.2f} ms/frame, {b["ms_per_gru_iter"]:.3f} ms per GRU iteration at BASELINE configs[1] (E_LU4_BD2, 640×480, B=1, 12 iters); `c4_strong` = configs[3] (global batch 64 in micro-batches of 8) on one GPU: {b["c4_strong"]["value"]:.1f} frames/s; rooflines; `cpu_baseline` (all cores + 1 thread, per-stage ms); `gpu_stage_ms`; `voxel_kernels` (K1 / K2) |
| `r04_rocprofv3_kernel_stats_c2only.csv` | `rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 30 --warmup 5 --no-extras` (top rows) | **batch-1 C2 only** (38 forwards): per-kernel totals — the frame's budget per kernel; every row is a kernel of this repository or torch's copy / fill plumbing |
| `r04_rocprofv3_kernel_stats.csv` | the same on the default command (`--no-cpu-baseline`) | both workloads of the bench (batch 1 and batch 8): use the C2-only file for averages |
| `r04_pmc_{{FETCH_SIZE,WRITE_SIZE,MFMA}}_<key>.csv` | `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` / `--pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE` (separate passes) on `tools/roofline_probe.py --key <key>` — the launchers `bench.py` times; last 5 launches; `..._calib.csv`: the same SQ / GRBM set on `tools/micro/fp8_cross` (pure MFMA streams: the 100 % mark) | fabric traffic and matrix-core busy cycles per launch of the three roofline kernels |
| `r04_pmc.json` | `tools/pmc_to_json.py` on those CSVs | bytes per launch incl. the gfx950 ×2 FETCH_SIZE correction (MI355X_MICROARCH.md §HBM), MFMA utilisation = busy / GRBM cycles relative to the calibration launch, the kernel-source hash |
| `r04_mfma_clock_fp8_cross.txt` | `tools/micro/fp8_cross` | (1) the fp8 K = 64 MFMA with the operand halves K5 uses is exact on realistic operands; (2) pure MFMA streams on random data, every SIMD busy: 3-pass split vs fp16 + fp8 cross terms — time per 32-channel block and the shader clock (`s_memtime` ÷ wall clock): the power-limited clock of the matrix pipes |
| `r04_store_patterns.txt` | `tools/micro/store_patterns` | a pure store stream of K5's shape: 4 B vs 16 B per lane, K5's item order vs lockstep, `nt`; linear fills; `hipMemset` — the write ceiling (≈ 0.6 of 8 TB/s) and that store width does not move it |
| `r04_k5_stamps_split8.txt`, `r04_k5_stamps_split.txt` | `BFLOW_HIP_LIB=…/libbflow_hip_stamps.so python tools/k5_probe.py --time-only --stamps --stamp-mode <mode>` (`tools/k5_ablate.sh stamps:-DSTREAM_STAMPS`) | per-workgroup cycle stamps of K5 at C2: prologue, cycles per 64-row pair, end times per XCD, sustained clock = cycles ÷ `s_memrealtime` |
| `r04_smi_roofline.txt`, `r04_smi_roofline_corr_build.txt` | `amd-smi metric --clock --power` every 0.25 s while `tools/roofline_probe.py --key <key> --reps 20000` repeats the launch | sclk and socket power under the layer-1 convolution / K5 (`r04_power_limit.txt`: `amd-smi static --limit`, the board's power cap) |
| `r04_k5_modes.txt` | `python tools/k5_modes_probe.py --big` | K5 per arithmetic / storage (`split`, `split8`, `f16/w`, `split/h`, `split8/h`, `f16`): error vs an fp64 GEMM and duration at C2 / C4 shard / C5 |
| `r04_corr_precision_e2e.txt` | `python tools/corr_precision_probe.py --c5` | end-to-end EPE vs the fp32 oracle and frame time per `corr_precision` at C2 (two inputs) and C5: the decomposition of the fp16 variant's error |
| `r04_lookup_conv_probe.txt` | `python tools/lookup_conv_probe.py --shapes c2,c4`, `... --stamps` on the `tools/lookup_conv_stamps.sh` build, `BFLOW_LOOKUP_CONV=1 python bench.py` vs default | the fused look-up + convc1 launch (opt-in): in-graph duration against the two separate launches, per-wave cycle stamps of its phases, and the A/B inside the captured forward (DESIGN.md §8 item 6) |
| `r04_gru_conv_probe.txt` | `python tools/gru_conv_probe.py` | a batch-1 GRU gate convolution: plain fp32 output vs fused gate epilogue, full input [h \| M] vs one half — the numbers behind the input-split experiment (DESIGN.md §8 item 6) |
| `r04_halo12_ab.txt`, `r04_spread_dma_ab.txt`, `r04_thin_head.txt`, `r04_half_tile_ab.txt`, `r04_nt_stores_ab.txt` | `tools/r04_h12.sh`, `tools/r04_thin.sh`, `tools/r04_thin2.sh`, `tools/r04_ht.sh`, `tools/r04_nt.sh`, `tools/r04_nt2.sh` (alternating runs on one box each) | this round's A/B experiments (DESIGN.md §8): the 12-wave small-grid kernel, LDS-DMA pieces spread over a step's taps, the thin head on the matrix cores (2×16 vs 2×10 patches vs the vector-ALU kernel), the half-tile variant of the encoder's 96-channel layers, non-temporal stores (look-up: adopted; conv epilogues / encoder: neutral) |
| `r04_iteration_launches.txt` | `tools/trace_iteration.py` on the `--kernel-trace` of `bench.py --no-extras` | one steady-state update iteration launch by launch: kernel, queue, workgroups, threads, µs — the two-queue form of rounds 2–3 (`BFLOW_NO_ONE_QUEUE=1` now) |
| `r04_encoder_conv_probes.txt` | `tools/enc_conv_probe.py` with probe builds of `conv_split.hip` (flags quoted in the file; not in the tree) | the encoder's 3×3 convolutions alone, product vs a third fewer LDS fragment reads vs no weight streaming: −0…4 % / −5…10 % (DESIGN §8 item 14) |
| `r04_frame_encoder_launches.txt`, `r04_frame_tail_launches.txt` | `tools/trace_frame.py <kernel_trace.csv> encoder\|tail` (`tools/r04_frame.sh`, `tools/r04_tail.sh`) | every launch of one steady-state frame OUTSIDE the update loop: encoders + K5 + pooling on the two queues (before item 12's trims), and the last iteration + mask head + up-sampling (after them) |
| `r04_iteration_launches_one_queue.txt` (`tools/collect_profiles.sh`: the product), `r04_one_queue_pairs_ab.txt` | `tools/r04_pair.sh` | the same iteration as ten launches on ONE queue (look-up ‖ im2col rider, `conv_split_pair_kernel`, `conv_halo8_pair_kernel`: DESIGN §8 item 11) and its alternating same-box A/B against the side-stream form (−1.0…−1.5 % per frame), the 10×16 pair variant and the rider placement |
| `r04_k5_balanced_split.txt` | `tools/r04_k5bal.sh` on the variant with a chip-wide equal work split (not in the tree) | K5: balanced split vs the lockstep split — durations per arithmetic, per-workgroup cycle stamps, bench A/B: slower |
| `r04_stem_norm_in_ab.txt`, `r04_residual_epilogue_ab.txt`, `r04_halo_occupancy_probe.txt`, `r04_thin_head_crossover.txt` | `tools/r04_stemnin.sh`, `tools/r04_res.sh`, `tools/r04_occ.sh`, `tools/r04_thinmax.sh` | encoder launch / traffic reductions that paid (the stem's norm never materialised: −0.65 %; the context encoder's `relu(x + y)` as conv2's epilogue: −1.3 %), the occupancy probe of the halo kernel (1 vs 2 workgroups per CU), the thin-head cross-over at batch 8 (neutral) |
| `r04_k7_ablation.txt` | `BFLOW_LOOKUP_ABL=<bits> python tools/k7_abl_probe.py` | K7 with phases switched off (timing only): the phases add up to the total at C2 and on the C4 shard |
| `r04_stamp_timeline.txt` | `python tools/stamp_timeline.py` | stage boundaries INSIDE the captured graph (no tracer): {tl} |
| `r04_train_probe.txt` | `BFLOW_TRAIN_PROBE_GRAPH=1 python tools/train_probe.py 10` | training path (SURVEY §8 f-4), unchanged this round, re-measured for regressions: {train} |

## Round-4 numbers (C2 = E_LU4_BD2 events-only, DSEC 640×480, batch 1, 12 iterations)

| Quantity | Round 4 | Round 3 | Round 2 | Round 1 |
|---|---|---|---|---|
| frames/s, 1 GPU, hipGraph replay (`value`: ONE workload at every N) | **{b["value"]:.1f}** ({b["ms_per_step"]:.2f} ms/frame) | 271.4–276.6 | 244.3 (4.09 ms) | 236.8 (4.22 ms) |
| the same frame with the correlation on three fp16 passes (`value_split`: fp32 class everywhere) | {vs["value"]:.1f} ({vs["ms_per_step"]:.2f} ms) | — | — | — |
| ms per GRU iteration (marginal, under replay) | **{b["ms_per_gru_iter"]:.3f}** | 0.159–0.163 | 0.171 | 0.181–0.186 |
| fixed part (encoders + volume + pyramid + up-sampling) | {b["ms_fixed_part"]:.2f} ms | 1.66–1.74 ms | 2.06 ms | 1.94–2.1 ms |
| whole frame vs the split format's matrix peak (`roofline_frame`: {rf["flop_per_frame"]/1e9:.0f} GFLOP as executed) | {rf["achieved"]:.0f} TFLOP/s-equiv. = **{rf["frac"]:.3f}** of 833 | 0.19 (judge's figure) | | |
| one update iteration vs the same peak (`roofline_update_iter`: {ru["flop_per_iteration"]/1e9:.1f} GFLOP) | {ru["achieved"]:.0f} TFLOP/s-equiv. = **{ru["frac"]:.3f}** | 0.17 | | |
| configs[3] global batch 64 on ONE GPU (8 micro-batches of 8, two at a time as parallel branches of one graph) | {b["c4_strong"]["value"]:.1f} frames/s | 408–422 | 363 | (batch 8: 300–314) |
| two batch-1 frames in flight (`c2_two_in_flight`, next to `value`, never `value`) | {b["c2_two_in_flight"]["value"]:.1f} frames/s | 322–328 | 290 | — |
| CPU baseline (oracle = op-for-op port, torch CPU fp32), {cb["cores"]} threads / 1 thread | {cb["value"]:.2f} frames/s ({cb["ms_per_frame"]:.0f} ms/frame) / {one["value"]:.3f} frames/s ({one["ms_per_frame"]:.0f} ms/frame) | | 0.77 | 0.75 |

| Kernel (as `bench.py` launches it) | bound | achieved | peak | frac | launch | PMC traffic vs algorithmic | MFMA utilisation (PMC) |
|---|---|---|---|---|---|---|---|
| `roofline`: `conv_halo8_pair_kernel<3,3>` on convc2 ‖ convf2 (3×3, 256→192 and 128→64, 1×60×80, one launch) — the small-grid 3×3 family has the largest total time of the frame | fp16 MFMA / 3 | {r["achieved"]:.0f} TFLOP/s-equiv. | 833 | **{r["frac"]:.2f}** | {r["avg_launch_ms"]*1e3:.1f} µs | {tr(r)} | {mf(r)} |
| `roofline_encoder`: `conv_halo_kernel<2,3,3,TR>` encoder layer1 3×3 (rounds 1–3 reported this one as `roofline`) | fp16 MFMA / 3 | {re_["achieved"]:.0f} TFLOP/s-equiv. | 833 | **{re_["frac"]:.2f}** (round 3: 0.36, round 2: 0.31) | {re_["avg_launch_ms"]*1e3:.0f} µs | {tr(re_)} | {mf(re_)} |
| `roofline_corr_build`: K5, the product launch (split8: hi·hi fp16 + fp8 cross terms, tiled planes) | HBM (2nd roof: matrix, 2 units / product) | {rk["achieved"]/1e3:.2f} TB/s; {rk.get("tflops_equivalent", 0):.0f} TFLOP/s-equiv. | 8 TB/s; 1250 | **{rk["frac"]:.2f}**; {rk.get("frac_of_store_ceiling", 0):.2f} of the best pure store stream measured in the same process ({rk.get("store_ceiling_gbs", 0)/1e3:.2f} TB/s); matrix {rk.get("frac_mfma", 0):.2f} | {rk["avg_launch_ms"]*1e3:.0f} µs | {tr(rk)} | {mf(rk)} |
| `roofline_corr_build_split`: K5 with three fp16 passes (the fp32-class number) | HBM / matrix (3 units / product) | {rks["achieved"]/1e3:.2f} TB/s | 8 TB/s; 833 | **{rks["frac"]:.2f}**; matrix {rks.get("frac_mfma", 0):.2f} | {rks["avg_launch_ms"]*1e3:.0f} µs | {tr(rks)} | {mf(rks)} |
| `roofline_corr_build_c5`: K5 at BASELINE configs[4] (1024², 6 targets) with the arithmetic that config selects (`f16/w`) | HBM | {rk5["achieved"]/1e3:.2f} TB/s | 8 | **{rk5["frac"]:.2f}** | {rk5["avg_launch_ms"]*1e3:.0f} µs | — | — |
| `roofline_lookup`: K7 at C2 (batch 1) | HBM (gather) | {rl["achieved"]/1e3:.2f} TB/s | 8 | **{rl["frac"]:.2f}** algorithmic; {rl.get("frac_of_line_granular_cap", 0):.2f} at 128-B line granularity ({rl.get("line_bytes", 0)/1e6:.1f} MB) | {rl["avg_launch_ms"]*1e3:.1f} µs | {tr(rl)} | — |
| `roofline_lookup_c4_shard`: K7 on C4's per-GPU shard (batch 8) | HBM (gather) | {rl4["achieved"]/1e3:.2f} TB/s | 8 | **{rl4["frac"]:.2f}**; {rl4.get("frac_of_line_granular_cap", 0):.2f} at line granularity | {rl4["avg_launch_ms"]*1e3:.1f} µs | — | — |

Per-stage milliseconds under the reference's CudaTimer names (`raft.py:116-186`):

| stage | HIP path (eager + hipEvents: upper bounds) | CPU oracle, {cb["cores"]} threads | CPU oracle, 1 thread |
|---|---|---|---|
''' + "".join(f'| `{k}` | {gs.get(k, "—")} | {cb["stage_ms"].get(k, "—")} | {one["stage_ms"].get(k, "—")} |\n'
              for k in ("fnet_ev", "cnet", "corr computation", "all iters", "1 iter", "get_flow (per iter)", "corr lookup (per iter)", "update (per iter)")) + f'''
K1 / K2 (`voxel_kernels`, 2 M synthetic events into the 15 × 480 × 640 grid): float x/y {vk["k1_float_xy"]["ms"]:.3f} ms = {vk["k1_float_xy"]["events_per_s"]/1e9:.2f} G events/s
({vk["k1_float_xy"]["atomics_per_s"]/1e9:.1f} G atomics/s, {vk["k1_float_xy"]["algorithmic_gb_s"]:.0f} GB/s algorithmic); int x/y {vk["k1_int_xy"]["ms"]:.3f} ms = {vk["k1_int_xy"]["events_per_s"]/1e9:.2f} G events/s;
K2 {vk["k2_norm"]["ms"]*1e3:.0f} µs ({vk["k2_norm"]["algorithmic_gb_s"]:.0f} GB/s over its four passes).

Other probes (numbers quoted in DESIGN.md §8 / §9): `tools/k5_probe.py [--big] [--f16] [--stamps]`, `tools/k5_ablate.sh` (timing-only ablation
builds of K5), `tools/k7_probe.py`, `tools/micro/load_paths.hip` (LDS-DMA vs register loads per CU), `tools/c5_check.py [--f16]`.
'''
open(os.path.join(P, "README.md"), "w").write(txt)
print(txt)
