#!/usr/bin/env python
"""Regenerates profiles/README.md from profiles/r05_bench.json + r05_pmc.json (tools only)."""
import json, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
b = json.load(open(os.path.join(P, "r05_bench.json")))
pm = json.load(open(os.path.join(P, "r05_pmc.json")))
p = pm["kernels"]
S = b["summary"]
cb, one, gs, vk = b["cpu_baseline"], b["cpu_baseline"]["single_thread"], b["gpu_stage_ms"], b["voxel_kernels"]


def rd(name):
    f = os.path.join(P, name)
    return open(f).read().strip() if os.path.exists(f) else ""


tl = " · ".join(" ".join(l.split()) for l in rd("r05_stamp_timeline.txt").splitlines() if " us " in l)
train = " / ".join(l.strip() for l in rd("r05_train_probe.txt").replace("train step ", "").splitlines())


def mf(k):
    m = p.get(k["kernel"], {}).get("mfma", {})
    return f'{m["mfma_utilisation"]:.2f}' if m.get("mfma_utilisation") is not None else "—"


def tr(k):
    e = p.get(k["kernel"])
    if not e or e.get("traffic") is None:
        return "—"
    return f'{e["traffic"]/1e6:.1f} MB vs {e["algorithmic_bytes"]/1e6:.1f} MB (reads {e["fetch_corrected"]/1e6:.1f}, writes {e["write"]/1e6:.1f})'


def row(key, what, bound):
    k = b[key]
    unit = "TFLOP/s-equiv." if k["bound"] == "mfma" else "GB/s"
    extra = ""
    if k.get("frac_of_line_granular_cap"):
        extra = f'; {k["frac_of_line_granular_cap"]:.2f} at 128-B line granularity'
    if k.get("model_cap"):
        extra += f'; cap of this arithmetic {k["model_cap"]["frac"]:.2f} (matrix {k["model_cap"]["t_mfma_us"]} us at {k["model_cap"]["sustained_clock_ghz"]} GHz, stores {k["model_cap"]["t_store_us"]} us)'
    if k.get("frac_mfma"):
        extra += f'; matrix {k["frac_mfma"]:.2f}'
    return f'| `{key}`: {what} | {bound} | {k["achieved"]:.0f} {unit} | **{k["frac"]:.3f}**{extra} | {k["avg_launch_ms"]*1e3:.1f} µs | {tr(k)} | {mf(k)} |\n'


txt = f'''# profiles/ — measured evidence, MI355X (gfx950)

Round-5 files (`r05_*`): `tools/collect_profiles.sh` on one `gpurun` MI355X box writes the bench line, the rocprofv3 summaries, the PMC passes and
the probes listed first; the A/B files further down were written by the commands quoted inside them while the kernels were developed (each on
ONE box, alternating).  This page: `tools/make_profiles_readme.py`.  `r01_*` … `r04_*` are the earlier rounds (index: `README_r04.md`).
`r05_pmc.json` records a hash of the kernel sources it was collected on; `bench.py` quotes its `traffic` only while the sources still hash to
the same value.

| File | Command | What it shows |
|---|---|---|
| `r05_bench.json` | `python bench.py --steps 30 --warmup 5` | the BENCH line: {b["value"]:.1f} frames/s, {b["ms_per_step"]:.2f} ms/frame, {b["ms_per_gru_iter"]:.3f} ms per GRU iteration at BASELINE configs[1]; `c4_strong` {S["c4_strong"]} frames/s; `c3_batch8`, `c5`, `c4_rank_shape_at_n8`, `pipeline_from_events`; rooflines; `cpu_baseline`; `gpu_stage_ms` (in-graph stamps); `voxel_kernels`; `summary` (the scalars, last) |
| `r05_rocprofv3_kernel_stats_c2only.csv` | `rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 30 --warmup 5 --no-extras` (top rows) | batch-1 C2 only: the frame's budget per kernel.  `roofline` = the kernel with the largest total here (`conv_halo_stream_kernel<true>`: the encoder's persistent 3×3, normalise-on-load form; `bench.py` times its plain form on the layer-1 launch) |
| `r05_rocprofv3_kernel_stats.csv` | the same on the default command (`--no-cpu-baseline`) | every workload of the bench line (batch 1, batch 8, C3, C5, K1 …) |
| `r05_iteration_launches.txt`, `r05_frame_encoder_launches.txt`, `r05_frame_tail_launches.txt` | `tools/trace_iteration.py` / `tools/trace_frame.py` on the C2-only kernel trace | one steady-state update iteration, the encoder phase and the frame's tail launch by launch (the tracer serialises the two queues: use `r05_stamp_timeline.txt` for overlap) |
| `r05_stamp_timeline.txt` | `python tools/stamp_timeline.py` | stage boundaries INSIDE the captured graph (`bflow_clock_stamp`, no tracer): {tl} |
| `r05_pmc_{{FETCH_SIZE,WRITE_SIZE,MFMA}}_<key>.csv`, `r05_pmc.json` | `rocprofv3 --pmc …` (separate passes) on `tools/roofline_probe.py --key <key>`; `tools/pmc_to_json.py` | fabric traffic per launch (gfx950 ×2 FETCH_SIZE correction) and matrix-core utilisation = `SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES` relative to the calibration launch (`tools/micro/fp8_cross`; round 4 normalised by `GRBM_GUI_ACTIVE`, which spans more than a short kernel — kept as `mfma_utilisation_over_gui_active`) |
| `r05_k1_probe.txt` | `python tools/k1_probe.py`; `rocprofv3 --kernel-trace --stats -- python tools/k1_probe.py 0 f` | K1 (tile-binned, LDS fixed-point, deterministic): whole calls per grid / event count, and the four kernels of one call |
| `r05_enc_stream_probe.txt`, `r05_enc_stream_ab.txt`, `r05_enc_stream_ablation.txt`, `r05_stream_frame_ab.txt` | `tools/enc_stream_probe.py` (`ENC_PROBE_NIN=1`), `tools/enc_stream_ablate.sh`, `tools/ab_bench.sh "BFLOW_CONV_STREAM=0" "BFLOW_CONV_STREAM=1"` | the encoder's persistent 3×3 kernel against the per-item kernel per shape (plain / normalise-on-load), its anatomy (timing-only ablation builds) and the whole-frame A/B |
| `r05_k7_tp_probe.txt`, `r05_k7_tp_sweep.txt` | `BFLOW_LOOKUP_TP=<2,4,8> python tools/k7_probe.py --shapes c2,c4` | K7 per pixels-per-workgroup: fewer, larger workgroups are slower — why the head + look-up fusion was not built |
| `r05_enc_stream_clock.txt` | `tools/enc_stream_clock.py` on `-DH8_STAMPS` builds (`tools/build_flag_variant.sh st_<x> "-DH8_STAMPS [-DCSTREAM_ABL=n]" conv_split.hip`) | **the persistent kernel is power-limited**: cycles AND clock per workgroup (`s_memtime` / `s_memrealtime`) — 0.75 of the matrix pipes busy in cycles at 1.31–1.38 GHz; fragment reads cost clock, LDS-DMA and stores cost cycles; older vs younger workgroup of a CU; one workgroup per CU; uneven ranges; the predictive ablations 6 / 7 (what a 64×64 wave tile / one weight tile per CU are worth) |
| `r05_enc_stream_pin_prio.txt`, `r05_enc_stream_fastpath.txt` | `tools/enc_stream_probe.py` on `-DCSTREAM_PIN=1` / `-DCSTREAM_PRIO=1` builds; before / after the interior-patch halo offsets | scheduling barriers, `s_setprio` and 35 % fewer instructions on the halo step: all neutral — issue slots are not the limit |
| `r05_k7_instruction_diet.txt`, `r05_k7_trimmed_gather.txt` | `tools/k7_probe.py` alternating with the previous library; `tools/micro/gather_lines` | K7 is instruction-bound at batch 8: per-pair gather records −4 % at every shape; a pure gather of the kernel's shape runs at 6.1–6.9 TB/s; fetching 26 % fewer lines is neutral |
| `r05_flag_edge.txt` | `tools/experiments/r05_edge_probe.py` on the tree + `tools/experiments/r05_flag_edge.patch` | the flag-synchronised edge (q convolution → next z\|r convolution as ONE launch with per-patch flags): parity-checked, 41.5 µs (65.6 with agent-scope fences) against 32.8 µs for the product's two launches — not kept |
| `r05_stream_share_frame_ab.txt` | `tools/ab_bench.sh "BFLOW_CONV_STREAM_SHARE=50" "BFLOW_CONV_STREAM_SHARE=57" 3` | older : younger range lengths of the persistent encoder kernel in the FRAME: no gain (the launch alone: −2…−3 %) — default 50 : 50 |
| `r05_mfma_clock_fp8_cross.txt`, `r05_store_patterns.txt`, `r05_k5_stamps_split8.txt`, `r05_k5_stamps_split.txt`, `r05_k5_modes.txt`, `r05_corr_precision_e2e.txt` | as in round 4 (`README_r04.md`) | K5 is unchanged this round: re-collected on this round's box (the sustained clock `bench.py` uses for `roofline_corr_build.model_cap` comes from the stamps) |
| `r05_train_probe.txt` | `BFLOW_TRAIN_PROBE_GRAPH=1 python tools/train_probe.py 10` | training path (SURVEY §8 f-4), unchanged, re-measured for regressions: {train} |

## Round-5 numbers (C2 = E_LU4_BD2 events-only, DSEC 640×480, batch 1, 12 iterations)

| Quantity | Round 5 | Round 4 | Round 3 | Round 2 | Round 1 |
|---|---|---|---|---|---|
| frames/s, 1 GPU, hipGraph replay (`value`) | **{b["value"]:.1f}** ({b["ms_per_step"]:.2f} ms/frame) | 289.8 (driver) / 298.9 | 271.4–276.6 | 244.3 | 236.8 |
| the same frame with the correlation on three fp16 passes (`value_split`) | {S["value_split"]} | 288.9 | — | — | — |
| ms per GRU iteration (marginal, under replay) | **{b["ms_per_gru_iter"]:.3f}** | 0.148–0.150 | 0.159–0.163 | 0.171 | 0.181–0.186 |
| fixed part (encoders + volume + pyramid + up-sampling) | {b["ms_fixed_part"]:.2f} ms | 1.55 ms | 1.66–1.74 | 2.06 | 1.94–2.1 |
| configs[3] global batch 64 on ONE GPU (`c4_strong`) | {S["c4_strong"]} frames/s | 433.7 | 408–422 | 363 | — |
| per-rank shape of configs[3] at N = 8 (8 frames as 2 × 4 in flight) | {S["c4_rank_shape_at_n8"]} frames/s | — | | | |
| configs[2] (events + images, batch 8) / configs[4] (1024², degree 10, 20 iterations) | {S["c3_batch8"]} / {S["c5"]} frames/s | — / ≈ 48 | | | |
| raw events → 2 × K1 → merge → K2 → forward (`pipeline_from_events`) | {S["pipeline_from_events"]} frames/s | — (K1 alone: 2 × 0.77 ms) | | | |
| two batch-1 frames in flight (`c2_two_in_flight`, never `value`) | {S["c2_two_in_flight"]} frames/s | 330.4 | 322–328 | 290 | — |
| K1, 2 M float-xy events into 15 × 480 × 640 | **{vk["k1_float_xy"]["ms"]*1e3:.0f} µs** = {vk["k1_float_xy"]["frac"]:.3f} of 8 TB/s, bit-identical run to run | 769 µs = 0.029, atomics | | | |
| CPU baseline (oracle, torch CPU fp32), {cb["cores"]} threads / 1 thread | {cb["value"]:.2f} / {one["value"]:.3f} frames/s | 0.78–0.80 / 0.21–0.22 | | 0.77 | 0.75 |

| Kernel (as `bench.py` launches it) | bound | achieved | frac of the roof | launch | PMC traffic vs algorithmic | MFMA utilisation (PMC, SQ_BUSY-normalised) |
|---|---|---|---|---|---|---|
''' + row("roofline", "`conv_halo_stream_kernel<false>` on the encoder's layer-1 launch (64→64 3×3 on 5×240×320; rounds 1–4: `conv_halo_kernel`, 0.31–0.36)", "fp16 MFMA / 3") \
    + row("roofline_update_conv", "`conv_halo8_pair_kernel<3,3>` convc2 ‖ convf2 at batch 1 (round 4's `roofline`)", "fp16 MFMA / 3") \
    + row("roofline_corr_build", "K5, the product launch (split8)", "HBM (2nd roof: matrix)") \
    + row("roofline_corr_build_split", "K5 on three fp16 passes (fp32 class)", "HBM / matrix") \
    + row("roofline_corr_build_c5", "K5 at BASELINE configs[4] (`f16/w`)", "HBM") \
    + row("roofline_lookup", f'K7 at C2 on SURVEY 8(d)\'s 24.33 MB; the product launch with its im2col rider: {b["roofline_lookup"]["product_launch_with_rider"]["avg_launch_ms"]*1e3:.1f} µs (+{b["roofline_lookup"]["product_launch_with_rider"]["rider_extra_us"]} µs, {b["roofline_lookup"]["product_launch_with_rider"]["rider_bytes"]/1e6:.2f} MB)', "HBM (gather)") \
    + row("roofline_lookup_c4_shard", "K7 on C4's per-GPU shard (batch 8)", "HBM (gather)") + f'''
Per-stage milliseconds under the reference's CudaTimer names (`raft.py:116-186`); the HIP column is read from in-graph clock stamps of the replay `value` is measured on:

| stage | HIP path (in-graph stamps) | CPU oracle, {cb["cores"]} threads | CPU oracle, 1 thread |
|---|---|---|---|
''' + "".join(f'| `{k}` | {gs.get(k, "—")} | {cb["stage_ms"].get(k, "—")} | {one["stage_ms"].get(k, "—")} |\n'
              for k in ("fnet_ev", "cnet", "corr computation", "all iters", "1 iter", "corr lookup (per iter)", "update (per iter)")) + f'''
K1 / K2 (`voxel_kernels`, 2 M synthetic events into the 15 × 480 × 640 grid): float x/y {vk["k1_float_xy"]["ms"]*1e3:.0f} µs = {vk["k1_float_xy"]["events_per_s"]/1e9:.1f} G events/s
({vk["k1_float_xy"]["frac"]:.3f} of 8 TB/s on SURVEY 8(d)'s bytes); int x/y {vk["k1_int_xy"]["ms"]*1e3:.0f} µs ({vk["k1_int_xy"]["frac"]:.3f}); K2 {vk["k2_norm"]["ms"]*1e3:.0f} µs ({vk["k2_norm"]["frac"]:.3f}).
'''
open(os.path.join(P, "README.md"), "w").write(txt)
print(txt[:3000])
