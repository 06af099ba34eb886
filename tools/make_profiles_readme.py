#!/usr/bin/env python
"""Regenerates profiles/README.md from profiles/r02_bench.json + r02_pmc.json (tools only)."""
import json, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
b = json.load(open(os.path.join(ROOT, "profiles", "r02_bench.json")))
p = json.load(open(os.path.join(ROOT, "profiles", "r02_pmc.json")))["kernels"]
r, rk, rl, cb = b["roofline"], b["roofline_corr_build"], b["roofline_lookup"], b["cpu_baseline"]
train = " / ".join(l.strip() for l in open(os.path.join(ROOT, "profiles", "r02_train_probe.txt")).read().strip().replace("train step ", "").splitlines())
txt = f'''# profiles/ — measured evidence, MI355X (gfx950)

Round-2 files (`r02_*`) were produced on a `gpurun` MI355X box by `tools/collect_profiles.sh` (the only writer of these files) and this
page by `tools/make_profiles_readme.py`; round-1 files (`r01_*`) are kept for comparison.  `gpurun_out/` is scratch, these are the copies
to be judged.  `r02_pmc.json` records a hash of the kernel sources it was collected on; `bench.py` quotes its `traffic` only while the
sources still hash to the same value.

| File | Command | What it shows |
|---|---|---|
| `r02_bench.json` | `python bench.py --steps 30 --warmup 5` | the BENCH line: {b["value"]:.1f} frames/s, {b["ms_per_step"]:.2f} ms/frame, {b["ms_per_gru_iter"]:.3f} ms per GRU iteration at BASELINE configs[1] (E_LU4_BD2, 640×480, B=1, 12 iters); `c4_strong` = configs[3] (global batch 64 in micro-batches of 8) on one GPU: {b["c4_strong"]["value"]:.1f} frames/s; rooflines; CPU baseline {cb["value"]:.2f} frames/s on the box's {cb["cores"]}-core quota |
| `r02_rocprofv3_kernel_stats.csv` | `rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 30 --warmup 5 --no-cpu-baseline` (top 45 rows) | per-kernel totals/averages over the same command (both workloads of the bench: batch 1 and batch 8, so the averages mix the two): every row is a kernel of this repository or torch's copy / fill plumbing — no library convolution or GEMM |
| `r02_pmc_{{FETCH,WRITE}}_SIZE_<key>.csv` | `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes) on `tools/roofline_probe.py --key <key>` — the same launchers `bench.py` times (`tools/roofline_kernels.py`); last 5 launches of the kernel | fabric-side traffic per launch of the three roofline kernels |
| `r02_pmc.json` | `tools/pmc_to_json.py` on the six CSVs | bytes per launch incl. the gfx950 ×2 FETCH_SIZE correction for wide coalesced streams (MI355X_MICROARCH.md §HBM) + the kernel-source hash |
| `r02_train_probe.txt`, `r02_train_rocprofv3_kernel_stats.csv` | `BFLOW_TRAIN_PROBE_GRAPH=1 python tools/train_probe.py 10`; `BFLOW_TRAIN_PROBE_GRAPH=engine rocprofv3 --kernel-trace --stats -- python tools/train_probe.py 5` | training path (SURVEY §8 f-4) with the convolutions (forward, dgrad, wgrad) on the conv engine, eagerly (host-bound) and as one hipGraph per step (`training.GraphedTrainStep`), the latter also on torch / MIOpen convolutions (`conv_train.ENABLED = False`): {train} |

## Round-2 numbers (C2 = E_LU4_BD2 events-only, DSEC 640×480, batch 1, 12 iterations)

| Quantity | Round 2 | Round 1 |
|---|---|---|
| frames/s, 1 GPU, hipGraph replay | **{b["value"]:.1f}** ({b["ms_per_step"]:.2f} ms/frame) | 236.8 (4.22 ms) |
| ms per GRU iteration (marginal, under replay) | **{b["ms_per_gru_iter"]:.3f}** | 0.181–0.186 |
| fixed part (encoders + volume + pyramid + up-sampling) | {b["ms_fixed_part"]:.2f} ms | 1.94–2.1 ms |
| configs[3] global batch 64 on ONE GPU (8 micro-batches of 8, two at a time as parallel branches of one graph) | {b["c4_strong"]["value"]:.1f} frames/s | (batch 8: 300–314) |
| two batch-1 frames in flight (`c2_two_in_flight`, next to `value`, never `value`) | {b["c2_two_in_flight"]["value"]:.1f} frames/s | — |
| CPU baseline (oracle = op-for-op port, torch CPU fp32, {cb["cores"]} threads) | {cb["value"]:.2f} frames/s ({cb["ms_per_frame"]:.0f} ms/frame) | 0.75 |

| Kernel (as `bench.py` launches it) | bound | achieved | peak | frac | launch | PMC traffic vs algorithmic |
|---|---|---|---|---|---|---|
| `conv_halo_kernel<2,3,3>` encoder layer1 3×3 (dominant kernel) | fp16 MFMA / 3 | {r["achieved"]:.0f} TFLOP/s-equiv. | 833 | **{r["frac"]:.2f}** | {r["avg_launch_ms"]*1e3:.0f} µs | {p[r["kernel"]]["traffic"]/1e6:.0f} MB vs 197 MB |
| `corr_stream_kernel<8,true>` K5 (was the 256×128 tile kernel: 0.23) | HBM | {rk["achieved"]/1e3:.2f} TB/s | 8 | **{rk["frac"]:.2f}** | {rk["avg_launch_ms"]*1e3:.0f} µs | {p[rk["kernel"]]["traffic"]/1e6:.0f} MB vs 393 MB: writes exact ({p[rk["kernel"]]["write"]/1e6:.0f} MB); reads = the reference slice re-streamed by every workgroup + panel loads, served by L2 / Infinity Cache (counted at the fabric) |
| `corr_lookup_tile_kernel<float,2,256>` K7 on tiled planes (was the one-plane row kernel: 0.15) | HBM (gather) | {rl["achieved"]/1e3:.2f} TB/s | 8 | **{rl["frac"]:.2f}** | {rl["avg_launch_ms"]*1e3:.1f} µs | {p[rl["kernel"]]["traffic"]/1e6:.1f} MB vs 24.3 MB |

Other probes of this round (numbers quoted in DESIGN.md §8): `tools/k5_probe.py [--big] [--f16] [--stamps]` (K5 at every BASELINE shape, per-workgroup
cycle stamps, ablation builds via `tools/k5_ablate.sh`), `tools/k7_probe.py` (look-up: row-major one-plane kernel vs tile kernel vs fp16 planes;
`BFLOW_LOOKUP_ABL` phase ablation), `tools/c5_check.py [--f16]` (C5 frame rate and peak memory).
'''
open(os.path.join(ROOT, "profiles", "README.md"), "w").write(txt)
print(txt)
