#!/usr/bin/env python
"""Folds rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter CSVs (one pair per roofline kernel) into profiles/r02_pmc.json.
usage: pmc_to_json.py <dir with <key>_{FETCH,WRITE}_SIZE.csv> <out.json>"""
import csv, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
d, outp = sys.argv[1], sys.argv[2]
NAMES = {"roofline": ("conv_halo_kernel<2,3,3> (encoder layer1 3x3 64->64, 5x240x320)", "conv_halo_kernel", True, 196755456.0),
         "roofline_corr_build": ("corr_stream_kernel<8, true> (bflow_corr_build_split, D = 256)", "corr_stream_kernel", True, 393216000.0),
         "roofline_lookup": ("corr_lookup_tile_kernel<float, 2, 256> (fused bezier, tiled planes, split out)", "corr_lookup_tile_kernel", True, 24326400.0)}
def avg(path, counter, regex):
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter and regex in r["Kernel_Name"]]
    vals = vals[-5:]                      # the probe's own launches are the last ones (the warm-up forward also runs these kernels)
    return sum(vals) / len(vals) * 1024.0 if vals else None   # counters are reported in KB
kern = {}
for key, (name, regex, wide, alg) in NAMES.items():
    f, w = avg(os.path.join(d, f"{key}_FETCH_SIZE.csv"), "FETCH_SIZE", regex), avg(os.path.join(d, f"{key}_WRITE_SIZE.csv"), "WRITE_SIZE", regex)
    fc = 2 * f if (wide and f is not None) else None
    kern[name] = {"fetch_raw": f, "fetch_corrected": fc, "write": w, "traffic": (fc if fc is not None else f) + w, "algorithmic_bytes": alg}
from bench import kernel_source_hash
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on tools/roofline_probe.py, MI355X, round 2",
           "kernel_source_hash": kernel_source_hash(),
           "units": "bytes per launch; counters are reported in KB (x1024). gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reads exactly 1/2 of the bytes of wide coalesced (16 B/lane) streams, global_load and buffer_load...lds alike -> 'fetch_corrected' = 2 x raw for the LDS-DMA kernels; the look-up gather uses narrow 4-B loads (uncalibrated, raw kept)",
           "kernels": kern}, open(outp, "w"), indent=1)
print(json.dumps(kern, indent=1))
