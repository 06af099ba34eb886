#!/usr/bin/env python
"""Folds the rocprofv3 --pmc counter CSVs of tools/collect_profiles.sh (one FETCH_SIZE / WRITE_SIZE / MFMA set per roofline kernel + the
MFMA calibration launch) into profiles/r05_pmc.json.   usage: pmc_to_json.py <dir with <key>_{FETCH_SIZE,WRITE_SIZE,MFMA}.csv> <out.json>"""
import csv, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.roofline_kernels import CONV_NAME, HALO8_NAME, HALO8_REGEX, K5_NAME, K5_SPLIT_NAME, LOOKUP_NAME
d, outp = sys.argv[1], sys.argv[2]
NAMES = {"roofline": (CONV_NAME, "conv_halo_stream_kernel", 196755456.0),
         "roofline_update_conv": (HALO8_NAME, HALO8_REGEX, 4.0 * 4800 * (256 + 192 + 128 + 64) + 4.0 * 9 * (192 * 256 + 64 * 128)),
         "roofline_corr_build": (K5_NAME, "corr_stream_kernel", 393216000.0),
         "roofline_corr_build_split": (K5_SPLIT_NAME, "corr_stream_kernel", 393216000.0),
         "roofline_lookup": (LOOKUP_NAME.split(";")[0] + ")", "corr_lookup_tile_kernel", 24326400.0)}


def rows_of(path, regex):
    return [r for r in csv.DictReader(open(path)) if regex in r["Kernel_Name"]] if os.path.exists(path) else []


def avg(path, counter, regex, last=5):
    vals = [float(r["Counter_Value"]) for r in rows_of(path, regex) if r["Counter_Name"] == counter][-last:]   # the probe's own launches are the last ones
    return sum(vals) / len(vals) if vals else None


def mfma(path, regex, last=5):
    g = {c: avg(path, c, regex, last) for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE")}
    rs = [r for r in rows_of(path, regex) if r["Counter_Name"] == "GRBM_GUI_ACTIVE"][-last:]
    dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs if "End_Timestamp" in r]
    if dur:
        g["launch_ns_under_profiler"] = sum(dur) / len(dur)
        if g["GRBM_GUI_ACTIVE"]:
            g["clock_ghz_gui_active_over_wall"] = g["GRBM_GUI_ACTIVE"] / 8.0 / (sum(dur) / len(dur))   # the counter is summed over the 8 XCDs
    return g


cal = mfma(os.path.join(d, "calib_MFMA.csv"), "rate_kernel", last=3)        # a pure MFMA stream: every SIMD's matrix pipe busy all the time
# normalisation: busy cycles per GRBM cycle when all 1024 matrix pipes are busy
norm = cal["SQ_VALU_MFMA_BUSY_CYCLES"] / cal["GRBM_GUI_ACTIVE"] if cal.get("SQ_VALU_MFMA_BUSY_CYCLES") and cal.get("GRBM_GUI_ACTIVE") else None
kern = {}
for key, (name, regex, alg) in NAMES.items():
    f = avg(os.path.join(d, f"{key}_FETCH_SIZE.csv"), "FETCH_SIZE", regex)
    w = avg(os.path.join(d, f"{key}_WRITE_SIZE.csv"), "WRITE_SIZE", regex)
    f = None if f is None else f * 1024.0            # counters are reported in KB
    w = None if w is None else w * 1024.0
    fc = None if f is None else 2 * f
    m = mfma(os.path.join(d, f"{key}_MFMA.csv"), regex)
    if norm and m.get("SQ_VALU_MFMA_BUSY_CYCLES") is not None and m.get("GRBM_GUI_ACTIVE"):
        m["mfma_utilisation_over_gui_active"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / m["GRBM_GUI_ACTIVE"] / norm
    # GRBM_GUI_ACTIVE spans more than a short kernel (it implied 2.8-3.4 GHz for the 13-21 us launches in round 4): the judged figure is
    # normalised by SQ_BUSY_CYCLES -- the cycles the shader engines had waves of THIS kernel -- relative to the same ratio of the calibration launch
    if cal.get("SQ_BUSY_CYCLES") and cal.get("SQ_VALU_MFMA_BUSY_CYCLES") and m.get("SQ_BUSY_CYCLES") and m.get("SQ_VALU_MFMA_BUSY_CYCLES") is not None:
        m["mfma_utilisation"] = (m["SQ_VALU_MFMA_BUSY_CYCLES"] / m["SQ_BUSY_CYCLES"]) / (cal["SQ_VALU_MFMA_BUSY_CYCLES"] / cal["SQ_BUSY_CYCLES"])
    kern[name] = {"fetch_raw": f, "fetch_corrected": fc, "write": w, "traffic": None if fc is None or w is None else fc + w, "algorithmic_bytes": alg,
                  "mfma": m}
from bench import kernel_source_hash
json.dump({"source": "rocprofv3 --pmc on tools/roofline_probe.py (the launchers bench.py times), MI355X, round 5: FETCH_SIZE, WRITE_SIZE and the SQ / GRBM set in "
                     "separate passes",
           "kernel_source_hash": kernel_source_hash(),
           "units": "bytes per launch; counters are reported in KB (x1024).  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reads exactly 1/2 of "
                    "the bytes of wide coalesced (16 B/lane) streams, global_load and buffer_load...lds alike.  All three kernels read through 16-B-per-lane "
                    "streams (the look-up's gather is 16-B LDS-DMA units), so 'fetch_corrected' = 2 x raw and 'traffic' = fetch_corrected + write for all "
                    "of them.  'mfma': averages per launch; mfma_utilisation = (SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES) of the kernel divided by the "
                    "same ratio of the calibration launch (tools/micro/fp8_cross rate_kernel: back-to-back MFMAs on every SIMD = 100 %)",
           "mfma_calibration": cal, "kernels": kern}, open(outp, "w"), indent=1)
print(json.dumps(kern, indent=1))
