#!/usr/bin/env python
"""Folds the rocprofv3 --pmc counter CSVs of tools/collect_profiles.sh (one FETCH_SIZE / WRITE_SIZE / MFMA set per roofline kernel + the
MFMA calibration launch + the PROBE line of tools/roofline_probe.py per key) into profiles/r06_pmc.json.
   usage: pmc_to_json.py <dir with <key>_{FETCH_SIZE,WRITE_SIZE,MFMA}.csv, <key>_meta.json, calib_MFMA.csv> <out.json>
FAILS (exit 1) when the calibration pass or any key's counters are missing: a table with holes is not evidence."""
import csv, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
d, outp = sys.argv[1], sys.argv[2]


def rows_of(path, regex):
    return [r for r in csv.DictReader(open(path)) if regex in r["Kernel_Name"]] if os.path.exists(path) else []


def avg(path, counter, regex, last):
    vals = [float(r["Counter_Value"]) for r in rows_of(path, regex) if r["Counter_Name"] == counter][-last:]   # the probe's own launches are the last ones
    return sum(vals) / len(vals) if vals else None


def mfma(path, regex, last):
    g = {c: avg(path, c, regex, last) for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE")}
    rs = [r for r in rows_of(path, regex) if r["Counter_Name"] == "GRBM_GUI_ACTIVE"][-last:]
    dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs if "End_Timestamp" in r]
    if dur:
        g["launch_ns_under_profiler"] = sum(dur) / len(dur)
    return g


missing = []
cal = mfma(os.path.join(d, "calib_MFMA.csv"), "rate_kernel", 3)        # a pure MFMA stream: every SIMD's matrix pipe busy all the time
if not (cal.get("SQ_BUSY_CYCLES") and cal.get("SQ_VALU_MFMA_BUSY_CYCLES")):
    missing.append("calibration (calib_MFMA.csv: tools/micro/fp8_cross rate_kernel under the SQ counters)")
kern = {}
for f in sorted(os.listdir(d)):
    if not f.endswith("_meta.json"):
        continue
    meta = json.load(open(os.path.join(d, f)))
    key, name, regex, reps = meta["key"], meta["name"], meta["regex"], int(meta.get("reps", 5))
    per = 2 if key == "roofline_corr_build_c5" else 1          # that key's launch = two kernel launches (event group + image group)
    fe = avg(os.path.join(d, f"{key}_FETCH_SIZE.csv"), "FETCH_SIZE", regex, reps * per)
    wr = avg(os.path.join(d, f"{key}_WRITE_SIZE.csv"), "WRITE_SIZE", regex, reps * per)
    m = mfma(os.path.join(d, f"{key}_MFMA.csv"), regex, reps * per)
    if fe is None or wr is None or not m.get("SQ_BUSY_CYCLES"):
        missing.append(f"{key} ({regex})")
        continue
    fe, wr = fe * 1024.0 * per, wr * 1024.0 * per            # counters are reported in KB; per LAUNCH of the bench's `launch()`
    fc = 2 * fe
    if cal.get("SQ_BUSY_CYCLES") and m.get("SQ_VALU_MFMA_BUSY_CYCLES") is not None:
        # busy cycles of the matrix pipes per cycle the shader engines had waves of THIS kernel, relative to the same ratio of the calibration launch
        m["mfma_utilisation"] = (m["SQ_VALU_MFMA_BUSY_CYCLES"] / m["SQ_BUSY_CYCLES"]) / (cal["SQ_VALU_MFMA_BUSY_CYCLES"] / cal["SQ_BUSY_CYCLES"])
    kern[name] = {"key": key, "fetch_raw": fe, "fetch_corrected": fc, "write": wr, "traffic": fc + wr, "algorithmic_bytes": meta["bytes"],
                  "traffic_over_algorithmic": (fc + wr) / meta["bytes"] if meta["bytes"] else None, "mfma": m}
if missing:
    print("pmc_to_json: MISSING " + "; ".join(missing), file=sys.stderr)
    sys.exit(1)
from bench import kernel_source_hash
json.dump({"source": "rocprofv3 --pmc on tools/roofline_probe.py (the launchers bench.py times), MI355X, round 6: FETCH_SIZE, WRITE_SIZE and the SQ / GRBM set in "
                     "separate passes",
           "kernel_source_hash": kernel_source_hash(),
           "units": "bytes per launch; counters are reported in KB (x1024).  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reads exactly 1/2 of "
                    "the bytes of wide coalesced (16 B/lane) streams, global_load and buffer_load...lds alike.  All of these kernels read through 16-B-per-lane "
                    "streams (the look-up's gather is 16-B LDS-DMA units), so 'fetch_corrected' = 2 x raw and 'traffic' = fetch_corrected + write.  'mfma': "
                    "averages per launch; mfma_utilisation = (SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES) of the kernel divided by the same ratio of the "
                    "calibration launch (tools/micro/fp8_cross rate_kernel: back-to-back MFMAs on every SIMD = 100 %)",
           "mfma_calibration": cal, "kernels": kern}, open(outp, "w"), indent=1)
print(json.dumps({k: {"traffic": v["traffic"], "alg": v["algorithmic_bytes"], "mfma_utilisation": v["mfma"].get("mfma_utilisation")} for k, v in kern.items()}, indent=1))
