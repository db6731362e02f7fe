#!/usr/bin/env python
"""Turn the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of
tools/pmc_workload.py into profiles/pmc_traffic.json.

Correction (MI355X_MICROARCH.md, HBM section): counters are KiB; the
calibration stream `k_calib_stream` reads a known 4*n bytes and writes 8*n
bytes, which gives the factor each counter has to be multiplied with for the
access widths these kernels use (measured: FETCH_SIZE x2.000, WRITE_SIZE x1.000).
Per kernel the bytes of ALL its dispatches are summed and divided by the number of
brutus_fit_batch calls of the workload (3), i.e. one row = HBM bytes that kernel
moves per call (several launches for k_top / k_fflux / k_offsets ...).  The row
"__total__" is the sum over all kernels: the real traffic of one call.

usage: PMC_COMMIT=<hash> pmc_to_json.py fetch.db write.db config batch calib_n [out.json] [ncalls]
(the commit the table was measured on travels with it: bench.py echoes it as
roofline.traffic_from_commit, so a table that has gone stale is visible)
"""
import json
import os
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"(?:void )?([A-Za-z0-9_:]+)", name)
    return m.group(1) if m else name


def per_kernel(db, counter):
    c = sqlite3.connect(db)
    acc = {}
    for name, did, val in c.execute(
            "select name, dispatch_id, counter_value from pmc_events where counter_name=?",
            (counter,)):
        acc.setdefault((short(name), did), 0.0)
        acc[(short(name), did)] += val          # sum over the per-SE/XCD rows
    out = {}
    for (k, did), v in acc.items():
        out[k] = out.get(k, 0.0) + v
    return out


def main():
    fdb, wdb, config, batch, n = sys.argv[1:6]
    out = sys.argv[6] if len(sys.argv) > 6 else "profiles/pmc_traffic.json"
    ncalls = float(sys.argv[7]) if len(sys.argv) > 7 else 3.0
    config, batch, n = int(config), int(batch), int(n)
    fetch = per_kernel(fdb, "FETCH_SIZE")
    write = per_kernel(wdb, "WRITE_SIZE")
    # the calibration stream is launched 3 times by tools/pmc_workload.py
    f_fac = (3 * 4.0 * n / 1024.0) / fetch["k_calib_stream"]
    w_fac = (3 * 8.0 * n / 1024.0) / write["k_calib_stream"]
    rows = []
    if os.path.exists(out):
        rows = [r for r in json.load(open(out))["rows"]
                if not (r["config"] == config and r["batch"] == batch)]
    trd = twr = 0.0
    for k in sorted(set(fetch) | set(write)):
        if not k.startswith("k_") or k in ("k_calib_stream", "k_relayout"):
            continue
        rd = fetch.get(k, 0.0) * f_fac * 1024.0 / ncalls
        wr = write.get(k, 0.0) * w_fac * 1024.0 / ncalls
        trd += rd
        twr += wr
        rows.append({"kernel": k, "config": config, "batch": batch,
                     "read_bytes_per_launch": round(rd), "write_bytes_per_launch": round(wr),
                     "hbm_bytes_per_launch": round(rd + wr)})
    rows.append({"kernel": "__total__", "config": config, "batch": batch,
                 "read_bytes_per_launch": round(trd), "write_bytes_per_launch": round(twr),
                 "hbm_bytes_per_launch": round(trd + twr)})
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    import bench
    json.dump({"commit": os.environ.get("PMC_COMMIT", "unknown"), "csrc_sha16": bench.csrc_sha16(),
               "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on "
                         "tools/pmc_workload.py; counters in KiB, corrected with k_calib_stream; bytes per brutus_fit_batch call",
               "fetch_correction": f_fac, "write_correction": w_fac, "rows": rows},
              open(out, "w"), indent=1)
    print("fetch x%.4f write x%.4f -> %s (%d rows)" % (f_fac, w_fac, out, len(rows)))


if __name__ == "__main__":
    main()
