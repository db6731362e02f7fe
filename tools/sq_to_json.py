#!/usr/bin/env python
"""Vector-instruction table for bench.py's `roofline.valu`, from an SQ counter pass and the ISA:

    python tools/sq_to_json.py <sq_a summary .txt> <config> <batch> <calls> <out.json> <file.s> [<file.s> ...]

* the summary is what tools/pmc_sq.sh writes from `rocprofv3 --pmc SQ_INSTS_VALU ...` of
  tools/pmc_workload.py (`calls` = brutus_fit_batch calls in that workload: 3): per kernel the
  dynamic number of vector wave-instructions per call;
* the .s files are the gfx950 assembly of the library's translation units (hipcc -S, or
  -save-temps): per kernel the STATIC mix of its vector instructions -- float64 arithmetic,
  float32 transcendentals, the rest -- which stands in for the dynamic mix (the hot loops
  dominate both).  bench.py prices the three classes with the issue times it measures on the
  box (brutus_calibrate_issue) and reports issue time / wall time.

Rows are merged into <out.json> by (kernel, config, batch); the file carries the fingerprint
of the kernel sources it was made from (bench.py flags a table older than the kernels)."""
import collections
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

TRANS32 = re.compile(r"^v_(exp|log|rcp|rsq|sqrt|sin|cos)(_legacy|_iflag)?_f32")


def isa_mix(paths):
    """{demangled short kernel name: (f64 share, trans32 share)} of the static vector instructions."""
    mix = {}
    for path in paths:
        L = open(path).read().split("\n")
        heads = [(k, l.split(":")[0]) for k, l in enumerate(L) if re.match(r"^_Z\S*:", l)]
        names = subprocess.run(["c++filt"], input="\n".join(h for _, h in heads), text=True,
                               capture_output=True).stdout.splitlines()
        for (k, _), dem in zip(heads, names):
            end = k
            while end < len(L) and ".end_amdhsa_kernel" not in L[end]:
                end += 1
            c = collections.Counter()
            for t in L[k:end]:
                t = t.strip()
                m = re.match(r"(v_[a-z_0-9]+)\s", t + " ")
                if not m:
                    continue
                o = m.group(1)
                if "_f64" in o and "cvt" not in o:
                    c["f64"] += 1
                elif TRANS32.match(o):
                    c["trans32"] += 1
                else:
                    c["other"] += 1
            tot = float(sum(c.values()))
            if tot:
                short = dem.replace("(anonymous namespace)::", "")
                short = re.sub(r"^void ", "", short).split("(")[0]
                mix[short] = (c["f64"] / tot, c["trans32"] / tot)
    return mix


def sq_counts(path):
    """{short kernel name: (dispatches, SQ_INSTS_VALU summed over the instances, per dispatch)}"""
    disp, valu = {}, {}
    for line in open(path):
        m = re.match(r"(\S.*?)\s+(\d+)\s+[\d.]+\s+[\d.]+\s+[\d.]+\s+[\d.]+\s+[\d.]+\s*$", line)
        if m and not line.startswith("#") and "SQ_" not in line:
            disp[m.group(1).strip()] = int(m.group(2))
        m = re.match(r"(\S.*?)\s+SQ_INSTS_VALU\s+(\d+)\s+([\d.]+)\s", line)
        if m:
            valu[m.group(1).strip()] = (int(m.group(2)), float(m.group(3)))
    out = {}
    for k, (rows, avg) in valu.items():
        d = disp.get(k)
        if d:
            out[k] = (d, avg * rows / d)
    return out


def main():
    sq, config, batch, calls, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    mix = isa_mix(sys.argv[6:])
    import bench
    table = {"rows": []}
    if os.path.exists(out):
        table = json.load(open(out))
    table["rows"] = [r for r in table["rows"] if not (r["config"] == config and r["batch"] == batch)]
    total = 0.
    for k, (d, per_dispatch) in sorted(sq_counts(sq).items(), key=lambda kv: -kv[1][1] * kv[1][0]):
        if not k.startswith("k_") or k.startswith("k_calib") or k.startswith("k_relayout"):
            continue
        f64, tr = mix.get(k, (0., 0.))
        per_call = per_dispatch * d / float(calls)
        total += per_call
        table["rows"].append({"kernel": k, "config": config, "batch": batch,
                              "dispatches_per_call": d / float(calls),
                              "valu_wave_insts_per_call": per_call,
                              "f64_share": round(f64, 4), "trans32_share": round(tr, 4),
                              "mix_known": k in mix})
    table["rows"].append({"kernel": "__total__", "config": config, "batch": batch,
                          "valu_wave_insts_per_call": total})
    table["csrc_sha16"] = bench.csrc_sha16()
    table["commit"] = os.environ.get("PMC_COMMIT", "unknown")
    table["note"] = ("SQ_INSTS_VALU of tools/pmc_workload.py (rocprofv3, summed over the counter's "
                     "instances) per brutus_fit_batch call; f64 / transcendental shares from the "
                     "static ISA of each kernel")
    json.dump(table, open(out, "w"), indent=1)
    print("%s: config %d batch %d: %.4g vector wave-instructions per call, %d kernels"
          % (out, config, batch, total, len([r for r in table["rows"] if r["config"] == config]) - 1))


if __name__ == "__main__":
    main()
