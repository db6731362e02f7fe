#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd SQLite) result: per-kernel calls / total / avg /
min / max duration, plus PMC counter sums when present.  Used to turn the
`gpurun_out/*.db` scratch files into the text summaries kept under profiles/.
    python tools/rocpd_summary.py gpurun_out/prof/x_results.db > profiles/x.txt
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\[clone .*\]", "", name)
    m = re.match(r"(?:void )?([A-Za-z0-9_:]+(?:<[^>]*>)?)", name)
    return m.group(1) if m else name[:60]


def main(path):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    rows = c.execute("select name, start, end from kernels").fetchall()
    agg = {}
    for name, st, en in rows:
        k = short(name)
        d = (en - st) / 1e3   # ns -> us
        a = agg.setdefault(k, [0, 0.0, 1e30, 0.0])
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values()) or 1.0
    print("# rocprofv3 --kernel-trace summary of %s" % path)
    print("%-44s %7s %12s %11s %11s %11s %6s" % ("kernel", "calls", "total_us",
                                                 "avg_us", "min_us", "max_us", "%"))
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%-44s %7d %12.1f %11.2f %11.2f %11.2f %6.1f"
              % (k, a[0], a[1], a[1] / a[0], a[2], a[3], 100 * a[1] / tot))
    try:
        pm = c.execute("select name, counter_name, counter_value, duration "
                       "from pmc_events").fetchall()
    except sqlite3.Error:
        pm = []
    if pm:
        acc = {}
        for name, ctr, val, dur in pm:
            a = acc.setdefault((short(name), ctr), [0, 0.0, 0.0, 0.0])
            a[0] += 1
            a[1] += val
            a[2] = max(a[2], val)
            a[3] += dur / 1e3
        print("\n# PMC counters per dispatch (raw counter units; FETCH_SIZE/WRITE_SIZE are KiB)")
        print("%-44s %-14s %7s %16s %16s %11s" % ("kernel", "counter", "calls",
                                                   "avg/dispatch", "max/dispatch", "avg_us"))
        for (k, ctr), a in sorted(acc.items(), key=lambda kv: -kv[1][1]):
            print("%-44s %-14s %7d %16.1f %16.1f %11.2f"
                  % (k, ctr, a[0], a[1] / a[0], a[2], a[3] / a[0]))


if __name__ == "__main__":
    main(sys.argv[1])
