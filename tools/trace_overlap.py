"""Concurrency of a rocprofv3 kernel trace: python tools/trace_overlap.py <dir> [skip_fraction]
Union of the kernel intervals vs their sum, time with 1 / 2 / 3+ kernels resident, and the
stretch of every kernel name (mean duration here)."""
import sqlite3, glob, sys, re, collections
db = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
con = sqlite3.connect(db)
tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t]
names = dict(con.execute("select id, kernel_name from %s" % ks[0]).fetchall())
rows = con.execute("select start, end, kernel_id from %s order by start" % kd).fetchall()
rows = rows[int(len(rows) * skip):]
ev = []
for s, e, k in rows:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
depth = 0; last = ev[0][0]; hist = collections.Counter()
for t, d in ev:
    hist[min(depth, 3)] += t - last
    last = t; depth += d
span = rows[-1][1] - rows[0][0]
tot = sum(e - s for s, e, _ in rows)
print("span %.2f ms, sum of kernels %.2f ms, idle %.2f, one %.2f, two %.2f, three+ %.2f" % (
    span / 1e6, tot / 1e6, hist[0] / 1e6, hist[1] / 1e6, hist[2] / 1e6, hist[3] / 1e6))
by = collections.defaultdict(list)
for s, e, k in rows:
    by[re.sub(r"_ZN12_GLOBAL__N_1\d+", "", names.get(k, str(k)))[:30]].append((e - s) / 1e3)
for n, v in sorted(by.items(), key=lambda kv: -sum(kv[1]))[:14]:
    print("%-32s n %4d  mean %8.1f us  total %8.2f ms" % (n, len(v), sum(v) / len(v), sum(v) / 1e3))
