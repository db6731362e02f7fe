"""Share of a kernel trace's span during which none of the device-filling kernels is resident:
    python tools/trace_big.py <dir> [skip_fraction]   (rocprofv3 --kernel-trace output directory)"""
import collections, glob, re, sqlite3, sys
db = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
con = sqlite3.connect(db)
tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t]
names = dict(con.execute("select id, kernel_name from %s" % ks[0]).fetchall())
rows = con.execute("select start, end, kernel_id from %s order by start" % kd).fetchall()
rows = rows[int(len(rows) * skip):]
BIG = ("k_post_mc", "k_mt_bits", "k_mt_jump", "k_fflux", "k_pre32", "k_derive", "k_mt_emit")
big = sorted((s, e) for s, e, k in rows if any(b in names.get(k, "") for b in BIG))
span0, span1 = rows[0][0], max(e for _, e, _ in rows)
cov = 0; cur_s, cur_e = big[0]
gaps = []
for s, e in big[1:]:
    if s > cur_e:
        cov += cur_e - cur_s
        gaps.append((cur_e, s))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
cov += cur_e - cur_s
print("span %.1f ms, a device-filling kernel resident %.1f ms (%.1f %%), none %.1f ms" %
      ((span1 - span0) / 1e6, cov / 1e6, 100. * cov / (span1 - span0), (span1 - span0 - cov) / 1e6))
# what runs in the gaps
inside = collections.Counter()
for s, e, k in rows:
    n = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", names.get(k, str(k)))[:28]
    if any(b in n for b in BIG):
        continue
    for g0, g1 in gaps:
        lo, hi = max(s, g0), min(e, g1)
        if hi > lo:
            inside[n] += hi - lo
for n, t in inside.most_common(10):
    print("  in the gaps: %-30s %.2f ms" % (n, t / 1e6))
