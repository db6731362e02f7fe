import sqlite3, glob, sys
db = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
con = sqlite3.connect(db)
tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t]
cols = [r[1] for r in con.execute("pragma table_info(%s)" % kd)]
rows = con.execute("select start, end, kernel_id from %s order by start" % kd).fetchall()
names = dict(con.execute("select id, kernel_name from %s" % ks[0]).fetchall()) if ks else {}
# second half only (second fit run)
rows = rows[len(rows)//2:]
t0 = rows[0][0]
busy = sum(e - s for s, e, _ in rows); span = rows[-1][1] - t0
print("span %.1f ms busy(sum) %.1f ms" % (span/1e6, busy/1e6))
# biggest gaps
gaps = []
last_end = rows[0][1]
for s, e, k in rows[1:]:
    if s > last_end: gaps.append((s - last_end, names.get(k, str(k))[:40]))
    last_end = max(last_end, e)
gaps.sort(reverse=True)
print("total gap %.1f ms in %d gaps" % (sum(g for g, _ in gaps)/1e6, len(gaps)))
for g, n in gaps[:25]: print("%.2f ms before %s" % (g/1e6, n))
