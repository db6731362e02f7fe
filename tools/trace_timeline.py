"""Timeline of the last part of a rocprofv3 kernel trace: python tools/trace_timeline.py <dir> [n]"""
import sqlite3, glob, sys, re
db = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 120
con = sqlite3.connect(db)
tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t]
names = dict(con.execute("select id, kernel_name from %s" % ks[0]).fetchall())
rows = con.execute("select start, end, kernel_id from %s order by start" % kd).fetchall()
mc = [t for t in tabs if "memory_copy" in t]
if mc:
    try:
        rows += [(s, e, -1) for s, e in con.execute("select start, end from %s" % mc[0]).fetchall()]
        rows.sort()
    except Exception as ex:
        print("no copies:", ex)
rows = rows[-n:]
t0 = rows[0][0]
last = t0
for s, e, k in rows:
    nm = "COPY" if k == -1 else re.sub(r"_ZN12_GLOBAL__N_1\d+", "", names.get(k, str(k)))[:28]
    print("%9.3f  +%7.3f gap %7.3f  %s" % ((s - t0) / 1e6, (e - s) / 1e6, (s - last) / 1e6, nm))
    last = max(last, e)
