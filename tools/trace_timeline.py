"""Print the kernel timeline of the LAST fit in a rocprofv3 --kernel-trace sqlite db: python tools/trace_timeline.py db [marker_kernel]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
c = db.cursor()
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kt = [t for t in tabs if 'kernel_dispatch' in t][0]
st = [t for t in tabs if 'kernel_symbol' in t][0]
rows = c.execute(f"select s.kernel_name, k.start, k.end from {kt} k join {st} s on k.kernel_id=s.id order by k.start").fetchall()
# last fit = from the last build_train_kernel on
idx = max(i for i, r in enumerate(rows) if 'build_train_kernel' in r[0])
rows = rows[idx:]
t0 = rows[0][1]
prev_end = t0
busy = 0
for name, s, e in rows:
    short = name.split('(')[0][:40]
    print("%9.1f us  +gap %6.1f  dur %8.1f  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, short))
    busy += e - s
    prev_end = e
print("span %.1f us, busy %.1f us" % ((prev_end - t0) / 1e3, busy / 1e3))
