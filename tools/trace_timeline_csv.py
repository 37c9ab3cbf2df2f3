"""Kernel timeline of the LAST fit in a rocprofv3 --kernel-trace --output-format csv run:
python tools/trace_timeline_csv.py <dir> [max_rows]   (prints start, gap to the previous kernel's end, duration, queue, name)"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:44], r.get("Queue_Id", "?")))
rows.sort()
idx = max(i for i, r in enumerate(rows) if "build_train" in r[2])
rows = rows[idx:]
t0 = rows[0][0]
prev_end = t0
busy = 0
lim = int(sys.argv[2]) if len(sys.argv) > 2 else 10 ** 9
for n, (s, e, name, q) in enumerate(rows):
    if n < lim:
        print("%9.1f us  +gap %7.1f  dur %8.1f  q%-3s %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, q, name))
    busy += e - s
    prev_end = max(prev_end, e)
print("span %.1f us, sum of kernel durations %.1f us" % ((prev_end - t0) / 1e3, busy / 1e3))
