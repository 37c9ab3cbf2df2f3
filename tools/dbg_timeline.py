"""Plain-run timeline of the factorization's chain kernels from BOCF_DBG_TL (device s_memrealtime stamps, 100 MHz):
BOCF_DBG_TL=/tmp/tl.txt BOCF_OPTIONS=lookahead=2 python tools/fit_only.py 4096 4 ; python tools/dbg_timeline.py /tmp/tl.txt [max_rows]
(the file holds the LAST factorization; one row per launch: workgroups whose start stamps lie within 3 us of each other)"""
import sys
names = {1: "potrf", 2: "tile128", 3: "gate", 4: "signal"}
rec = [tuple(int(x) for x in l.split()) for l in open(sys.argv[1])]
t00 = min(r[3] for r in rec)
rec.sort(key=lambda r: (r[0], r[3]))
launches = []
for kid in names:
    rs = [r for r in rec if r[0] == kid]
    cur = None
    for r in rs:
        if cur is None or r[3] - cur[1] > 300:           # 3 us in 10-ns ticks from the launch's first stamp
            cur = [kid, r[3], r[3], r[4], 1]
            launches.append(cur)
        else:
            cur[3] = max(cur[3], r[4]); cur[4] += 1
launches.sort(key=lambda l: l[1])
lim = int(sys.argv[2]) if len(sys.argv) > 2 else 10 ** 9
prev = {}
for n, (kid, first, _, last, cnt) in enumerate(launches):
    if n < lim:
        print("%9.1f us  dur %7.1f  wgs %4d  %s" % ((first - t00) / 100.0, (last - first) / 100.0, cnt, names[kid]))
po = [l for l in launches if l[0] == 1]
print("potrf starts (us):", " ".join("%.0f" % ((l[1] - t00) / 100.0) for l in po[::2]))
print("span %.1f us" % ((max(l[3] for l in launches) - t00) / 100.0))
