"""Summarise the task timeline of the team factorization (probes build: BOCF_PROBES=1 BOCF_TEAM_TL=/tmp/tl.txt python tools/team_check.py 1024).
python tools/team_timeline.py /tmp/tl.txt      (stamps are s_memrealtime ticks, 100 MHz: 0.01 us each)"""
import sys


def main():
    path = sys.argv[1]
    T = nb = m = None
    rec = []
    for line in open(path):
        if line.startswith("#"):
            _, _, T, _, nb, _, m = line.split()
            T, nb, m = int(T), int(nb), int(m)
            continue
        b, code, t0, t1, t2 = (int(x) for x in line.split())
        rec.append((b, code, t0, t1, t2))
    if not rec:
        print("no records")
        return
    base = min(r[2] for r in rec if r[2])
    us = lambda t: (t - base) * 0.01
    print("T %d nb %d m %d, %d records; output 0 only" % (T, nb, m, len(rec)))
    rows = []
    for b, code, t0, t1, t2 in rec:
        if b // T != 0:
            continue
        kind, rest = code // 1000000, code % 1000000
        p, r, c = rest // 10000, (rest // 100) % 100, rest % 100
        name = {1: "potrf", 2: "solveU", 3: "solveR", 4: "updA", 5: "updT", 6: "kinv", 7: "stream", 8: "last16"}.get(kind, "?%d" % kind)
        rows.append((t0, "wg%-3d %-6s p=%d (%d,%d)  start %8.2f  wait %6.2f  work %6.2f  end %8.2f" % (b % T, name, p, r, c, us(t0), (t1 - t0) * 0.01, (t2 - t1) * 0.01, us(t2))))
    crit = [x for x in rows if "potrf" in x[1] or "stream" in x[1] or "last16" in x[1] or any(("(%d,%d)" % (q, q + 1)) in x[1] and "solveU" in x[1] for q in range(nb)) or any(("(%d,%d)" % (q, q)) in x[1] and "updA" in x[1] and ("p=%d" % (q - 1)) in x[1] for q in range(nb))]
    print("---- critical chain")
    for _, s in sorted(crit):
        print(s)
    if len(sys.argv) > 2:
        print("---- all tasks")
        for _, s in sorted(rows):
            print(s)
    print("total %.2f us" % us(max(r[4] for r in rec)))


if __name__ == "__main__":
    main()
