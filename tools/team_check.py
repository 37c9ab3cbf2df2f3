"""Team schedule (option team_fit, chol_team.hip) against the launched schedule on the same inputs: factor, alpha, posterior, timing.
python tools/team_check.py [N ...]      (BOCF_PROBES=1 BOCF_TEAM_TL=<file> adds the task timeline of the last fit)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bocf_amd as B
from bocf_amd import synthetic as R


def run(N, m, d, team):
    p = R.synthetic_problem(N, d, m, 256, 8, 1237)
    kern = [B.kern.RBF(d, variance=1.0, lengthscale=p["lengthscales"][j], ARD=True) for j in range(m)]
    model = B.multi_outputGP(m, kernel=kern, noise_var=p["noise"], fixed_hyps=True)
    model.incremental = False
    model.set_option("team_fit", team)
    model.updateModel(p["X"], p["Y"])
    ts = []
    for _ in range(8):
        t0 = time.perf_counter()
        model.updateModel(p["X"], p["Y"])
        ts.append((time.perf_counter() - t0) * 1e3)
    L, a = zip(*[model.get_factor(j) for j in range(m)])
    mu, var = model.predict(p["Xc"])
    ctx = model._context()
    return dict(L=np.array(L), a=np.array(a), mu=mu, var=var, ms=min(ts), sched=ctx.stat("last_schedule"), to=ctx.stat("sched_timeouts"))


def main():
    sizes = [int(x) for x in sys.argv[1:]] or [256, 300, 512, 1024]
    m, d = int(os.environ.get("M", "4")), 6
    for N in sizes:
        a = run(N, m, d, 0)
        b = run(N, m, d, 1)
        rel = lambda x, y: float(np.abs(x - y).max() / max(np.abs(y).max(), 1e-300))
        print("N=%d m=%d: launched %.3f ms (schedule %d) | team %.3f ms (schedule %d, time-outs %d) | rel diff L %.2e alpha %.2e mean %.2e | abs diff var %.2e (var max %.2e)"
              % (N, m, a["ms"], a["sched"], b["ms"], b["sched"], b["to"], rel(b["L"], a["L"]), rel(b["a"], a["a"]), rel(b["mu"], a["mu"]),
                 float(np.abs(b["var"] - a["var"]).max()), float(a["var"].max())), flush=True)


if __name__ == "__main__":
    main()
