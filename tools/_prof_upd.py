import cProfile, pstats, sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
import bocf_amd as B
from bocf_amd import synthetic as R
N, d, m = int(sys.argv[1]), 4, 4
p = R.synthetic_problem(N, d, m, 8, 8, 1240)
model = B.multi_outputGP(m, exact_feval=[True] * m, fixed_hyps=False)
np.random.seed(0); model.updateModel(p["X"], p["Y"])
np.random.seed(1)
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter(); model.updateModel(p["X"], p["Y"]); t = time.perf_counter() - t0
pr.disable()
print("updateModel %.3f s" % t)
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
