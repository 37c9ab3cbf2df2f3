"""cProfile of one learning-mode updateModel (host-side view of where a hyper-parameter update spends its time)."""
import cProfile
import os
import pstats
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bocf_amd as B
from bocf_amd import synthetic as R

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
p = R.synthetic_problem(N, 4, 4, 8, 8, 1240)
model = B.multi_outputGP(4, exact_feval=[True] * 4, fixed_hyps=False)
np.random.seed(0)
model.updateModel(p["X"], p["Y"])
pr = cProfile.Profile()
np.random.seed(1)
pr.enable()
model.updateModel(p["X"], p["Y"])
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
