"""Debug: factor with the scalar and the MFMA diagonal-block kernels, print where the factors differ."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bocf_amd as B
N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
rng = np.random.RandomState(1)
d = 2
X = rng.uniform(size=(N, d))
Y = [np.sin(3 * X.sum(1))[:, None]]
out = {}
for scalar in (1, 0):
    m = B.multi_outputGP(1, kernel=[B.kern.RBF(d, variance=2.0, lengthscale=np.array([0.3, 0.4]), ARD=True)], noise_var=[1e-3], fixed_hyps=True)
    m.set_option("potrf_scalar", scalar)
    m.updateModel(X, Y)
    L, a = m.get_factor(0)
    out[scalar] = (L, a, m.predict(X[:5]))
L1, L0 = out[1][0], out[0][0]
bad = np.abs(L1 - L0) > 1e-9 * (1 + np.abs(L1))
print("N", N, "mismatches", bad.sum(), "max abs", np.abs(L1 - L0).max())
rows, cols = np.nonzero(bad)
if bad.sum():
    print("rows (L row = U column):", sorted(set(rows.tolist()))[:40])
    print("cols (L col = U row):", sorted(set(cols.tolist()))[:40])
    for r, c in list(zip(rows, cols))[:10]:
        print(r, c, L1[r, c], L0[r, c])
print("alpha diff", np.abs(out[1][1] - out[0][1]).max(), "var diff", np.abs(out[1][2][1] - out[0][2][1]).max())
