"""CPU emulation of an error-free int8 split ("Ozaki scheme") of the variance contraction V = R^T K*, sum_r V^2 at BASELINE configs[2]
(numerics half): would s slices of 7 bits per operand, multiplied exactly on the I8 matrix pipe (int32 accumulation)
and recombined in fp64, hold the variance parity that the fp64 MFMA contraction holds?  Exact integer arithmetic in NumPy stands in for
the I8 MFMA.  Truth: oracle/truth_ld.c (long double).  TEST / PROBE INFRASTRUCTURE (lives under oracle/: nothing shipped imports it): python oracle/ozaki_numerics_probe.py"""
import os
import sys
import time

import numpy as np
from scipy.linalg import lapack

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import cpu_ref as R, truth as T


def split(A, axis, s, bits=7):
    """Scale every vector along `axis` by a power of two so that |.| < 1, then peel s signed slices of `bits` bits:
    A ~ 2^e * sum_i slice_i * 2^(-bits (i + 1)), slice_i integer in [-2^(bits-1), 2^(bits-1)] (what an int8 holds)."""
    mx = np.abs(A).max(axis=axis, keepdims=True)
    e = np.ceil(np.log2(np.where(mx > 0, mx, 1.0))) + 1
    rem = A / np.exp2(e)
    out = []
    for i in range(s):
        q = np.rint(rem * 2.0 ** (bits * (i + 1)))
        out.append(q.astype(np.int64))
        rem = rem - q / 2.0 ** (bits * (i + 1))
    return out, e


def main():
    N, d, m, C = 4096, 8, 4, 64
    p = R.synthetic_problem(N, d, m, C, 8, 1237)
    j = 0
    X, y, ls = p["X"], p["Y"][j][:, 0], p["lengthscales"][j]
    Xc = p["Xc"][:C]
    tru = T.posterior_truth("rbf", X, y, 1.0, ls, 1e-6, Xc)["raw_var"]
    K = R.kern_K_direct("rbf", X, None, 1.0, ls)
    K[np.diag_indices_from(K)] += 1e-6 + 1e-8
    L = lapack.dpotrf(K, lower=1)[0]
    Rinv = lapack.dtrtri(np.triu(L.T), lower=0)[0]             # R = U^-1 (upper), the device's operand
    Ks = R.kern_K_direct("rbf", X, Xc, 1.0, ls)                # (N, C)
    v64 = Rinv.T @ Ks
    var64 = 1.0 - np.square(v64).sum(0)
    print("fp64 GEMM (NumPy):                    max |dvar| %.3e   (variance %.2e..%.2e)" % (np.abs(var64 - tru).max(), tru.min(), tru.max()))
    for s in (5, 6, 7, 8, 9):
        t0 = time.time()
        A, ea = split(Rinv, 0, s)          # columns of R (output row r of V) share an exponent
        B, eb = split(Ks, 0, s)            # columns of K*
        V = np.zeros((N, C))
        ngemm = 0
        for g in range(s):                 # slice pairs with i + j = g: accumulate EXACTLY (int64 here, int32 on the matrix pipe)
            acc = np.zeros((N, C), dtype=np.int64)
            for i in range(g + 1):
                jx = g - i
                if i < s and jx < s:
                    acc += A[i].T @ B[jx]
                    ngemm += 1
            assert np.abs(acc).max() < 2 ** 31, "int32 accumulator would overflow"
            V += acc.astype(np.float64) * 2.0 ** (-7 * (g + 2))
        V *= np.exp2(ea).T * np.exp2(eb)
        var = 1.0 - np.square(V).sum(0)
        print("int8 split, %d slices (%2d int8 GEMMs, i + j < %d): max |dvar| %.3e   [%.0f s]" % (s, ngemm, s, np.abs(var - tru).max(), time.time() - t0))


if __name__ == "__main__":
    main()
