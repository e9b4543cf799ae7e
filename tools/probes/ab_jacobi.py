"""A/B of the eigen-solver behind kpop_ca: blocked Jacobi steps against the plain ones (kpop_tune("dbg", 32)), wall of kpop_ca
on a 131,328 x 1,024 and a 65,664 x 1,636 table (small I: the transfers stay small and the solver shows)."""
import sys, time
import numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
import kpop_amd
from kpop_amd import api
kpop_amd.init(0)
for I, J in ((65664, 1636),):
    rng = np.random.RandomState(J)
    base = rng.gamma(2.0, 1.0, size=I)
    N = np.empty((I, J))
    for j in range(J):
        N[:, j] = rng.poisson(base * rng.lognormal(0, 0.5, size=I) * 3)
    res = {}
    for dbg in (1, 65, 2, 32):
        api.tune("dbg", dbg)
        kpop_amd.ca(N[:4096])
        t0 = time.time(); tw, inertia, T = kpop_amd.ca(N); t1 = time.time()
        res[dbg] = (t1 - t0, tw, inertia)
        print("I=%d J=%d %s: %.3f s wall" % (I, J, ("blocked, %d inner sweeps, %s" % (dbg & 15, "looped" if dbg & 64 else "rows in registers")) if dbg != 32 else "plain  ", t1 - t0), flush=True)
    api.tune("dbg", 0)
    a, b = res[2], res[32]
    print("   inertia max rel diff %.2e; |twisted| max abs diff over the leading half %.2e" % (
        np.max(np.abs(a[2] - b[2]) / np.maximum(b[2], 1e-300)), np.max(np.abs(np.abs(a[1][:, :J // 2]) - np.abs(b[1][:, :J // 2])))))
