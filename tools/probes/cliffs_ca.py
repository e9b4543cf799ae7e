#!/usr/bin/env python3
"""kpop_ca (host table in, twister out) over numbers of classes J at I = 65,664 k-mers: wall seconds -- looking for the sizes
where the eigen-solver changes kernels (rows in registers up to 2,048 columns, the factor route up to 2,048)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import kpop_amd
    kpop_amd.init(0)
    I = 65664
    for J in [int(x) for x in os.environ.get("JS", "64,256,1024,1636,2048,2049,2304,3000").split(",")]:
        rng = np.random.RandomState(J)
        base = rng.gamma(2.0, 1.0, size=I)
        N = rng.poisson(np.outer(base, rng.lognormal(0, 0.5, size=J)) * 3).astype(np.float64)
        kpop_amd.ca(N[:4096])
        t0 = time.time()
        tw, inertia, T = kpop_amd.ca(N)
        t1 = time.time()
        print("I = %d  J = %5d  %.3f s wall  (%d dimensions)" % (I, J, t1 - t0, len(inertia)), flush=True)


if __name__ == "__main__":
    main()
