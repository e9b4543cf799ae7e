#!/usr/bin/env python3
"""Times kpop_ca (twister generation) at SARS-CoV-2-like sizes and checks it against numpy on the same table."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import kpop_amd
    from oracle import ca_ref
    kpop_amd.init(0)
    for I, J in ((131328, 256), (524800, 512), (524800, 1636)):
        rng = np.random.RandomState(J)
        base = rng.gamma(2.0, 1.0, size=I)
        N = np.empty((I, J))
        for j in range(J):
            N[:, j] = rng.poisson(base * rng.lognormal(0, 0.5, size=I) * 3)
        t0 = time.time()
        tw, inertia, T = kpop_amd.ca(N)
        t1 = time.time()
        flops = 2.0 * I * J * J * 0.5 + 2.0 * I * J * (J - 1)  # upper half of S'S + S*W
        print("kpop_ca I=%d J=%d: %.2f s wall (incl. %.1f GB H2D, %.1f GB D2H, Jacobi on the GPU); GEMM flops %.2f T" % (I, J, t1 - t0, N.nbytes / 1e9, T.nbytes / 1e9, flops / 1e12), flush=True)
        if I * J <= 131328 * 256:
            tw_o, in_o, T_o = ca_ref.ca(N)
            print("   vs numpy: inertia max rel %.1e, twisted max abs (sign-aligned, leading half) %.1e" % (
                np.max(np.abs(inertia - in_o) / in_o), np.max(np.abs(ca_ref.align_signs(tw, tw_o, 1)[:, :J // 2] - tw_o[:, :J // 2]))))


if __name__ == "__main__":
    main()
