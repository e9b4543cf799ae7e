#!/usr/bin/env python3
"""kpop_ca on a table with near-duplicate classes (several pivots of the Gram matrix's factorisation at the rounding
floor): the leading dimensions against the numpy restatement of R's ca, with the factor route (no fallback before this
round's change) and as it runs now.  KPOP_JACOBI_TRACE=1 shows the route taken."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import kpop_amd
    from oracle import ca_ref
    kpop_amd.init(0)
    rng = np.random.RandomState(3)
    I, J = 4000, 160
    base = rng.gamma(2.0, 1.0, size=I)
    N = rng.poisson(np.outer(base, rng.lognormal(0, 0.5, size=J)) * 30).astype(np.float64)
    # thirty classes that are copies of others, ten that are copies up to one count in a few thousand
    for j in range(30):
        N[:, 100 + j] = N[:, j]
    for j in range(10):
        N[:, 130 + j] = N[:, 40 + j]
        N[rng.randint(0, I, size=3), 130 + j] += 1
    tw, inertia, T = kpop_amd.ca(N, True)
    tw_o, in_o, T_o = ca_ref.ca(N, True)
    lead = 60
    a = ca_ref.align_signs(tw[:, :lead], tw_o[:, :lead], axis=1)
    scale = np.abs(tw_o[:, :lead]).max()
    print("inertia, leading %d: max rel diff %.2e" % (lead, np.max(np.abs(inertia[:lead] - in_o[:lead]) / in_o[:lead])))
    print("class positions, leading %d dimensions: max abs diff %.2e of a largest entry of %.2e" % (lead, np.max(np.abs(a - tw_o[:, :lead])), scale))


if __name__ == "__main__":
    main()
