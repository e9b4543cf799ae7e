#!/usr/bin/env python3
"""Randomised shapes through kpop_ca (against the numpy restatement of R's ca) and through the rescaled median (against the
oracle): column counts around every switch of the kernels, fewer k-mers than spectra, duplicated and empty spectra."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import kpop_amd
    from oracle import ca_ref
    from oracle import oracle as O
    kpop_amd.init(0)
    rng = np.random.RandomState(2026)
    bad = 0
    shapes = [(I, J) for J in (31, 32, 33, 63, 64, 65, 100, 127, 128, 129, 255, 256, 257, 300) for I in (J // 2 + 1, J, 3 * J, 2000)]
    for I, J in shapes:
        base = rng.gamma(2.0, 1.0, size=I)
        N = np.array([rng.poisson(base * rng.lognormal(0, 0.6, size=I) * 5) for _ in range(J)], dtype=np.float64).T.copy()
        N[:, 0] += 1.0  # no empty spectrum
        N[N.sum(axis=1) == 0, 0] = 1.0
        if J > 40:
            N[:, 7] = N[:, 3]  # a duplicated spectrum: an exact linear dependence
        for normalize in (True, False):
            tw, inertia, T = kpop_amd.ca(N, normalize)
            tw_o, in_o, T_o = ca_ref.ca(N, normalize)
            nd = min(I, J) - 1
            keep = in_o > 1e-9 * in_o[0]   # dimensions that carry anything
            ok = np.allclose(inertia[keep], in_o[keep], rtol=1e-7, atol=1e-13)
            lead = max(1, int(keep.sum() * 0.8))
            ta = ca_ref.align_signs(tw, tw_o, axis=1)
            ok = ok and np.max(np.abs(ta[:, :lead] - tw_o[:, :lead])) <= 1e-6 * np.max(np.abs(tw_o))
            x = N / N.sum(axis=0, keepdims=True) if normalize else None
            if normalize:
                ok = ok and np.allclose(T[:lead] @ x, tw.T[:lead], rtol=0, atol=1e-8 * np.max(np.abs(tw)))
            if not ok:
                bad += 1
                print("kpop_ca MISMATCH at I=%d J=%d normalize=%s" % (I, J, normalize), flush=True)
    print("kpop_ca: %d shapes x 2, %d mismatches" % (len(shapes), bad), flush=True)
    bad_m = 0
    for n_cols in (33, 64, 65, 128, 129, 256, 257, 511, 512, 513, 1024, 1025, 1500, 2048, 2049):
        for n_rows in (1, 17, 333, 1000):
            density = rng.choice([0.01, 0.3, 1.0])
            table = (rng.rand(n_cols, n_rows) < density) * rng.poisson(5.0, size=(n_cols, n_rows))
            cols = [c.astype(np.int32) for c in table]
            col_sum = np.maximum(table.sum(axis=1), 1).astype(np.float64) + rng.randint(0, 2, n_cols)
            sel = list(rng.permutation(n_cols)[: rng.randint(max(1, n_cols // 2), n_cols + 1)])
            out, norm = kpop_amd.counter_combine(cols, sel, col_sum, 1)
            want, wnorm = O.counter_combine(cols, sel, col_sum, 1)
            if not np.array_equal(out, want):
                bad_m += 1
                print("median MISMATCH at %d spectra x %d k-mers (%d selected)" % (n_cols, n_rows, len(sel)), flush=True)
    print("median: %d mismatches" % bad_m, flush=True)
    return 1 if bad or bad_m else 0


if __name__ == "__main__":
    sys.exit(main())
