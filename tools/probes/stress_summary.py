#!/usr/bin/env python3
"""Randomised shapes through the summary kernels against the oracle (a soak, not a test of the suite): small first operands
(the wave kernel with its tail / whole-CU-LDS / rows routes, the block kernel), and rows of 70,000-300,000 distances through
the one-pass, two-pass and no-rows paths.  Stops at the first disagreement."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def check(kpop, O, m1, m2, metric, kind, p, keep, cap, tag, exact=True):
    st_o, offs, idx_o, dist_o, z_o = O.distance_summary(m1, m2, metric, kind, p, True, keep)
    st, n, idx, dist, z = kpop.distance_summary(m1, m2, metric, kind, p, True, keep, max_neighbours=cap)
    np.testing.assert_allclose(st[:, :2], st_o[:, :2], rtol=1e-10, atol=1e-13, err_msg=tag)
    if kind != 2:
        assert np.array_equal(st[:, 2:], st_o[:, 2:]), (tag, st[:, 2:], st_o[:, 2:])
    else:
        np.testing.assert_allclose(st[:, 2:], st_o[:, 2:], rtol=1e-10, atol=1e-13, err_msg=tag)
    for j in range(m2.shape[0]):
        a, b = int(offs[j]), int(offs[j + 1])
        if kind == 2:  # (pow() differs from the host's by ulps: tie groups may split differently)
            continue
        assert n[j] == b - a, (tag, j, n[j], b - a)
        m = min(n[j], cap)
        if kind != 2:
            assert idx[j, :m].tolist() == idx_o[a:a + m].tolist(), (tag, j)
            assert np.array_equal(dist[j, :m], dist_o[a:a + m]), (tag, j)


def main():
    import kpop_amd as kpop
    from kpop_amd import api
    from oracle import oracle as O
    kpop.init(0)
    rng = np.random.RandomState(int(os.environ.get("SEED", "1")))
    n_small, n_large = int(os.environ.get("N_SMALL", "150")), int(os.environ.get("N_LARGE", "12"))
    for it in range(n_small):
        r1 = int(rng.choice([rng.randint(1, 80), rng.randint(60, 140), rng.randint(120, 280), rng.randint(250, 600), rng.randint(600, 3000)]))
        d = int(rng.choice([1, 3, 9, 16, 40, 64, 65, 100, 200]))
        r2 = int(rng.randint(1, 400))
        kind = int(rng.choice([0, 0, 1, 2]))
        p = float(rng.choice([1.0, 1.5, 2.0, 3.0]))
        keep = int(rng.choice([0, 1, 2, 7, 50]))
        grid = rng.rand() < 0.5
        m1 = np.round(rng.randn(r1, d), 1) if grid else rng.randn(r1, d)
        m2 = np.round(rng.randn(r2, d), 1) if grid else rng.randn(r2, d)
        if r1 > 4 and rng.rand() < 0.5:
            m1[r1 - 1] = m1[0]
            m1[r1 // 2] = m1[1]
            m2[r2 // 2] = m1[r1 - 1]
        metric = O.metric_powers(O.synth_inertia(d))
        tag = "small it=%d r1=%d d=%d r2=%d kind=%d p=%g keep=%d grid=%s" % (it, r1, d, r2, kind, p, keep, grid)
        check(kpop, O, m1, m2, metric, kind, p, keep, min(r1, 64), tag)
        if kind != 2 and r1 <= 600:  # the same rows as a ready distance matrix (summarize_distances: the kernels' PRE variants)
            dm = kpop.distance_rowwise(m1, m2, metric, kind, p, True)
            st, n, idx, dist, z = kpop.distance_summary(m1, m2, metric, kind, p, True, keep, max_neighbours=min(r1, 64))
            st2, n2, idx2, dist2, z2 = kpop.summarize_distances(dm, keep_at_most=keep, max_neighbours=min(r1, 64))
            for x, y in ((st, st2), (n, n2)):
                assert np.array_equal(x, y, equal_nan=True), (tag, "summarize_distances")
            for j in range(r2):
                mm = min(int(n[j]), min(r1, 64))
                assert np.array_equal(idx[j, :mm], idx2[j, :mm]) and np.array_equal(dist[j, :mm], dist2[j, :mm]) and np.array_equal(z[j, :mm], z2[j, :mm], equal_nan=True), (tag, j)
        if it % 25 == 0:
            print("ok", tag, flush=True)
    for it in range(n_large):
        r1 = int(rng.randint(70000, 300000))
        d = int(rng.choice([4, 12, 33]))
        r2 = int(rng.randint(1, 9))
        keep = int(rng.choice([1, 2, 100, 1500]))
        shape = rng.choice(["random", "classes", "grid", "few"])
        if shape == "random":
            m1 = rng.randn(r1, d)
        elif shape == "classes":
            c = rng.randn(7, d) * 3
            m1 = np.repeat(c, [r1 // 7] * 6 + [r1 - 6 * (r1 // 7)], axis=0) + rng.randn(r1, d) * 0.2
        elif shape == "grid":
            m1 = np.round(rng.randn(r1, d), 0)
        else:
            m1 = rng.randn(50, d)[rng.randint(0, 50, size=r1)]  # fifty distinct rows: tie groups of thousands
        m2 = rng.randn(r2, d) if shape != "grid" else np.round(rng.randn(r2, d), 0)
        m2[0] = m1[17]
        metric = O.metric_powers(O.synth_inertia(d))
        for mode in (1, 3, 2):
            api.tune("summary2", mode)
            tag = "large it=%d r1=%d d=%d r2=%d keep=%d shape=%s summary2=%d" % (it, r1, d, r2, keep, shape, mode)
            check(kpop, O, m1, m2, metric, 0, 2.0, keep, 256, tag)
        api.tune("summary2", 1)
        print("ok", tag, flush=True)
    print("all agree")


if __name__ == "__main__":
    main()
