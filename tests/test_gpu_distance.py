"""GPU parity: norms, rowwise distances, summaries (lib/Space.ml:150-205, lib/Matrix.ml:42-76,191-266,632-766).

BASELINE.json asks for 1e-5 relative on distances; the kernels keep the reference's operation order, so we
hold them to 1e-12 (pow() of the Minkowski distance differs from libm by a few ulp: 1e-11)."""
import numpy as np
import pytest

from conftest import load_golden, unhex

pytestmark = pytest.mark.gpu


def rel_err(got, want):
    return np.max(np.abs(got - want) / np.maximum(np.abs(want), 1e-300)) if got.size else 0.0


KINDS = [("euclidean", 0, 2.0, 1e-12), ("cosine", 1, 2.0, 1e-12), ("minkowski(1)", 2, 1.0, 1e-11),
         ("minkowski(2.5)", 2, 2.5, 1e-11)]


def test_distance_golden_vectors(kpop):
    g = load_golden("distance_small.json")
    d = g["n_dims"]
    m1 = unhex(g["m1"], (g["m1_rows"], d))
    m2 = unhex(g["m2"], (g["m2_rows"], d))
    metric = kpop.metric_compute(unhex(g["inertia"]))
    assert np.array_equal(metric, unhex(g["metric_powers_1_1_2"]))
    for case in g["cases"]:
        want = unhex(case["dmatrix"], (g["m2_rows"], g["m1_rows"]))
        got = kpop.distance_rowwise(m1, m2, metric, case["kind"], case["p"], case["normalize"])
        tol = 1e-11 if case["kind"] == 2 else 1e-13
        assert np.max(np.abs(got - want)) <= tol * np.max(np.abs(want)), case["distance"]
        if case["kind"] != 2:
            assert np.array_equal(got, want), case["distance"]  # same IEEE operations in the same order
        for s in case["summaries"]:
            st, n, idx, dist, z = kpop.distance_summary(m1, m2, metric, case["kind"], case["p"], case["normalize"],
                                                        s["keep_at_most"], max_neighbours=g["m1_rows"])
            wst = unhex(s["stats"], (g["m2_rows"], 4))
            np.testing.assert_allclose(st, wst, rtol=1e-10, atol=1e-13)
            offs = s["offsets"]
            assert n.tolist() == [offs[j + 1] - offs[j] for j in range(g["m2_rows"])]
            for j in range(g["m2_rows"]):
                assert idx[j, :n[j]].tolist() == s["idx"][offs[j]:offs[j + 1]]
                np.testing.assert_allclose(dist[j, :n[j]], unhex(s["dist"][offs[j]:offs[j + 1]]), rtol=1e-10, atol=1e-13)


@pytest.mark.parametrize("name,kind,p,tol", KINDS)
@pytest.mark.parametrize("r1,r2,d", [(1, 1, 1), (65, 300, 64), (10, 77, 9), (130, 129, 100), (3, 5, 1635)])
def test_distance_rowwise_vs_oracle(kpop, oracle, name, kind, p, tol, r1, r2, d):
    rng = np.random.RandomState(r1 * 7 + r2 + d)
    m1 = rng.normal(size=(r1, d))
    m2 = rng.normal(size=(r2, d))
    if r2 > 2:
        m2[1] = 0.0
        m2[2] = m1[0]
    metric = oracle.metric_powers(oracle.synth_inertia(d))
    for normalize in (True, False):
        want = oracle.distance_rowwise(m1, m2, metric, kind, p, normalize)
        got = kpop.distance_rowwise(m1, m2, metric, kind, p, normalize)
        assert got.shape == (r2, r1)  # rows = second operand (lib/Matrix.ml:264-266)
        assert np.max(np.abs(got - want)) <= tol * max(np.max(np.abs(want)), 1e-300)
        if kind != 2:
            assert np.array_equal(got, want)


@pytest.mark.parametrize("name,kind,p,tol", KINDS[:3])
@pytest.mark.parametrize("r1,keep", [(1, 2), (2, 1), (65, 2), (65, 0), (300, 5), (1000, 300)])
def test_distance_summary_vs_oracle(kpop, oracle, name, kind, p, tol, r1, keep):
    rng = np.random.RandomState(r1 + keep)
    d, r2 = 16, 40
    m1 = np.round(rng.normal(size=(r1, d)), 1)  # coarse grid -> exact ties between distances do occur
    m2 = np.round(rng.normal(size=(r2, d)), 1)
    if r1 > 3:
        m1[3] = m1[1]  # a guaranteed tie group
    metric = oracle.metric_powers(oracle.synth_inertia(d))
    st_o, offs, idx_o, dist_o, z_o = oracle.distance_summary(m1, m2, metric, kind, p, True, keep)
    st, n, idx, dist, z = kpop.distance_summary(m1, m2, metric, kind, p, True, keep, max_neighbours=r1)
    np.testing.assert_allclose(st, st_o, rtol=max(tol, 1e-10), atol=1e-13)
    for j in range(r2):
        a, b = int(offs[j]), int(offs[j + 1])
        assert n[j] == b - a
        if kind != 2:
            assert idx[j, :n[j]].tolist() == idx_o[a:b].tolist()
            assert np.array_equal(dist[j, :n[j]], dist_o[a:b])
        np.testing.assert_allclose(z[j, :n[j]], z_o[a:b], rtol=1e-8, atol=1e-10)


def test_summary_truncates_at_max_neighbours(kpop, oracle):
    m1 = np.ones((6, 3))
    m2 = np.zeros((2, 3)) + 0.5
    metric = oracle.metric_flat(3)
    st, n, idx, dist, z = kpop.distance_summary(m1, m2, metric, normalize=False, keep_at_most=1, max_neighbours=2)
    assert n.tolist() == [6, 6]  # one tie group of six: eff_len = 6 (lib/Matrix.ml:648-649)
    assert idx[:, :2].tolist() == [[0, 1], [0, 1]]
    assert st[0, 1] == 0.0 and np.isnan(z[:, :2]).all()  # sd = 0: z unguarded (:688)


def test_distance_errors(kpop):
    with pytest.raises(ValueError):
        kpop.distance_rowwise(np.ones((2, 3)), np.ones((2, 4)), np.ones(3))  # Incompatible_geometries
    with pytest.raises(kpop.KPopError):
        kpop.distance_rowwise(np.ones((2, 3)), np.ones((2, 3)), np.ones(3), kind=7)
    with pytest.raises(kpop.KPopError):
        kpop.distance_rowwise(np.ones((2, 3)), np.ones((2, 3)), np.ones(3), kind=2, p=-1.0)
    out = kpop.distance_rowwise(np.ones((0, 3)), np.ones((2, 3)), np.ones(3))
    assert out.shape == (2, 0)


def test_pipeline_headline_shape_sample(kpop, oracle):
    """count -> twist -> distance vs classes on 20k synthetic reads, k=12, D=64, C=65; whole pipeline
    against oracle.pipeline on a 300-read sample (twister restricted to the sample's k-mers)."""
    k, d, n, L, C = 12, 64, 20000, 150, 65
    tw = kpop.Twister.synth(0x5EED, k, d)
    bases, offs = oracle.synth_reads(0x4B506F70, n, L)
    cb, co = oracle.synth_reads(0xC1A55, C, 500)
    classes = tw.count_twist(cb, co)
    twisted = tw.count_twist(bases, offs)
    metric = kpop.metric_compute(oracle.synth_inertia(d))
    dist = kpop.distance_rowwise(classes, twisted, metric)
    assert dist.shape == (n, C)
    pick = np.arange(0, n, n // 300)[:300]
    sb = np.concatenate([bases[int(offs[r]):int(offs[r + 1])] for r in pick])
    so = np.arange(len(pick) + 1, dtype=np.uint64) * np.uint64(L)
    allb = np.concatenate([sb, cb])
    allo = np.concatenate([so, co[1:] + so[-1]])
    h, c, o = oracle.count_reads(allb, allo, k)
    cols = np.unique(h)
    T = oracle.synth_twister(0x5EED, d, cols)
    cls_o = oracle.twist(T, cols, h, c.astype(np.float64), o)[len(pick):]
    tw_o, di_o, _ = oracle.pipeline(sb, so, k, T, cols, cls_o, metric)
    assert np.array_equal(classes, cls_o)
    assert np.array_equal(twisted[pick], tw_o)
    assert rel_err(dist[pick], di_o) <= 1e-12


def test_readme_known_answer_through_the_gpu(kpop):
    """README.md:649 -> README.md:660 with the HIP summary kernel (KPopTwistDB -S path, lib/Matrix.ml:767-810)."""
    kat = load_golden("readme_kat.json")
    row = np.array([[float(x) for x in kat["distance_row_text"]]])
    st, n, idx, d, z = kpop.summarize_distances(row, keep_at_most=kat["keep_at_most"], max_neighbours=10)
    want = kat["summary_line"].split("\t")
    np.testing.assert_allclose(st[0], [float(x) for x in want[1:5]], rtol=1e-12)
    assert n[0] == 2
    assert [kat["distance_header"][int(i)] for i in idx[0, :2]] == ["2", "10"]
    assert d[0, :2].tolist() == [float(want[6]), float(want[9])]
    np.testing.assert_allclose(z[0, :2], [float(want[7]), float(want[10])], rtol=1e-11)


def test_summarize_distances_vs_oracle(kpop, oracle):
    rng = np.random.RandomState(5)
    dm = np.round(rng.rand(50, 300), 2)  # ties
    st, n, idx, d, z = kpop.summarize_distances(dm, keep_at_most=3, max_neighbours=300)
    for j in range(dm.shape[0]):
        so, io, do, zo = oracle.summarize_row(dm[j], 3)
        np.testing.assert_allclose(st[j], so, rtol=1e-12)
        assert n[j] == len(io) and idx[j, :n[j]].tolist() == io.tolist()
        assert np.array_equal(d[j, :n[j]], do)


@pytest.mark.parametrize("r1,keep,kind", [(5000, 2, 0), (20000, 300, 0), (4097, 0, 1), (9000, 7, 2)])
def test_distance_summary_large_reference_set(kpop, oracle, r1, keep, kind):
    """Relatedness-engine shape (README.md:1101): r1 beyond the LDS sort -> radix-select path.  mean/sd are tree
    sums there, so statistics are held to 1e-10; neighbours (indices, distances) must match exactly."""
    rng = np.random.RandomState(r1)
    d, r2 = 16, 12
    m1 = np.round(rng.normal(size=(r1, d)), 1)
    m2 = np.round(rng.normal(size=(r2, d)), 1)
    m1[7] = m1[3]
    m2[5] = m1[11]  # a zero distance
    metric = oracle.metric_powers(oracle.synth_inertia(d))
    p = 1.5
    st_o, offs, idx_o, dist_o, z_o = oracle.distance_summary(m1, m2, metric, kind, p, True, keep)
    cap = 2048
    st, n, idx, dist, z = kpop.distance_summary(m1, m2, metric, kind, p, True, keep, max_neighbours=cap)
    np.testing.assert_allclose(st, st_o, rtol=1e-10, atol=1e-13)
    for j in range(r2):
        a, b = int(offs[j]), int(offs[j + 1])
        assert n[j] == b - a  # the reference's eff_len, even beyond what is returned
        m = min(n[j], cap)
        if kind != 2:
            assert idx[j, :m].tolist() == idx_o[a:a + m].tolist()
            assert np.array_equal(dist[j, :m], dist_o[a:a + m])
        else:
            np.testing.assert_allclose(dist[j, :m], dist_o[a:a + m], rtol=1e-11)
        np.testing.assert_allclose(z[j, :m], z_o[a:a + m], rtol=1e-8, atol=1e-9)


@pytest.mark.parametrize("kind,p", [(0, 2.0), (1, 2.0), (2, 1.5)])
def test_row_norms_and_distances_with_supplied_norms(kpop, oracle, kind, p):
    """kpop_dev_row_norms = Base.get_normalizations (lib/Matrix.ml:42-76) against the oracle, and kpop_dev_distance_rowwise_norms
    with those norms supplied = kpop_dev_distance_rowwise, bit for bit (what the streaming pipeline does with its class vectors)"""
    import torch
    from kpop_amd import _lib, api
    lib = _lib.load()
    rng = np.random.RandomState(kind + 3)
    d, r1, r2 = 64, 65, 3000
    m1, m2 = rng.normal(size=(r1, d)), rng.normal(size=(r2, d))
    m1[4] = 0.0  # a zero row: norm 0 -> 1
    metric = oracle.metric_powers(oracle.synth_inertia(d))
    dev = torch.device("cuda", 0)
    t1, t2, tm = torch.from_numpy(m1).to(dev), torch.from_numpy(m2).to(dev), torch.from_numpy(metric).to(dev)
    norms = torch.zeros(r1, dtype=torch.float64, device=dev)
    api.check(lib.kpop_dev_row_norms(t1.data_ptr(), r1, d, tm.data_ptr(), kind, p, norms.data_ptr(), None))
    want = oracle.normalizations(m1, metric, kind, p)
    got = norms.cpu().numpy()
    if kind == 2:
        np.testing.assert_allclose(got, want, rtol=1e-12)
    else:
        assert np.array_equal(got, want)
    assert got[4] == 1.0
    work = torch.empty(api.dev_distance_workspace_bytes(r1, r2, d), dtype=torch.uint8, device=dev)
    a = torch.zeros(r2, r1, dtype=torch.float64, device=dev)
    b = torch.zeros(r2, r1, dtype=torch.float64, device=dev)
    api.dev_distance_rowwise(t1.data_ptr(), r1, t2.data_ptr(), r2, d, tm.data_ptr(), work.data_ptr(), a.data_ptr(), kind=kind, p=p)
    api.check(lib.kpop_dev_distance_rowwise_norms(t1.data_ptr(), r1, norms.data_ptr(), t2.data_ptr(), r2, d, tm.data_ptr(), kind, p, 1, work.data_ptr(),
                                                  b.data_ptr(), None))
    torch.cuda.synchronize()
    assert np.array_equal(a.cpu().numpy(), b.cpu().numpy())


@pytest.mark.parametrize("case", ["random", "sorted", "ties", "constant", "clustered"])
@pytest.mark.parametrize("keep", [2, 300])
def test_summarize_distances_two_pass_path(kpop, oracle, case, keep):
    """rows of 65,536 distances and more take the brackets-from-a-sample paths (summary_large.hip, round 3; one pass: the
    candidates of the median and of the MAD's bands together, a certificate for the MAD -- or two passes): exact
    medians, MADs and neighbour lists whatever the sample saw -- a matrix in sorted order, heavy ties, constant rows and a
    clustered one (the bracket misses: the row is redone by the one-block-per-row kernel) included; mean / sd to 1e-10"""
    from kpop_amd import api
    rng = np.random.RandomState(len(case) + keep)
    r2, r1 = 5, 200003
    if case == "random":
        dm = np.abs(rng.normal(1.0, 0.2, size=(r2, r1)))
    elif case == "sorted":
        dm = np.sort(np.abs(rng.normal(1.0, 0.2, size=(r2, r1))), axis=1)
        dm[1] = dm[1][::-1]
    elif case == "ties":
        dm = np.round(np.abs(rng.normal(1.0, 0.2, size=(r2, r1))), 2)  # ~100 distinct values: tie groups of thousands
        dm[2, :5000] = 0.0
    elif case == "constant":
        dm = np.full((r2, r1), 0.75)
        dm[3, 17] = 0.5
    else:  # the first half of every row far from the second: a sample in runs still brackets; an adversary would not
        dm = np.concatenate([rng.normal(1.0, 0.01, size=(r2, r1 // 2)), rng.normal(5.0, 0.01, size=(r2, r1 - r1 // 2))], axis=1)
        dm[4, ::2] = 9.0  # ... and a row that alternates
    cap = 512
    for mode in (1, 3, 0):  # one pass over the rows (default), two passes (the round's first version), one block per row
        api.tune("summary2", mode)
        st, n, idx, d, z = kpop.summarize_distances(dm, keep_at_most=keep, max_neighbours=cap)
        for j in range(r2):
            so, io, do, zo = oracle.summarize_row(dm[j], keep)
            np.testing.assert_allclose(st[j, :2], so[:2], rtol=1e-10, atol=1e-13)
            assert st[j, 2] == so[2] and st[j, 3] == so[3], (case, mode, j, st[j], so)  # median and MAD are order statistics: exact
            assert n[j] == len(io)
            m = min(n[j], cap)
            assert idx[j, :m].tolist() == io[:m].tolist() and np.array_equal(d[j, :m], do[:m])
    api.tune("summary2", 1)


def test_distance_summary_two_pass_on_twisted_rows(kpop, oracle):
    """the same path from twisted rows (distances computed on the device into the chunk's rows): 70,000 reference rows"""
    rng = np.random.RandomState(9)
    d, r1, r2 = 16, 70000, 6
    m1 = np.round(rng.normal(size=(r1, d)), 2)
    m2 = np.round(rng.normal(size=(r2, d)), 2)
    m2[1] = m1[5]
    metric = oracle.metric_powers(oracle.synth_inertia(d))
    for keep in (1, 50):
        st_o, offs, idx_o, dist_o, z_o = oracle.distance_summary(m1, m2, metric, 0, 2.0, True, keep)
        st, n, idx, dist, z = kpop.distance_summary(m1, m2, metric, 0, 2.0, True, keep, max_neighbours=256)
        np.testing.assert_allclose(st, st_o, rtol=1e-10, atol=1e-13)
        for j in range(r2):
            a, b = int(offs[j]), int(offs[j + 1])
            assert n[j] == b - a
            m = min(n[j], 256)
            assert idx[j, :m].tolist() == idx_o[a:a + m].tolist() and np.array_equal(dist[j, :m], dist_o[a:a + m])


@pytest.mark.parametrize("case", ["random", "classes", "constant", "grid"])
def test_distance_summary_without_distance_rows(kpop, oracle, case):
    """kpop_tune("summary2", 2), 131,072 reference rows and more: the distances are computed and reduced in one kernel
    (summary_fused_pass_kernel), no r2 x r1 rows in HBM.  Exact medians, MADs and neighbour lists against the oracle, and against the two-pass path
    (kpop_tune("summary2", 1)) bit for bit; a database sorted by class, a constant one (every distance equal: the bands hold
    everything, the rows are redone from distance rows) and a coarse grid (heavy ties) included"""
    from kpop_amd import api
    rng = np.random.RandomState(len(case))
    d, r1, r2 = 12, 150001, 7
    if case == "random":
        m1 = rng.normal(size=(r1, d))
    elif case == "classes":  # five classes, the database sorted by class: a stripe of 2,048 rows is all one class
        centres = rng.normal(size=(5, d)) * 3
        m1 = np.repeat(centres, [r1 // 5] * 4 + [r1 - 4 * (r1 // 5)], axis=0) + rng.normal(size=(r1, d)) * 0.3
    elif case == "constant":
        m1 = np.tile(rng.normal(size=(1, d)), (r1, 1))
    else:
        m1 = np.round(rng.normal(size=(r1, d)), 0)
    m2 = rng.normal(size=(r2, d)) if case != "grid" else np.round(rng.normal(size=(r2, d)), 0)
    m2[1] = m1[5]  # a zero distance
    m1[77] = m1[5]  # ... twice
    metric = oracle.metric_powers(oracle.synth_inertia(d))
    for keep in (1, 300):
        st_o, offs, idx_o, dist_o, z_o = oracle.distance_summary(m1, m2, metric, 0, 2.0, True, keep)
        res = {}
        for mode in (2, 1):
            api.tune("summary2", mode)
            res[mode] = kpop.distance_summary(m1, m2, metric, 0, 2.0, True, keep, max_neighbours=512)
        api.tune("summary2", 1)
        st, n, idx, dist, z = res[2]
        np.testing.assert_allclose(st[:, :2], st_o[:, :2], rtol=1e-10, atol=1e-13)
        assert np.array_equal(st[:, 2:], st_o[:, 2:]), (case, keep, st, st_o)  # median and MAD: order statistics, exact
        assert np.array_equal(st[:, 2:], res[1][0][:, 2:])
        for j in range(r2):
            a, b = int(offs[j]), int(offs[j + 1])
            assert n[j] == b - a == res[1][1][j]
            m = min(n[j], 512)
            assert idx[j, :m].tolist() == idx_o[a:a + m].tolist() and np.array_equal(dist[j, :m], dist_o[a:a + m]), (case, keep, j)
            if case != "constant":  # (sd = 0 there: the reference's z is 0/0, the tree sums' sd a few ulps of nothing)
                np.testing.assert_allclose(z[j, :m], z_o[a:a + m], rtol=1e-8, atol=1e-9)


def test_large_reference_set_long_neighbour_lists(kpop, oracle):
    """against more than 4,096 rows the summary kernels return at most 2,048 neighbours per row; longer lists -- keep_at_most =
    all (req_len = r1, lib/Matrix.ml:723-726), or a tie group of thousands (:648-649) -- are completed by the host entry
    points (the row sorted by (distance, column) on the device): against the oracle, from the operands and from a given
    distance matrix, with room for everything and with a stride that truncates"""
    rng = np.random.RandomState(1)
    m1, m2 = rng.normal(size=(5000, 8)), rng.normal(size=(3, 8))
    m1[100:130] = m1[7]  # ties inside the list
    metric = oracle.metric_powers(oracle.synth_inertia(8))
    for normalize in (True, False):
        st, n, idx, dist, z = kpop.distance_summary(m1, m2, metric, normalize=normalize, keep_at_most=0, max_neighbours=5000)
        so, offs, io, do, zo = oracle.distance_summary(m1, m2, metric, normalize=normalize, keep_at_most=0)
        assert n.tolist() == [5000] * 3
        np.testing.assert_allclose(st, so, rtol=1e-10)
        for j in range(3):
            a = int(offs[j])
            assert idx[j].tolist() == io[a:a + 5000].tolist()
            assert np.array_equal(dist[j], do[a:a + 5000])
            np.testing.assert_allclose(z[j], zo[a:a + 5000], rtol=1e-8, atol=1e-9)
    dm = kpop.distance_rowwise(m1, m2, metric)
    st, n, idx, dist, z = kpop.summarize_distances(dm, keep_at_most=0, max_neighbours=5000)
    for j in range(3):
        so, io, do, zo = oracle.summarize_row(dm[j], 5000)
        assert n[j] == 5000 and idx[j].tolist() == io.tolist() and np.array_equal(dist[j], do)
        np.testing.assert_allclose(z[j], zo, rtol=1e-8, atol=1e-9)
    st, n, idx, dist, z = kpop.distance_summary(m1, m2, metric, keep_at_most=0, max_neighbours=3000)  # truncated by the caller's own stride
    so, offs, io, do, zo = oracle.distance_summary(m1, m2, metric, keep_at_most=0)
    assert n.tolist() == [5000] * 3
    for j in range(3):
        a = int(offs[j])
        assert idx[j].tolist() == io[a:a + 3000].tolist() and np.array_equal(dist[j], do[a:a + 3000])
    # a tie group of 3,000 among 20,000 reference rows: keep_at_most = 5 lists the whole group
    dm = rng.rand(4, 20000) + 1.0
    for j in range(4):
        tie = rng.choice(20000, size=3000, replace=False)
        dm[j, tie] = 0.5
        dm[j, rng.choice(np.setdiff1d(np.arange(20000), tie), size=2, replace=False)] = 0.25
    st, n, idx, dist, z = kpop.summarize_distances(dm, keep_at_most=5, max_neighbours=4000)
    for j in range(4):
        so, io, do, zo = oracle.summarize_row(dm[j], 5)
        assert n[j] == 3002 == len(io)
        assert idx[j, :3002].tolist() == io.tolist() and np.array_equal(dist[j, :3002], do)
        np.testing.assert_allclose(st[j], so, rtol=1e-10)
        np.testing.assert_allclose(z[j, :3002], zo, rtol=1e-8, atol=1e-9)


def test_summarize_distances_large_row_with_ties(kpop, oracle):
    rng = np.random.RandomState(2)
    dm = np.round(rng.rand(6, 10000), 3)  # ~10 entries per distinct value: tie groups everywhere
    st, n, idx, d, z = kpop.summarize_distances(dm, keep_at_most=25, max_neighbours=200)
    for j in range(dm.shape[0]):
        so, io, do, zo = oracle.summarize_row(dm[j], 25)
        np.testing.assert_allclose(st[j], so, rtol=1e-10)
        assert n[j] == len(io) and idx[j, :n[j]].tolist() == io.tolist()
        assert np.array_equal(d[j, :n[j]], do)


@pytest.mark.parametrize("kind,p", [(0, 2.0), (1, 2.0), (2, 1.5), (2, 3.0)])
@pytest.mark.parametrize("rows,d", [(1, 1), (65, 9), (1000, 64), (300, 257)])
def test_embeddings_vs_oracle(kpop, oracle, kind, p, rows, d):
    """Base.get_embeddings (lib/Matrix.ml:78-128): euclidean/cosine bit-exact, minkowski through the device pow()."""
    rng = np.random.RandomState(rows * 7 + d + kind)
    m = rng.standard_normal((rows, d))
    if rows > 2:
        m[1] = 0.0                                     # zero norm: the row is left as it is
    metric = oracle.metric_powers(oracle.synth_inertia(d))
    for normalize in (True, False):
        got = kpop.embeddings(m, metric, kind, p, normalize)
        want = oracle.embeddings(m, metric, kind, p, normalize)
        if kind == 2 and normalize:
            np.testing.assert_allclose(got, want, rtol=1e-12, atol=0)
        else:
            assert np.array_equal(got, want)


def test_config4_one_million_reads_pipeline(kpop, oracle):
    """BASELINE config 4 (1M x 150 bp, k = 12) through the device-resident pipeline on one GPU -- the work 8 ranks would
    split by reads with no collective: count -> twist -> distances to 65 class vectors.  A spread of reads is checked
    bit for bit against the oracle (twister restricted to their k-mers); shards of the read range reproduce the rows."""
    import torch
    from kpop_amd import api
    from kpop_amd.pipeline import DevicePipeline
    from kpop_amd.shard import shard_bounds
    k, d, n, L, C = 12, 64, 1_000_000, 150, 65
    dev = torch.device("cuda", 0)
    sp = torch.cuda.current_stream().cuda_stream
    tw = kpop.Twister.synth(0x7457, k, d)
    metric = kpop.metric_compute(oracle.synth_inertia(d))
    pipe = DevicePipeline(tw, metric, dev)
    bases = torch.empty(n * L, dtype=torch.uint8, device=dev)
    offs = torch.empty(n + 1, dtype=torch.int64, device=dev)
    api.dev_synth_reads(0x4B506F70, n, L, bases.data_ptr(), offs.data_ptr(), stream=sp)
    cb = torch.empty(C * 500, dtype=torch.uint8, device=dev)
    co = torch.empty(C + 1, dtype=torch.int64, device=dev)
    api.dev_synth_reads(0xC1A55, C, 500, cb.data_ptr(), co.data_ptr(), stream=sp)
    classes = pipe.count_twist(cb, co, 500)
    twisted = pipe.count_twist(bases, offs, L)
    dist = pipe.distance_rowwise(classes, twisted)
    torch.cuda.synchronize()
    assert dist.shape == (n, C)
    pick = np.array([0, 1, 124_999, 125_000, 500_000, 999_999])
    hb = np.concatenate([bases[r * L:(r + 1) * L].cpu().numpy() for r in pick])
    ho = np.arange(len(pick) + 1, dtype=np.uint64) * L
    chb, cho = oracle.synth_reads(0xC1A55, C, 500)
    h, c, o = oracle.count_reads(np.concatenate([hb, chb]), np.concatenate([ho, cho[1:] + ho[-1]]), k)
    cols = np.unique(h)
    want = oracle.twist(oracle.synth_twister(0x7457, d, cols), cols, h, c.astype(np.float64), o)
    assert np.array_equal(twisted[torch.from_numpy(pick).to(dev)].cpu().numpy(), want[:len(pick)])
    assert np.array_equal(classes.cpu().numpy(), want[len(pick):])
    wd = oracle.distance_rowwise(want[len(pick):], want[:len(pick)], metric)
    assert np.array_equal(dist[torch.from_numpy(pick).to(dev)].cpu().numpy(), wd)
    # rank r of 8 twists reads [lo, hi) only: the same rows, no exchange
    for r in (0, 3, 7):
        lo, hi = shard_bounds(n, r, 8)
        part = pipe.count_twist(bases[lo * L:hi * L], offs[lo:hi + 1] - offs[lo], L)
        assert torch.equal(part, twisted[lo:hi])


@pytest.mark.parametrize("kind,p", [(0, 2.0), (1, 2.0), (2, 1.5)])
@pytest.mark.parametrize("r1,r2,d", [(7, 5, 40_000), (65, 65, 70_001), (1, 3, 524_800)])
def test_distance_rowwise_long_rows(kpop, oracle, kind, p, r1, r2, d):
    """Spectral distances (KPopCountDB --distances): a few rows over very many dimensions take the slab-summed path
    (n_dims >= 32768); sums of slab sums differ from the reference's running sum by rounding only."""
    rng = np.random.RandomState(r1 * 31 + r2 + d % 97)
    m1 = rng.poisson(3.0, size=(r1, d)).astype(np.float64)
    m2 = rng.poisson(3.0, size=(r2, d)).astype(np.float64)
    m1 /= m1.sum(axis=1, keepdims=True)
    m2 /= m2.sum(axis=1, keepdims=True)
    m2[0] = m1[0]                                   # one coincident pair: distance exactly 0
    metric = np.ones(d)
    for normalize in (True, False):
        got = kpop.distance_rowwise(m1, m2, metric, kind, p, normalize)
        want = oracle.distance_rowwise(m1, m2, metric, kind, p, normalize)
        assert got.shape == (r2, r1) and got[0, 0] == 0.0
        np.testing.assert_allclose(got, want, rtol=1e-11, atol=0)


def test_distance_rowwise_more_rows_than_grid_y(kpop, oracle):
    """4.2 M second-operand rows against 200 reference rows: the rows ride on grid.y (65,535 blocks of 60), so the
    launch is split; rows from both sides of the split are checked against the oracle."""
    import torch
    from kpop_amd import api
    r1, r2, d = 200, 4_200_000, 4
    dev = torch.device("cuda", 0)
    sp = torch.cuda.current_stream().cuda_stream
    rng = np.random.RandomState(3)
    m1 = rng.standard_normal((r1, d))
    m2 = torch.randn(r2, d, dtype=torch.float64, device=dev)
    metric = np.array([0.4, 0.3, 0.2, 0.1])
    t1, tm = torch.from_numpy(m1).to(dev), torch.from_numpy(metric).to(dev)
    work = torch.empty(api.dev_distance_workspace_bytes(r1, r2, d), dtype=torch.uint8, device=dev)
    out = torch.empty(r2, r1, dtype=torch.float64, device=dev)
    api.dev_distance_rowwise(t1.data_ptr(), r1, m2.data_ptr(), r2, d, tm.data_ptr(), work.data_ptr(), out.data_ptr(), stream=sp)
    torch.cuda.synchronize()
    pick = torch.tensor([0, 1, 3_932_099, 3_932_100, 3_932_101, 4_199_999], device=dev)
    want = oracle.distance_rowwise(m1, m2[pick].cpu().numpy(), metric)
    assert np.array_equal(out[pick].cpu().numpy(), want)


def test_reference_distance_iterator_known_answer_on_the_gpu(kpop):
    """The reference's test/DistanceIterator known answer (see tests/test_oracle_golden.py) through kpop_distance_rowwise:
    Minkowski(1) goes through the device's pow(), so the digits are held to 1e-12 instead of %.15g text."""
    g = load_golden("distance_iterator.json")
    pts = np.array(g["points"], dtype=np.float64).reshape(-1, 1)
    d = kpop.distance_rowwise(pts, pts, np.array([g["metric_weight"]]), kpop.MINKOWSKI, 1.0, normalize=False)
    for i, j, text in g["pairs"]:
        assert d[j, i] == pytest.approx(float(text), rel=1e-12, abs=1e-300)
    n, listed = len(pts), {(min(i, j), max(i, j)) for i, j, _ in g["pairs"]}
    for i in range(n):
        for j in range(i + 1, n):  # |0.1 - 0.4| is one ulp above the bound: a tolerance band around it, not a sharp cut
            assert (d[j, i] <= 0.3 * (1 + 1e-12)) if (i, j) in listed else (d[j, i] >= 0.3 * (1 - 1e-12))


@pytest.mark.parametrize("r1", [1, 2, 63, 64, 65, 66, 72, 73, 128, 129, 136, 137, 257, 264, 265, 300, 512])
def test_wave_summary_equals_block_summary(kpop, oracle, r1):
    """the one-wavefront-per-row summary (small first operand) and the one-block-per-row summary are the same operations in
    the same order: identical bits, ties and sd = 0 included; `-S` on a ready distance matrix likewise"""
    from kpop_amd import api
    rng = np.random.RandomState(r1)
    d = 9 if r1 > 128 else 64
    m1 = rng.randn(r1, d)
    m2 = rng.randn(700, d)
    if r1 >= 3:
        m1[2] = m1[0]  # a tie group
        m2[5] = m1[1]  # a zero distance
    if r1 >= 66:  # 64 R + a few columns: the last ones are placed by rank among the sorted 64 R -- ties with those, and among themselves
        m1[r1 - 1] = m1[0]
        m1[r1 - 2] = m1[5]
        m2[7] = m1[r1 - 1]
        if r1 % 64 >= 4:
            m1[r1 - 3] = m1[r1 - 4]
    metric = kpop.metric_compute(oracle.synth_inertia(d))
    for kind, p in ((kpop.EUCLIDEAN, 2.0), (kpop.COSINE, 2.0), (kpop.MINKOWSKI, 1.5)):
        for keep in (2, 0, 7):
            res = {}
            for dbg in (0, 4):  # 4: force the block kernel
                api.tune("dbg", dbg)
                res[dbg] = kpop.distance_summary(m1, m2, metric, kind=kind, p=p, keep_at_most=keep, max_neighbours=min(r1, 40))
            api.tune("dbg", 0)
            for x, y in zip(res[0], res[4]):
                assert np.array_equal(x, y, equal_nan=True), (r1, kind, keep)
    dm = kpop.distance_rowwise(m1, m2, metric)
    dm[3] = dm[3, 0]  # a row whose distances are all equal: sd = 0
    res = {}
    for dbg in (0, 4):
        api.tune("dbg", dbg)
        res[dbg] = kpop.summarize_distances(dm, keep_at_most=2, max_neighbours=min(r1, 16))
    api.tune("dbg", 0)
    for x, y in zip(res[0], res[4]):
        assert np.array_equal(x, y, equal_nan=True), r1


@pytest.mark.parametrize("r1", [65, 70, 131])
def test_wave_summary_many_rows_per_wavefront(kpop, oracle, r1):
    """100,001 rows: every wavefront walks over a dozen of them, so the tail columns' distances are worked out for eight rows
    at a time (distance_summary_wave_kernel, TAIL); against the block kernel bit for bit, and rows against the oracle"""
    from kpop_amd import api
    rng = np.random.RandomState(r1)
    d, r2 = (64 if r1 < 128 else 20), 100001
    m1 = np.round(rng.randn(r1, d), 2)
    m2 = np.round(rng.randn(r2, d), 2)
    m1[r1 - 1] = m1[3]
    m2[99999] = m1[r1 - 1]
    metric = kpop.metric_compute(oracle.synth_inertia(d))
    res = {}
    for dbg in (0, 4, 32768):  # 4: the block kernel; 32768: the wave kernel with the doubled network
        api.tune("dbg", dbg)
        res[dbg] = kpop.distance_summary(m1, m2, metric, keep_at_most=2, max_neighbours=8)
    api.tune("dbg", 0)
    for other in (4, 32768):
        for x, y in zip(res[0], res[other]):
            assert np.array_equal(x, y, equal_nan=True), (r1, other)
    rows = np.concatenate([np.arange(0, r2, 9973), [99999, r2 - 1]])
    st_o, offs, idx_o, dist_o, z_o = oracle.distance_summary(m1, m2[rows], metric, 0, 2.0, True, 2)
    st, n, idx, dist, z = res[0]
    np.testing.assert_allclose(st[rows], st_o, rtol=1e-10, atol=1e-13)
    for q, j in enumerate(rows):
        a, b = int(offs[q]), int(offs[q + 1])
        m = min(n[j], 8)
        assert n[j] == b - a and idx[j, :m].tolist() == idx_o[a:a + m].tolist() and np.array_equal(dist[j, :m], dist_o[a:a + m])


@pytest.mark.parametrize("r1,d", [(100, 64), (128, 64), (200, 96), (500, 64), (100, 300), (512, 40)])
def test_wave_summary_first_operands_beyond_half_a_cu_of_lds(kpop, oracle, r1, d):
    """100-200 class vectors of 64+ dimensions: one block per CU with the whole LDS; what does not fit even so (500 x 64,
    100 x 300) goes through distance rows in the workspace and the wave kernel over them.  The block kernel's bits."""
    from kpop_amd import api
    rng = np.random.RandomState(r1 + d)
    m1 = np.round(rng.randn(r1, d), 1)
    m2 = np.round(rng.randn(3001, d), 1)
    m1[9] = m1[4]
    m2[11] = m1[r1 - 1]
    metric = kpop.metric_compute(oracle.synth_inertia(d))
    for kind, p in ((kpop.EUCLIDEAN, 2.0), (kpop.MINKOWSKI, 1.5)):
        res = {}
        for dbg in (0, 4):
            api.tune("dbg", dbg)
            res[dbg] = kpop.distance_summary(m1, m2, metric, kind=kind, p=p, keep_at_most=3, max_neighbours=12)
        api.tune("dbg", 0)
        for x, y in zip(res[0], res[4]):
            assert np.array_equal(x, y, equal_nan=True), (r1, d, kind)
    st_o, offs, idx_o, dist_o, z_o = oracle.distance_summary(m1, m2[:40], metric, 0, 2.0, True, 3)
    st, n, idx, dist, z = kpop.distance_summary(m1, m2, metric, keep_at_most=3, max_neighbours=12)
    np.testing.assert_allclose(st[:40], st_o, rtol=1e-10, atol=1e-13)
    for j in range(40):
        a, b = int(offs[j]), int(offs[j + 1])
        m = min(n[j], 12)
        assert n[j] == b - a and idx[j, :m].tolist() == idx_o[a:a + m].tolist() and np.array_equal(dist[j, :m], dist_o[a:a + m])


def test_splits_gaps_golden_and_random(kpop, pyref):
    """kpop_splits_gaps (per-dimension radix sorts + one stable sort of all gaps) against the restatement of
    lib/Matrix.ml:527-600: same gaps bit for bit, same member sets, same order, ties included"""
    from conftest import load_golden
    g = load_golden("splits_small.json")
    rng = np.random.RandomState(8)
    cases = [([[float.fromhex(x) for x in row] for row in c["emb"]], c["keep"]) for c in g["cases"]]
    big = np.round(rng.normal(size=(3000, 7)), 2)  # rounded: many equal coordinates and equal gaps
    cases.append((big.tolist(), 500))
    for emb, keep in cases:
        gap, dim, idx, perm = kpop.splits_gaps(np.array(emb), keep)
        want = pyref.splits_gaps(emb, keep)
        assert len(gap) == len(want)
        for s, (wg, wm) in enumerate(want):
            assert gap[s].hex() == wg.hex(), s
            assert sorted(perm[dim[s], :idx[s] + 1].tolist()) == wm, s


@pytest.mark.parametrize("case", ["random", "classes", "near_duplicates", "grid", "offset", "constant"])
@pytest.mark.parametrize("kind,d", [(0, 64), (1, 64), (0, 12), (0, 100), (0, 200), (1, 130)])
def test_distance_summary_on_the_matrix_cores(kpop, oracle, case, kind, d):
    """65,536 reference rows and more, euclidean / cosine: the distances are f64 MFMAs that LOCATE (|a|^2 + |b|^2 - 2 a.b: last
    bits differ from the reference's chain), what is reported -- neighbours, median, MAD -- is recomputed with the chain
    (distance_mfma.hip; beyond 128 dimensions -- 200, 130 here -- the tiled contraction, 128 x 128 rows a block, any number of dimensions).
    Exact against the oracle and bit for bit against the vector-pipe path (kpop_tune("summary_mfma", 0)),
    where cancellation bites included: reference rows equal to a query and 1e-12 .. 1e-6 away from it (near_duplicates),
    every row far from the origin (offset: the bands hold too much, the rows go to the fall-back), ties by the thousand
    (grid, constant)"""
    from kpop_amd import api
    rng = np.random.RandomState(len(case) * 131 + kind * 7 + d)
    r1, r2 = 70001, 9
    if case == "classes":
        centres = rng.normal(size=(5, d)) * 3
        m1 = np.repeat(centres, [r1 // 5] * 4 + [r1 - 4 * (r1 // 5)], axis=0) + rng.normal(size=(r1, d)) * 0.3
    elif case == "constant":
        m1 = np.tile(rng.normal(size=(1, d)), (r1, 1))
    elif case == "grid":
        m1 = np.round(rng.normal(size=(r1, d)), 0)
    else:
        m1 = rng.normal(size=(r1, d))
    m2 = rng.normal(size=(r2, d)) if case != "grid" else np.round(rng.normal(size=(r2, d)), 0)
    if case == "offset":
        m1 += 1000.0
        m2 += 1000.0
    m2[1] = m1[5]  # a zero distance
    m1[77] = m1[5]  # ... twice
    if case == "near_duplicates":
        for t, eps in enumerate((0.0, 1e-12, 1e-9, 1e-6, 1e-3)):
            rows = np.arange(1000 + 40 * t, 1000 + 40 * t + 40)
            m1[rows] = m2[2] * (1.0 + eps * rng.normal(size=(40, d)))
            m1[rows + 30000] = m2[3] + eps * rng.normal(size=(40, d))
    metric = oracle.metric_powers(oracle.synth_inertia(d))
    for normalize in (False, True) if case in ("random", "near_duplicates") else (True,):
        for keep in (1, 300):
            st_o, offs, idx_o, dist_o, z_o = oracle.distance_summary(m1, m2, metric, kind, 2.0, normalize, keep)
            res = {}
            # 3: the summary's pass inside the contraction, no distance row written (kpop_tune("summary_mfma", 2); up to 128 dimensions, beyond: as 1);
            # 1: the default, approximate rows then the pass over them; 2: ... its refinement scanning the rows instead of reading the lists
            api.tune("summary_audit", 1)
            api.summary_fallbacks()
            for mode in (3, 1, 2, 0):
                api.tune("summary_mfma", {3: 2, 1: 1, 2: 1, 0: 0}[mode])
                api.tune("summary_mfma_lists", 0 if mode == 2 else 1)
                res[mode] = kpop.distance_summary(m1, m2, metric, kind, 2.0, normalize, keep, max_neighbours=512)
                left = api.summary_fallbacks()
                # rows left to the exact fall-back change no result and cost milliseconds: on plain data there must be none (a fall-back
                # that ran hid a record-count mismatch of the in-contraction pass for a while: the results were right, the call three times slower)
                if case in ("random", "classes") and mode != 0:
                    assert left == 0, (case, mode, keep, left)
            api.tune("summary_audit", 0)
            api.tune("summary_mfma", 1)
            api.tune("summary_mfma_lists", 1)
            for a_, b_ in zip(res[1], res[2]):
                assert np.array_equal(a_, b_, equal_nan=True)
            for a_, b_ in zip(res[3][1:4], res[1][1:4]):  # (counts, columns, distances: exact either way; mean and sd are sums of approximate values)
                assert np.array_equal(a_, b_, equal_nan=True)
            assert np.array_equal(res[3][0][:, 2:], res[1][0][:, 2:])
            np.testing.assert_allclose(res[3][0][:, :2], res[1][0][:, :2], rtol=1e-10, atol=1e-13)
            st, n, idx, dist, z = res[3]
            np.testing.assert_allclose(st[:, :2], st_o[:, :2], rtol=1e-10, atol=1e-13)
            assert np.array_equal(st[:, 2:], st_o[:, 2:]), (case, keep, st, st_o)  # median and MAD: order statistics, exact
            assert np.array_equal(st[:, 2:], res[0][0][:, 2:])
            for j in range(r2):
                a, b = int(offs[j]), int(offs[j + 1])
                assert n[j] == b - a == res[0][1][j], (case, keep, j, n[j], b - a)
                m = min(n[j], 512)
                assert idx[j, :m].tolist() == idx_o[a:a + m].tolist() and np.array_equal(dist[j, :m], dist_o[a:a + m]), (case, keep, j)
                if case != "constant":
                    np.testing.assert_allclose(z[j, :m], z_o[a:a + m], rtol=1e-8, atol=1e-9)


@pytest.mark.parametrize("case", ["random", "near_duplicates", "offset", "grid", "constant"])
@pytest.mark.parametrize("kind,r1,r2,d", [(0, 1636, 2000, 1635), (1, 1636, 1700, 1635), (0, 3000, 8000, 200), (1, 300, 60000, 256), (0, 6000, 4000, 200)])
def test_distance_rowwise_on_the_matrix_cores(kpop, oracle, case, kind, r1, r2, d):
    """kpop_dev_distance_rowwise of 2^32 products and more (the reference's own job: 650 K samples x 1,636 classes x 1,635 dimensions,
    README.md:1054-1060; lib/Matrix.ml:191-266) as a tiled contraction on the f64 matrix cores: every distance within 1e-12 of the
    oracle's, relatively -- pairs whose d^2 is a small part of |a|^2 + |b|^2 (rows equal to each other, 1e-12 .. 1e-3 apart, everything
    far from the origin) recomputed with the reference's chain, i.e. bit for bit; kpop_tune("distance_mfma", 0) gives the
    vector pipe's bits for every pair"""
    from kpop_amd import api
    rng = np.random.RandomState(len(case) * 17 + kind + d)
    if case == "constant":
        m1 = np.tile(rng.normal(size=(1, d)), (r1, 1))
    elif case == "grid":
        m1 = np.round(rng.normal(size=(r1, d)), 0)
    else:
        m1 = rng.normal(size=(r1, d))
    m2 = rng.normal(size=(r2, d)) if case != "grid" else np.round(rng.normal(size=(r2, d)), 0)
    if case == "offset":
        m1 += 1000.0
        m2 += 1000.0
    m2[1] = m1[5]
    if case == "near_duplicates":
        for t, eps in enumerate((0.0, 1e-12, 1e-9, 1e-6, 1e-3, 0.1, 0.3)):
            m2[10 + t] = m1[7] * (1.0 + eps * rng.normal(size=d))
            m2[30 + t] = m1[8 + t] + eps * rng.normal(size=d)
    metric = oracle.metric_powers(oracle.synth_inertia(d))
    sub = np.unique(np.concatenate([np.arange(min(r2, 64)), rng.randint(0, r2, size=200)]))  # (the oracle on a sample of the second operand's rows when it is long)
    for normalize in (True, False) if case in ("random", "near_duplicates") else (True,):
        got = kpop.distance_rowwise(m1, m2, metric, kind, 2.0, normalize)
        want = oracle.distance_rowwise(m1, m2[sub], metric, kind, 2.0, normalize)
        g = got[sub]
        err = np.abs(g - want) / np.maximum(np.abs(want), 1e-300)
        err[want == g] = 0.0
        assert err.max() <= 1e-12, (case, normalize, err.max())
        assert g[1, 5] == want[1, 5] == 0.0 or sub[1] != 1
        if case == "offset":
            assert np.array_equal(g, want)  # (every pair cancels: every pair went through the chain)
        if case == "near_duplicates":
            assert np.array_equal(sub[:64], np.arange(64))
            assert np.array_equal(g[10:15, 7], want[10:15, 7]) and all(g[30 + t, 8 + t] == want[30 + t, 8 + t] for t in range(5))  # (the chain's bits for the pairs that cancel)
        if r2 <= 8000:
            api.tune("distance_mfma", 0)
            try:
                exact = kpop.distance_rowwise(m1, m2, metric, kind, 2.0, normalize)
            finally:
                api.tune("distance_mfma", 1)
            assert np.array_equal(exact[sub], want)
            if case == "random":
                assert not np.array_equal(exact, got)  # (the matrix cores did run)


def test_distance_summary_on_the_matrix_cores_in_several_chunks(kpop, oracle):
    """more query rows than one chunk of distance rows holds (8 GiB of them: 15,232 rows against 70,001), a last chunk that is
    short, a number of reference rows that is no multiple of 16 and of dimensions that is none of 4: every row against the
    vector-pipe path bit for bit (medians, MADs, neighbours), a handful against the oracle"""
    from kpop_amd import api
    rng = np.random.RandomState(11)
    d, r1, r2 = 7, 70001, 16000
    m1 = rng.normal(size=(r1, d))
    m2 = rng.normal(size=(r2, d))
    m2[15999] = m1[70000]
    m2[15232] = m1[3]
    metric = oracle.metric_powers(oracle.synth_inertia(d))
    res = {}
    for mode in (2, 1, 0):
        api.tune("summary_mfma", mode)
        res[mode] = kpop.distance_summary(m1, m2, metric, 0, 2.0, True, 3, max_neighbours=16)
    api.tune("summary_mfma", 1)
    assert np.array_equal(res[2][0][:, 2:], res[1][0][:, 2:]) and np.array_equal(res[2][1], res[1][1])
    a, b = res[2], res[0]
    assert np.array_equal(a[0][:, 2:], b[0][:, 2:])
    np.testing.assert_allclose(a[0][:, :2], b[0][:, :2], rtol=1e-11)
    assert np.array_equal(a[1], b[1])
    keep = np.arange(16)[None, :] < np.minimum(a[1], 16)[:, None]
    assert np.array_equal(a[2][keep], b[2][keep]) and np.array_equal(a[3][keep], b[3][keep])
    pick = [0, 1, 15231, 15232, 15233, 15999]
    st_o, offs, idx_o, dist_o, z_o = oracle.distance_summary(m1, m2[pick], metric, 0, 2.0, True, 3)
    for t, j in enumerate(pick):
        lo, hi = int(offs[t]), int(offs[t + 1])
        assert a[1][j] == hi - lo
        assert a[2][j, :hi - lo].tolist() == idx_o[lo:hi].tolist() and np.array_equal(a[3][j, :hi - lo], dist_o[lo:hi])
        assert a[0][j, 2] == st_o[t, 2] and a[0][j, 3] == st_o[t, 3]
    assert a[3][15999, 0] == 0.0 and a[2][15999, 0] == 70000


@pytest.mark.parametrize("kind,d,r2", [(0, 24, 640), (1, 64, 515), (0, 200, 600)])
def test_distance_summary_on_the_matrix_cores_many_query_rows(kpop, oracle, kind, d, r2):
    """Hundreds of query rows against 70,001 (whole tiles of query rows in the matrix-core kernels; 512 and more: the two-stream form under
    kpop_tune("summary_lanes", 2)): the default, the pass inside the contraction, two lanes and the vector pipe agree bit for bit on
    medians, MADs and neighbours, no row goes to the fall-back, a handful of rows against the oracle"""
    from kpop_amd import api
    rng = np.random.RandomState(d + r2)
    r1 = 70001
    m1 = rng.normal(size=(r1, d))
    m2 = rng.normal(size=(r2, d))
    m2[::7] = m1[rng.randint(0, r1, size=len(m2[::7]))]  # (a copy of a reference row every seventh: a zero distance each)
    metric = oracle.metric_powers(oracle.synth_inertia(d))
    res = {}
    api.tune("summary_audit", 1)
    api.summary_fallbacks()
    # (copy: the reference set divided by its norms into a copy first, as until late in round 6 -- the default takes it as it is; general_pass: the
    # pass over the approximate rows written for any caller's rows)
    for name, mfma, lanes, rawref, plain_pass in (("default", 1, 1, 1, 1), ("select", 2, 1, 1, 1), ("lanes", 1, 2, 1, 1), ("copy", 1, 1, 0, 1), ("general_pass", 1, 1, 1, 0),
                                                  ("vector", 0, 1, 1, 1)):
        api.tune("summary_mfma", mfma)
        api.tune("summary_lanes", lanes)
        api.tune("summary_rawref", rawref)
        api.tune("summary_pass", plain_pass)
        res[name] = kpop.distance_summary(m1, m2, metric, kind, 2.0, True, 20, max_neighbours=32)
        left = api.summary_fallbacks()
        assert left == 0, (name, left)
    api.tune("summary_audit", 0)
    api.tune("summary_mfma", 1)
    api.tune("summary_lanes", 1)
    api.tune("summary_rawref", 1)
    api.tune("summary_pass", 1)
    ref = res["vector"]
    for name in ("default", "select", "lanes", "copy", "general_pass"):
        got = res[name]
        assert np.array_equal(got[0][:, 2:], ref[0][:, 2:]), name
        np.testing.assert_allclose(got[0][:, :2], ref[0][:, :2], rtol=1e-10)
        assert np.array_equal(got[1], ref[1]), name
        keep = np.arange(32)[None, :] < np.minimum(got[1], 32)[:, None]
        assert np.array_equal(got[2][keep], ref[2][keep]) and np.array_equal(got[3][keep], ref[3][keep]), name
    for a_, b_ in zip(res["lanes"], res["default"]):
        assert np.array_equal(a_, b_, equal_nan=True)  # (the lanes change nothing at all)
    pick = [0, 1, 7, 255, 256, r2 - 1]
    st_o, offs, idx_o, dist_o, z_o = oracle.distance_summary(m1, m2[pick], metric, kind, 2.0, True, 20)
    for t, j in enumerate(pick):
        lo, hi = int(offs[t]), int(offs[t + 1])
        got = res["default"]
        assert got[1][j] == hi - lo
        assert got[2][j, :hi - lo].tolist() == idx_o[lo:hi].tolist() and np.array_equal(got[3][j, :hi - lo], dist_o[lo:hi])
        assert got[0][j, 2] == st_o[t, 2] and got[0][j, 3] == st_o[t, 3]



@pytest.mark.parametrize("members,d", [(100, 32), (500, 32), (100, 136)])
def test_distance_summary_against_a_database_laid_out_lineage_by_lineage(kpop, oracle, members, d):
    """A reference set of clusters of near-identical rows, cluster after cluster (what a database of genomes sorted by lineage is): neighbouring
    elements of a distance row are near-copies of each other.  The brackets come from the distances to a sample of the reference ROWS at
    even spacing, so they hold whatever the layout: no row goes through a slow path (with runs of the distance rows as the sample,
    kpop_tune("summary_sample", 0), most do -- asserted, so that the test keeps meaning something), and the results are those of the vector pipe"""
    from kpop_amd import api
    rng = np.random.RandomState(members + d)
    r1, r2 = 131072, 300
    centres = rng.normal(size=((r1 + members - 1) // members, d))
    m1 = np.repeat(centres, members, axis=0)[:r1] + 1e-3 * rng.normal(size=(r1, d))
    m2 = m1[rng.randint(0, r1, size=r2)].copy() + 1e-4 * rng.normal(size=(r2, d))
    metric = oracle.metric_powers(oracle.synth_inertia(d))
    res, left = {}, {}
    api.tune("summary_audit", 1)
    api.summary_fallbacks()
    for name, mfma, sample in (("rows", 1, 1), ("runs", 1, 0), ("vector", 0, 1)):
        api.tune("summary_mfma", mfma)
        api.tune("summary_sample", sample)
        res[name] = kpop.distance_summary(m1, m2, metric, 0, 2.0, True, 20, max_neighbours=32)
        left[name] = api.summary_fallbacks()
    api.tune("summary_audit", 0)
    api.tune("summary_mfma", 1)
    api.tune("summary_sample", 1)
    assert left["rows"] == 0, left
    assert left["runs"] > r2 // 10, left  # (the failure this sample is there for)
    for name in ("rows", "runs"):
        got, ref = res[name], res["vector"]
        assert np.array_equal(got[0][:, 2:], ref[0][:, 2:]), name
        np.testing.assert_allclose(got[0][:, :2], ref[0][:, :2], rtol=1e-10)
        assert np.array_equal(got[1], ref[1]), name
        keep = np.arange(32)[None, :] < np.minimum(got[1], 32)[:, None]
        assert np.array_equal(got[2][keep], ref[2][keep]) and np.array_equal(got[3][keep], ref[3][keep]), name
    pick = [0, 1, 150, 299]
    st_o, offs, idx_o, dist_o, z_o = oracle.distance_summary(m1, m2[pick], metric, 0, 2.0, True, 20)
    for t, j in enumerate(pick):
        lo, hi = int(offs[t]), int(offs[t + 1])
        got = res["rows"]
        assert got[1][j] == hi - lo
        assert got[2][j, :hi - lo].tolist() == idx_o[lo:hi].tolist() and np.array_equal(got[3][j, :hi - lo], dist_o[lo:hi])
        assert got[0][j, 2] == st_o[t, 2] and got[0][j, 3] == st_o[t, 3]
