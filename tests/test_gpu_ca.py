"""GPU parity of the twister generator (kpop_ca: correspondence analysis, MFMA f64 GEMMs + host Jacobi)
against the numpy restatement of R's `ca` as KPopTwist uses it (src/KPopTwist:93-116).  Dimension signs are
arbitrary (they are in R too), so dimensions are sign-aligned before comparing."""
import numpy as np
import pytest

from oracle import ca_ref

pytestmark = pytest.mark.gpu


def synthetic_table(rng, n_kmers, n_classes, depth=200):
    base = rng.gamma(2.0, 1.0, size=n_kmers)
    classes = [np.maximum(base * rng.lognormal(0.0, 0.6, size=n_kmers), 0) for _ in range(n_classes)]
    return np.array([rng.poisson(c / c.sum() * depth * n_kmers) for c in classes], dtype=np.float64).T.copy()


@pytest.mark.parametrize("I,J,normalize", [(512, 10, True), (512, 10, False), (2080, 33, True), (8256, 130, True), (300, 2, True)])
def test_ca_vs_numpy_svd(kpop, I, J, normalize):
    rng = np.random.RandomState(I + J)
    N = synthetic_table(rng, I, J)
    N[5] = 0.0  # a k-mer that occurs nowhere: no mass, zero twister column
    tw_o, in_o, T_o = ca_ref.ca(N, normalize)
    tw, inertia, T = kpop.ca(N, normalize)
    nd = min(I, J) - 1
    assert tw.shape == (J, nd) and inertia.shape == (nd,) and T.shape == (nd, I)
    np.testing.assert_allclose(inertia, in_o, rtol=1e-9, atol=1e-14)
    assert abs(inertia.sum() - 1.0) < 1e-12 and np.all(np.diff(inertia) <= 1e-15)
    tw_a = ca_ref.align_signs(tw, tw_o, axis=1)
    T_a = ca_ref.align_signs(T, T_o, axis=0)
    scale_tw, scale_T = np.max(np.abs(tw_o)), np.max(np.abs(T_o))
    # the Gram route loses accuracy on dimensions with tiny singular values: hold the leading 90 % tightly
    lead = max(1, int(0.9 * nd))
    assert np.max(np.abs(tw_a[:, :lead] - tw_o[:, :lead])) <= 1e-8 * scale_tw
    assert np.max(np.abs(T_a[:lead] - T_o[:lead])) <= 1e-8 * scale_T
    assert np.max(np.abs(tw_a - tw_o)) <= 1e-6 * scale_tw
    assert np.all(T[:, 5] == 0.0)
    # transition formula: twisting a class's own (normalised) spectrum through the twister gives its position
    x = N / N.sum(axis=0, keepdims=True)
    np.testing.assert_allclose(T @ x, tw.T, rtol=0, atol=1e-9 * scale_tw)


def test_generated_twister_feeds_the_hot_path(kpop, oracle):
    """End to end without R: class spectra -> kpop_ca -> twister -> kpop_count_twist classifies reads
    drawn from the classes (nearest class = true class)."""
    k, n_classes, glen = 6, 8, 4000
    rng = np.random.RandomState(3)
    genomes = ["".join(rng.choice(list("ACGT"), size=glen)) for _ in range(n_classes)]
    cols = oracle.enumerate_kmers(k)
    col_of = {int(h): i for i, h in enumerate(cols)}
    from conftest import concat
    gb, go = concat(genomes)
    h, c, o = oracle.count_reads(gb, go, k)
    N = np.zeros((len(cols), n_classes))
    for j in range(n_classes):
        for hh, cc in zip(h[int(o[j]):int(o[j + 1])], c[int(o[j]):int(o[j + 1])]):
            N[col_of[int(hh)], j] = cc
    twisted_classes, inertia, T = kpop.ca(N)
    tw = kpop.Twister.load(T, cols, k)
    # class vectors through the twister == CA's own class positions
    got = tw.count_twist(gb, go)
    np.testing.assert_allclose(got, twisted_classes, rtol=0, atol=1e-9 * np.max(np.abs(twisted_classes)))
    reads, truth = [], []
    for j, g in enumerate(genomes):
        for _ in range(20):
            s = int(rng.randint(0, glen - 600))
            reads.append(g[s:s + 600])
            truth.append(j)
    rb, ro = concat(reads)
    t = tw.count_twist(rb, ro)
    metric = kpop.metric_compute(inertia)
    d = kpop.distance_rowwise(twisted_classes, t, metric)
    assert (np.argmin(d, axis=1) == np.array(truth)).mean() >= 0.95


def test_ca_several_dim_slabs_and_device_entry(kpop):
    """More dimensions than one slab of the twister pipeline holds (kCaDimSlab = 256 in ca.hip), with a ragged last
    slab; and kpop_dev_ca on device pointers gives the bytes kpop_ca gives."""
    import torch
    from kpop_amd import api
    I, J = 3000, 600
    rng = np.random.RandomState(11)
    N = synthetic_table(rng, I, J, depth=50)
    tw, inertia, T = kpop.ca(N, True)
    nd = J - 1
    assert T.shape == (nd, I)
    tw_o, in_o, T_o = ca_ref.ca(N, True)
    np.testing.assert_allclose(inertia, in_o, rtol=1e-8, atol=1e-14)
    lead = nd // 2
    T_a = ca_ref.align_signs(T, T_o, axis=0)
    assert np.max(np.abs(T_a[:lead] - T_o[:lead])) <= 1e-8 * np.max(np.abs(T_o))
    x = N / N.sum(axis=0, keepdims=True)
    np.testing.assert_allclose(T @ x, tw.T, rtol=0, atol=1e-9 * np.max(np.abs(tw)))
    dev = torch.device("cuda", 0)
    dN = torch.from_numpy(N).to(dev)
    before = dN.clone()
    work = torch.empty(api.dev_ca_workspace_bytes(I, J), dtype=torch.uint8, device=dev)
    d_tw = torch.zeros(J, nd, dtype=torch.float64, device=dev)
    d_in = torch.zeros(nd, dtype=torch.float64, device=dev)
    d_T = torch.zeros(nd, I, dtype=torch.float64, device=dev)
    got = api.dev_ca(dN.data_ptr(), I, J, work.data_ptr(), d_tw.data_ptr(), d_in.data_ptr(), d_T.data_ptr(), normalize=True,
                     stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert got == nd and torch.equal(dN, before)
    assert np.array_equal(d_in.cpu().numpy(), inertia)
    assert np.array_equal(d_tw.cpu().numpy(), tw)
    assert np.array_equal(d_T.cpu().numpy(), T)


def test_device_table_helpers_and_in_place_analysis(kpop):
    """kpop_dev_table_row_sums / _col_sums / _gather_rows against numpy, and kpop_dev_ca with d_work == d_counts (the table
    standardised where it stands) against the out-of-place call: the same bytes."""
    import ctypes as C

    import torch
    from kpop_amd import _lib, api
    L = _lib.load()
    dev = torch.device("cuda", 0)
    rng = np.random.RandomState(5)
    I, J = 4099, 77
    N = synthetic_table(rng, I, J, depth=30)
    dN = torch.from_numpy(N).to(dev)
    rs = torch.zeros(I, dtype=torch.float64, device=dev)
    cs = torch.zeros(J, dtype=torch.float64, device=dev)
    assert L.kpop_dev_table_row_sums(dN.data_ptr(), I, J, rs.data_ptr(), None) == 0
    assert L.kpop_dev_table_col_sums(dN.data_ptr(), I, J, cs.data_ptr(), None) == 0
    torch.cuda.synchronize()
    np.testing.assert_allclose(rs.cpu().numpy(), N.sum(axis=1), rtol=1e-13)
    np.testing.assert_allclose(cs.cpu().numpy(), N.sum(axis=0), rtol=1e-13)
    assert np.array_equal(rs.cpu().numpy(), N.sum(axis=1))  # counts are integers: every order of summation is exact
    rows = rng.permutation(I)[:1500].astype(np.uint64)
    d_rows = torch.from_numpy(rows.view(np.int64)).to(dev)
    picked = torch.zeros(len(rows), J, dtype=torch.float64, device=dev)
    assert L.kpop_dev_table_gather_rows(dN.data_ptr(), J, d_rows.data_ptr(), len(rows), picked.data_ptr(), None) == 0
    torch.cuda.synchronize()
    assert np.array_equal(picked.cpu().numpy(), N[rows.astype(np.int64)])
    nd = J - 1
    outs = []
    for in_place in (False, True):
        table = dN.clone()
        work = table if in_place else torch.empty(api.dev_ca_workspace_bytes(I, J), dtype=torch.uint8, device=dev)
        d_tw = torch.zeros(J, nd, dtype=torch.float64, device=dev)
        d_in = torch.zeros(nd, dtype=torch.float64, device=dev)
        d_T = torch.zeros(nd, I, dtype=torch.float64, device=dev)
        assert api.dev_ca(table.data_ptr(), I, J, work.data_ptr(), d_tw.data_ptr(), d_in.data_ptr(), d_T.data_ptr()) == nd
        torch.cuda.synchronize()
        assert torch.equal(table, dN) != in_place  # in place: the counts are gone; otherwise untouched
        outs.append((d_tw.cpu().numpy(), d_in.cpu().numpy(), d_T.cpu().numpy()))
    for a, b in zip(outs[0], outs[1]):
        assert np.array_equal(a, b)
