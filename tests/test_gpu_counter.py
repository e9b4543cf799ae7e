"""GPU parity of the k-mer database operations (kpop_counter_*: lib/KMerDB.ml statistics, transformations and
class combination) against the oracle, through the C ABI.

Bars: counts, combined spectra and every statistic of the default linear transformation (threshold 1, power 1)
are bit-exact.  With power != 1 the column sums are tree reductions of rounded pow() values and device pow/log
are not glibc's: relative 1e-12 there (north_star allows 1e-5 for floating point)."""
import numpy as np
import pytest

from conftest import load_golden, unhex

pytestmark = pytest.mark.gpu

NAMES = {"binary": 0, "power": 1, "clr": 2, "pseudocounts": 3}


def same(a, b, exact, rtol=1e-12):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape
    if exact:
        assert np.array_equal(a, b, equal_nan=True)
    else:
        assert np.array_equal(np.isnan(a), np.isnan(b))
        fin = np.isfinite(a) & np.isfinite(b)
        assert np.array_equal(a[~fin & ~np.isnan(a)], b[~fin & ~np.isnan(b)])
        np.testing.assert_allclose(a[fin], b[fin], rtol=rtol, atol=0)


def test_counter_golden(kpop):
    g = load_golden("counter_small.json")
    cols = [np.array(v, dtype=np.int32) for v in g["columns"]]
    n_rows, n_cols = g["n_rows"], g["n_cols"]
    for st in g["stats"]:
        cs, rs = kpop.counter_stats(cols, st["threshold"], st["power"])
        exact = st["power"] == 1.0  # non_zero, max, sum; sum_log goes through the device's log()
        wc, wr = unhex(st["col_stats"], (n_cols, 4)), unhex(st["row_stats"], (n_rows, 4))
        same(cs[:, :3], wc[:, :3], exact)
        same(rs[:, :3], wr[:, :3], exact)
        same(cs[:, 3], wc[:, 3], False, rtol=1e-12)
        same(rs[:, 3], wr[:, 3], False, rtol=1e-12)
    for tr in g["transforms"]:
        # statistics from the oracle's side of the fixture so the transformation is tested on its own
        st = next(s for s in g["stats"] if s["threshold"] == tr["threshold"] and s["power"] == tr["power"])
        cs = unhex(st["col_stats"], (n_cols, 4))
        want = unhex(tr["table"], (n_rows, n_cols))
        for kmer_major in (True, False):
            got = kpop.counter_transform(cols, cs, NAMES[tr["which"]], tr["threshold"], tr["power"], kmer_major=kmer_major)
            exact = tr["which"] == "binary" or (tr["which"] == "power" and tr["power"] == 1.0)
            same(got if kmer_major else got.T, want, exact)
    lin = unhex(g["stats"][0]["col_stats"], (n_cols, 4))[:, 2]
    for cb in g["combines"]:
        out, norm = kpop.counter_combine(cols, cb["sel"], lin, 0 if cb["criterion"] == "mean" else 1)
        assert out.tolist() == cb["out"], (cb["sel"], cb["criterion"])
        assert norm == pytest.approx(float.fromhex(cb["norm"]), rel=1e-12)


@pytest.mark.parametrize("n_rows,n_cols,lam", [(5000, 70, 3.0), (100_003, 9, 20.0), (777, 300, 1.0), (64, 1, 5.0), (1, 5, 2.0),
                                                (257, 1000, 0.3), (129, 1100, 2.0), (1500, 130, 0.05), (300, 4200, 0.7)])
def test_counter_random_vs_oracle(kpop, oracle, n_rows, n_cols, lam):
    rng = np.random.default_rng(n_rows * 31 + n_cols)
    depth = rng.uniform(0.2, 3.0, size=n_cols)
    cols = [rng.poisson(lam * d, n_rows).astype(np.int32) for d in depth]
    if n_cols > 3:
        cols[2][:] = 0
    for thr, pw in ((1.0, 1.0), (2.0, 1.0), (1e-4, 1.0)):
        cs, rs = kpop.counter_stats(cols, thr, pw)
        ocs, ors = oracle.counter_stats(cols, thr, pw)
        same(cs[:, :3], ocs[:, :3], True)
        same(rs[:, :3], ors[:, :3], True)
        same(cs[:, 3], ocs[:, 3], False, rtol=1e-11)
        same(rs[:, 3], ors[:, 3], False, rtol=1e-11)
    cs, _ = kpop.counter_stats(cols, 1.0, 0.5, rows=False)
    ocs, _ = oracle.counter_stats(cols, 1.0, 0.5)
    same(cs, ocs, False, rtol=1e-11)
    lin, _ = kpop.counter_stats(cols, 1.0, 1.0, rows=False)
    col_sum = lin[:, 2]
    sels = [list(range(n_cols)), list(range(n_cols - 1, -1, -1)), list(rng.permutation(n_cols)[: max(1, n_cols // 2)])]
    for sel in sels:
        for crit in (0, 1):
            out, norm = kpop.counter_combine(cols, sel, col_sum, crit)
            want, wnorm = oracle.counter_combine(cols, sel, col_sum, crit)
            assert np.array_equal(out, want), (sel[:5], crit)
            assert norm == pytest.approx(wnorm, rel=1e-12)
    if n_rows * n_cols <= 400_000:
        for which, thr, pw in ((1, 1.0, 1.0), (0, 2.0, 1.0), (2, 1.0, 1.0), (3, 1.0, 1.0), (3, 2.0, 0.5), (3, 3.0, 0.0), (1, 0.001, 2.0)):
            ocs, _ = oracle.counter_stats(cols, thr, pw)
            got = kpop.counter_transform(cols, ocs, which, thr, pw, kmer_major=True)
            want = oracle.counter_transform(cols, ocs, which, thr, pw, kmer_major=True)
            same(got, want, which == 0 or (which == 1 and pw == 1.0), rtol=1e-11)


def test_combine_properties_at_scale(kpop):
    """k = 12 sized database (8.39 M canonical k-mers, 12 spectra): size-independent properties."""
    n_rows = 8_390_656
    rng = np.random.default_rng(12)
    base = rng.poisson(3.0, n_rows).astype(np.int32)
    cols = [base, base * 2, base * 3] + [rng.poisson(2.0, n_rows).astype(np.int32) for _ in range(9)]
    cs, rs = kpop.counter_stats(cols, 1.0, 1.0)
    assert np.array_equal(cs[:, 2], [float(v.astype(np.int64).sum()) for v in cols])
    assert np.array_equal(cs[:, 0], [float(np.count_nonzero(v)) for v in cols])
    tot = np.zeros(n_rows, dtype=np.int64)
    for v in cols:
        tot += v
    assert np.array_equal(rs[:, 2], tot.astype(np.float64))
    col_sum = cs[:, 2]
    # three rescaled copies of one spectrum: each rescales to 3*base exactly (max_norm/norm = 3, 3/2, 1)
    out, _ = kpop.counter_combine(cols, [0, 1, 2], col_sum, 0)
    assert np.array_equal(out, base.astype(np.int64) * 9)
    med, _ = kpop.counter_combine(cols, [0, 1, 2], col_sum, 1)
    assert np.array_equal(med, base.astype(np.int64) * 9)
    # the median does not depend on the visiting order; the mean of a single spectrum is the spectrum
    a, _ = kpop.counter_combine(cols, list(range(12)), col_sum, 1)
    b, _ = kpop.counter_combine(cols, list(range(11, -1, -1)), col_sum, 1)
    assert np.array_equal(a, b)
    one, norm = kpop.counter_combine(cols, [5], col_sum, 0)
    assert np.array_equal(one, cols[5]) and norm == col_sum[5]


def test_counter_errors(kpop):
    cols = [np.arange(10, dtype=np.int32)]
    with pytest.raises(Exception, match="Unknown_combination_criterion"):
        kpop.counter_combine(cols, [0], [45.0], 7)
    with pytest.raises(Exception, match="unknown transformation"):
        kpop.counter_transform(cols, np.zeros((1, 4)), 9)
    with pytest.raises(Exception, match="Invalid_transformation"):
        kpop.counter_stats(cols, -1.0, 1.0)
    cs, rs = kpop.counter_stats([], 1.0, 1.0)
    assert cs.shape == (0, 4) and rs.shape == (0, 4)


def test_division_by_reciprocal_is_exact(kpop):
    """div_rn (reciprocal + two FMA corrections) against the hardware IEEE division, bit for bit: the operands the
    combination produces (count * max_norm over an integer norm) and adversarial ones (divisors next to powers of two,
    all-ones mantissas, random mantissas over a wide exponent range)."""
    import ctypes as C

    import torch
    from kpop_amd import _lib
    L = _lib.load()
    rng = np.random.default_rng(7)
    n = 1 << 22
    sets = []
    counts = rng.integers(0, 5000, n).astype(np.float64)
    max_norm = float(rng.integers(10**6, 10**12))
    sets.append((counts * max_norm, rng.integers(1, 10**12, n).astype(np.float64)))
    sets.append((rng.integers(1, 2**31, n).astype(np.float64) * float(2**40 + 12345), rng.integers(1, 2**53, n, dtype=np.int64).astype(np.float64)))
    pw = 2.0 ** rng.integers(1, 52, n)
    sets.append((rng.integers(1, 2**53, n, dtype=np.int64).astype(np.float64), np.maximum(pw + rng.integers(-2, 3, n), 1.0)))
    ones = np.ldexp(2.0 - 2.0**-52, rng.integers(-20, 60, n))            # all-ones mantissa: the hardware route
    sets.append((rng.integers(1, 2**53, n, dtype=np.int64).astype(np.float64), ones))
    mant = rng.integers(2**52, 2**53, (2, n), dtype=np.int64).astype(np.float64)
    sets.append((np.ldexp(mant[0], rng.integers(-200, 200, n)), np.ldexp(mant[1], rng.integers(-200, 200, n))))
    dev = torch.device("cuda", 0)
    for a, b in sets:
        da, db = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
        fast, exact = torch.empty_like(da), torch.empty_like(da)
        assert L.kpop_dev_division_probe(da.data_ptr(), db.data_ptr(), n, fast.data_ptr(), exact.data_ptr(), None) == 0
        torch.cuda.synchronize()
        f, e = fast.cpu().numpy(), exact.cpu().numpy()
        assert np.array_equal(e, a / b)          # the hardware division is IEEE
        assert np.array_equal(f, e), int(np.count_nonzero(f != e))


def test_counter_more_spectra_than_grid_y(kpop, oracle):
    """70,000 spectra of 40 k-mers: spectra ride on grid.y (65,535 at most per launch), so statistics and the
    spectrum-major transformation go in two launches; the mean of all of them and the bit-serial median agree with
    the oracle."""
    rng = np.random.default_rng(70000)
    n_cols, n_rows = 70_000, 40
    table = rng.poisson(2.0, size=(n_cols, n_rows)).astype(np.int32)
    cols = list(table)
    cs, rs = kpop.counter_stats(cols, 1.0, 1.0)
    assert np.array_equal(cs[:, 2], table.sum(axis=1).astype(np.float64)) and np.array_equal(rs[:, 2], table.sum(axis=0).astype(np.float64))
    got = kpop.counter_transform(cols, cs, 1, 1.0, 1.0, kmer_major=False)
    assert np.array_equal(got, table.astype(np.float64))
    sel = list(range(n_cols))
    for crit in (0, 1):
        out, norm = kpop.counter_combine(cols, sel, cs[:, 2], crit)
        want, wnorm = oracle.counter_combine(cols, sel, cs[:, 2], crit)
        assert np.array_equal(out, want) and norm == pytest.approx(wnorm, rel=1e-12)


@pytest.mark.parametrize("n_cols", [65, 128, 129, 257, 512, 513, 1025, 1637, 2048, 2049])
def test_median_awkward_values(kpop, oracle, n_cols):
    """The rescaled median where selection has the least to hold on to: column counts next to the widths the kernels
    change at; rescaled values that agree to many digits without being equal (count c over norm n against c+1 over a
    norm a hair larger); runs of exact ties; counts that wrapped around int32 (negative) and the largest positive
    ones; spectra with nothing but zeros elsewhere."""
    rng = np.random.default_rng(n_cols)
    n_rows = 403
    table = rng.poisson(6.0, size=(n_cols, n_rows)).astype(np.int64)
    table[:, 0:40] = 7                                     # exact ties wherever the norms tie too
    table[:, 40:80] = np.arange(n_cols)[:, None] % 3 + 1000  # near ties once rescaled
    table[::7, 80:120] = -5                                # wrapped counts
    table[::11, 120:160] = 2**31 - 1
    table[:, 160:200] = 0
    table[::2, 200:240] = 0                                # exactly half zeros: the upper median sits on the edge
    cols = [c.astype(np.int32) for c in table]
    col_sum = np.abs(table).sum(axis=1).astype(np.float64) + rng.integers(0, 3, n_cols)  # many norms equal or adjacent
    col_sum[3] = col_sum[4]
    sel = list(rng.permutation(n_cols))
    out, norm = kpop.counter_combine(cols, sel, col_sum, 1)
    want, wnorm = oracle.counter_combine(cols, sel, col_sum, 1)
    assert np.array_equal(out, want), np.flatnonzero(out != want)[:10]
    assert norm == pytest.approx(wnorm, rel=1e-12)


def test_median_storage_not_16_byte_aligned(kpop, oracle):
    """kpop_dev_counter_combine on storage that starts 4 bytes off a 16-byte boundary: the tile loads fall back from
    16-byte fetches to single counts, same result."""
    import torch
    from kpop_amd import _lib
    L = _lib.load()
    rng = np.random.default_rng(99)
    n_rows, n_cols = 1000, 200
    table = rng.poisson(3.0, size=(n_cols, n_rows)).astype(np.int32)
    ld = int(L.kpop_dev_counter_ld(n_rows))
    dev = torch.device("cuda", 0)
    flat = torch.zeros(n_cols * ld + 8, dtype=torch.int32, device=dev)
    view = flat[1:1 + n_cols * ld].view(n_cols, ld)
    view[:, :n_rows] = torch.from_numpy(table).to(dev)
    assert view.data_ptr() % 16 == 4
    col_sum = table.sum(axis=1).astype(np.float64)
    sel = torch.arange(n_cols, dtype=torch.int32, device=dev)
    norm = torch.from_numpy(col_sum).to(dev)
    ws = torch.empty(int(L.kpop_dev_counter_workspace_bytes(n_cols, n_rows)), dtype=torch.uint8, device=dev)
    out = torch.empty(n_rows, dtype=torch.int32, device=dev)
    nrm = torch.empty(1, dtype=torch.float64, device=dev)
    rc = L.kpop_dev_counter_combine(view.data_ptr(), ld, n_rows, sel.data_ptr(), norm.data_ptr(), n_cols, n_cols, float(col_sum.max()), 1,
                                    ws.data_ptr(), out.data_ptr(), nrm.data_ptr(), None)
    assert rc == 0, L.kpop_last_error()
    torch.cuda.synchronize()
    want, _ = oracle.counter_combine(list(table), list(range(n_cols)), col_sum, 1)
    assert np.array_equal(out.cpu().numpy(), want)
