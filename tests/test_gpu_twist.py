"""GPU parity: twist (lib/Twister.ml:146-188) and the fused count->twist, through the C ABI vs the oracle.

Tolerance: BASELINE.json asks for 1e-5 relative on twisted coordinates.  The kernels keep the reference's
order of operations (ascending column, unfused multiply-add), so we hold them to RTOL = 1e-12 of the
row's scale and additionally report bit-exactness where it is expected."""
import numpy as np
import pytest

from conftest import concat, load_golden, unhex

pytestmark = pytest.mark.gpu

RTOL = 1e-12


def _rc(h, k):
    r = 0
    for _ in range(k):
        r = (r << 2) | (3 - (h & 3))
        h >>= 2
    return r


def assert_close(got, want, scale=None):
    got, want = np.asarray(got), np.asarray(want)
    assert got.shape == want.shape
    s = np.max(np.abs(want)) if scale is None else scale
    assert np.max(np.abs(got - want)) <= RTOL * max(s, 1e-300), np.max(np.abs(got - want))


def test_twister_synth_matches_oracle(kpop, oracle):
    """Device-generated twister == oracle-generated twister loaded through kpop_twister_load."""
    k, d = 6, 9
    cols = oracle.enumerate_kmers(k)
    T = oracle.synth_twister(42, d, cols)
    a = kpop.Twister.synth(42, k, d)
    b = kpop.Twister.load(T, cols, k)
    assert a.info()["n_cols"] == b.info()["n_cols"] == len(cols)
    bases, offs = oracle.synth_reads(9, 200, 100)
    ta, tb = a.count_twist(bases, offs), b.count_twist(bases, offs)
    assert np.array_equal(ta, tb)
    h, c, o = oracle.count_reads(bases, offs, k)
    assert_close(ta, oracle.twist(T, cols, h, c.astype(np.float64), o))


def test_twist_golden_vectors(kpop, oracle):
    g = load_golden("twist_small.json")
    k, d = g["k"], g["n_dims"]
    cols = np.array(g["col_hash"], dtype=np.uint64)
    T = unhex(g["twister_dims_major"], (d, len(cols)))
    tw = kpop.Twister.load(T, cols, k)
    bases, offs = concat(g["reads"])
    h, c, o = oracle.count_reads(bases, offs, k)
    for normalize in (True, False):
        want = unhex(g["twisted_normalize_%s" % str(normalize).lower()], (len(g["reads"]), d))
        assert_close(tw.count_twist(bases, offs, normalize=normalize), want)
        assert_close(tw.twist(h, c.astype(np.float64), o, normalize=normalize), want)
    ds = g["dup_spectrum"]
    got = tw.twist(np.array(ds["hash"], dtype=np.uint64), unhex(ds["value"]), np.array([0, len(ds["hash"])], dtype=np.uint64))
    assert_close(got, unhex(ds["twisted"], (1, d)))


@pytest.mark.parametrize("k,d", [(5, 1), (8, 9), (10, 64), (10, 100), (12, 256), (9, 1635)])
def test_count_twist_vs_oracle(kpop, oracle, k, d):
    rng = np.random.RandomState(k * 1000 + d)
    seqs = ["", "ACG", "N" * 40] + ["".join(rng.choice(list("ACGTN"), size=int(rng.randint(k, 200)), p=[.245] * 4 + [.02]))
                                   for _ in range(150)]
    bases, offs = concat(seqs)
    h, c, o = oracle.count_reads(bases, offs, k)
    # a general (R-produced) twister: knows ~70 % of the k-mers that occur plus some that do not,
    # in shuffled column order
    seen = np.unique(h)
    known = seen[rng.rand(len(seen)) < 0.7] if k > 5 else seen
    extra = rng.randint(0, 4 ** k, size=100).astype(np.uint64)
    extra = np.array([x for x in extra if int(x) <= _rc(int(x), k)], dtype=np.uint64)  # canonical ones only
    cols = np.unique(np.concatenate([known, extra]))
    cols = cols[rng.permutation(len(cols))]
    T = oracle.synth_twister(77, d, cols)
    tw = kpop.Twister.load(T, cols, k)
    for normalize in (True, False):
        want = oracle.twist(T, cols, h, c.astype(np.float64), o, normalize)
        got = tw.count_twist(bases, offs, normalize=normalize)
        assert_close(got, want)
        assert_close(tw.twist(h, c.astype(np.float64), o, normalize=normalize), want)
    assert np.array_equal(got[:3], np.zeros((3, d)))  # no k-mer -> zero row


@pytest.mark.parametrize("d", [64, 33, 65, 72, 81, 96, 100, 130, 160])
def test_count_twist_bit_exact_when_columns_ascend(kpop, oracle, d):
    """With columns in ascending hash order the fused kernel adds the same terms in the same order as the
    oracle, unfused: the result is bit-identical -- also where the last block of dimensions is short (65, 72, 81, 96, 130,
    160: several rows per load instruction, the products handed down the lanes in row order)."""
    k = 8
    cols = oracle.enumerate_kmers(k)
    T = oracle.synth_twister(3, d, cols)
    tw = kpop.Twister.load(T, cols, k)
    bases, offs = oracle.synth_reads(123, 500, 150)
    h, c, o = oracle.count_reads(bases, offs, k)
    want = oracle.twist(T, cols, h, c.astype(np.float64), o)
    assert np.array_equal(tw.count_twist(bases, offs), want)


def test_count_twist_k_above_lut_limit(kpop, oracle):
    """k > 16: name -> column by bisection over the sorted hashes."""
    k, d = 21, 16
    bases, offs = oracle.synth_reads(8, 60, 120)
    h, c, o = oracle.count_reads(bases, offs, k)
    cols = np.unique(h)[::2].copy()  # the twister knows every other k-mer that occurs
    rng = np.random.RandomState(0)
    cols = cols[rng.permutation(len(cols))]
    T = oracle.synth_twister(5, d, cols)
    tw = kpop.Twister.load(T, cols, k)
    want = oracle.twist(T, cols, h, c.astype(np.float64), o)
    assert_close(tw.count_twist(bases, offs), want)
    assert_close(tw.twist(h, c.astype(np.float64), o), want)


@pytest.mark.parametrize("k", [15, 16])
def test_count_twist_k15_k16_through_the_block_index(kpop, oracle, k):
    """k = 15, 16: name -> row through rank-select BLOCKS of one 64-byte sector (480 presence bits + a prefix; twister.h) instead
    of 16-byte words -- a twister that knows some of the k-mers that occur and some that do not, reads and a few assemblies (the
    streaming kernel: the tile route reads rank words and is off at these k), against the oracle; hashes at both ends of the range
    and next to block boundaries among the columns"""
    d = 24
    rng = np.random.RandomState(k)
    bases, offs = oracle.synth_reads(8 + k, 300, 150)
    genomes = ["".join(rng.choice(list("ACGT"), size=int(n))) for n in (3000, 700, 5200)]
    gb, go = concat(genomes)
    bases = np.concatenate([bases, gb])
    offs = np.concatenate([offs, offs[-1] + go[1:]])
    h, c, o = oracle.count_reads(bases, offs, k)
    present = np.unique(h)
    extra = np.array([0, 1, 479, 480, 481, 959, 960, (1 << (2 * k)) - 1, (1 << (2 * k)) - 480, 12345678], dtype=np.uint64)
    # (names that are not canonical k-mers are never looked up: they only take their bits of the index, next to the block edges)
    cols = np.unique(np.concatenate([present[rng.rand(len(present)) < 0.6], extra]).astype(np.uint64))
    cols = cols[rng.permutation(len(cols))]
    T = oracle.synth_twister(5, d, cols)
    tw = kpop.Twister.load(T, cols, k)
    want = oracle.twist(T, cols, h, c.astype(np.float64), o)
    assert_close(tw.count_twist(bases, offs), want)
    assert_close(tw.twist(h, c.astype(np.float64), o), want)


def test_twist_long_spectrum(kpop, oracle):
    """A 30 kb genome's spectrum (config 3 shape) through the CSR twist."""
    from conftest import GOLDEN
    seq = "".join(l.strip() for l in open(GOLDEN + "/wuhan.fasta") if not l.startswith(">"))
    k, d = 10, 64
    bases, offs = concat([seq, seq[:5000]])
    h, c, o = oracle.count_reads(bases, offs, k)
    cols = oracle.enumerate_kmers(k)
    T = oracle.synth_twister(11, d, cols)
    tw = kpop.Twister.load(T, cols, k)
    assert_close(tw.twist(h, c.astype(np.float64), o), oracle.twist(T, cols, h, c.astype(np.float64), o))


@pytest.mark.parametrize("d,normalize", [(64, True), (100, False), (9, True)])
def test_twist_few_very_long_spectra_in_segments(kpop, oracle, d, normalize):
    """a handful of spectra of tens of thousands of lines (class spectra, genomes) are cut into stretches of 8,192 lines, a
    wavefront each, and the stretches' sums added in order (twist_csr_kernel<.., SEG>): against the oracle, and bit for bit
    against the one-wavefront-per-spectrum kernel, which walks the same stretches (kpop_tune("dbg", 1 << 28)); ragged
    lengths, an empty spectrum, unknown k-mers, fractional values"""
    from kpop_amd import api
    rng = np.random.RandomState(d)
    k = 9
    allk = oracle.enumerate_kmers(k)
    cols = allk[rng.rand(len(allk)) < 0.9]
    T = oracle.synth_twister(5, d, cols)
    tw = kpop.Twister.load(T, cols, k)
    lens = [len(allk), 70000, 0, 16384, 33333, 5]
    hs, vs, offs = [], [], [0]
    for n in lens:
        hh = np.sort(rng.choice(allk, size=min(n, len(allk)), replace=False)) if n else np.zeros(0, dtype=np.uint64)
        hs.append(hh)
        vs.append(np.round(rng.rand(len(hh)) * 50, 3) + 0.125)
        offs.append(offs[-1] + len(hh))
    h, v, o = np.concatenate(hs).astype(np.uint64), np.concatenate(vs), np.array(offs, dtype=np.uint64)
    want = oracle.twist(T, cols, h, v, o, normalize=normalize)
    got = tw.twist(h, v, o, normalize=normalize)
    api.tune("dbg", 1 << 28)
    try:
        plain = tw.twist(h, v, o, normalize=normalize)
    finally:
        api.tune("dbg", 0)
    scale = max(1.0, np.max(np.abs(want)))
    assert np.max(np.abs(got - want)) <= 1e-12 * scale and np.max(np.abs(plain - want)) <= 1e-12 * scale
    # both kernels form a spectrum's sums per stretch of 8,192 lines and add the stretches in order: the same bits
    assert np.array_equal(got, plain)
    # kpop_dev_twist with the longest spectrum UNDERSTATED (20,000 where two spectra are longer): those two come back as rows of
    # NaNs from the segmented launch too, never as silently shortened sums; the others are as before
    import torch
    dev = torch.device("cuda", 0)
    dh, dv, do = (torch.from_numpy(a).to(dev) for a in (h.view(np.int64), v, o.view(np.int64)))
    out = torch.zeros(len(lens), d, dtype=torch.float64, device=dev)
    api.dev_twist(tw, dh.data_ptr(), dv.data_ptr(), do.data_ptr(), len(lens), 20000, out.data_ptr(), normalize=normalize)
    torch.cuda.synchronize()
    out = out.cpu().numpy()
    long_ = [i for i, n in enumerate(lens) if min(n, len(allk)) > 3 * 8192]
    assert len(long_) == 3 and all(np.isnan(out[i]).all() for i in long_)
    rest = [i for i in range(len(lens)) if i not in long_]
    assert np.array_equal(out[rest], got[rest])


def test_headline_shape_sample_vs_oracle(kpop, oracle):
    """BASELINE headline shape: 100k x 150 bp, k=12, D=64, full synthetic twister (4.3 GB in HBM).
    Size-independent checks: determinism, normalised = unnormalised / n_kmers; plus a 400-read sample
    against the oracle, which sees a twister restricted to the sample's k-mers (identical by
    lib/Twister.ml:167-169: unknown k-mers do not contribute)."""
    k, d, n, L = 12, 64, 100000, 150
    tw = kpop.Twister.synth(0x5EED, k, d)
    assert tw.info()["n_cols"] == (4 ** 12 + 4 ** 6) // 2
    bases, offs = oracle.synth_reads(0x4B506F70, n, L)
    t1 = tw.count_twist(bases, offs)
    t2 = tw.count_twist(bases, offs)
    assert np.array_equal(t1, t2)
    raw = tw.count_twist(bases, offs, normalize=False)
    np.testing.assert_allclose(raw / (L - k + 1), t1, rtol=1e-13, atol=1e-15)
    rng = np.random.RandomState(1)
    pick = np.sort(rng.choice(n, size=400, replace=False))
    sb, so = concat([bytes(bases[int(offs[r]):int(offs[r + 1])]).decode() for r in pick])
    h, c, o = oracle.count_reads(sb, so, k)
    cols = np.unique(h)
    T = oracle.synth_twister(0x5EED, d, cols)
    want = oracle.twist(T, cols, h, c.astype(np.float64), o)
    assert_close(t1[pick], want)
    assert np.array_equal(t1[pick], want)  # ascending columns on both sides: bit-exact


@pytest.mark.parametrize("k,d", [(10, 64), (12, 9), (7, 100), (21, 16)])
def test_count_twist_long_sequences(kpop, oracle, k, d):
    """Genomes (config 3 shape): sequences beyond 512 windows take the streaming kernel -- several 16384-window
    segments, Ns inside, mixed in one batch with short reads that stay on the per-read kernel."""
    from conftest import GOLDEN
    rng = np.random.RandomState(k + d)
    wuhan = "".join(l.strip() for l in open(GOLDEN + "/wuhan.fasta") if not l.startswith(">"))
    long1 = "".join(rng.choice(list("ACGT"), size=100000))
    long1 = long1[:40000] + "N" * 37 + long1[40037:70000] + "nnnRYK" + long1[70006:]
    seqs = [wuhan, "ACGTTGCA" * 20, long1, "", wuhan[:513 + k - 1], wuhan[100:612 + k - 1], "".join(rng.choice(list("ACGT"), size=150))]
    bases, offs = concat(seqs)
    h, c, o = oracle.count_reads(bases, offs, k)
    if k <= 12:
        cols = oracle.enumerate_kmers(k)
    else:
        cols = np.unique(h)[::2].copy()
    T = oracle.synth_twister(21, d, cols)
    tw = kpop.Twister.load(T, cols, k)
    for normalize in (True, False):
        want = oracle.twist(T, cols, h, c.astype(np.float64), o, normalize)
        got = tw.count_twist(bases, offs, normalize=normalize)
        for r in range(len(seqs)):
            scale = max(np.max(np.abs(want[r])), 1e-300) if normalize else max(np.max(np.abs(want[r])), 1.0)
            assert np.max(np.abs(got[r] - want[r])) <= 1e-12 * scale * (1 if normalize else 100), (r, normalize)
    # determinism of the segment combine
    assert np.array_equal(tw.count_twist(bases, offs), tw.count_twist(bases, offs))


def test_count_twist_many_genomes(kpop, oracle):
    """A batch of 200 x 30 kb genomes: sum over instances / acc vs the oracle's dedupe-normalise-multiply."""
    k, d, n, L = 12, 64, 200, 30000
    bases, offs = oracle.synth_reads(77, n, L)
    tw = kpop.Twister.synth(0x5EED, k, d)
    got = tw.count_twist(bases, offs)
    pick = [0, 57, 199]
    sb, so = concat([bytes(bases[int(offs[r]):int(offs[r + 1])]).decode() for r in pick])
    h, c, o = oracle.count_reads(sb, so, k)
    cols = np.unique(h)
    T = oracle.synth_twister(0x5EED, d, cols)
    want = oracle.twist(T, cols, h, c.astype(np.float64), o)
    assert np.max(np.abs(got[pick] - want)) <= 1e-12 * np.max(np.abs(want))


def test_device_pipeline_all_vs_all_single_rank(kpop, oracle):
    """The device-resident register flow (kpop_amd/pipeline.py) incl. the all-vs-all block, world size 1."""
    import torch
    from kpop_amd.pipeline import DevicePipeline
    k, d, n, L = 8, 64, 300, 120
    dev = torch.device("cuda", 0)
    bases, offs = oracle.synth_reads(4, n, L)
    tw = kpop.Twister.synth(9, k, d)
    metric = kpop.metric_compute(oracle.synth_inertia(d))
    pipe = DevicePipeline(tw, metric, dev)
    t = pipe.count_twist(torch.from_numpy(bases).to(dev), torch.from_numpy(offs.astype(np.int64)).to(dev), L)
    lo, hi, block = pipe.all_vs_all_rows(t, n)
    torch.cuda.synchronize()
    cols = oracle.enumerate_kmers(k)
    T = oracle.synth_twister(9, d, cols)
    h, c, o = oracle.count_reads(bases, offs, k)
    want_t = oracle.twist(T, cols, h, c.astype(np.float64), o)
    assert np.array_equal(t.cpu().numpy(), want_t)
    assert (lo, hi) == (0, n)
    assert np.array_equal(block.cpu().numpy(), oracle.distance_rowwise(want_t, want_t, metric))


@pytest.mark.parametrize("k,d,world", [(6, 5, 2), (8, 33, 3), (9, 64, 8)])
def test_kmer_row_sharded_twister(kpop, oracle, k, d, world):
    """The k = 15 / large-D layout of SURVEY.md 8e at a size the oracle can check: each 'rank' holds the k-mer rows of
    one hash range plus the all-ones dimension, twists every read without normalising, and the summed partials
    divided by the summed `acc` are the twisted rows.  Slices of the device-generated twister and of an uploaded one."""
    import torch

    from kpop_amd.shard import kmer_slice_bounds, reduce_partial_twists
    n, L = 300, 150
    bases, offs = oracle.synth_reads(0x4B506F70, n, L)
    bases = bases.copy()
    bases[offs[3]:offs[4]] = ord("N")
    cols = oracle.enumerate_kmers(k)
    T = oracle.synth_twister(11, d, cols)
    h, c, o = oracle.count_reads(bases, offs, k)
    want = oracle.twist(T, cols, h, c.astype(np.float64), o)
    full = kpop.Twister.synth(11, k, d)
    for make in ("synth", "load"):
        total = np.zeros((n, d + 1))
        rows = 0
        for r in range(world):
            rng_ = kmer_slice_bounds(k, r, world)
            tw = kpop.Twister.synth(11, k, d, hash_range=rng_, acc_dim=True) if make == "synth" else kpop.Twister.load_slice(T, cols, k, rng_)
            info = tw.info()
            assert info["n_dims"] == d + 1
            rows += info["n_cols"]
            total += tw.count_twist(bases, offs, normalize=False)
            tw.free()
        assert rows == len(cols)
        got = reduce_partial_twists(torch.from_numpy(total)).numpy()   # world size 1: no collective, the division only
        assert np.all(got[3] == 0.0)
        np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-15)
    # the accumulator column is the number of k-mers of the read that the slice knows: they add up to all windows
    assert np.array_equal(total[:, -1], np.array([c[o[i]:o[i + 1]].sum() for i in range(n)], dtype=np.float64))
    np.testing.assert_allclose(full.count_twist(bases, offs), want, rtol=1e-12, atol=1e-15)


@pytest.mark.parametrize("k,d,world", [(9, 16, 3), (10, 5, 2), (9, 31, 4)])
def test_a_slice_of_the_twister_keeps_its_rows_at_their_hashes(kpop, oracle, k, d, world):
    """a k-mer-row shard of up to 32 dimensions (a rank of BASELINE config 5's multi-GPU form) also keeps its rows at the address the
    hash names, for its own range of hashes: a window whose k-mer is another rank's is not looked at, one of its own costs one
    miss (kpop_tune("direct", 1) builds the table at any k).  Against the index slice by slice (1e-12: the additions are grouped
    differently), the same windows counted; the slices' sum against the oracle."""
    from kpop_amd import api
    from kpop_amd.shard import kmer_slice_bounds
    n, L = 400, 150
    bases, offs = oracle.synth_reads(0x4B506F70, n, L)
    bases = bases.copy()
    bases[offs[7]:offs[8]] = ord("N")
    cols = oracle.enumerate_kmers(k)
    T = oracle.synth_twister(11, d, cols)
    h, c, o = oracle.count_reads(bases, offs, k)
    want = oracle.twist(T, cols, h, c.astype(np.float64), o)
    total = np.zeros((n, d + 1))
    for r in range(world):
        rng_ = kmer_slice_bounds(k, r, world)
        part = {}
        for mode in (1, 0):
            api.tune("direct", mode)
            try:
                tw = kpop.Twister.synth(11, k, d, hash_range=rng_, acc_dim=True)
            finally:
                api.tune("direct", 2)
            assert (tw.info()["direct_bytes"] > 0) == (mode == 1)
            if mode == 1:
                assert tw.info()["direct_bytes"] == (rng_[1] - rng_[0]) * ((d + 1 + 15) // 16 * 16) * 8  # (its own hashes only, rows padded to 16 doubles)
            part[mode] = tw.count_twist(bases, offs, normalize=False)
            tw.free()
        # (through the index the k-mers of other ranks drop out of the list before the rows are dealt to the lane groups, at their hashes
        # they stay in it as rows that add nothing: another grouping of the same additions)
        np.testing.assert_allclose(part[1], part[0], rtol=1e-12, atol=1e-14)
        assert np.array_equal(part[0][:, -1], part[1][:, -1])  # (the windows counted: integers)
        total += part[1]
    acc = total[:, -1:]
    got = np.where(acc != 0, total[:, :-1] / np.where(acc != 0, acc, 1.0), total[:, :-1])
    assert np.all(got[7] == 0.0)
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-15)


def test_row_sharded_pipeline_with_rccl_world_size_1(kpop, oracle):
    """count_twist_row_sharded through torch.distributed's nccl (= RCCL) backend on one rank: the collective path is the
    one N > 1 ranks take; the result must be the plain twist."""
    import socket

    import torch
    import torch.distributed as dist
    from kpop_amd.pipeline import DevicePipeline
    from kpop_amd.shard import kmer_slice_bounds
    k, d, n, L = 8, 16, 200, 150
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)
    try:
        bases, offs = oracle.synth_reads(21, n, L)
        tw = kpop.Twister.synth(9, k, d, hash_range=kmer_slice_bounds(k, 0, 1), acc_dim=True)
        pipe = DevicePipeline(tw, kpop.metric_compute(oracle.synth_inertia(d)), dev, row_sharded=True)
        t = pipe.count_twist_row_sharded(torch.from_numpy(bases).to(dev), torch.from_numpy(offs.astype(np.int64)).to(dev), L)
        dist.barrier()
        torch.cuda.synchronize()
        cols = oracle.enumerate_kmers(k)
        h, c, o = oracle.count_reads(bases, offs, k)
        want = oracle.twist(oracle.synth_twister(9, d, cols), cols, h, c.astype(np.float64), o)
        assert t.shape == (n, d)
        np.testing.assert_allclose(t.cpu().numpy(), want, rtol=1e-12, atol=1e-15)
        with pytest.raises(ValueError):
            pipe.count_twist(torch.from_numpy(bases).to(dev), torch.from_numpy(offs.astype(np.int64)).to(dev), L)
    finally:
        if created:
            dist.destroy_process_group()


def test_config5_k15_full_twister(kpop, oracle):
    """BASELINE config 5 at its full size: every canonical 15-mer (536,870,912 rows x 16 dims, 69 GB of HBM) resident;
    10k reads and three 30 kb genomes twisted; EVERY row checked against the oracle with the twister restricted to the
    batch's own k-mers (the synthetic coefficients are a pure function of (dimension, hash)); the row-sharded layout
    (two hash-range slices with accumulator dimensions) must give the same rows."""
    import torch
    from kpop_amd.shard import kmer_slice_bounds, reduce_partial_twists
    free, total = torch.cuda.mem_get_info(0)
    if free < 150e9:
        pytest.skip("needs 150 GB of free HBM")
    k, d, n, L = 15, 16, 10000, 150
    tw = kpop.Twister.synth(0x5EED, k, d)
    info = tw.info()
    assert info["n_cols"] == 4 ** 15 // 2 and info["device_bytes"] > 68e9
    bases, offs = oracle.synth_reads(0x4B506F70, n, L)
    gb, go = oracle.synth_reads(0xABCDE, 3, 30000)
    allb = np.concatenate([bases, gb])
    allo = np.concatenate([offs, go[1:] + offs[-1]])
    got = tw.count_twist(allb, allo)
    tw.free()
    pick = list(range(n + 3))  # every read and every genome against the oracle
    sb, so = allb, allo
    h, c, o = oracle.count_reads(sb, so, k)
    cols = np.unique(h)
    want = oracle.twist(oracle.synth_twister(0x5EED, d, cols), cols, h, c.astype(np.float64), o)
    # D <= 32 packs two k-mer rows per wavefront load (the order of summation differs from the oracle's), and genomes
    # go through the streaming kernel: rounding only
    np.testing.assert_allclose(got[pick], want, rtol=1e-12, atol=1e-15)
    # properties over all 10k reads: coordinates of a normalised spectrum are convex combinations of coefficients in [-1,1)
    assert np.all(np.abs(got) < 1.0) and np.all(np.isfinite(got))
    total = np.zeros((n + 3, d + 1))
    for r in range(2):
        sl = kpop.Twister.synth(0x5EED, k, d, hash_range=kmer_slice_bounds(k, r, 2), acc_dim=True)
        total += sl.count_twist(allb, allo, normalize=False)
        sl.free()
    sharded = reduce_partial_twists(torch.from_numpy(total)).numpy()
    np.testing.assert_allclose(sharded, got, rtol=1e-12, atol=1e-15)
    assert np.array_equal(total[:n, -1], np.full(n, L - k + 1.0))           # every window of an N-free read is counted once


def test_config3_50k_genomes_full_size(kpop, oracle):
    """BASELINE config 3 at its full size: 50,000 sequences of 30 kb (1.5 GB of bases, generated on the device), k = 12,
    D = 64.  A sample of genomes -- first, last and a spread -- is copied back and checked against the oracle with the
    twister restricted to the sample's k-mers; every row must be finite and inside the coefficient range."""
    import torch
    from kpop_amd import api
    k, d, n, L = 12, 64, 50000, 30000
    dev = torch.device("cuda", 0)
    sp = torch.cuda.current_stream().cuda_stream
    tw = kpop.Twister.synth(0x7457, k, d)
    bases = torch.empty(n * L, dtype=torch.uint8, device=dev)
    offs = torch.empty(n + 1, dtype=torch.int64, device=dev)
    api.dev_synth_reads(0xC0FFEE, n, L, bases.data_ptr(), offs.data_ptr(), stream=sp)
    out = torch.zeros(n, d, dtype=torch.float64, device=dev)
    api.dev_count_twist(tw, bases.data_ptr(), offs.data_ptr(), n, n * L, L, out.data_ptr(), stream=sp)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    assert np.all(np.isfinite(got)) and np.all(np.abs(got) < 1.0)
    pick = sorted(set([0, 1, 777, 24999, 49998, 49999] + list(range(3, n, 251))))  # 200+ genomes spread over the batch
    sb = np.concatenate([bases[r * L:(r + 1) * L].cpu().numpy() for r in pick])
    so = np.arange(len(pick) + 1, dtype=np.uint64) * L
    first, _ = oracle.synth_reads(0xC0FFEE, 2, L)
    assert np.array_equal(sb[:2 * L], first)                                  # the device generator is the oracle's
    h, c, o = oracle.count_reads(sb, so, k)
    cols = np.unique(h)
    want = oracle.twist(oracle.synth_twister(0x7457, d, cols), cols, h, c.astype(np.float64), o)
    np.testing.assert_allclose(got[pick], want, rtol=1e-12, atol=1e-15)
    assert all(int(o[i + 1] - o[i]) > 29000 for i in range(len(pick)))     # ~29,989 distinct-ish 12-mers per genome
    # ... and EVERY row, through linearity: without normalisation the rows add up to the twist of the batch's merged
    # spectrum (bin/KPopCount.ml:60 over all 50,000 sequences: 1.5 G windows through the LDS-staged histogram), so one
    # number per dimension checks all 3.2 M coordinates
    api.dev_count_twist(tw, bases.data_ptr(), offs.data_ptr(), n, n * L, L, out.data_ptr(), normalize=False, stream=sp)
    torch.cuda.synchronize()
    total = out.sum(dim=0).cpu().numpy()
    hb = bases.cpu().numpy()
    ho = offs.cpu().numpy().astype(np.uint64)
    mh, mc, mo = kpop.count_reads(hb, ho, k, per_read=False, capacity=(4 ** k + 2 ** k) // 2 + 1)
    assert int(mc.astype(np.int64).sum()) == n * (L - k + 1)
    merged = tw.twist(mh, mc.astype(np.float64), mo, normalize=False)[0]
    assert np.max(np.abs(total - merged)) <= 1e-9 * np.max(np.abs(merged)), np.max(np.abs(total - merged))


def test_config3_50k_assemblies_of_one_organism_full_size(kpop, oracle):
    """BASELINE config 3 as it is labelled -- the matrix-core path -- at its full size: 50,000 assemblies of ONE organism (a 30 kb
    genome with 0.3 % substitutions per copy, made on the device), k = 12, D = 64, the default dispatch: 46,000 chunks through
    count_twist_tile_kernel.  A sample of genomes against the oracle (twister restricted to the sample's k-mers); EVERY row through
    linearity (without normalisation the rows add up to the twist of the batch's merged spectrum, itself counted by the LDS
    (hash, count) route); and the same bits from a second call"""
    import torch
    from kpop_amd import api
    k, d, n, L = 12, 64, 50000, 30000
    dev = torch.device("cuda", 0)
    sp = torch.cuda.current_stream().cuda_stream
    tw = kpop.Twister.synth(0x7457, k, d)
    ref = torch.empty(L, dtype=torch.uint8, device=dev)
    ro = torch.empty(2, dtype=torch.int64, device=dev)
    api.dev_synth_reads(0xBEEF, 1, L, ref.data_ptr(), ro.data_ptr(), stream=sp)
    torch.cuda.synchronize()
    bases = ref.repeat(n)
    acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    step = 1 << 27
    for lo in range(0, n * L, step):
        hi = min(n * L, lo + step)
        hit = torch.rand(hi - lo, device=dev, generator=g) < 0.003
        sub = acgt[torch.randint(0, 4, (hi - lo,), device=dev, generator=g)]
        bases[lo:hi] = torch.where(hit, sub, bases[lo:hi])
    offs = torch.arange(n + 1, dtype=torch.int64, device=dev) * L
    out = torch.zeros(n, d, dtype=torch.float64, device=dev)
    api.dev_count_twist(tw, bases.data_ptr(), offs.data_ptr(), n, n * L, L, out.data_ptr(), stream=sp)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    out.zero_()
    api.dev_count_twist(tw, bases.data_ptr(), offs.data_ptr(), n, n * L, L, out.data_ptr(), stream=sp)
    torch.cuda.synchronize()
    assert np.array_equal(got, out.cpu().numpy())
    assert np.all(np.isfinite(got)) and np.all(np.abs(got) < 1.0)
    pick = sorted(set([0, 1, 63, 64, 777, 24999, 49998, 49999] + list(range(3, n, 401))))
    sb = np.concatenate([bases[r * L:(r + 1) * L].cpu().numpy() for r in pick])
    so = np.arange(len(pick) + 1, dtype=np.uint64) * L
    h, c, o = oracle.count_reads(sb, so, k)
    cols = np.unique(h)
    want = oracle.twist(oracle.synth_twister(0x7457, d, cols), cols, h, c.astype(np.float64), o)
    np.testing.assert_allclose(got[pick], want, rtol=1e-12, atol=1e-15)
    api.tune("dense", 0)
    try:
        api.dev_count_twist(tw, bases.data_ptr(), offs.data_ptr(), n, n * L, L, out.data_ptr(), stream=sp)
        torch.cuda.synchronize()
    finally:
        api.tune("dense", 2)
    plain = out.cpu().numpy()
    assert not np.array_equal(plain, got) and np.max(np.abs(plain - got)) <= 1e-12 * np.max(np.abs(plain))  # (the tile route did run, and agrees)
    api.dev_count_twist(tw, bases.data_ptr(), offs.data_ptr(), n, n * L, L, out.data_ptr(), normalize=False, stream=sp)
    torch.cuda.synchronize()
    total = out.sum(dim=0).cpu().numpy()
    mh, mc, mo = kpop.count_reads(bases.cpu().numpy(), offs.cpu().numpy().astype(np.uint64), k, per_read=False, capacity=(4 ** k + 2 ** k) // 2 + 1)
    assert int(mc.astype(np.int64).sum()) == n * (L - k + 1)
    merged = tw.twist(mh, mc.astype(np.float64), mo, normalize=False)[0]
    assert np.max(np.abs(total - merged)) <= 1e-9 * np.max(np.abs(merged)), np.max(np.abs(total - merged))


@pytest.mark.parametrize("k,d,n", [(5, 64, 300), (7, 9, 129), (8, 100, 40), (9, 256, 70), (6, 200, 65)])
def test_dense_twist_on_the_matrix_cores_equals_the_sparse_one(kpop, oracle, k, d, n):
    """kpop_dev_twist_dense (spectra as a dense matrix times the twister's rows, f64 MFMA) against kpop_dev_twist: the same
    sums in the GEMM's order, equal to rounding; unknown k-mers, duplicate lines, an empty spectrum, normalisation on and off"""
    import torch
    from kpop_amd import api
    rng = np.random.RandomState(k * d)
    cols = oracle.enumerate_kmers(k)
    cols = cols[rng.rand(len(cols)) < 0.8]
    tw = kpop.Twister.load(oracle.synth_twister(5, d, cols), cols, k)
    allk = oracle.enumerate_kmers(k)
    hs, vs, offs = [], [], [0]
    for s in range(n):
        m = 0 if s == 3 else int(rng.randint(1, min(len(allk), 600)))
        pick = rng.choice(allk, size=m, replace=True)  # duplicates and k-mers the twister does not hold
        hs.append(pick)
        vs.append(rng.randint(1, 50, size=m).astype(np.float64))
        offs.append(offs[-1] + m)
    h, v, o = np.concatenate(hs).astype(np.uint64), np.concatenate(vs), np.array(offs, dtype=np.uint64)
    dev = torch.device("cuda", 0)
    dh, dv, do = torch.from_numpy(h.view(np.int64)).to(dev), torch.from_numpy(v).to(dev), torch.from_numpy(o.view(np.int64)).to(dev)
    work = torch.empty(api.dev_twist_dense_workspace_bytes(tw, n), dtype=torch.uint8, device=dev)
    for normalize in (True, False):
        a = torch.zeros(n, d, dtype=torch.float64, device=dev)
        b = torch.full((n, d), 7.0, dtype=torch.float64, device=dev)
        api.dev_twist(tw, dh.data_ptr(), dv.data_ptr(), do.data_ptr(), n, 0, a.data_ptr(), normalize=normalize)
        api.dev_twist_dense(tw, dh.data_ptr(), dv.data_ptr(), do.data_ptr(), n, work.data_ptr(), b.data_ptr(), normalize=normalize)
        torch.cuda.synchronize()
        a, b = a.cpu().numpy(), b.cpu().numpy()
        assert np.all(b[3] == 0.0)
        assert np.max(np.abs(a - b)) <= 1e-12 * max(np.max(np.abs(a)), 1.0)
        want = oracle.twist(oracle.synth_twister(5, d, cols), cols, h, v, o, normalize=normalize)
        assert np.max(np.abs(b - want)) <= 1e-12 * max(np.max(np.abs(want)), 1.0)
    # the fused form takes lines that ascend by hash (repeats and unknown k-mers stay): sort every spectrum's lines
    order = np.concatenate([offs[i] + np.argsort(h[offs[i]:offs[i + 1]], kind="stable") for i in range(n)]) if len(h) else np.zeros(0, dtype=np.int64)
    hs_, vs_ = h[order], v[order]
    dh2, dv2 = torch.from_numpy(hs_.view(np.int64)).to(dev), torch.from_numpy(vs_).to(dev)
    for normalize in (True, False):
        c = torch.full((n, d), 7.0, dtype=torch.float64, device=dev)
        api.dev_twist_dense_sorted(tw, dh2.data_ptr(), dv2.data_ptr(), do.data_ptr(), n, work.data_ptr(), c.data_ptr(), normalize=normalize)
        torch.cuda.synchronize()
        c = c.cpu().numpy()
        want = oracle.twist(oracle.synth_twister(5, d, cols), cols, hs_, vs_, o, normalize=normalize)
        assert np.all(c[3] == 0.0)
        assert np.max(np.abs(c - want)) <= 1e-12 * max(np.max(np.abs(want)), 1.0), np.max(np.abs(c - want))
    # lines that do NOT ascend: that spectrum (and only it) comes back as NaNs
    bad = next(i for i in range(n) if offs[i + 1] - offs[i] >= 3 and len(np.unique(hs_[offs[i]:offs[i + 1]])) >= 2)
    hb = hs_.copy()
    seg = hb[offs[bad]:offs[bad + 1]]
    seg[0], seg[-1] = seg[-1], seg[0]
    c = torch.zeros(n, d, dtype=torch.float64, device=dev)
    api.dev_twist_dense_sorted(tw, torch.from_numpy(hb.view(np.int64)).to(dev).data_ptr(), dv2.data_ptr(), do.data_ptr(), n, work.data_ptr(), c.data_ptr())
    torch.cuda.synchronize()
    c = c.cpu().numpy()
    assert np.isnan(c[bad]).all() and not np.isnan(np.delete(c, bad, axis=0)).any()


@pytest.mark.parametrize("k,d,content", [(5, 64, 0), (7, 9, 0), (8, 64, 0), (7, 130, 1), (6, 256, 0)])
def test_count_twist_through_the_dense_image(kpop, oracle, k, d, content):
    """kpop_dev_count_twist_dense: sequences -> one u32 counter per twister row in LDS -> the contraction on the f64 matrix
    cores, against the oracle (count + twist) and the fused sparse kernels: genomes, short and empty sequences, Ns, a twister
    that lacks some k-mers, normalisation on and off"""
    import torch
    from kpop_amd import api
    rng = np.random.RandomState(k * 7 + d)
    seqs = ["", "ACG", "N" * 50, "ACGT" * 3] + ["".join(rng.choice(list("ACGTN"), size=int(n), p=[.2475] * 4 + [.01])) for n in rng.randint(k, 9000, size=90)]
    bases, offs = concat(seqs)
    cols = oracle.enumerate_kmers(k, content)
    if k >= 7:
        cols = cols[rng.rand(len(cols)) < 0.9]
    T = oracle.synth_twister(11, d, cols)
    tw = kpop.Twister.load(T, cols, k)
    dev = torch.device("cuda", 0)
    db, do = torch.from_numpy(bases).to(dev), torch.from_numpy(offs.view(np.int64)).to(dev)
    n = len(seqs)
    work = torch.empty(api.dev_count_twist_dense_workspace_bytes(tw, n), dtype=torch.uint8, device=dev)
    h, c, o = oracle.count_reads(bases, offs, k, content)
    for normalize in (True, False):
        out = torch.full((n, d), 7.0, dtype=torch.float64, device=dev)
        api.dev_count_twist_dense(tw, db.data_ptr(), do.data_ptr(), n, work.data_ptr(), out.data_ptr(), content=content, normalize=normalize)
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        want = oracle.twist(T, cols, h, c.astype(np.float64), o, normalize=normalize)
        assert np.max(np.abs(got - want)) <= 1e-12 * max(np.max(np.abs(want)), 1.0), np.max(np.abs(got - want))
        assert np.all(got[0] == 0.0) and np.all(got[2] == 0.0)
    if k == 8:  # the host entry point picks the dense image for batches of assemblies (the default; kpop_tune("dense", 0) opts out)
        long_ = [s_ for s_ in seqs if len(s_) > 7000] * 8
        lb, lo = concat(long_)
        got = tw.count_twist(lb, lo)
        api.tune("dense", 0)
        try:
            ref = tw.count_twist(lb, lo)
        finally:
            api.tune("dense", 2)
        assert len(long_) >= 64 and np.max(np.abs(got - ref)) <= 1e-12 * np.max(np.abs(ref)) and not np.array_equal(got, ref)
    big = kpop.Twister.synth(1, 9, 8)
    with pytest.raises(kpop.KPopError):  # 131,072 k-mers: no dense image
        api.dev_count_twist_dense(big, db.data_ptr(), do.data_ptr(), n, work.data_ptr(), out.data_ptr())


@pytest.mark.parametrize("k,d,rate", [(12, 64, 0.002), (10, 40, 0.001), (13, 130, 0.003), (12, 64, 0.01), (12, 64, 0.03), (11, 16, 0.1)])
def test_assemblies_of_one_organism_through_the_tile_kernel(kpop, oracle, k, d, rate):
    """The DEFAULT path for sequences of more than 512 windows: stretches of 64 assemblies go through count_twist_tile_kernel
    (the rows of four seed sequences in an LDS set = the consensus, counted into an LDS matrix and multiplied on the matrix
    cores; the rows private to a sequence listed and gathered by tile_residual_kernel), stretches that share little with
    their seeds and unrelated sequences are left to the streaming kernel, short reads to the wave kernel -- against the
    oracle and against kpop_tune("dense", 0), normalised or not, at divergences from 0.1 % to 10 % (lib/Twister.ml:146-188)"""
    from kpop_amd import api
    rng = np.random.RandomState(k + d)
    ref = rng.choice(list("ACGT"), size=7000)
    seqs = []
    for i in range(150):  # mutants: substitutions, a few Ns, some with a deletion (the windows shift), ragged ends
        m = ref.copy()
        hit = rng.rand(len(m)) < rate
        m[hit] = rng.choice(list("ACGTN"), size=int(hit.sum()), p=[.24, .24, .24, .24, .04])
        if i % 7 == 0:
            cut = int(rng.randint(100, 6000))
            m = np.concatenate([m[:cut], m[cut + 5:]])
        if i % 31 == 5:
            m[1000:1700] = "N"  # a masked stretch (an amplicon that dropped out)
        seqs.append("".join(m[: len(m) - int(rng.randint(0, 900))]))
    seqs += ["".join(rng.choice(list("ACGT"), size=int(rng.randint(3000, 9000)))) for _ in range(70)]  # unrelated: nothing in common with the seeds
    seqs += ["ACGTTGCA" * 10, "", "ACG", "A" * 2000] + ["".join(rng.choice(list("ACGT"), size=150)) for _ in range(20)]
    order = rng.permutation(len(seqs))
    seqs = [seqs[i] for i in order[:40]] + seqs[:150] + [seqs[i] for i in order[40:] if i >= 150]  # a run of mutants in the middle
    bases, offs = concat(seqs)
    h, c, o = oracle.count_reads(bases, offs, k)
    cols = np.unique(h)
    cols = cols[rng.rand(len(cols)) < 0.95]
    allc = oracle.enumerate_kmers(k) if k <= 10 else None
    if allc is not None:
        cols = allc
    T = oracle.synth_twister(3, d, cols)
    tw = kpop.Twister.load(T, cols, k)
    for normalize in (True, False):
        want = oracle.twist(T, cols, h, c.astype(np.float64), o, normalize=normalize)
        got = tw.count_twist(bases, offs, normalize=normalize)
        again = tw.count_twist(bases, offs, normalize=normalize)
        api.tune("dense", 0)
        try:
            ref_rows = tw.count_twist(bases, offs, normalize=normalize)
        finally:
            api.tune("dense", 2)
        scale = max(np.max(np.abs(want)), 1.0)
        assert np.max(np.abs(ref_rows - want)) <= 1e-12 * scale
        assert np.max(np.abs(got - want)) <= 1e-12 * scale, np.max(np.abs(got - want))
        assert np.array_equal(got, again)  # (no order of additions depends on the run)
        if rate <= 0.03:
            assert not np.array_equal(got, ref_rows)  # (another order of additions: the tile kernel did run)
        # (at 10 % fewer than a third of the windows survive: tile_group_probe_kernel leaves the groups to the streaming kernel)


def _one_organism(rng, n, L, rate, unrelated=8, indel_every=7):
    """n mutants of one random L-base sequence (substitutions at `rate`, a few Ns, deletions, ragged ends) + a few strangers and short reads"""
    ref = rng.choice(list("ACGT"), size=L)
    seqs = []
    for i in range(n):
        m = ref.copy()
        hit = rng.rand(len(m)) < rate
        m[hit] = rng.choice(list("ACGTN"), size=int(hit.sum()), p=[.24, .24, .24, .24, .04])
        if indel_every and i % indel_every == 0:
            cut = int(rng.randint(100, L - 1000))
            m = np.concatenate([m[:cut], m[cut + 5:]])
        if i % 31 == 5:
            m[1000:1700] = "N"
        seqs.append("".join(m[: len(m) - int(rng.randint(0, 900))]))
    seqs += ["".join(rng.choice(list("ACGT"), size=int(rng.randint(3000, 5000)))) for _ in range(unrelated)]
    seqs += ["ACGTTGCA" * 10, "", "ACG", "A" * 2000] + ["".join(rng.choice(list("ACGT"), size=150)) for _ in range(20)]
    order = rng.permutation(len(seqs))
    return [seqs[i] for i in order[:20]] + seqs[:n] + [seqs[i] for i in order[20:] if i >= n]


@pytest.mark.parametrize("k,d,rate", [(12, 256, 0.003), (10, 1635, 0.002), (11, 100, 0.01), (12, 72, 0.03), (13, 130, 0.001), (9, 65, 0.003)])
def test_assemblies_through_more_than_64_dimensions(kpop, oracle, k, d, rate):
    """beyond 64 dimensions the pipelined tile kernel runs three stages a block (tile_pipe.h, WIDE): producers prepare the next chunk, four
    MFMA wavefronts multiply the current one's X against unit after unit of 16 columns of the members' rows, four gather wavefronts add
    the chunk before's residual rows to its slots -- at the reference's own 1,635 dimensions (README.md:1029), at 256, at widths that end inside
    a slab (100, 72, 130) and one column past a slab (65): against the oracle, against the streaming kernel, the same bits twice,
    and against round 4's kernel (phases one after the other, the residual rows in a launch of their own).  lib/Twister.ml:146-188"""
    from kpop_amd import api
    rng = np.random.RandomState(1000 * k + d)
    seqs = _one_organism(rng, 150, 7000, rate)
    bases, offs = concat(seqs)
    h, c, o = oracle.count_reads(bases, offs, k)
    cols = np.unique(h)
    cols = cols[rng.rand(len(cols)) < 0.95]
    T = oracle.synth_twister(3, d, cols)
    tw = kpop.Twister.load(T, cols, k)
    for normalize in (True, False):
        want = oracle.twist(T, cols, h, c.astype(np.float64), o, normalize=normalize)
        got = tw.count_twist(bases, offs, normalize=normalize)
        again = tw.count_twist(bases, offs, normalize=normalize)
        api.tune("dense", 0)
        try:
            ref_rows = tw.count_twist(bases, offs, normalize=normalize)
        finally:
            api.tune("dense", 2)
        scale = max(np.max(np.abs(want)), 1.0)
        assert np.max(np.abs(ref_rows - want)) <= 1e-12 * scale
        assert np.max(np.abs(got - want)) <= 1e-12 * scale, np.max(np.abs(got - want))
        assert np.array_equal(got, again)
        assert not np.array_equal(got, ref_rows)  # (another order of additions: the tile kernel did run)
    api.tune("tilepipe", 0)
    try:
        r4 = tw.count_twist(bases, offs, normalize=False)
    finally:
        api.tune("tilepipe", 1)
    assert np.max(np.abs(r4 - got)) <= 1e-12 * scale
    tw.free()


@pytest.mark.parametrize("k,d,rate", [(12, 64, 0.002), (10, 40, 0.01), (12, 64, 0.03)])
def test_the_slab_by_slab_kernel_at_up_to_64_dimensions_gives_the_other_kernels_bits(kpop, oracle, k, d, rate):
    """kpop_tune("tilewide", 1) sends a twister of up to 64 dimensions through the three-stage kernel (MFMA wavefronts + gather
    wavefronts): the members' rows are multiplied in the same order and a sequence's residual rows added in the same (window) order, so
    the rows must equal the exchanging kernel's bit for bit -- the hand-overs between the three stages, the sums leaving from the
    accumulators' registers and their meeting the gather's in the slots all have to be right for that."""
    from kpop_amd import api
    rng = np.random.RandomState(k * d)
    seqs = _one_organism(rng, 200, 9000, rate)
    bases, offs = concat(seqs)
    h, c, o = oracle.count_reads(bases, offs, k)
    cols = np.unique(h)
    cols = cols[rng.rand(len(cols)) < 0.95]
    T = oracle.synth_twister(5, d, cols)
    tw = kpop.Twister.load(T, cols, k)
    narrow = tw.count_twist(bases, offs)
    api.tune("tilewide", 1)
    try:
        wide = tw.count_twist(bases, offs)
    finally:
        api.tune("tilewide", 0)
    want = oracle.twist(T, cols, h, c.astype(np.float64), o)
    assert np.max(np.abs(narrow - want)) <= 1e-12 * max(np.max(np.abs(want)), 1.0)
    assert np.array_equal(narrow, wide)
    tw.free()


def test_a_batch_past_the_tile_routes_bound_goes_through_in_sub_batches(kpop, oracle):
    """9,000 assemblies of 6 kb through 72 dimensions with the bound on the route's per-slot tables lowered to 40 MiB
    (kpop_tune("tilecap_mb")): two sub-batches of sequences, each a call of its own -- rows against the streaming kernel's, a
    sample against the oracle, the same bits twice"""
    import torch
    from kpop_amd import api
    k, d, n, L = 11, 72, 9000, 6000
    dev = torch.device("cuda", 0)
    sp = torch.cuda.current_stream().cuda_stream
    tw = kpop.Twister.synth(0x51AB, k, d)
    ref = torch.empty(L, dtype=torch.uint8, device=dev)
    ro = torch.empty(2, dtype=torch.int64, device=dev)
    api.dev_synth_reads(0xBEEF, 1, L, ref.data_ptr(), ro.data_ptr(), stream=sp)
    torch.cuda.synchronize()
    acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    g = torch.Generator(device=dev)
    g.manual_seed(11)
    hit = torch.rand(n * L, device=dev, generator=g) < 0.004
    bases = torch.where(hit, acgt[torch.randint(0, 4, (n * L,), device=dev, generator=g)], ref.repeat(n))
    offs = torch.arange(n + 1, dtype=torch.int64, device=dev) * L
    out = torch.zeros(n, d, dtype=torch.float64, device=dev)

    def run():
        out.fill_(float("nan"))
        api.dev_count_twist(tw, bases.data_ptr(), offs.data_ptr(), n, n * L, L, out.data_ptr(), stream=sp)
        torch.cuda.synchronize()
        return out.clone()

    whole = run()
    api.tune("tilecap_mb", 40)
    try:
        split, split2 = run(), run()
    finally:
        api.tune("tilecap_mb", 0)
    api.tune("dense", 0)
    try:
        stream_rows = run()
    finally:
        api.tune("dense", 2)
    assert torch.equal(split, split2)
    assert not torch.equal(whole, stream_rows)
    scale = float(stream_rows.abs().max())
    assert float((whole - stream_rows).abs().max()) <= 1e-12 * scale
    assert float((split - stream_rows).abs().max()) <= 1e-12 * scale
    m = (40 << 20) // ((L // 512 + 2) * (d * 8 + 16))  # sequences a sub-batch (count_twist.hip)
    assert 4096 <= m < n and not torch.equal(split, whole)  # (other groups share a consensus set: another order of additions)
    pick = [0, 1, m - 1, m, m + 1, n - 1]
    sb = np.concatenate([bases[r * L:(r + 1) * L].cpu().numpy() for r in pick])
    so = np.arange(len(pick) + 1, dtype=np.uint64) * L
    h, c, o = oracle.count_reads(sb, so, k)
    cols = np.unique(h)
    want = oracle.twist(oracle.synth_twister(0x51AB, d, cols), cols, h, c.astype(np.float64), o)
    np.testing.assert_allclose(split.cpu().numpy()[pick], want, rtol=1e-12, atol=1e-15)
    tw.free()


@pytest.mark.parametrize("d", [64, 136])
def test_assemblies_with_low_complexity_runs_and_shifted_copies(kpop, oracle, d):
    """what the pipelined tile kernel (tile_pipe.h, up to 64 dimensions) leaves to the streaming kernel or takes the slow way:
    stretches in which one k-mer occurs more than 255 times (a poly-A of 400, an (AT)n of 500, in every mutant: the one-byte
    counts of X would overflow -- the row-sum check must send those chunks away, not return garbage), and assemblies that carry
    an insertion early on (every window after it differs from the set's reference at its coordinate: hashed one by one) --
    against the oracle, and the same bits twice"""
    from kpop_amd import api
    rng = np.random.RandomState(5)
    k = 12  # (d = 136: the three-stage kernel of more than 64 dimensions -- a chunk it turns away is never handed to its MFMA and gather wavefronts)
    ref = rng.choice(list("ACGT"), size=9000)
    ref[1500:1900] = "A"
    ref[4000:4500] = list("AT" * 250)
    ref[6100:6400] = list("ACG" * 100)  # (period three: 100 of a k-mer per strand at most, no overflow)
    seqs = []
    for i in range(200):
        m = ref.copy()
        hit = rng.rand(len(m)) < 0.002
        m[hit] = rng.choice(list("ACGT"), size=int(hit.sum()))
        if i % 5 == 1:
            at = int(rng.randint(50, 400))
            m = np.concatenate([m[:at], rng.choice(list("ACGT"), size=int(rng.randint(1, 9))), m[at:]])  # an insertion: shifted from there on
        seqs.append("".join(m))
    bases, offs = concat(seqs)
    h, c, o = oracle.count_reads(bases, offs, k)
    cols = np.unique(h)
    cols = cols[rng.rand(len(cols)) < 0.97]  # (some k-mers have no row: members without one, misses without one)
    T = oracle.synth_twister(8, d, cols)
    tw = kpop.Twister.load(T, cols, k)
    want = oracle.twist(T, cols, h, c.astype(np.float64), o)
    got, again = tw.count_twist(bases, offs), tw.count_twist(bases, offs)
    api.tune("tilepipe", 0)
    try:
        r4 = tw.count_twist(bases, offs)  # round 4's kernel, phases one after the other
    finally:
        api.tune("tilepipe", 1)
    scale = max(np.max(np.abs(want)), 1.0)
    assert np.max(np.abs(got - want)) <= 1e-12 * scale, np.max(np.abs(got - want))
    assert np.max(np.abs(r4 - want)) <= 1e-12 * scale
    assert np.array_equal(got, again)


@pytest.mark.parametrize("d", [24, 90])
def test_few_assemblies_among_many_reads_and_tiny_batches(kpop, oracle, d):
    """the tile route's groups are cut from the sequences that HAVE segments: three assemblies of one organism among 3,000 reads
    (fewer than the route bothers with), then forty of them scattered among the reads (one group), against the oracle"""
    rng = np.random.RandomState(77)
    k = 11
    ref = rng.choice(list("ACGT"), size=5000)
    def mutant():
        m = ref.copy()
        hit = rng.rand(len(m)) < 0.004
        m[hit] = rng.choice(list("ACGT"), size=int(hit.sum()))
        return "".join(m)
    reads = ["".join(rng.choice(list("ACGT"), size=150)) for _ in range(3000)]
    cols = oracle.enumerate_kmers(k)
    T = oracle.synth_twister(9, d, cols)
    tw = kpop.Twister.load(T, cols, k)
    for n_asm in (3, 40):
        seqs = list(reads)
        for i in range(n_asm):
            seqs.insert(int(rng.randint(0, len(seqs))), mutant())
        bases, offs = concat(seqs)
        h, c, o = oracle.count_reads(bases, offs, k)
        want = oracle.twist(T, cols, h, c.astype(np.float64), o)
        got = tw.count_twist(bases, offs)
        assert np.max(np.abs(got - want)) <= 1e-12 * max(np.max(np.abs(want)), 1.0)


@pytest.mark.parametrize("d", [64, 200])
def test_first_sequence_of_a_group_is_the_odd_one_out(kpop, oracle, d):
    """sequence 0 of a group seeds the consensus set; when it is a stranger (a contaminant among 80 assemblies of one organism)
    the other seeds find fewer than half of their rows there and the set is started again from the next seed: the group still goes
    through the tile kernel (another order of additions than kpop_tune("dense", 0)), and the stranger's own rows are right"""
    from kpop_amd import api
    rng = np.random.RandomState(21)
    k = 12
    ref = rng.choice(list("ACGT"), size=6000)
    def mutant():
        m = ref.copy()
        hit = rng.rand(len(m)) < 0.003
        m[hit] = rng.choice(list("ACGT"), size=int(hit.sum()))
        return "".join(m)
    seqs = ["".join(rng.choice(list("ACGT"), size=6000))] + [mutant() for _ in range(80)]
    bases, offs = concat(seqs)
    h, c, o = oracle.count_reads(bases, offs, k)
    cols = np.unique(h)
    T = oracle.synth_twister(4, d, cols)
    tw = kpop.Twister.load(T, cols, k)
    want = oracle.twist(T, cols, h, c.astype(np.float64), o)
    got = tw.count_twist(bases, offs)
    api.tune("dense", 0)
    try:
        plain = tw.count_twist(bases, offs)
    finally:
        api.tune("dense", 2)
    scale = max(np.max(np.abs(want)), 1.0)
    assert np.max(np.abs(got - want)) <= 1e-12 * scale and np.max(np.abs(plain - want)) <= 1e-12 * scale
    assert not np.array_equal(got[1:64], plain[1:64])  # (the mutants of group 0 went through the tile kernel)


def test_understated_max_len_yields_nan_rows_not_stale_memory(kpop, oracle):
    """kpop_dev_count_twist trusts the caller's max_len to schedule the long-sequence pass; a read longer than it says
    must come back as NaNs, and kpop_dev_distance_rowwise refuses a null workspace for very long rows"""
    import torch
    from kpop_amd import api
    k, d = 10, 64
    tw = kpop.Twister.synth(3, k, d)
    seqs = ["ACGT" * 30, "ACGT" * 400, "TTGACC" * 20]
    bases, offs = concat(seqs)
    dev = torch.device("cuda", 0)
    b, o = torch.from_numpy(bases).to(dev), torch.from_numpy(offs.view(np.int64)).to(dev)
    out = torch.full((3, d), 7.0, dtype=torch.float64, device=dev)
    api.dev_count_twist(tw, b.data_ptr(), o.data_ptr(), 3, b.numel(), 150, out.data_ptr())  # 150 < 1600
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    assert np.isnan(got[1]).all() and not np.isnan(got[0]).any() and not np.isnan(got[2]).any()
    api.dev_count_twist(tw, b.data_ptr(), o.data_ptr(), 3, b.numel(), 1600, out.data_ptr())
    torch.cuda.synchronize()
    assert not np.isnan(out.cpu().numpy()).any()
    m = torch.zeros(2, 40000, dtype=torch.float64, device=dev)
    metric = torch.ones(40000, dtype=torch.float64, device=dev)
    res = torch.zeros(2, 2, dtype=torch.float64, device=dev)
    with pytest.raises(kpop.KPopError):
        api.dev_distance_rowwise(m.data_ptr(), 2, m.data_ptr(), 2, 40000, metric.data_ptr(), None, res.data_ptr(), normalize=False)


@pytest.mark.parametrize("d", [16, 24, 5])
def test_count_twist_with_the_rows_at_their_hashes(kpop, oracle, d):
    """A nearly complete twister of large k and few dimensions also keeps its rows at the address the hash names (twister.h,
    `direct`): the fused kernel for reads then looks nothing up.  k = 13: (a) every canonical 13-mer (the layout is built by its
    own rule) -- the same bits as without it (kpop_tune("direct", 0)); (b) a twister that knows SOME of the k-mers that occur
    (forced: kpop_tune("direct", 1)) -- rows that do not exist must drop out of the sum AND of the normaliser; reads with Ns,
    reads that hold nothing the twister knows, and assemblies (the streaming kernel keeps the index) in the batch; against the
    oracle, normalised and not."""
    from kpop_amd import api
    k = 13
    rng = np.random.RandomState(d)
    bases, offs = oracle.synth_reads(77 + d, 400, 150)
    bases = bases.copy()
    bases[offs[3] + 40] = ord("N")
    bases[offs[5]:offs[6]] = ord("N")  # a read of Ns: no window at all
    genomes = ["".join(rng.choice(list("ACGT"), size=int(n))) for n in (2500, 900)]
    gb, go = concat(genomes)
    allb = np.concatenate([bases, gb])
    allo = np.concatenate([offs, offs[-1] + go[1:]])
    h, c, o = oracle.count_reads(allb, allo, k)
    try:
        # (a) complete
        res = {}
        for mode in (2, 0):
            api.tune("direct", mode)
            tw = kpop.Twister.synth(0x5EED, k, d)
            assert (tw.info()["direct_bytes"] > 0) == (mode == 2)
            res[mode] = tw.count_twist(allb, allo)
            tw.free()
        assert np.array_equal(res[2], res[0])
        cols = np.unique(h)
        want = oracle.twist(oracle.synth_twister(0x5EED, d, cols), cols, h, c.astype(np.float64), o)
        assert_close(res[2], want)
        # (b) partial, forced
        present = np.unique(h)
        keep = present[rng.rand(len(present)) < 0.6]
        gone_read = np.unique(h[o[7]:o[8]])  # read 7 holds nothing the twister knows
        keep = np.setdiff1d(keep, gone_read)
        cols = np.unique(np.concatenate([keep, np.array([0, (1 << (2 * k)) - 1], dtype=np.uint64)]).astype(np.uint64))
        cols = cols[rng.permutation(len(cols))]
        T = oracle.synth_twister(5, d, cols)
        for normalize in (True, False):
            want = oracle.twist(T, cols, h, c.astype(np.float64), o, normalize=normalize)
            got = {}
            for mode in (1, 0):
                api.tune("direct", mode)
                tw = kpop.Twister.load(T, cols, k)
                assert (tw.info()["direct_bytes"] > 0) == (mode == 1)
                got[mode] = tw.count_twist(allb, allo, normalize=normalize)
                tw.free()
            assert_close(got[1], want)
            assert_close(got[0], want)
            assert np.all(got[1][7] == 0.0) and np.all(got[1][5] == 0.0)
    finally:
        api.tune("direct", 2)


@pytest.mark.parametrize("rate,d", [(0.003, 64), (0.01, 64), (0.003, 256)])
def test_many_chunks_a_block_give_the_same_bits_every_call(kpop, oracle, rate, d):
    """8,000 assemblies of one 30 kb organism through the pipelined tile kernel, five calls: every block works through some thirty
    chunks with its producers two chunks ahead of its consumers and the consumers' halves out of step -- where a hand-over that
    is wrong shows (one was: the release counter of round 5's first barriers by half) -- and the bits must be the same every
    time; a sample of rows against the oracle."""
    import torch
    from kpop_amd import api
    k, n, L = 12, 8000, 30000
    dev = torch.device("cuda", 0)
    sp = torch.cuda.current_stream().cuda_stream
    tw = kpop.Twister.synth(0x7457, k, d)
    ref = torch.empty(L, dtype=torch.uint8, device=dev)
    ro = torch.empty(2, dtype=torch.int64, device=dev)
    api.dev_synth_reads(0xF00D, 1, L, ref.data_ptr(), ro.data_ptr(), stream=sp)
    torch.cuda.synchronize()
    bases = ref.repeat(n)
    acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    hit = torch.rand(n * L, device=dev, generator=g) < rate
    bases = torch.where(hit, acgt[torch.randint(0, 4, (n * L,), device=dev, generator=g)], bases)
    offs = torch.arange(n + 1, dtype=torch.int64, device=dev) * L
    out = torch.zeros(n, d, dtype=torch.float64, device=dev)
    first = None
    for _ in range(5):
        out.zero_()
        api.dev_count_twist(tw, bases.data_ptr(), offs.data_ptr(), n, n * L, L, out.data_ptr(), stream=sp)
        torch.cuda.synchronize()
        if first is None:
            first = out.clone()
        else:
            assert torch.equal(first, out)
    pick = [0, 1, 63, 64, 3999, 7998, 7999]
    sb = np.concatenate([bases[r * L:(r + 1) * L].cpu().numpy() for r in pick])
    so = np.arange(len(pick) + 1, dtype=np.uint64) * L
    h, c, o = oracle.count_reads(sb, so, k)
    cols = np.unique(h)
    want = oracle.twist(oracle.synth_twister(0x7457, d, cols), cols, h, c.astype(np.float64), o)
    np.testing.assert_allclose(first.cpu().numpy()[pick], want, rtol=1e-12, atol=1e-15)
    tw.free()


@pytest.mark.parametrize("k,d,tk", [(3, 64, 8), (3, 9, 30), (6, 40, 15), (7, 100, 30), (12, 65, 30), (2, 300, 5)])
def test_protein_through_the_fused_kernels(kpop, oracle, k, d, tk):
    """-C protein (bin/KPopCount.ml:246-248, five bits a residue) through the fused count->twist entry points: the one-wavefront
    kernel for sequences of up to 512 windows, the streaming kernel for longer ones, kpop_spectra_twist (count, then the line-by-line
    twist: bit for bit what the counting and the twisting calls give one after the other) and the pipeline.  The twister is loaded
    with the k its names' width implies (tk: 2 tk bits hold the 5 k of a hash)."""
    from kpop_amd import api
    rng = np.random.RandomState(100 * k + d)
    aa = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWYXBZ*", dtype=np.uint8)
    p = np.array([1.0] * 20 + [0.06, 0.03, 0.03, 0.03])
    p /= p.sum()
    lens = [0, 1, k - 1, k, k + 1, 511 + k, 512 + k, 513 + k, 1500, 9000] + [int(x) for x in rng.randint(0, 600, size=200)]
    prot = [bytes(aa[rng.choice(len(aa), size=max(n, 0), p=p)]).decode() for n in lens]
    bases, offs = concat(prot)
    h, c, o = oracle.count_reads(bases, offs, k, oracle.PROTEIN)
    seen = np.unique(h)
    cols = seen[rng.rand(len(seen)) < 0.7] if len(seen) > 50 else seen
    cols = cols[rng.permutation(len(cols))]
    T = oracle.synth_twister(9, d, cols)
    tw = kpop.Twister.load(T, cols, tk)
    tw.set_count_k(k)
    short = np.array([i for i, n in enumerate(lens) if n - k + 1 <= 512])
    sb, so = concat([prot[i] for i in short])
    for normalize in (True, False):
        want = oracle.twist(T, cols, h, c.astype(np.float64), o, normalize)
        got = tw.count_twist(bases, offs, content=kpop.PROTEIN, normalize=normalize)
        assert_close(got, want)
        api.tune("dense", 0)  # (kpop_twist line by line, not its matrix-core route: what kpop_spectra_twist does on the device)
        line_by_line = tw.twist(h, c.astype(np.float64), o, normalize=normalize)
        api.tune("dense", 2)
        assert np.array_equal(tw.spectra_twist(bases, offs, k, content=kpop.PROTEIN, normalize=normalize), line_by_line)
        got_s = tw.count_twist(sb, so, content=kpop.PROTEIN, normalize=normalize)
        assert_close(got_s, want[short])
        assert np.array_equal(got_s, got[short])  # (the wave kernel's rows do not depend on what else the batch holds)
    assert np.array_equal(got[:3], np.zeros((3, d)))
    pl = kpop.Pipeline(tw, outputs=api.OUT_TWISTED, content=kpop.PROTEIN, normalize_counts=True)
    res = pl.run(sb, so)
    assert_close(res["twisted"], oracle.twist(T, cols, h, c.astype(np.float64), o, True)[short])
    pl.close()
    with pytest.raises(kpop.KPopError):
        tw.set_count_k(k)
        tw.count_twist_packed(*api.pack_bases(sb), so, content=kpop.PROTEIN)  # the packed form is DNA
