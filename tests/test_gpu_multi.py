"""Several GPUs behind the C ABI (kpop_init_devices / kpop_sharded_*, kpop_amd/csrc/multi.hip): one process, one host
thread per device slot, reads cut by kpop_shard_bounds -- the in-library replacement of `-T` (bin/KPopTwistDB.ml:103)
and the fork()ed workers of lib/Twister.ml:90-196.

The GPU tests run n = 1, 2, 3 and 8 device SLOTS ALIASED TO GPU 0 (the boxes these tests see have one GPU): every
code path of the multi-device job runs -- per-slot contexts, threads, replicas, pipelines, the pushes of the all-gather
(device-local copies here, xGMI peer copies on a real node), the barriers -- and only the physical link is missing.
Results must equal the one-device results bit for bit (every row depends on its own read only)."""
import ctypes as C

import numpy as np
import pytest

from conftest import concat


def test_shard_bounds_match_the_python_mirror_and_partition():
    """CPU: kpop_shard_bounds is host arithmetic -- contiguous, balanced, covering, same as kpop_amd/shard.py"""
    import kpop_amd
    from kpop_amd.shard import shard_bounds as py
    for world in (1, 2, 3, 8, 16):
        for n in (0, 1, 2, 7, 8, 9, 100000, 1000003):
            prev = 0
            sizes = []
            for r in range(world):
                lo, hi = kpop_amd.shard_bounds(n, r, world)
                assert (lo, hi) == py(n, r, world) and lo == prev and hi >= lo
                sizes.append(hi - lo)
                prev = hi
            assert prev == n and max(sizes) - min(sizes) <= 1
    with pytest.raises(kpop_amd.KPopError):
        kpop_amd.shard_bounds(10, 3, 3)


def _merge_like_the_library(n, world, rows_of):
    """what kpop_sharded_run must produce: shard r's rows at [lo_r, hi_r)"""
    import kpop_amd
    out = []
    for r in range(world):
        lo, hi = kpop_amd.shard_bounds(n, r, world)
        out.append(rows_of(lo, hi))
    return np.concatenate(out) if out else np.zeros((0,))


def test_shard_merge_logic_on_cpu(oracle):
    """CPU: sharding reads by kpop_shard_bounds and concatenating per-shard oracle results == the oracle on the batch,
    at n = 2, 3, 8 (the merge the library performs by writing shard rows in place)"""
    k, d = 7, 9
    rng = np.random.RandomState(3)
    seqs = ["".join(rng.choice(list("ACGT"), size=int(rng.randint(1, 120)))) for _ in range(101)]
    bases, offs = concat(seqs)
    cols = oracle.enumerate_kmers(k)
    T = oracle.synth_twister(5, d, cols)
    h, c, o = oracle.count_reads(bases, offs, k)
    whole = oracle.twist(T, cols, h, c.astype(np.float64), o)
    for world in (2, 3, 8):
        def rows_of(lo, hi):
            hh, cc, oo = oracle.count_reads(bases, offs[lo:hi + 1], k)
            return oracle.twist(T, cols, hh, cc.astype(np.float64), oo)
        assert np.array_equal(_merge_like_the_library(len(seqs), world, rows_of), whole)


@pytest.fixture
def slots(request):
    import kpop_amd
    n = request.param
    kpop_amd.init_devices([0] * n)
    yield n
    kpop_amd.init(0)


def _job(oracle, k, d, C_, n, seed):
    rng = np.random.RandomState(seed)
    seqs = ["", "AC"] + ["".join(rng.choice(list("ACGTN"), size=int(rng.randint(1, 220)), p=[.2475] * 4 + [.01])) for _ in range(n - 2)]
    bases, offs = concat(seqs)
    cols = oracle.enumerate_kmers(k)
    T = oracle.synth_twister(seed + 1, d, cols)
    cb, co = oracle.synth_reads(seed + 2, C_, 400)
    hc, cc, oc = oracle.count_reads(cb, co, k)
    classes = oracle.twist(T, cols, hc, cc.astype(np.float64), oc)
    metric = oracle.metric_powers(oracle.synth_inertia(d))
    return bases, offs, cols, T, classes, metric


@pytest.mark.gpu
@pytest.mark.parametrize("slots", [1, 2, 3, 8], indirect=True)
def test_sharded_run_equals_one_device_and_oracle(slots, oracle):
    import kpop_amd as kpop
    assert kpop.device_slots() == slots
    k, d, C_, n = 9, 64, 7, 333
    bases, offs, cols, T, classes, metric = _job(oracle, k, d, C_, n, 11)
    tw = kpop.Twister.load(T, cols, k)
    sh = kpop.Sharded(tw, classes, metric, outputs=kpop.OUT_TWISTED | kpop.OUT_DISTANCES | kpop.OUT_SUMMARY, keep_at_most=2,
                      max_neighbours=C_, chunk_reads=17)
    assert sh.slots == slots
    out = sh.run(bases, offs)
    h, c, o = oracle.count_reads(bases, offs, k)
    want_tw = oracle.twist(T, cols, h, c.astype(np.float64), o)
    assert np.array_equal(out["twisted"], want_tw)
    want_di = oracle.distance_rowwise(classes, want_tw, metric)
    assert np.max(np.abs(out["distances"] - want_di) / want_di) <= 1e-12
    one = kpop.Pipeline(tw, classes, metric, outputs=kpop.OUT_TWISTED | kpop.OUT_DISTANCES | kpop.OUT_SUMMARY, keep_at_most=2,
                        max_neighbours=C_).run(bases, offs)
    for name in ("twisted", "distances", "stats", "n_neighbours"):
        assert np.array_equal(out[name], one[name], equal_nan=True), name
    for j in range(n):
        m = min(int(one["n_neighbours"][j]), C_)
        assert np.array_equal(out["nb_index"][j, :m], one["nb_index"][j, :m])
    # host matrices over the slots
    assert np.array_equal(kpop.sharded_distance_rowwise(classes, want_tw, metric), kpop.distance_rowwise(classes, want_tw, metric))
    a = kpop.sharded_distance_summary(classes, want_tw, metric, keep_at_most=2, max_neighbours=C_)
    b = kpop.distance_summary(classes, want_tw, metric, keep_at_most=2, max_neighbours=C_)
    assert np.array_equal(a[0], b[0], equal_nan=True) and np.array_equal(a[1], b[1])
    sh.close()


@pytest.mark.gpu
@pytest.mark.parametrize("slots", [1, 2, 3, 8], indirect=True)
def test_resident_step_all_gather_and_all_vs_all(slots, oracle):
    """config 4 in small: reads resident per slot, twist in chunks with the pushes of the all-gather under it, distances
    to the classes, then the all-vs-all summary on the gathered matrix -- against the one-device entry points"""
    import kpop_amd as kpop
    from kpop_amd import _lib
    lib = _lib.load()
    k, d, C_, n, L = 10, 64, 5, 1003, 150
    tw = kpop.Twister.synth(0x5EED, k, d)
    cb, co = oracle.synth_reads(0xC1A55, C_, 600)
    classes = tw.count_twist(cb, co)
    metric = kpop.metric_compute(oracle.synth_inertia(d))
    sh = kpop.Sharded(tw, classes, metric, outputs=kpop.OUT_TWISTED | kpop.OUT_DISTANCES)
    d_bases, d_offs, n_reads, n_bases = [], [], [], []
    for s in range(slots):
        lo, hi = kpop.shard_bounds(n, s, slots)
        kpop.use_device(s)
        pb, po = C.c_void_p(), C.c_void_p()
        kpop.check(lib.kpop_dev_malloc(C.byref(pb), max((hi - lo) * L, 8)))
        kpop.check(lib.kpop_dev_malloc(C.byref(po), (hi - lo + 1) * 8))
        kpop.check(lib.kpop_dev_synth_reads(0x4B506F70, hi - lo, L, lo, pb, po, None))
        kpop.check(lib.kpop_synchronize(None))
        d_bases.append(pb.value)
        d_offs.append(po.value)
        n_reads.append(hi - lo)
        n_bases.append((hi - lo) * L)
    kpop.use_device(0)
    bases, offs = oracle.synth_reads(0x4B506F70, n, L)
    want_tw = tw.count_twist(bases, offs)
    want_di = kpop.distance_rowwise(classes, want_tw, metric)
    for chunks, gather in ((1, True), (4, True), (3, False)):
        sh.resident_step(d_bases, d_offs, n_reads, n_bases, L, chunks=chunks, gather=gather)
        for s in range(slots):
            full, first, rows, dist = sh.resident_buffers(s)
            lo, hi = kpop.shard_bounds(n, s, slots)
            assert (first, rows) == (lo, hi - lo)
            kpop.use_device(s)
            if gather:  # every slot holds every row
                got = np.zeros((n, d))
                kpop.check(lib.kpop_memcpy_d2h(got.ctypes.data, full, got.nbytes))
                assert np.array_equal(got, want_tw), (slots, s, chunks)
            else:
                got = np.zeros((hi - lo, d))
                if hi > lo:
                    kpop.check(lib.kpop_memcpy_d2h(got.ctypes.data, full + lo * d * 8, got.nbytes))
                assert np.array_equal(got, want_tw[lo:hi])
            gd = np.zeros((hi - lo, C_))
            if hi > lo:
                kpop.check(lib.kpop_memcpy_d2h(gd.ctypes.data, dist, gd.nbytes))
            assert np.array_equal(gd, want_di[lo:hi])
            t = sh.timings(s)
            assert t["ms_compute"] > 0 and t["ms_exposed_comm"] >= 0
        kpop.use_device(0)
    sh.resident_step(d_bases, d_offs, n_reads, n_bases, L, chunks=2, gather=True)
    q, stats, nn, idx, dd, z = sh.all_vs_all_summary(queries_per_slot=9, keep_at_most=3, max_neighbours=8)
    assert len(q) == sum(min(9, x) for x in n_reads)
    b = kpop.distance_summary(want_tw, want_tw[q.astype(np.int64)], metric, keep_at_most=3, max_neighbours=8)
    assert np.array_equal(stats, b[0], equal_nan=True) and np.array_equal(nn, b[1])
    for j in range(len(q)):
        m = min(int(nn[j]), 8)
        assert np.array_equal(idx[j, :m], b[2][j, :m]) and np.array_equal(dd[j, :m], b[3][j, :m])
        assert dd[j, 0] == 0.0 and int(q[j]) in idx[j, :m].tolist()  # a read is its own nearest neighbour
    with pytest.raises(kpop.KPopError):
        sh.all_vs_all_summary(queries_per_slot=0, capacity=3)  # every row is a query: 1003 > 3
    for s in range(slots):
        kpop.use_device(s)
        lib.kpop_dev_free(C.c_void_p(d_bases[s]))
        lib.kpop_dev_free(C.c_void_p(d_offs[s]))
    kpop.use_device(0)
    sh.close()
