"""GPU parity of the streaming host pipeline (kpop_pipeline_*, kpop_amd/csrc/pipeline.hip): reads in host memory ->
twisted rows / distances / summary in host memory on three streams, against the oracle and against the one-call entry
points it replaces (README.md:606 + :641/:656: bin/KPopCount.ml:36-50 -> lib/Twister.ml:146-188 -> lib/Matrix.ml:191-266,
691-766).  Chunking must not show: every row depends on its own read only.

Tolerances: twisted rows 1e-12 of the matrix scale (bit-exact where the fused kernel is, D > 32); distances 1e-12
relative; summary neighbours identical, statistics 1e-10."""
import numpy as np
import pytest

from conftest import concat

pytestmark = pytest.mark.gpu


def _problem(oracle, k, d, n_classes, seed, n_reads, max_len=200, genome=None):
    rng = np.random.RandomState(seed)
    seqs = ["", "ACG", "N" * 30] + ["".join(rng.choice(list("ACGTN"), size=int(rng.randint(1, max_len)), p=[.2475] * 4 + [.01]))
                                   for _ in range(n_reads - 3)]
    if genome:
        seqs[len(seqs) // 2] = "".join(rng.choice(list("ACGT"), size=genome))
    bases, offs = concat(seqs)
    cols = oracle.enumerate_kmers(k)
    T = oracle.synth_twister(seed + 1, d, cols)
    cb, co = oracle.synth_reads(seed + 2, n_classes, 500)
    hc, cc, oc = oracle.count_reads(cb, co, k)
    classes = oracle.twist(T, cols, hc, cc.astype(np.float64), oc)
    metric = oracle.metric_powers(oracle.synth_inertia(d))
    return bases, offs, cols, T, classes, metric


def _oracle_outputs(oracle, bases, offs, k, cols, T, classes, metric, keep_at_most):
    h, c, o = oracle.count_reads(bases, offs, k)
    tw = oracle.twist(T, cols, h, c.astype(np.float64), o)
    di = oracle.distance_rowwise(classes, tw, metric)
    return tw, di, oracle.distance_summary(classes, tw, metric, keep_at_most=keep_at_most)


def _check_summary(out, want, n):
    st_o, offs_o, idx_o, dist_o, z_o = want
    assert np.allclose(out["stats"], st_o, rtol=1e-10, atol=1e-300, equal_nan=True)
    for j in range(n):
        lo, hi = int(offs_o[j]), int(offs_o[j + 1])
        assert int(out["n_neighbours"][j]) == hi - lo
        m = min(hi - lo, out["nb_index"].shape[1])
        assert out["nb_index"][j, :m].tolist() == idx_o[lo:lo + m].tolist()
        assert np.allclose(out["nb_distance"][j, :m], dist_o[lo:lo + m], rtol=1e-12, atol=0)


@pytest.mark.parametrize("k,d,chunk_reads,depth", [(8, 64, 7, 2), (10, 64, 64, 3), (6, 9, 5, 4), (12, 100, 0, 0)])
def test_pipeline_vs_oracle_ragged(kpop, oracle, k, d, chunk_reads, depth):
    """ragged batch (empty reads, reads shorter than k, Ns) in many small chunks, ring wrapped several times"""
    n, C = 300, 11
    bases, offs, cols, T, classes, metric = _problem(oracle, k, d, C, 1000 + k, n)
    tw = kpop.Twister.load(T, cols, k)
    pl = kpop.Pipeline(tw, classes, metric, outputs=kpop.OUT_TWISTED | kpop.OUT_DISTANCES | kpop.OUT_SUMMARY,
                       keep_at_most=2, max_neighbours=C, chunk_reads=chunk_reads, depth=depth)
    pb = kpop.host_empty(len(bases), np.uint8)
    pb[:] = bases
    po = kpop.host_empty(len(offs), np.uint64)
    po[:] = offs
    out = pl.run(pb, po)
    st = pl.stats()
    assert st["pinned"] and (chunk_reads == 0 or st["chunks"] == -(-n // chunk_reads))
    want_tw, want_di, want_sum = _oracle_outputs(oracle, bases, offs, k, cols, T, classes, metric, 2)
    scale = np.max(np.abs(want_tw))
    assert np.max(np.abs(out["twisted"] - want_tw)) <= 1e-12 * scale
    if d > 32:
        assert np.array_equal(out["twisted"], want_tw)  # the fused kernel keeps the reference's order of additions
    ok = want_di > 0
    assert np.max(np.abs(out["distances"][ok] - want_di[ok]) / want_di[ok]) <= 1e-12
    _check_summary(out, want_sum, n)
    # and identical, bit for bit, to the one-call entry points on the whole batch
    one_tw = tw.count_twist(bases, offs)
    assert np.array_equal(out["twisted"], one_tw)
    assert np.array_equal(out["distances"], kpop.distance_rowwise(classes, one_tw, metric))
    # pageable buffers give the same results (HIP stages them)
    out2 = pl.run(bases, offs, pinned_outputs=False)
    assert not pl.stats()["pinned"]
    for name in ("twisted", "distances", "stats", "n_neighbours"):
        assert np.array_equal(out[name], out2[name], equal_nan=True), name
    for j in range(n):  # (neighbour slots past n_neighbours are not written)
        m = min(int(out["n_neighbours"][j]), C)
        assert np.array_equal(out["nb_index"][j, :m], out2["nb_index"][j, :m])
        assert np.array_equal(out["nb_distance"][j, :m], out2["nb_distance"][j, :m])
    pl.close()


def test_pipeline_selected_outputs_and_tickets_in_flight(kpop, oracle):
    """-d and -s callers never pull the twisted rows; several batches in flight share the ring"""
    k, d, C, n = 9, 64, 5, 400
    bases, offs, cols, T, classes, metric = _problem(oracle, k, d, C, 77, n)
    tw = kpop.Twister.load(T, cols, k)
    want_tw, want_di, want_sum = _oracle_outputs(oracle, bases, offs, k, cols, T, classes, metric, 1)
    pb = kpop.host_empty(len(bases), np.uint8)
    pb[:] = bases
    po = kpop.host_empty(len(offs), np.uint64)
    po[:] = offs
    for outputs in (kpop.OUT_TWISTED, kpop.OUT_DISTANCES, kpop.OUT_SUMMARY, kpop.OUT_DISTANCES | kpop.OUT_SUMMARY):
        pl = kpop.Pipeline(tw, classes, metric, outputs=outputs, keep_at_most=1, max_neighbours=C, chunk_reads=33, depth=3)
        outs = [pl.alloc_outputs(n) for _ in range(6)]
        for o in outs:
            for a in o.values():
                a[...] = 0
        tickets = [pl.submit(pb, po, o) for o in outs]  # six batches enqueued before any is collected
        for t in reversed(tickets):
            pl.collect(t)
        for o in outs:
            assert set(o) == set(pl.alloc_outputs(1))
            if "twisted" in o:
                assert np.array_equal(o["twisted"], want_tw)
            if "distances" in o:
                ok = want_di > 0
                assert np.max(np.abs(o["distances"][ok] - want_di[ok]) / want_di[ok]) <= 1e-12
            if "stats" in o:
                _check_summary(o, want_sum, n)
        pl.close()
    # twisted only: no classes needed
    pl = kpop.Pipeline(tw, outputs=kpop.OUT_TWISTED)
    assert np.array_equal(pl.run(bases, offs)["twisted"], want_tw)
    pl.close()
    with pytest.raises(kpop.KPopError):
        kpop.Pipeline(tw, outputs=kpop.OUT_DISTANCES)  # distances without classes
    with pytest.raises(kpop.KPopError):
        kpop.Pipeline(tw, classes, metric, outputs=0)


def test_pipeline_library_chosen_chunks_with_batches_in_flight(kpop, oracle):
    """chunk_reads = 0: the first batch is cut into chunks (a short first one), the batches submitted while it is in flight
    go as ONE chunk each -- whatever the cut, every batch comes back with the same bits as a batch run alone"""
    k, d, C, n = 9, 16, 5, 40000
    bases, offs, cols, T, classes, metric = _problem(oracle, k, d, C, 31, n, max_len=70)
    tw = kpop.Twister.load(T, cols, k)
    want_tw, want_di, _ = _oracle_outputs(oracle, bases, offs, k, cols, T, classes, metric, 1)
    pb = kpop.host_empty(len(bases), np.uint8)
    pb[:] = bases
    po = kpop.host_empty(len(offs), np.uint64)
    po[:] = offs
    pl = kpop.Pipeline(tw, classes, metric, outputs=kpop.OUT_TWISTED | kpop.OUT_DISTANCES)
    alone = pl.run(pb, po)
    assert pl.stats()["chunks"] >= 3  # 4,096 + 16,384 + ...
    assert np.max(np.abs(alone["twisted"] - want_tw)) <= 1e-12 * max(1.0, np.max(np.abs(want_tw)))  # (D <= 32: the packed gather's tree)
    outs = [pl.alloc_outputs(n) for _ in range(5)]
    tickets = [pl.submit(pb, po, o) for o in outs]
    chunks_last = pl.stats()["chunks"]
    pl.collect(tickets[-1])
    assert chunks_last == 1  # the fifth batch went up while the fourth was in flight
    for o in outs:
        assert np.array_equal(o["twisted"], alone["twisted"]) and np.array_equal(o["distances"], alone["distances"])
    ok = want_di > 0
    assert np.max(np.abs(alone["distances"][ok] - want_di[ok]) / want_di[ok]) <= 1e-12
    pl.close()


def test_pipeline_genomes_and_reads_mixed(kpop, oracle):
    """a genome among short reads: its chunk takes the streaming kernel (per-stream segment partials), the others the
    one-wavefront-per-read kernel; chunk_bases cuts chunks by size"""
    k, d, C, n = 10, 64, 4, 120
    bases, offs, cols, T, classes, metric = _problem(oracle, k, d, C, 5, n, genome=40000)
    tw = kpop.Twister.load(T, cols, k)
    want_tw, want_di, _ = _oracle_outputs(oracle, bases, offs, k, cols, T, classes, metric, 1)
    for chunk_bases in (0, 3000):
        pl = kpop.Pipeline(tw, classes, metric, outputs=kpop.OUT_TWISTED | kpop.OUT_DISTANCES, chunk_reads=50, chunk_bases=chunk_bases)
        out = pl.run(bases, offs)
        if chunk_bases:
            assert pl.stats()["chunks"] > 3
        # the genome kernel sums instances in sequence order: equal to the reference's sum up to rounding
        assert np.max(np.abs(out["twisted"] - want_tw)) <= 1e-12 * np.max(np.abs(want_tw))
        short = np.diff(offs.astype(np.int64)) <= 512 + k - 1
        assert np.array_equal(out["twisted"][short], want_tw[short])
        ok = want_di > 0
        assert np.max(np.abs(out["distances"][ok] - want_di[ok]) / want_di[ok]) <= 1e-11
        pl.close()
    # an empty batch is a ticket that completes at once
    pl = kpop.Pipeline(tw, classes, metric)
    out = pl.run(np.zeros(0, dtype=np.uint8), np.zeros(1, dtype=np.uint64))
    assert out["twisted"].shape == (0, d) and out["distances"].shape == (0, C)
    pl.close()


def test_pipeline_headline_shape(kpop, oracle):
    """BASELINE headline: 100k x 150 bp, k=12, D=64, C=65 through the pipeline from page-locked buffers; every row against
    the oracle (bit-exact twisted rows, distances to 1e-12), and a checksum against the device-resident path"""
    k, d, C, n, L = 12, 64, 65, 100000, 150
    tw = kpop.Twister.synth(0x5EED, k, d)
    bases, offs = oracle.synth_reads(0x4B506F70, n, L)
    cb, co = oracle.synth_reads(0xC1A55, C, 30000)
    classes = tw.count_twist(cb, co)
    metric = kpop.metric_compute(oracle.synth_inertia(d))
    pb = kpop.host_empty(len(bases), np.uint8)
    pb[:] = bases
    po = kpop.host_empty(len(offs), np.uint64)
    po[:] = offs
    pl = kpop.Pipeline(tw, classes, metric, outputs=kpop.OUT_TWISTED | kpop.OUT_DISTANCES)
    out = pl.run(pb, po)
    assert pl.stats() == {"chunks": 5, "pinned": True, "depth": 4}  # chunks that grow: 11, 14, 18, 22, 35 % of the batch
    cols = oracle.enumerate_kmers(k)
    T = oracle.synth_twister(0x5EED, d, cols)
    want_tw, want_di, _ = oracle.pipeline(bases, offs, k, T, cols, classes, metric)
    assert np.array_equal(out["twisted"], want_tw)
    assert np.max(np.abs(out["distances"] - want_di) / want_di) <= 1e-12
    pl.close()


def test_packed_bases_give_the_ascii_paths_bits(kpop, oracle):
    """2.25 bits a base at the boundary (kpop_pack_bases, kpop_count_twist_packed, kpop_dev_count_twist_packed,
    kpop_pipeline_submit_packed; packed.hip): reads with Ns, lower case, IUPAC codes and dashes, empty and short ones, two
    assemblies of one organism's size among them -- the rows of the one-byte-a-base entry points bit for bit, through the one
    call, the device-resident call and the streaming pipeline (whose chunks start in the middle of a packed word), and against
    the oracle (bin/KPopCount.ml:36-50, 242-245 -> lib/Twister.ml:146-188)"""
    import torch
    from kpop_amd import api
    rng = np.random.RandomState(21)
    k, d = 9, 40
    alphabet, prob = list("ACGTacgtNRYKM-"), [.22] * 4 + [.02] * 4 + [.01] * 4 + [.0, .0]
    prob[-2:] = [(1.0 - sum(prob[:-2])) / 2] * 2
    seqs = ["", "ACG", "N" * 30, "acgtacgtacgtacgt"] + ["".join(rng.choice(alphabet, size=int(rng.randint(1, 400)), p=prob)) for _ in range(1500)]
    seqs[700] = "".join(rng.choice(list("ACGT"), size=9000))
    seqs[900] = "".join(rng.choice(list("ACGTN"), size=7013, p=[.2475] * 4 + [.01]))
    bases, offs = concat(seqs)
    cols = oracle.enumerate_kmers(k)
    T = oracle.synth_twister(4, d, cols)
    tw = kpop.Twister.load(T, cols, k)
    codes, invalid = api.pack_bases(bases, threads=2)
    want = tw.count_twist(bases, offs)
    h, c, o = oracle.count_reads(bases, offs, k)
    np.testing.assert_allclose(want, oracle.twist(T, cols, h, c.astype(np.float64), o), rtol=1e-12, atol=1e-15)
    assert np.array_equal(tw.count_twist_packed(codes, invalid, offs), want)
    # device-resident: unpack alone, then the packed twin
    dev = torch.device("cuda", 0)
    dc, dm = torch.from_numpy(codes.view(np.int32)).to(dev), torch.from_numpy(invalid.view(np.int32)).to(dev)
    do = torch.from_numpy(offs.astype(np.int64)).to(dev)
    spread = torch.zeros(len(bases), dtype=torch.uint8, device=dev)
    api.dev_unpack_bases(dc.data_ptr(), dm.data_ptr(), len(bases), spread.data_ptr())
    torch.cuda.synchronize()
    up = np.frombuffer(bytes(bases), dtype=np.uint8) & 0xDF
    expect = np.where(np.isin(up, np.frombuffer(b"ACGT", dtype=np.uint8)), up, ord("N")).astype(np.uint8)
    assert np.array_equal(spread.cpu().numpy(), expect)
    out = torch.zeros(len(seqs), d, dtype=torch.float64, device=dev)
    api.dev_count_twist_packed(tw, dc.data_ptr(), dm.data_ptr(), do.data_ptr(), len(seqs), len(bases), max(len(s_) for s_ in seqs), out.data_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), want)
    # the streaming pipeline, in chunks that start anywhere in a word
    for chunk_reads in (0, 97, 400):
        pl = kpop.Pipeline(tw, outputs=kpop.OUT_TWISTED, chunk_reads=chunk_reads, depth=3)
        res = pl.alloc_outputs(len(seqs), pinned=False)
        pl.collect(pl.submit_packed(codes, invalid, np.ascontiguousarray(offs, dtype=np.uint64), res))
        assert np.array_equal(res["twisted"], want), chunk_reads
        pl.close()
    tw.free()


@pytest.mark.parametrize("k,d,max_len", [(12, 64, 150), (9, 16, 60), (21, 33, 300), (5, 100, 516)])
def test_reads_twisted_straight_from_the_packed_words(kpop, oracle, k, d, max_len):
    """a batch whose reads all fit the one-wavefront-per-read kernel never becomes bytes on the device: count_twist_wave_kernel<...,
    PACKED> stages a read's 2-bit codes from the words of the batch (BASELINE north_star: "coalesced HBM loads of packed bases") --
    the byte kernel's rows bit for bit, Ns, lower case and IUPAC codes included, reads that start anywhere in a word"""
    import torch
    from kpop_amd import api
    rng = np.random.RandomState(k * d)
    alphabet, prob = list("ACGTacgtNRY-"), [.23] * 4 + [.015] * 4 + [.005] * 4
    seqs = ["", "AC", "N" * 20] + ["".join(rng.choice(alphabet, size=int(rng.randint(1, max_len + 1)), p=prob)) for _ in range(3000)]
    seqs[5] = "".join(rng.choice(list("ACGT"), size=max_len))
    bases, offs = concat(seqs)
    h, c, o = oracle.count_reads(bases, offs, k)
    cols = np.unique(h)
    T = oracle.synth_twister(9, d, cols)
    tw = kpop.Twister.load(T, cols, k)
    codes, invalid = api.pack_bases(bases)
    dev = torch.device("cuda", 0)
    dc, dm = torch.from_numpy(codes.view(np.int32)).to(dev), torch.from_numpy(invalid.view(np.int32)).to(dev)
    do = torch.from_numpy(offs.astype(np.int64)).to(dev)
    db = torch.from_numpy(np.frombuffer(bytes(bases), dtype=np.uint8).copy()).to(dev)
    longest = max(len(s_) for s_ in seqs)
    for normalize in (True, False):
        # (the byte kernel through the device-resident entry point: the host one sends batches that fill a small twister through the dense image)
        ref = torch.full((len(seqs), d), float("nan"), dtype=torch.float64, device=dev)
        api.dev_count_twist(tw, db.data_ptr(), do.data_ptr(), len(seqs), len(bases), longest, ref.data_ptr(), normalize=normalize)
        out = torch.full((len(seqs), d), float("nan"), dtype=torch.float64, device=dev)
        api.dev_count_twist_packed(tw, dc.data_ptr(), dm.data_ptr(), do.data_ptr(), len(seqs), len(bases), longest, out.data_ptr(), normalize=normalize)
        torch.cuda.synchronize()
        assert torch.equal(out, ref)
        if normalize:
            want = ref.cpu().numpy()
            np.testing.assert_allclose(want, oracle.twist(T, cols, h, c.astype(np.float64), o), rtol=1e-12, atol=1e-15)
            assert np.array_equal(tw.count_twist_packed(codes, invalid, offs), want)
    tw.free()
