"""CPU tests of the N>1 path: world_size-2 gloo.  The per-shard compute is the oracle standing in for the
HIP kernels (tests only); what is under test is the sharding, the all-gather and the row bookkeeping."""
import os
import socket

import numpy as np
import pytest

from kpop_amd.shard import merge_labelled_rows, shard_bounds, shard_reads


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 100000, 100003):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(10, 2, 2)


def test_shard_reads_ragged():
    offs = np.array([0, 5, 5, 12, 40, 41], dtype=np.uint64)
    lo, hi, local, b0, b1 = shard_reads(offs, 1, 2)
    assert (lo, hi) == (3, 5) and local.tolist() == [0, 28, 29] and (b0, b1) == (12, 41)


def test_merge_labelled_rows():
    labels, rows = merge_labelled_rows(["b", "a", "B"], np.arange(6).reshape(3, 2))
    assert labels == ["B", "a", "b"] and rows.tolist() == [[4, 5], [2, 3], [0, 1]]  # bytewise order
    with pytest.raises(ValueError):
        merge_labelled_rows(["x", "x"], np.zeros((2, 1)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, ret):
    import torch
    import torch.distributed as dist

    from kpop_amd.shard import all_gather_rows
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        k, d, L = 6, 5, 60
        bases, offs = O.synth_reads(0x4B506F70, n, L)
        cols = O.enumerate_kmers(k)
        T = O.synth_twister(3, d, cols)
        metric = O.metric_powers(O.synth_inertia(d))
        lo, hi, local_offs, b0, b1 = shard_reads(offs, rank, world)
        h, c, o = O.count_reads(bases[b0:b1], local_offs, k)  # stand-in for kpop_count_twist on this rank's GPU
        mine = O.twist(T, cols, h, c.astype(np.float64), o)
        full = all_gather_rows(torch.from_numpy(mine), n).numpy()
        # all-vs-all: this rank's block of rows = distances of its reads (second operand) to everyone
        block = O.distance_rowwise(full, mine, metric)
        ret[rank] = (lo, hi, full, block)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n", [11, 64])
def test_world_size_2_all_gather_and_distance_blocks(oracle, n):
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, n, ret), nprocs=world, join=True)
    k, d, L = 6, 5, 60
    bases, offs = oracle.synth_reads(0x4B506F70, n, L)
    cols = oracle.enumerate_kmers(k)
    T = oracle.synth_twister(3, d, cols)
    h, c, o = oracle.count_reads(bases, offs, k)
    want = oracle.twist(T, cols, h, c.astype(np.float64), o)
    dm = oracle.distance_rowwise(want, want, oracle.metric_powers(oracle.synth_inertia(d)))
    rows = []
    for r in range(world):
        lo, hi, full, block = ret[r]
        assert np.array_equal(full, want)            # every rank holds every twisted vector after the gather
        assert np.array_equal(block, dm[lo:hi])      # and owns its contiguous block of distance rows
        rows.append(block)
    assert np.array_equal(np.concatenate(rows), dm)  # shards concatenate to the single-GPU result


def _row_shard_worker(rank, world, port, n, ret):
    import torch
    import torch.distributed as dist

    from kpop_amd.shard import kmer_slice_bounds, reduce_partial_twists
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        k, d, L = 6, 5, 80
        bases, offs = O.synth_reads(0x4B506F70, n, L)
        bases = bases.copy()
        bases[offs[1]:offs[2]] = ord("N")            # a read with no k-mer at all: acc = 0 on every rank
        cols = O.enumerate_kmers(k)
        T = O.synth_twister(3, d, cols)
        lo, hi = kmer_slice_bounds(k, rank, world)
        keep = (cols >= lo) & (cols < hi)
        Ts = np.vstack([T[:, keep], np.ones((1, int(keep.sum())))])   # what Twister.load_slice uploads
        h, c, o = O.count_reads(bases, offs, k)                         # every rank sees every read
        partial = O.twist(Ts, cols[keep], h, c.astype(np.float64), o, normalize=False)  # stand-in for the rank's GPU
        ret[rank] = (int(keep.sum()), reduce_partial_twists(torch.from_numpy(partial)).numpy())
    finally:
        dist.destroy_process_group()


def test_world_size_2_kmer_row_sharded_twist(oracle):
    """k-mer rows of the twister cut over 2 ranks, one all-reduce of the partial sums (k=15 case)."""
    import torch.multiprocessing as mp
    world, port, n = 2, _free_port(), 23
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_row_shard_worker, args=(world, port, n, ret), nprocs=world, join=True)
    k, d, L = 6, 5, 80
    bases, offs = oracle.synth_reads(0x4B506F70, n, L)
    bases = bases.copy()
    bases[offs[1]:offs[2]] = ord("N")
    cols = oracle.enumerate_kmers(k)
    T = oracle.synth_twister(3, d, cols)
    h, c, o = oracle.count_reads(bases, offs, k)
    want = oracle.twist(T, cols, h, c.astype(np.float64), o)
    assert ret[0][0] + ret[1][0] == len(cols) and ret[0][0] > 0 and ret[1][0] > 0
    for r in range(world):
        got = ret[r][1]
        assert got.shape == want.shape and np.all(got[1] == 0.0)
        # sum of slice sums divided by acc vs the reference's sum of (v/acc) terms: rounding only
        np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-15)


# ---------------------------------------------------------------------------------------------------------
# BASELINE config 4's flow (bench.py --gpus N): ShardedJob = chunked all-gather + all-vs-all summary
# ---------------------------------------------------------------------------------------------------------
def test_chunked_gather_layout():
    from kpop_amd.shard import ChunkedGather
    for n, world, chunks in ((0, 2, 3), (1, 2, 4), (11, 2, 3), (64, 3, 4), (1000, 8, 4), (7, 8, 2)):
        lay = ChunkedGather(n, world, chunks)
        pos = lay.position_of_global()
        assert len(set(pos.tolist())) == n and (n == 0 or pos.max() < lay.n_chunks * world * lay.chunk_rows)
        for r, (lo, hi) in enumerate(lay.bounds):
            assert hi - lo <= lay.per_pad
            spans = [lay.chunk_span(c, hi - lo) for c in range(lay.n_chunks)]
            assert spans[0][0] == 0 and spans[-1][1] == hi - lo and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))


class _OracleCompute:
    """the oracle standing in for DeviceCompute (tests only): same three operations on CPU tensors"""

    def __init__(self, O, bases, offs, k, T, cols, classes, metric):
        self.O, self.bases, self.offs, self.k, self.T, self.cols, self.classes, self.metric = O, bases, offs, k, T, cols, classes, metric

    def count_twist(self, first, n, out):
        import torch
        if n:
            o = self.offs[first:first + n + 1]
            h, c, oo = self.O.count_reads(self.bases[int(o[0]):int(o[-1])], o - o[0], self.k)
            out.copy_(torch.from_numpy(self.O.twist(self.T, self.cols, h, c.astype(np.float64), oo)))

    def distance_to_classes(self, twisted, out):
        import torch
        if twisted.shape[0]:
            out.copy_(torch.from_numpy(self.O.distance_rowwise(self.classes, twisted.numpy(), self.metric)))

    def summary(self, m1, m2, keep_at_most, max_neighbours):
        st, offs, idx, dd, z = self.O.distance_summary(m1.numpy(), m2.numpy(), self.metric, keep_at_most=keep_at_most)
        return st, offs, idx, dd, z


def _config4_worker(rank, world, port, n, chunks, ret):
    import torch
    import torch.distributed as dist

    from kpop_amd.pipeline import ShardedJob
    from kpop_amd.shard import ChunkedGather
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        k, d, L, C = 6, 5, 60, 4
        bases, offs = O.synth_reads(0x4B506F70, n, L)
        cols = O.enumerate_kmers(k)
        T = O.synth_twister(3, d, cols)
        metric = O.metric_powers(O.synth_inertia(d))
        cb, co = O.synth_reads(0xC1A55, C, 200)
        hc, cc, oc = O.count_reads(cb, co, k)
        classes = O.twist(T, cols, hc, cc.astype(np.float64), oc)
        lay = ChunkedGather(n, world, chunks)
        lo, hi, local_offs, b0, b1 = shard_reads(offs, rank, world)
        comp = _OracleCompute(O, bases[b0:b1], local_offs, k, T, cols, classes, metric)
        job = ShardedJob(torch, comp, lay, rank, d, C, torch.device("cpu"))
        job.step()
        job.step()  # a second pass overwrites in place
        qid, st, so, idx, dd, z = job.all_vs_all_summary(3)
        ret[rank] = (lo, hi, job.gathered().numpy(), job.dmat.numpy(), qid, st, so, idx, dd)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n,chunks", [(11, 3), (64, 4), (5, 1)])
def test_world_size_2_config4_job(oracle, n, chunks):
    """What `bench.py --gpus 2` runs, with the oracle doing the arithmetic: shards twisted chunk by chunk, one
    all-gather per chunk, rows back in read order, distances to the classes, all-vs-all summary on the gathered matrix."""
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_config4_worker, args=(world, port, n, chunks, ret), nprocs=world, join=True)
    k, d, L, C = 6, 5, 60, 4
    bases, offs = oracle.synth_reads(0x4B506F70, n, L)
    cols = oracle.enumerate_kmers(k)
    T = oracle.synth_twister(3, d, cols)
    metric = oracle.metric_powers(oracle.synth_inertia(d))
    h, c, o = oracle.count_reads(bases, offs, k)
    want = oracle.twist(T, cols, h, c.astype(np.float64), o)
    cb, co = oracle.synth_reads(0xC1A55, C, 200)
    hc, cc, oc = oracle.count_reads(cb, co, k)
    classes = oracle.twist(T, cols, hc, cc.astype(np.float64), oc)
    dm = oracle.distance_rowwise(classes, want, metric)
    st_w, so_w, idx_w, dd_w, _ = oracle.distance_summary(want, want, metric, keep_at_most=2)
    for r in range(world):
        lo, hi, full, dmat, qid, st, so, idx, dd = ret[r]
        assert np.array_equal(full, want)
        assert np.array_equal(dmat, dm[lo:hi])
        assert qid.tolist() == list(range(lo, min(lo + 3, hi)))
        for j, g in enumerate(qid):
            assert np.array_equal(st[j], st_w[g])
            a, b = int(so[j]), int(so[j + 1])
            aw, bw = int(so_w[g]), int(so_w[g + 1])
            assert idx[a:b].tolist() == idx_w[aw:bw].tolist() and np.array_equal(dd[a:b], dd_w[aw:bw])
            assert dd[a] == 0.0 and g in idx[a:b].tolist()  # every read finds itself at distance 0
