"""Device-resident orchestration of count -> twist -> distance for one rank (one GPU).

Mirrors the KPopTwistDB register flow (bin/KPopTwistDB.ml:410-417,489-554) on tensors that stay in HBM:
twisted register <- count_twist(reads); distances <- distance_rowwise(register, operand).  torch supplies
device memory, the stream and (for all-vs-all) the RCCL all-gather; every computation is a kpop_dev_* call.
"""
import numpy as np

from . import api
from .shard import all_gather_rows, reduce_partial_twists, shard_bounds


class DevicePipeline:
    def __init__(self, twister, metric, device, kind=api.EUCLIDEAN, p=2.0, normalize_counts=True,
                 normalize_distance=True, row_sharded=False):
        """row_sharded: `twister` is this rank's k-mer-row slice with the all-ones accumulator dimension
        (Twister.synth(..., hash_range=kmer_slice_bounds(k, rank, world), acc_dim=True) or Twister.load_slice)."""
        import torch
        self.torch = torch
        self.tw = twister
        self.dev = device
        self.kind, self.p = kind, p
        self.normalize_counts, self.normalize_distance = normalize_counts, normalize_distance
        self.row_sharded = row_sharded
        self.tw_dims = twister.info()["n_dims"]
        self.n_dims = self.tw_dims - (1 if row_sharded else 0)
        self.metric = torch.as_tensor(np.ascontiguousarray(metric, dtype=np.float64)).to(device)
        self._work = None

    def _stream(self):
        return self.torch.cuda.current_stream(self.dev).cuda_stream

    def count_twist(self, bases, offsets, max_len, out=None):
        """bases: uint8 tensor, offsets: int64 tensor [n+1] (both on the device) -> twisted [n, D] f64."""
        n = offsets.numel() - 1
        if self.row_sharded:
            raise ValueError("a k-mer-row shard twists through count_twist_row_sharded (every rank sees every read)")
        if out is None:
            out = self.torch.zeros(max(n, 1), self.n_dims, dtype=self.torch.float64, device=self.dev)[:n]
        api.dev_count_twist(self.tw, bases.data_ptr(), offsets.data_ptr(), n, bases.numel(), int(max_len), out.data_ptr(),
                            normalize=self.normalize_counts, stream=self._stream())
        return out

    def count_twist_row_sharded(self, bases, offsets, max_len, group=None):
        """The k = 15 / large-D layout: ALL reads against this rank's k-mer rows, un-normalised, then ONE
        all-reduce of the [n, D+1] partial sums (RCCL over xGMI) and the division by the reduced `acc` column."""
        n = offsets.numel() - 1
        partial = self.torch.zeros(max(n, 1), self.tw_dims, dtype=self.torch.float64, device=self.dev)[:n]
        api.dev_count_twist(self.tw, bases.data_ptr(), offsets.data_ptr(), n, bases.numel(), int(max_len), partial.data_ptr(),
                            normalize=False, stream=self._stream())
        return reduce_partial_twists(partial, normalize=self.normalize_counts, group=group)

    def _workspace(self, r1, r2):
        need = api.dev_distance_workspace_bytes(r1, r2, self.n_dims)
        if self._work is None or self._work.numel() < need:
            self._work = self.torch.empty(need, dtype=self.torch.uint8, device=self.dev)
        return self._work

    def distance_rowwise(self, m1, m2, out=None):
        """-> [r2, r1]: rows = m2 (the operand), cols = m1 (the register); lib/Matrix.ml:253,264-266."""
        r1, r2 = m1.shape[0], m2.shape[0]
        if out is None:
            out = self.torch.empty(r2, r1, dtype=self.torch.float64, device=self.dev)
        api.dev_distance_rowwise(m1.data_ptr(), r1, m2.data_ptr(), r2, self.n_dims, self.metric.data_ptr(),
                                 self._workspace(r1, r2).data_ptr(), out.data_ptr(), kind=self.kind, p=self.p,
                                 normalize=self.normalize_distance, stream=self._stream())
        return out

    def all_vs_all_rows(self, twisted_local, n_total, group=None):
        """All-vs-all distances, sharded: ONE all-gather of the twisted vectors (RCCL over xGMI), then this rank's
        [n_local, n_total] block of rows.  Returns (lo, hi, block)."""
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            full = all_gather_rows(twisted_local, n_total, group)
            lo, hi = shard_bounds(n_total, dist.get_rank(group), dist.get_world_size(group))
        else:
            full, lo, hi = twisted_local, 0, n_total
        return lo, hi, self.distance_rowwise(full, twisted_local)


class DeviceCompute:
    """The kpop_dev_* calls a ShardedJob needs, on reads and class vectors resident in this rank's HBM."""

    def __init__(self, pipeline, bases, offsets, read_len, classes):
        self.p, self.bases, self.offsets, self.read_len, self.classes = pipeline, bases, offsets, int(read_len), classes
        self.torch = pipeline.torch
        self._summary = None

    def count_twist(self, first, n, out):
        """reads [first, first+n) of the shard -> out[0:n] (enqueued on the current stream)"""
        if n:
            api.dev_count_twist(self.p.tw, self.bases.data_ptr(), self.offsets.data_ptr() + 8 * first, n, self.bases.numel(),
                                self.read_len, out.data_ptr(), normalize=self.p.normalize_counts, stream=self.p._stream())

    def distance_to_classes(self, twisted, out):
        if twisted.shape[0]:
            self.p.distance_rowwise(self.classes, twisted, out)

    def summary(self, m1, m2, keep_at_most, max_neighbours):
        """Matrix.summarize_rowwise (lib/Matrix.ml:691-766) of the rows of m2 against all of m1, never forming r2 x r1"""
        t, dev = self.torch, self.p.dev
        r1, r2 = m1.shape[0], m2.shape[0]
        stats = t.zeros(max(r2, 1), 4, dtype=t.float64, device=dev)[:r2]
        n = t.zeros(max(r2, 1), dtype=t.int32, device=dev)[:r2]
        idx = t.zeros(max(r2, 1), max_neighbours, dtype=t.int32, device=dev)[:r2]
        dd = t.zeros(max(r2, 1), max_neighbours, dtype=t.float64, device=dev)[:r2]
        z = t.zeros(max(r2, 1), max_neighbours, dtype=t.float64, device=dev)[:r2]
        if r2:
            api.dev_distance_summary(m1.data_ptr(), r1, m2.data_ptr(), r2, self.p.n_dims, self.p.metric.data_ptr(),
                                     self.p._workspace(r1, r2).data_ptr(), stats.data_ptr(), n.data_ptr(), idx.data_ptr(),
                                     dd.data_ptr(), z.data_ptr(), keep_at_most=keep_at_most, max_neighbours=max_neighbours,
                                     kind=self.p.kind, p=self.p.p, normalize=self.p.normalize_distance, stream=self.p._stream())
        return stats, n, idx, dd, z


class ShardedJob:
    """BASELINE config 4 on one rank: this rank's reads -> count->twist in chunks, the all-gather of chunk c (RCCL
    over xGMI, on `comm_stream`) under the twist of chunk c+1, distances of the rank's rows to the class set, and --
    on the gathered matrix -- the all-vs-all summary of any of its rows (lib/Matrix.ml:691-766; the N x N matrix is never
    formed).  `compute` does the arithmetic (DeviceCompute on a GPU; the CPU tests put the oracle there),
    so what this class owns is the order of operations, the streams and the row bookkeeping."""

    def __init__(self, torch, compute, layout, rank, n_dims, n_classes, device, comm_stream=None, group=None):
        from .shard import ChunkedGather  # noqa: F401  (layout is one)
        self.torch, self.compute, self.layout, self.rank, self.group = torch, compute, layout, rank, group
        self.lo, self.hi = layout.bounds[rank]
        self.n_local = self.hi - self.lo
        self.local = layout.local_buffer(torch, n_dims, device)
        self.full = layout.full_buffer(torch, n_dims, device) if layout.world > 1 or self._dist_on() else None
        self.dmat = torch.zeros(max(self.n_local, 1), n_classes, dtype=torch.float64, device=device)[:self.n_local]
        self.comm_stream = comm_stream
        self.is_cuda = getattr(device, "type", str(device)) == "cuda"

    @staticmethod
    def _dist_on():
        import torch.distributed as dist
        return dist.is_available() and dist.is_initialized()

    def step(self, events=None):
        """One pass over the shard.  events (optional): dict of lists the caller reads after synchronising:
        'twist' [(start, end)] per chunk on the compute stream, 'gather' [(start, end)] per chunk on the comm stream."""
        t, L = self.torch, self.layout
        cur = t.cuda.current_stream() if self.is_cuda else None
        for c in range(L.n_chunks):
            a, b = L.chunk_span(c, self.n_local)
            ev = self._mark(events, "twist", cur)
            self.compute.count_twist(a, b - a, self.local[a:b])
            self._mark_end(ev, cur)
            if self.full is None:
                continue
            if self.is_cuda and self.comm_stream is not None:
                self.comm_stream.wait_stream(cur)
                with t.cuda.stream(self.comm_stream):
                    ev = self._mark(events, "gather", self.comm_stream)
                    L.gather_chunk(c, self.local, self.full, self.group)
                    self._mark_end(ev, self.comm_stream)
            else:
                L.gather_chunk(c, self.local, self.full, self.group)
        ev = self._mark(events, "distance", cur)
        self.compute.distance_to_classes(self.local[:self.n_local], self.dmat)
        self._mark_end(ev, cur)
        if self.is_cuda and self.comm_stream is not None and self.full is not None:
            # the step ends when every rank's rows have arrived; what the compute stream still waits for here is the
            # EXPOSED part of the exchange (the rest travelled under the twist)
            ev = self._mark(events, "exposed", cur)
            cur.wait_stream(self.comm_stream)
            self._mark_end(ev, cur)

    def _mark(self, events, key, stream):
        if events is None or not self.is_cuda:
            return None
        pair = (self.torch.cuda.Event(enable_timing=True), self.torch.cuda.Event(enable_timing=True))
        events.setdefault(key, []).append(pair)
        pair[0].record(stream)
        return pair

    @staticmethod
    def _mark_end(pair, stream):
        if pair is not None:
            pair[1].record(stream)

    def gathered(self):
        """[n_total, D] in read order, on this rank"""
        if self.full is None:
            return self.local[:self.n_local]
        return self.layout.global_order(self.full)

    def all_vs_all_summary(self, n_queries, keep_at_most=2, max_neighbours=8):
        """The first n_queries rows of this rank's shard against EVERY twisted vector of the job.
        -> (global read numbers of the queries, stats, n, idx (global read numbers), dist, z)"""
        q = min(int(n_queries), self.n_local)
        full = self.gathered()
        stats, n, idx, dd, z = self.compute.summary(full, self.local[:q], keep_at_most, max_neighbours)
        return np.arange(self.lo, self.lo + q), stats, n, idx, dd, z
