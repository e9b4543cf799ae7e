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
