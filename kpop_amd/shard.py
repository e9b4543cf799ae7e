"""Sharding the hot path over the GPUs of one node: one process per GPU.

Every sequence is independent through count and twist (lib/Twister.ml:146-188 keeps no cross-spectrum
state except the duplicate-label check, :195), so reads are cut into contiguous ranges, one per rank,
with the twister, the class vectors and the metric replicated.

  * distances against a reference set (README.md:641,656): no exchange at all; rank r produces rows
    [lo_r, hi_r) of the result;
  * all-vs-all distances: ONE all-gather of the twisted vectors (RCCL over xGMI when the tensors are on
    GPUs, gloo on CPU), after which rank r computes its [hi_r-lo_r] x N block of rows.

  * twisters beyond one GPU's HBM (k = 15 with D >= 64 is 275 GB): the k-mer ROWS of the twister are cut into
    contiguous hash ranges, one per rank; every rank twists ALL reads against its slice without normalising,
    carrying its part of the normaliser in an extra all-ones dimension, and ONE all-reduce (sum) of the
    [n_reads x (D+1)] partials followed by a division gives the twisted rows on every rank.  This is the one place
    on the path with a genuine exchange of partial results.

torch.distributed is plumbing here (rendezvous + the collective); nothing in this file computes beyond the final
division of the reduced sums.
"""
import numpy as np


def shard_bounds(n_items, rank, world):
    """Contiguous, balanced ranges: the first n_items % world ranks get one extra item."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("rank %d outside world %d" % (rank, world))
    base, extra = divmod(int(n_items), world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_reads(offsets, rank, world):
    """-> (lo, hi, local_offsets, base_lo, base_hi): the rank's reads and the slice of `bases` they occupy;
    local_offsets start at 0."""
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    lo, hi = shard_bounds(len(offsets) - 1, rank, world)
    local = offsets[lo:hi + 1] - offsets[lo]
    return lo, hi, local, int(offsets[lo]), int(offsets[hi])


def all_gather_rows(local_rows, n_total, group=None):
    """All-gather row blocks of unequal height (shard_bounds order) into the full [n_total, D] matrix.
    local_rows: torch tensor [n_local, D] on the rank's device.  One collective, padded to the largest shard."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    d = local_rows.shape[1]
    sizes = [shard_bounds(n_total, r, world)[1] - shard_bounds(n_total, r, world)[0] for r in range(world)]
    if local_rows.shape[0] != sizes[rank]:
        raise ValueError("rank %d holds %d rows, its shard has %d" % (rank, local_rows.shape[0], sizes[rank]))
    pad = max(sizes) if sizes else 0
    send = torch.zeros(pad, d, dtype=local_rows.dtype, device=local_rows.device)
    send[:sizes[rank]] = local_rows
    recv = torch.empty(world * pad, d, dtype=local_rows.dtype, device=local_rows.device)
    dist.all_gather_into_tensor(recv, send, group=group)
    out = torch.empty(n_total, d, dtype=local_rows.dtype, device=local_rows.device)
    for r in range(world):
        lo, hi = shard_bounds(n_total, r, world)
        out[lo:hi] = recv[r * pad:r * pad + (hi - lo)]
    return out


def kmer_slice_bounds(k, rank, world):
    """Hash range [lo, hi) of the k-mer rows rank `rank` keeps: equal cuts of the 4^k hash space (canonical k-mers are
    denser at low hashes, so low ranks hold somewhat more rows; the cuts stay trivially computable by every rank)."""
    return shard_bounds(1 << (2 * int(k)), rank, world)


def reduce_partial_twists(partial, normalize=True, group=None):
    """partial: [n, D+1] un-normalised twist of every read against this rank's k-mer rows, last column = the rank's
    part of `acc` (lib/Twister.ml:158).  One all-reduce (sum), then t = sum / acc where acc <> 0 (:177-178).
    Works on torch tensors (GPU: RCCL, CPU: gloo); returns [n, D] on every rank."""
    import torch
    import torch.distributed as dist
    total = partial.clone()
    if dist.is_available() and dist.is_initialized():  # also at world size 1: the same code path as N > 1
        dist.all_reduce(total, op=dist.ReduceOp.SUM, group=group)
    out = total[:, :-1]
    if not normalize:
        return out.contiguous()
    acc = total[:, -1:]
    return torch.where(acc != 0, out / torch.where(acc != 0, acc, torch.ones_like(acc)), out).contiguous()


def merge_labelled_rows(labels, rows):
    """The reference returns twisted rows sorted by label (bytewise; lib/Twister.ml:197-204) and refuses
    duplicate labels (:195).  Host-side bookkeeping after the shards' rows are concatenated."""
    enc = [l.encode("utf-8", "surrogateescape") if isinstance(l, str) else bytes(l) for l in labels]
    order = sorted(range(len(enc)), key=lambda i: enc[i])
    for a, b in zip(order, order[1:]):
        if enc[a] == enc[b]:
            raise ValueError("Duplicate_label %r" % labels[a])
    return [labels[i] for i in order], np.asarray(rows)[order]


class ChunkedGather:
    """The all-gather of twisted vectors (BASELINE config 4) cut into chunks so that the exchange of
    chunk c travels over xGMI while the rank still twists chunk c+1 (the caller puts the two on different streams).

    Every rank holds `per_pad` rows (its shard, zero-padded to n_chunks * chunk_rows); chunk c of every rank lands in
    full[c] = [world][chunk_rows][D], so each collective reads and writes contiguous memory.  The gathered matrix is
    therefore chunk-major; `position_of_global()` maps a global read number to its row of full.view(-1, D) and
    `global_order(full)` returns the [n_total, D] matrix in read order (what lib/Twister.ml:197-204 would hold).

    staging="cpu" routes the collective through host tensors (gloo cannot move device tensors): test rigs only."""

    def __init__(self, n_total, world, n_chunks=1, staging=None):
        self.n_total, self.world = int(n_total), int(world)
        self.bounds = [shard_bounds(self.n_total, r, self.world) for r in range(self.world)]
        per = max(hi - lo for lo, hi in self.bounds) if self.bounds else 0
        self.n_chunks = max(1, min(int(n_chunks), max(per, 1)))
        self.chunk_rows = -(-max(per, 1) // self.n_chunks)
        self.per_pad = self.chunk_rows * self.n_chunks
        self.staging = staging

    def chunk_span(self, c, n_local):
        """rows [a, b) of the local shard that chunk c really holds (b - a < chunk_rows on the ragged edge)"""
        a = min(c * self.chunk_rows, n_local)
        return a, min(a + self.chunk_rows, n_local)

    def local_buffer(self, torch, n_dims, device, dtype=None):
        return torch.zeros(self.per_pad, n_dims, dtype=dtype or torch.float64, device=device)

    def full_buffer(self, torch, n_dims, device, dtype=None):
        return torch.zeros(self.n_chunks, self.world, self.chunk_rows, n_dims, dtype=dtype or torch.float64, device=device)

    def gather_chunk(self, c, local, full, group=None):
        """One collective: chunk c of every rank -> full[c].  Enqueued on the caller's current stream."""
        import torch.distributed as dist
        send = local[c * self.chunk_rows:(c + 1) * self.chunk_rows]
        recv = full[c].view(self.world * self.chunk_rows, -1)
        if not (dist.is_available() and dist.is_initialized()):
            if self.world != 1:
                raise RuntimeError("ChunkedGather over %d ranks needs an initialised process group" % self.world)
            recv.copy_(send)
            return
        if self.staging == "cpu":
            s = send.cpu()
            r = s.new_empty(self.world * self.chunk_rows, s.shape[1])
            dist.all_gather_into_tensor(r, s, group=group)
            recv.copy_(r)
        else:
            dist.all_gather_into_tensor(recv, send, group=group)

    def position_of_global(self):
        """int64[n_total]: row of full.view(-1, D) holding global read g"""
        pos = np.empty(self.n_total, dtype=np.int64)
        for r, (lo, hi) in enumerate(self.bounds):
            l = np.arange(hi - lo, dtype=np.int64)
            pos[lo:hi] = ((l // self.chunk_rows) * self.world + r) * self.chunk_rows + l % self.chunk_rows
        return pos

    def global_order(self, full):
        import torch
        idx = torch.from_numpy(self.position_of_global()).to(full.device)
        return full.view(-1, full.shape[-1]).index_select(0, idx)

    def bytes_received_per_rank(self, n_dims, itemsize=8):
        """payload a rank takes in from its peers per full gather"""
        return (self.world - 1) * self.per_pad * n_dims * itemsize
