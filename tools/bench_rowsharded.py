#!/usr/bin/env python3
"""BASELINE config 5 (k = 15, 10k sequences, 1 -> 8 GPUs): the twister's k-mer rows sharded over the ranks, every rank
twisting all reads against its slice, one RCCL all-reduce of the [n x (D+1)] partial sums per step.

    python tools/bench_rowsharded.py --dims 16                                   # one GPU: the whole 69 GB twister
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \\
        tools/bench_rowsharded.py --dims 64                                      # 275 GB of twister over 8 GPUs
Prints one JSON line on rank 0.  Not the round's headline bench (that is bench.py); a measurement aid."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-k", type=int, default=15)
    ap.add_argument("--dims", type=int, default=16)
    ap.add_argument("--reads", type=int, default=10000)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    a = ap.parse_args()
    import torch
    import torch.distributed as dist
    world, rank, local = (int(os.environ.get(v, d)) for v, d in (("WORLD_SIZE", "1"), ("RANK", "0"), ("LOCAL_RANK", "0")))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29544")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)  # RCCL prints its banner on fd 1 at times of its own choosing: stdout is stderr until the JSON line
    dist.init_process_group("nccl", device_id=dev)
    dist.barrier()
    torch.cuda.synchronize()
    import kpop_amd
    from kpop_amd import api
    from kpop_amd.pipeline import DevicePipeline
    from kpop_amd.shard import kmer_slice_bounds
    kpop_amd.init(local)
    n, L, k, d = a.reads, a.read_len, a.k, a.dims
    t0 = time.perf_counter()
    tw = kpop_amd.Twister.synth(0x7457, k, d, hash_range=kmer_slice_bounds(k, rank, world), acc_dim=True)
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t0
    info = tw.info()
    w = np.exp2(-np.arange(d, dtype=np.float64) / 8.0)
    pipe = DevicePipeline(tw, kpop_amd.metric_compute(w / w.sum()), dev, row_sharded=True)
    sp = torch.cuda.current_stream().cuda_stream
    bases = torch.empty(n * L, dtype=torch.uint8, device=dev)
    offsets = torch.empty(n + 1, dtype=torch.int64, device=dev)
    api.dev_synth_reads(0x4B506F70, n, L, bases.data_ptr(), offsets.data_ptr(), stream=sp)
    for _ in range(a.warmup):
        out = pipe.count_twist_row_sharded(bases, offsets, L)
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = pipe.count_twist_row_sharded(bases, offsets, L)
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    checksum = float(out.sum().item())
    if rank == 0:
        os.write(real_stdout, (json.dumps({"metric": "sequences/sec count->twist, k=%d, twister k-mer rows sharded over %d GPU(s)" % (k, world),
                          "value": n * a.steps / float(el.item()), "unit": "sequences/sec", "n_gpus": world, "steps": a.steps,
                          "warmup": a.warmup, "ms_per_step": float(el.item()) / a.steps * 1e3, "scaling": "strong", "dtype": "f64",
                          "data": "synthetic",
                          "config": {"workload": "%d reads x %d bp on every rank, k=%d, %d dims (+1 accumulator), rank 0 holds %d "
                                                 "k-mer rows = %.1f GB" % (n, L, k, d, info["n_cols"], info["device_bytes"] / 1e9),
                                     "collective": "one all-reduce(sum) of %d x %d f64 per step" % (n, d + 1)},
                          "twister_build_s": t_build, "checksum": checksum}) + "\n").encode())
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
