#!/usr/bin/env python3
"""bench.py -- sequences/sec of the count -> twist -> distance hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic reads already resident in HBM:
fused count->twist (kpop_dev_count_twist) then rowwise distances to the class vectors
(kpop_dev_distance_rowwise).  Workload = BASELINE.json's metric: 100k x 150 bp reads, k=12, with the
survey's headline synthetic shape D=64 dims, C=65 classes.  For N>1 the driver launches one
rank per GPU with torch.distributed.run; reads shard across ranks with no data-path collective (distances
are vs a replicated class set), every rank does the same per-GPU work: weak scaling.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
READ_SEED, TWISTER_SEED, CLASS_SEED = 0x4B506F70, 0x5EED, 0xC1A55


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--reads", type=int, default=100000, help="reads per GPU (weak) or in total (strong)")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("-k", type=int, default=12)
    ap.add_argument("--dims", type=int, default=64)
    ap.add_argument("--classes", type=int, default=65)
    ap.add_argument("--class-len", type=int, default=500)
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the cpu_baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-dist", action="store_true", help="initialise RCCL even at world size 1 (exercises the N>1 code path)")
    return ap.parse_args()


def cpu_baseline(args, metric, k_gpu_twisted, k_gpu_dist, classes_host, sample_offsets):
    """The oracle (CPU restatement of the reference's algorithm, kind "port") timed on this box's host
    cores on a bounded sample of the same workload; also used to check the GPU results of that sample."""
    from oracle import oracle as O
    threads = os.cpu_count() or 1
    k, d, L = args.k, args.dims, args.read_len
    t0 = time.time()
    cols = O.enumerate_kmers(k)
    T = O.synth_twister(TWISTER_SEED, d, cols)  # the reference's dims-major layout
    setup_s = time.time() - t0
    # calibrate on 2000 reads, then size the sample for ~cpu_seconds of wall time
    calib = 2000
    bases, offs = O.synth_reads(READ_SEED, calib, L)
    _, _, secs = O.pipeline(bases, offs, k, T, cols, classes_host, metric, threads=threads)
    rate = calib / max(secs, 1e-9)
    n = int(min(args.reads, max(calib, rate * args.cpu_seconds)))
    bases, offs = O.synth_reads(READ_SEED, n, L)
    tw, di, secs = O.pipeline(bases, offs, k, T, cols, classes_host, metric, threads=threads)
    out = {"value": n / secs, "unit": "sequences/sec", "cores": threads, "kind": "port",
           "sample": "first %d of the %d synthetic reads, count->twist->distance in oracle/kpop_oracle.c "
                     "(OpenMP over reads; twister in the reference's dims-major layout, generated in %.1f s "
                     "outside the timed region)" % (n, args.reads, setup_s)}
    m = min(n, k_gpu_twisted.shape[0])
    parity = {
        "reads_checked": m,
        "twisted_max_abs_err": float(np.max(np.abs(k_gpu_twisted[:m] - tw[:m]))),
        "twisted_bit_exact": bool(np.array_equal(k_gpu_twisted[:m], tw[:m])),
        "distance_max_rel_err": float(np.max(np.abs(k_gpu_dist[:m] - di[:m]) / np.maximum(di[:m], 1e-300))),
    }
    return out, parity


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d"
                     % (args.gpus, args.gpus))
        sys.exit("--gpus %d disagrees with WORLD_SIZE %d" % (args.gpus, world))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist
    # stdout must carry the one JSON line and nothing else, but RCCL prints a version banner on fd 1 whenever it feels
    # like it (communicator bring-up, first collective of a kind, teardown).  So fd 1 points at stderr for the whole
    # run, on every rank, and rank 0 writes its line to the saved descriptor at the very end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)  # RCCL; used for the barrier and the max over ranks only
        dist.barrier()
        torch.cuda.synchronize()

    import kpop_amd
    from kpop_amd import api
    kpop_amd.init(local_rank)

    k, d, L, C = args.k, args.dims, args.read_len, args.classes
    if args.scaling == "weak":
        n_local, first = args.reads, rank * args.reads
    else:
        per = (args.reads + world - 1) // world
        first = min(rank * per, args.reads)
        n_local = min(per, args.reads - first)
    n_total = args.reads * world if args.scaling == "weak" else args.reads

    stream = torch.cuda.current_stream()
    sp = stream.cuda_stream
    tw = kpop_amd.Twister.synth(TWISTER_SEED, k, d)
    bases = torch.empty(max(n_local * L, 1), dtype=torch.uint8, device=dev)
    offsets = torch.empty(n_local + 1, dtype=torch.int64, device=dev)
    api.dev_synth_reads(READ_SEED, n_local, L, bases.data_ptr(), offsets.data_ptr(), first_read=first, stream=sp)
    cbases = torch.empty(C * args.class_len, dtype=torch.uint8, device=dev)
    coffs = torch.empty(C + 1, dtype=torch.int64, device=dev)
    api.dev_synth_reads(CLASS_SEED, C, args.class_len, cbases.data_ptr(), coffs.data_ptr(), stream=sp)
    classes = torch.zeros(C, d, dtype=torch.float64, device=dev)
    api.dev_count_twist(tw, cbases.data_ptr(), coffs.data_ptr(), C, C * args.class_len, args.class_len, classes.data_ptr(),
                        stream=sp)
    # inertia of the synthetic twister: w_d ~ 2^(-d/8), sum 1; metric = powers(1,1,2) of it
    w = np.exp2(-np.arange(d, dtype=np.float64) / 8.0)
    metric_host = kpop_amd.metric_compute(w / w.sum())
    metric = torch.from_numpy(metric_host).to(dev)
    twisted = torch.zeros(max(n_local, 1), d, dtype=torch.float64, device=dev)
    dmat = torch.zeros(max(n_local, 1), C, dtype=torch.float64, device=dev)
    work = torch.empty(api.dev_distance_workspace_bytes(C, n_local, d), dtype=torch.uint8, device=dev)

    def step(ev=None):
        if ev:
            ev[0].record(stream)
        api.dev_count_twist(tw, bases.data_ptr(), offsets.data_ptr(), n_local, n_local * L, L, twisted.data_ptr(),
                            stream=sp)
        if ev:
            ev[1].record(stream)
        api.dev_distance_rowwise(classes.data_ptr(), C, twisted.data_ptr(), n_local, d, metric.data_ptr(),
                                 work.data_ptr(), dmat.data_ptr(), stream=sp)
        if ev:
            ev[2].record(stream)

    def barrier():
        if use_dist:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    events = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(events[i])
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        ms_fused = float(np.mean([e[0].elapsed_time(e[1]) for e in events]))
        ms_dist = float(np.mean([e[1].elapsed_time(e[2]) for e in events]))
        windows = max(L - k + 1, 0)
        bytes_per_read = L + windows * d * 8 + d * 8  # read L B, gather nnz*D*8 B, write D*8 B
        achieved = n_local * bytes_per_read / (ms_fused * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                key = "count_twist_wave_kernel:n=%d,L=%d,k=%d,D=%d" % (n_local, L, k, d)
                traffic = tj.get(key, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "sequences/sec end-to-end count->twist->distance, k=%d, %dk x %dbp" % (k, args.reads // 1000, L),
            "value": n_total * args.steps / elapsed,
            "unit": "sequences/sec",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "%d reads x %d bp per GPU, k=%d DNA-ds, twister %d canonical k-mers x %d dims (f64, "
                                   "synthetic), %d class vectors, euclidean, metric powers(1,1,2)"
                                   % (n_local, L, k, tw.info()["n_cols"], d, C),
                       "reads_per_gpu": n_local, "read_len": L, "k": k, "n_dims": d, "n_classes": C,
                       "sharding": "reads sharded across ranks, twister and classes replicated, no collective"},
            "roofline": {"kernel": "count_twist_wave_kernel", "bound": "hbm", "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": n_local * bytes_per_read, "avg_launch_ms": ms_fused},
            "kernels_ms": {"count_twist": ms_fused, "distance_rowwise(+norms)": ms_dist},
        }
        if not args.no_cpu_baseline and world == 1:  # the CPU leg runs on rank 0 at N=1 only
            m = min(n_local, 20000)
            cb, parity = cpu_baseline(args, metric_host, twisted[:m].cpu().numpy(), dmat[:m].cpu().numpy(),
                                      classes.cpu().numpy(), None)
            line["cpu_baseline"] = cb
            line["parity_check"] = parity
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(line) + "\n").encode())
    barrier()
    if use_dist:
        dist.destroy_process_group()
    os.close(real_stdout)


if __name__ == "__main__":
    main()
