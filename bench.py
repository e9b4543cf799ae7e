#!/usr/bin/env python3
"""bench.py -- sequences/sec of the count -> twist -> distance hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic reads already resident in HBM.

N = 1 (default): the workload BASELINE.json's metric is quoted on -- 100k x 150 bp reads, k=12, with the survey's
headline synthetic shape D=64 dims, C=65 classes: fused count->twist (kpop_dev_count_twist), then
rowwise distances to the class vectors (kpop_dev_distance_rowwise).  The same line also carries, measured in the same
run and never mixed into `value`: the PCIe-inclusive rate through the host-buffer entry points, the file-to-file rate
through the drop-in binaries, the CPU baseline, and BASELINE config 4 (1M reads) on this one GPU -- the N = 1 point of
the strong-scaling curve below.

N > 1: BASELINE config 4, STRONG scaling -- 1M x 150 bp reads IN TOTAL, cut into contiguous shards, one rank per GPU;
every rank twists its shard in chunks, the RCCL all-gather of chunk c (xGMI, own stream) travels under the twist of
chunk c+1, and the rank's rows go through the distances to the class set.  After the timed region the gathered matrix
is checked (an order-free checksum over all ranks) and used: every rank summarises a few of its rows against all
1M twisted vectors (the all-vs-all summary of lib/Matrix.ml:691-766, N x N never formed), timed separately.
`python bench.py --gpus N` starts its own ranks (a fresh `torch.distributed.run` child, before this process touches
any GPU); under an existing launcher (WORLD_SIZE set) it is one of the ranks.

`--workload headline --scaling weak` is round 1's mode: 100k reads per GPU, no collective (distances to a replicated
class set need none).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
READ_SEED, TWISTER_SEED, CLASS_SEED = 0x4B506F70, 0x5EED, 0xC1A55
MFMA_F64_PEAK_TFLOPS = 78.6  # MI355X f64 matrix peak (256 CUs x 4 SIMDs x 32 flop/clk x 2.4 GHz)
MALL_GATHER_PEAK_GBS = 8600.0  # random whole rows of a table that sits in the 256 MiB Infinity Cache (MI355X_MICROARCH.md, 'Indexed rows')
L2_PEAK_GBS = 34500.0  # MI355X_MICROARCH.md: aggregate L2 bandwidth, 34.5 TB/s
XGMI_DIRECT_ESTIMATE_MS = 0.42  # 64 MB per shard over 7 links x 153 GB/s, one shard per link


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=["auto", "headline", "config4", "config5"], default="auto",
                    help="auto = headline at 1 GPU, config4 (1M reads, all-gather) beyond; config5 = k=15, D=16, 10,000 reads, the twister's k-mer rows "
                         "sharded over the GPUs and ONE all-reduce of the partial rows (any --gpus; -k / --dims / --reads override)")
    ap.add_argument("--scaling", choices=["auto", "weak", "strong"], default="auto",
                    help="weak: --reads per GPU; strong: --reads in total (auto: weak for headline, strong for config4)")
    ap.add_argument("--reads", type=int, default=0, help="0 = 100,000 (headline) or 1,000,000 (config4)")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("-k", type=int, default=12)
    ap.add_argument("--dims", type=int, default=64)
    ap.add_argument("--classes", type=int, default=65)
    ap.add_argument("--class-len", type=int, default=30000, help="class genomes (SURVEY.md 8d: 30 kb)")
    ap.add_argument("--ag-chunks", type=int, default=0,
                    help="config4: pieces the all-gather is cut into; 0 = max(4, GPUs): the exposed tail is one piece's exchange")
    ap.add_argument("--in-process", action="store_true",
                    help="all GPUs from ONE process through the C ABI (kpop_init_devices + kpop_sharded_*: one host thread per "
                         "device, hipMemcpyPeerAsync all-gather) instead of one torch.distributed rank per GPU with RCCL")
    ap.add_argument("--queries", type=int, default=1024, help="config4: rows (in total) of the all-vs-all summary")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the cpu_baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the PCIe-inclusive, file-to-file and config-4 legs of the 1-GPU line")
    ap.add_argument("--no-children", action="store_true",
                    help="skip the legs that start child processes (the rocprofv3 --pmc traffic passes, the file-to-file CLIs): for running the whole line under a profiler")
    ap.add_argument("--f2f-reads", type=int, default=1000000, help="reads of the file-to-file leg")
    ap.add_argument("--force-dist", action="store_true", help="initialise RCCL even at world size 1 (exercises the N>1 code path)")
    ap.add_argument("--spawn", action="store_true", help="start the ranks as a child launcher even for --gpus 1 (tests the relay)")
    ap.add_argument("--timeout", type=float, default=900.0,
                    help="the launcher's watchdog: seconds after which it ends the ranks' process group, prints every rank's last lines and exits non-zero")
    ap.add_argument("--no-fallback", action="store_true",
                    help="the launcher does not retry a failed RCCL run through --in-process (also KPOP_BENCH_NO_FALLBACK=1)")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------------
# the launcher: no torch, no GPU call in this process
# ---------------------------------------------------------------------------------------------------------
def _tail(path, n=40):
    try:
        with open(path, "rb") as f:
            return b"\n".join(f.read().splitlines()[-n:]).decode("utf-8", "replace")
    except OSError:
        return "(no log)"


def _run_child(cmd, env, timeout_s):
    """start `cmd` in its own process group, relay its stdout looking for the JSON line, and end the whole group if it has not
    finished after timeout_s -> (rc or None when it timed out, the JSON line or None, seconds)"""
    import signal
    import threading
    t0 = time.time()
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, start_new_session=True)
    found = []

    def relay():
        for raw in proc.stdout:
            text = raw.decode("utf-8", "replace")
            at = text.find('{"metric"')  # (a launcher that tees the ranks' output puts a prefix in front)
            if at >= 0:
                found.append(text[at:])
            else:
                sys.stderr.write(text)

    th = threading.Thread(target=relay, daemon=True)
    th.start()
    try:
        rc = proc.wait(timeout=timeout_s)
    except subprocess.TimeoutExpired:
        rc = None
        for sig, grace in ((signal.SIGTERM, 5.0), (signal.SIGKILL, 5.0)):
            try:
                os.killpg(proc.pid, sig)  # (this process never touches a GPU: it may end its children as it likes)
            except ProcessLookupError:
                break
            try:
                proc.wait(timeout=grace)
                break
            except subprocess.TimeoutExpired:
                continue
    th.join(timeout=5.0)
    return rc, (found[-1] if found else None), time.time() - t0


def self_launch(args):
    """`bench.py --gpus N` from a plain shell: this process starts the ranks (torch.distributed.run, RCCL), watches them, and -- the
    first N > 1 run will be somebody else's, on hardware this code has never seen -- leaves something to debug from when they fail:
    a watchdog (--timeout), every rank's stderr kept in a file of its own and its tail printed, NCCL_DEBUG=WARN, and a second
    attempt through the one-process C-ABI form (--in-process: peer copies, no RCCL) whose line says that it is a fallback."""
    import tempfile
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    argv = [a for a in sys.argv[1:] if a != "--spawn"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL cannot share buffers across ranks without it
    env.setdefault("OMP_NUM_THREADS", "8")
    env.setdefault("NCCL_DEBUG", "WARN")
    logdir = tempfile.mkdtemp(prefix="kpop_bench_ranks_")
    env["KPOP_BENCH_LOGDIR"] = logdir
    rc, line, took = _run_child(cmd, env, args.timeout)
    logs = sorted(f for f in os.listdir(logdir) if f.endswith(".stderr"))
    ok = rc == 0 and line is not None
    for f in logs:  # the ranks' stderr: all of it (bounded) after a good run, the tail after a bad one
        text = _tail(os.path.join(logdir, f), 200 if ok else 40)
        if text.strip():
            sys.stderr.write("---- %s%s\n%s\n" % (f, "" if ok else " (last 40 lines)", text))
    if ok:
        sys.stdout.write(line)
        sys.stdout.flush()
        sys.exit(0)
    reason = ("no rank finished within --timeout %.0f s (the ranks' process group was ended)" % args.timeout if rc is None
              else "the ranks exited %d" % rc if rc != 0 else "the ranks exited 0 but printed no JSON line")
    sys.stderr.write("bench.py: %s after %.0f s; %d rank log(s) under %s\n" % (reason, took, len(logs), logdir))
    if args.no_fallback or os.environ.get("KPOP_BENCH_NO_FALLBACK") == "1":
        sys.exit(124 if rc is None else (rc or 1))
    # a FRESH child through the C ABI alone: one process, one host thread per GPU, hipMemcpyPeerAsync instead of RCCL
    sys.stderr.write("bench.py: trying the same job --in-process (kpop_init_devices + kpop_sharded_*: peer copies, no RCCL)\n")
    env2 = dict(os.environ)
    env2.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env2["KPOP_BENCH_FALLBACK_REASON"] = reason
    rc2, line2, _ = _run_child([sys.executable, os.path.abspath(__file__)] + [a for a in argv if a != "--in-process"] + ["--in-process"],
                               env2, args.timeout)
    if rc2 == 0 and line2 is not None:
        sys.stdout.write(line2)
        sys.stdout.flush()
        sys.exit(0)
    sys.stderr.write("bench.py: the --in-process attempt %s\n" % ("timed out" if rc2 is None else "exited %d" % rc2))
    sys.exit(124 if rc is None else (rc or 1))


def rank_preamble():
    """first thing in a rank started by self_launch: its stderr into a file of its own (the launcher prints the tails), and the
    test switch KPOP_BENCH_FAKE_HANG=<rank>|all -- a rank that never reaches the rendezvous -- before anything touches a GPU"""
    logdir, rank = os.environ.get("KPOP_BENCH_LOGDIR"), os.environ.get("RANK", "0")
    if logdir and os.path.isdir(logdir):
        fd = os.open(os.path.join(logdir, "rank%s.stderr" % rank), os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
        os.dup2(fd, 2)
        os.close(fd)
    hang = os.environ.get("KPOP_BENCH_FAKE_HANG")
    if hang is not None and hang in (rank, "all"):
        sys.stderr.write("rank %s: KPOP_BENCH_FAKE_HANG: sleeping instead of joining the rendezvous\n" % rank)
        sys.stderr.flush()
        while True:
            time.sleep(3600)


# ---------------------------------------------------------------------------------------------------------
# one rank
# ---------------------------------------------------------------------------------------------------------
class Rank:
    def __init__(self, args):
        t_start = time.perf_counter()
        import numpy as np
        import torch
        import torch.distributed as dist
        self.args, self.np, self.torch, self.dist = args, np, torch, dist
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if args.gpus != self.world:
            sys.exit("--gpus %d disagrees with WORLD_SIZE %d" % (args.gpus, self.world))
        if not torch.cuda.is_available():
            sys.exit("bench.py needs a GPU: the hot path has no CPU fallback")
        # KPOP_BENCH_SHARE_GPU=1: every rank on GPU 0 and the collectives through gloo + host staging -- a rig for
        # running the N>1 code on a 1-GPU box; what it measures is not a scaling number and the line says so
        self.shared_gpu = os.environ.get("KPOP_BENCH_SHARE_GPU") == "1"
        dev_index = 0 if self.shared_gpu else self.local_rank
        torch.cuda.set_device(dev_index)
        self.dev = torch.device("cuda", dev_index)
        self.use_dist = self.world > 1 or args.force_dist
        # stdout must carry the one JSON line and nothing else, but RCCL prints a version banner on fd 1 whenever it
        # feels like it.  So fd 1 points at stderr for the whole run, on every rank, and rank 0 writes its line to the
        # saved descriptor at the very end.
        sys.stdout.flush()
        self.real_stdout = os.dup(1)
        os.dup2(2, 1)
        if self.use_dist:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
            if self.shared_gpu:
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=self.dev)  # RCCL
            self.barrier()
        import kpop_amd
        from kpop_amd import api
        self.kpop, self.api = kpop_amd, api
        kpop_amd.init(dev_index)
        self.stream = torch.cuda.current_stream()
        self.sp = self.stream.cuda_stream
        if args.workload == "config5":  # (its twister is a slice of k-mer rows, built by run_config5; no class vectors)
            self.tw = None
            torch.cuda.synchronize()
            self.startup_s = time.perf_counter() - t_start
            return
        self.tw = kpop_amd.Twister.synth(TWISTER_SEED, args.k, args.dims)
        k, d, C = args.k, args.dims, args.classes
        cbases = torch.empty(C * args.class_len, dtype=torch.uint8, device=self.dev)
        coffs = torch.empty(C + 1, dtype=torch.int64, device=self.dev)
        api.dev_synth_reads(CLASS_SEED, C, args.class_len, cbases.data_ptr(), coffs.data_ptr(), stream=self.sp)
        self.classes = torch.zeros(C, d, dtype=torch.float64, device=self.dev)
        api.dev_count_twist(self.tw, cbases.data_ptr(), coffs.data_ptr(), C, C * args.class_len, args.class_len,
                            self.classes.data_ptr(), stream=self.sp)
        # inertia of the synthetic twister: w_d ~ 2^(-d/8), sum 1; metric = powers(1,1,2) of it
        w = np.exp2(-np.arange(d, dtype=np.float64) / 8.0)
        self.metric_host = kpop_amd.metric_compute(w / w.sum())
        self.metric = torch.from_numpy(self.metric_host).to(self.dev)
        torch.cuda.synchronize()
        # start-up of a rank (imports, RCCL rendezvous, the twister -- 4.3 GB synthesised on the device at k=12 -- and the
        # class vectors): outside every timed region, reported so that it is seen
        self.startup_s = time.perf_counter() - t_start

    # -- plumbing
    def barrier(self):
        if self.use_dist:
            self.dist.barrier()
            self.torch.cuda.synchronize()

    def max_over_ranks(self, x):
        if not self.use_dist:
            return x
        t = self.torch.tensor([x], dtype=self.torch.float64, device="cpu" if self.shared_gpu else self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def gather_floats(self, x):
        """one float per rank -> list on every rank"""
        if not self.use_dist:
            return [float(x)]
        t = self.torch.zeros(self.world, dtype=self.torch.float64, device="cpu" if self.shared_gpu else self.dev)
        t[self.rank] = float(x)
        self.dist.all_reduce(t)
        return [float(v) for v in t.cpu().tolist()]

    def synth_reads(self, n, first):
        t, L = self.torch, self.args.read_len
        bases = t.empty(max(n * L, 1), dtype=t.uint8, device=self.dev)
        offsets = t.empty(n + 1, dtype=t.int64, device=self.dev)
        self.api.dev_synth_reads(READ_SEED, n, L, bases.data_ptr(), offsets.data_ptr(), first_read=first, stream=self.sp)
        return bases, offsets

    def timed(self, step, steps, warmup):
        """warmup, then exactly `steps` steps between barrier + synchronize pairs; -> seconds, max over ranks"""
        for _ in range(warmup):
            step(None)
        self.torch.cuda.synchronize()
        self.barrier()
        self.torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            step(i)
        self.torch.cuda.synchronize()
        self.barrier()
        self.torch.cuda.synchronize()
        return self.max_over_ranks(time.perf_counter() - t0)

    def bytes_per_read(self):
        a = self.args
        windows = max(a.read_len - a.k + 1, 0)
        return a.read_len + windows * a.dims * 8 + a.dims * 8  # read L B, gather nnz*D*8 B, write D*8 B

    def roofline(self, n_reads_per_launch, avg_ms, launches_note=None):
        a = self.args
        alg = n_reads_per_launch * self.bytes_per_read()
        achieved = alg / (avg_ms * 1e-3) / 1e9
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                key = "count_twist_wave_kernel:n=%d,L=%d,k=%d,D=%d" % (n_reads_per_launch, a.read_len, a.k, a.dims)
                if key in tj:
                    traffic = tj[key].get("hbm_bytes_per_launch")
                    traffic_src = ("not measured in this run: PMC passes of the same launch shape, %s"
                                   % tj[key].get("source", "profiles/"))
            except Exception:
                pass
        out = {"kernel": "count_twist_wave_kernel", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
               "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
               "algorithmic_bytes_per_launch": alg, "avg_launch_ms": avg_ms}
        if launches_note:
            out["note"] = launches_note
        return out

    # -- the headline job: reads resident, count->twist then distances to the classes, no collective
    def run_headline(self, n_local, first, steps, warmup, want_outputs=0):
        t, a = self.torch, self.args
        d, C, L = a.dims, a.classes, a.read_len
        bases, offsets = self.synth_reads(n_local, first)
        twisted = t.zeros(max(n_local, 1), d, dtype=t.float64, device=self.dev)
        dmat = t.zeros(max(n_local, 1), C, dtype=t.float64, device=self.dev)
        work = t.empty(self.api.dev_distance_workspace_bytes(C, n_local, d), dtype=t.uint8, device=self.dev)
        events = [[t.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(steps)]

        def step(i):
            ev = events[i] if i is not None else None
            if ev:
                ev[0].record(self.stream)
            self.api.dev_count_twist(self.tw, bases.data_ptr(), offsets.data_ptr(), n_local, n_local * L, L,
                                     twisted.data_ptr(), stream=self.sp)
            if ev:
                ev[1].record(self.stream)
            self.api.dev_distance_rowwise(self.classes.data_ptr(), C, twisted.data_ptr(), n_local, d, self.metric.data_ptr(),
                                          work.data_ptr(), dmat.data_ptr(), stream=self.sp)
            if ev:
                ev[2].record(self.stream)

        elapsed = self.timed(step, steps, warmup)
        ms_fused = float(self.np.mean([e[0].elapsed_time(e[1]) for e in events]))
        ms_dist = float(self.np.mean([e[1].elapsed_time(e[2]) for e in events]))
        res = {"elapsed": elapsed, "ms_fused": ms_fused, "ms_dist": ms_dist}
        if want_outputs:
            m = min(n_local, want_outputs)
            res["twisted"] = twisted[:m].cpu().numpy()
            res["dmat"] = dmat[:m].cpu().numpy()
        return res

    # -- BASELINE config 4: reads in total, sharded; chunked all-gather under the twist; distances to the classes
    def run_config4(self, n_total, steps, warmup):
        from kpop_amd.pipeline import DevicePipeline, DeviceCompute, ShardedJob
        from kpop_amd.shard import ChunkedGather
        t, a, np = self.torch, self.args, self.np
        ag_chunks = a.ag_chunks if a.ag_chunks > 0 else max(4, self.world)
        layout = ChunkedGather(n_total, self.world, ag_chunks if self.world > 1 or self.use_dist else 1,
                               staging="cpu" if self.shared_gpu else None)
        lo, hi = layout.bounds[self.rank]
        bases, offsets = self.synth_reads(hi - lo, lo)
        pipe = DevicePipeline(self.tw, self.metric_host, self.dev)
        comp = DeviceCompute(pipe, bases, offsets, a.read_len, self.classes)
        comm = t.cuda.Stream(device=self.dev) if (self.world > 1 or self.use_dist) else None
        job = ShardedJob(t, comp, layout, self.rank, a.dims, a.classes, self.dev, comm_stream=comm)
        pipe._workspace(a.classes, job.n_local)  # allocated before the timed region
        per_step_events = []

        def step(i):
            ev = {} if i is not None else None
            job.step(ev)
            if ev is not None:
                per_step_events.append(ev)

        elapsed = self.timed(step, steps, warmup)
        res = {"elapsed": elapsed, "n_local": job.n_local, "chunks": layout.n_chunks, "chunk_rows": layout.chunk_rows}
        tw_ms = [p[0].elapsed_time(p[1]) for ev in per_step_events for p in ev.get("twist", [])]
        res["ms_twist_chunk"] = float(np.mean(tw_ms)) if tw_ms else None
        res["ms_twist_step"] = float(np.sum(tw_ms) / max(len(per_step_events), 1)) if tw_ms else None
        res["ms_dist"] = float(np.mean([p[0].elapsed_time(p[1]) for ev in per_step_events for p in ev.get("distance", [])]))
        ag = [p[0].elapsed_time(p[1]) for ev in per_step_events for p in ev.get("gather", [])]
        if ag:
            res["ms_allgather_step_on_its_stream"] = float(np.sum(ag) / len(per_step_events))
        ex = [p[0].elapsed_time(p[1]) for ev in per_step_events for p in ev.get("exposed", [])]
        mine = float(np.sum(ex) / max(len(per_step_events), 1)) if ex else 0.0
        res["ms_exposed_comm_per_rank"] = self.gather_floats(mine)
        res["ms_compute_per_rank"] = self.gather_floats(res["ms_twist_step"] or 0.0)
        # how many ranks the COLLECTIVE saw (a sum of ones through it: not the launcher's word for it)
        ones = t.ones(1, dtype=t.float64, device="cpu" if self.shared_gpu else self.dev)
        if self.use_dist:
            self.dist.all_reduce(ones)
        res["n_ranks_seen"] = int(ones.item())
        res["startup_s_per_rank"] = self.gather_floats(self.startup_s)
        if job.full is not None:
            # the exchange alone, nothing else running: all chunks back to back on the comm stream
            t.cuda.synchronize()
            self.barrier()
            reps, e0, e1 = 5, t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
            with t.cuda.stream(comm):
                for c in range(layout.n_chunks):
                    layout.gather_chunk(c, job.local, job.full)  # warm
                e0.record(comm)
                for _ in range(reps):
                    for c in range(layout.n_chunks):
                        layout.gather_chunk(c, job.local, job.full)
                e1.record(comm)
            t.cuda.synchronize()
            alone = self.max_over_ranks(e0.elapsed_time(e1) / reps)
            rx = layout.bytes_received_per_rank(a.dims)
            res["allgather"] = {"ms_alone": alone, "bytes_received_per_rank": rx,
                                "bytes_gathered_total": layout.world * layout.per_pad * a.dims * 8,
                                "rx_GBps_per_rank": rx / (alone * 1e-3) / 1e9 if alone > 0 else None,
                                "direct_xgmi_estimate_ms": XGMI_DIRECT_ESTIMATE_MS * (layout.per_pad * a.dims * 8) / 64e6,
                                "ranks_in_communicator": self.dist.get_world_size() if self.use_dist else 1,
                                "backend": "gloo+host staging (KPOP_BENCH_SHARE_GPU rig)" if self.shared_gpu else "nccl (RCCL)"}
            # is the gathered matrix the union of the shards?  order-free checksum of the f64 bit patterns
            mine = job.local[:job.n_local].view(t.int64).sum()
            got = job.full.view(t.int64).sum()  # padding rows are zeros
            tot = mine.clone().cpu() if self.shared_gpu else mine.clone()
            if self.use_dist:
                self.dist.all_reduce(tot)
            res["gather_checksum_ok"] = bool(int(tot.item()) == int(got.item()))
        # the gathered matrix in use: a few rows of every rank against all n_total twisted vectors
        q_rank = max(1, a.queries // self.world)
        ava_all = []
        for _ in range(4):  # the first call grows the library's workspace (an allocation, reported on its own); then three warm ones
            t.cuda.synchronize()
            self.barrier()
            t0 = time.perf_counter()
            qid, stats, nn, idx, dd, z = job.all_vs_all_summary(q_rank, keep_at_most=2, max_neighbours=8)
            t.cuda.synchronize()
            ava_all.append(self.max_over_ranks(time.perf_counter() - t0))
        ava = sorted(ava_all[1:])[1]
        idx_h, dd_h, nn_h = idx.cpu().numpy(), dd.cpu().numpy(), nn.cpu().numpy()
        # every read is its own nearest neighbour at distance 0 (identical reads would tie and also be listed)
        own = all((dd_h[j, 0] == 0.0) and (int(qid[j]) in idx_h[j, :min(int(nn_h[j]), idx_h.shape[1])].tolist())
                  for j in range(len(qid)))
        own_all = self.max_over_ranks(0.0 if own else 1.0) == 0.0
        mfma_flops = 2.0 * q_rank * self.world * n_total * a.dims  # the contraction a . b of every pair: what the matrix cores do of the call
        res["all_vs_all"] = {"queries_total": q_rank * self.world, "against": n_total, "seconds": ava,
                             "roofline": {"kernel": "distance_rows_mfma_kernel (+ the summary's sample / pass / finish / refine kernels: the WHOLE call is the time)", "bound": "mfma",
                                          "achieved": mfma_flops / ava / 1e12 if ava > 0 else None, "peak": MFMA_F64_PEAK_TFLOPS, "unit": "TFLOP/s",
                                          "frac": mfma_flops / ava / 1e12 / MFMA_F64_PEAK_TFLOPS if ava > 0 else None, "traffic": None,
                                          "algorithmic_flops_per_launch": mfma_flops, "avg_launch_ms": ava * 1e3,
                                          "note": "since round 5 the summary's distances are f64 MFMAs that locate; what is reported is recomputed with the reference's chain "
                                                  "(csrc/distance_mfma.hip). The MFMA kernel is ~0.4 of the call (kernel by kernel: profiles/r06_summary_lanes.txt); "
                                                  "the fraction here is the contraction's flops over the whole call"},
                             "seconds_is": "median of three warm calls", "first_call_seconds": ava_all[0],
                             "pairs_per_second": q_rank * self.world * n_total / ava if ava > 0 else None,
                             "every_query_finds_itself_at_distance_0": own_all,
                             "note": "after the timed region; kpop_dev_distance_summary on the gathered matrix, N x N never formed"}
        return res

    def finish(self, line):
        if self.rank == 0:
            sys.stdout.flush()
            os.write(self.real_stdout, (json.dumps(line) + "\n").encode())
        self.barrier()
        if self.use_dist:
            self.dist.destroy_process_group()
        os.close(self.real_stdout)


# ---------------------------------------------------------------------------------------------------------
# BASELINE configs 2, 3 and 5 on this GPU (the N = 1 line's extras; every one a device-resident timing like `value`)
# ---------------------------------------------------------------------------------------------------------
def _event_ms(R, fn, reps, warm=1):
    """median and all of `reps` HIP-event timings (ms) of fn() on the rank's stream, after `warm` untimed calls"""
    t = R.torch
    for _ in range(warm):
        fn()
    t.cuda.synchronize()
    ms = []
    for _ in range(reps):
        e0, e1 = t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
        e0.record(R.stream)
        fn()
        e1.record(R.stream)
        t.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    return float(sorted(ms)[len(ms) // 2]), ms


def config2_leg(R):
    """BASELINE config 2: 10,000 x 150 bp, k = 10, D = 64 -- count -L (one CSR spectrum per read) and the fused count->twist"""
    t, api, kp = R.torch, R.api, R.kpop
    k, d, n, L = 10, 64, 10000, 150
    tw = kp.Twister.synth(TWISTER_SEED, k, d)
    try:
        bases = t.empty(n * L, dtype=t.uint8, device=R.dev)
        offs = t.empty(n + 1, dtype=t.int64, device=R.dev)
        api.dev_synth_reads(READ_SEED, n, L, bases.data_ptr(), offs.data_ptr(), stream=R.sp)
        w = L - k + 1
        scratch = t.empty(api.dev_count_reads_scratch_bytes(n, L, k), dtype=t.uint8, device=R.dev)
        oh = t.empty(n * w, dtype=t.int64, device=R.dev)
        oc = t.empty(n * w, dtype=t.int32, device=R.dev)
        oo = t.empty(n + 1, dtype=t.int64, device=R.dev)
        out = t.zeros(n, d, dtype=t.float64, device=R.dev)
        ms_c, _ = _event_ms(R, lambda: api.dev_count_reads(bases.data_ptr(), offs.data_ptr(), n, L, k, scratch.data_ptr(), oh.data_ptr(),
                                                           oc.data_ptr(), oo.data_ptr(), stream=R.sp), 20, 3)
        ms_t, _ = _event_ms(R, lambda: api.dev_count_twist(tw, bases.data_ptr(), offs.data_ptr(), n, n * L, L, out.data_ptr(), stream=R.sp), 20, 3)
        alg_c = n * (L + w * 8)            # read L B, write (L - k + 1) x (u32 hash + u32 count)
        alg_t = n * (L + w * d * 8 + d * 8)  # read L B, gather nnz x D x 8 B, write D x 8 B
        table = tw.info()["device_bytes"]
        return {
            "workload": "%d reads x %d bp, k=%d DNA-ds, D=%d (twister %.2f GB: inside the 256 MB Infinity Cache + L2 after the first launch)" % (n, L, k, d, table / 1e9),
            "value": n / ((ms_c + ms_t) * 1e-3), "unit": "sequences/sec", "value_is": "count -L then the fused count->twist, both device-resident",
            "ms_per_step": ms_c + ms_t, "kernels_ms": {"count_reads (-L, CSR out)": ms_c, "count_twist (fused)": ms_t},
            "roofline": {"kernel": "count_twist_wave_kernel", "bound": "infinity_cache", "achieved": alg_t / (ms_t * 1e-3) / 1e9, "peak": MALL_GATHER_PEAK_GBS, "unit": "GB/s",
                         "frac": alg_t / (ms_t * 1e-3) / 1e9 / MALL_GATHER_PEAK_GBS, "traffic": None, "algorithmic_bytes_per_launch": alg_t, "avg_launch_ms": ms_t,
                         "note": "the 0.27 GB table is past the L2s (4 MiB each) and inside the 256 MiB Infinity Cache: every row is a random 512-B read of it. "
                                 "peak = what MI355X_MICROARCH.md measures for uniformly random rows of an Infinity-Cache-resident table (8.6 TB/s chip-wide at 38 MB, "
                                 "7.4-7.9 at 151 MB). The launch scales with the reads from 2,000 to 100,000 of them (0.027 / 0.107 / 0.295 / 0.94 ms: "
                                 "tools/probes/ab_small_batch_unroll.py) and does not change with the row loads in flight: bandwidth, not a latency chain"},
            "roofline_count": {"kernel": "count_wave_kernel", "bound": "valu", "achieved_GBps_on_algorithmic_bytes": alg_c / (ms_c * 1e-3) / 1e9,
                               "algorithmic_bytes_per_launch": alg_c, "avg_launch_ms": ms_c,
                               "note": "instruction-bound (a bitonic network per read): profiles/r04_sq_counters.txt has the VALU issue fraction at 100k reads"},
        }
    finally:
        tw.free()


def _mutants_on_device(R, n, L, rate, seed):
    """n copies of one synthetic L-base genome with point substitutions at `rate`, made on the device in slices"""
    t, api = R.torch, R.api
    ref = t.empty(L, dtype=t.uint8, device=R.dev)
    ro = t.empty(2, dtype=t.int64, device=R.dev)
    api.dev_synth_reads(seed, 1, L, ref.data_ptr(), ro.data_ptr(), stream=R.sp)
    t.cuda.synchronize()
    bases = ref.repeat(n)
    acgt = t.tensor(list(b"ACGT"), dtype=t.uint8, device=R.dev)
    g = t.Generator(device=R.dev)
    g.manual_seed(seed)
    step = 1 << 27
    for lo in range(0, n * L, step):
        hi = min(n * L, lo + step)
        hit = t.rand(hi - lo, device=R.dev, generator=g) < rate
        sub = acgt[t.randint(0, 4, (hi - lo,), device=R.dev, generator=g)]
        bases[lo:hi] = t.where(hit, sub, bases[lo:hi])
    offs = t.arange(n + 1, dtype=t.int64, device=R.dev) * L
    return bases, offs


def config3_leg(R):
    """BASELINE config 3: 50,000 x 30 kb, k = 12, D = 64, the default dispatch -- once on unrelated genomes (the streaming
    kernel: the HBM gather), once on assemblies of one organism (0.3 % divergence: the tile kernel, consensus on the matrix cores)"""
    t, api = R.torch, R.api
    k, d, n, L = R.args.k, R.args.dims, 50000, 30000
    free, _ = t.cuda.mem_get_info(R.dev)
    if free < 24e9:
        return {"skipped": "needs 24 GB of free HBM, %.1f GB free" % (free / 1e9)}
    out = t.zeros(n, d, dtype=t.float64, device=R.dev)
    windows = n * (L - k + 1)
    res = {"workload": "%d sequences x %d bp, k=%d DNA-ds, D=%d, default dispatch (kpop_tune untouched)" % (n, L, k, d)}
    # unrelated genomes
    bases = t.empty(n * L, dtype=t.uint8, device=R.dev)
    offs = t.empty(n + 1, dtype=t.int64, device=R.dev)
    api.dev_synth_reads(0xC1A55, n, L, bases.data_ptr(), offs.data_ptr(), stream=R.sp)
    call = lambda: api.dev_count_twist(R.tw, bases.data_ptr(), offs.data_ptr(), n, n * L, L, out.data_ptr(), stream=R.sp)
    ms, all_ms = _event_ms(R, call, 3, 1)
    alg = n * L + windows * d * 8 + n * d * 8
    res["unrelated_genomes"] = {
        "value": n / (ms * 1e-3), "unit": "sequences/sec", "ms_per_step": ms, "ms_all": all_ms,
        "roofline": {"kernel": "count_twist_stream_kernel", "bound": "hbm", "achieved": alg / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes_per_launch": alg, "avg_launch_ms": ms,
                     "note": "the whole call (probe, scans, streaming kernel, combine) timed as one launch: the streaming kernel is > 99 % of it "
                             "(profiles/r04_tile_kernel_ab.txt); every window's row is a random 512 B row of a 4.3 GB table"}}
    del bases
    # assemblies of one organism
    bases, offs = _mutants_on_device(R, n, L, 0.003, 0x0123)
    call = lambda: api.dev_count_twist(R.tw, bases.data_ptr(), offs.data_ptr(), n, n * L, L, out.data_ptr(), stream=R.sp)
    ms, all_ms = _event_ms(R, call, 3, 1)
    api.debug_counters(16)
    api.tune("dbg", 32 << 24)
    call()
    cnt = api.debug_counters(16)
    api.tune("dbg", 0)
    flops = 2.0 * 64 * d * cnt[15]
    res["one_organism_0.3pct"] = {
        "value": n / (ms * 1e-3), "unit": "sequences/sec", "ms_per_step": ms, "ms_all": all_ms,
        "roofline": {"kernel": "count_twist_tile_pipe_kernel", "bound": "mfma", "achieved": flops / (ms * 1e-3) / 1e12, "peak": MFMA_F64_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": flops / (ms * 1e-3) / 1e12 / MFMA_F64_PEAK_TFLOPS, "traffic": None, "algorithmic_flops_per_launch": flops, "avg_launch_ms": ms,
                     "chunks_taken": cnt[14], "set_rows_multiplied": cnt[15],
                     "note": "flops = 2 x 64 sequences x D x the rows of every chunk's consensus set as multiplied (counted by the kernel in an extra, "
                             "untimed call), over the WHOLE call's time: half the block's wavefronts multiply a chunk while the other half prepare the "
                             "next (tile_pipe.h; phase clocks and the MFMA-pipe counter: profiles/r05_tile_pipe_ab.txt, r05_tile_pipe_pmc.txt); the same batch through "
                             "the streaming kernel alone (kpop_tune(\"dense\", 0)) is L2-latency-bound at ~100 SIMD-cycles a window"}}
    api.tune("dense", 0)
    try:
        ms0, _ = _event_ms(R, call, 2, 1)
    finally:
        api.tune("dense", 2)
    res["one_organism_0.3pct"]["streaming_kernel_alone_ms"] = ms0
    res["one_organism_0.3pct"]["speedup_over_streaming_kernel"] = ms0 / ms
    # ... and the same batch through twisters of more dimensions: beyond 64 the tile kernel takes the columns unit by unit in three stages
    # a block (tile_pipe.h, WIDE) -- at 256 dimensions and at the 1,635 of the reference's own large run (README.md:1029; k = 10 as there)
    dims = {str(d): {"k": k, "ms": ms, "roofline": dict(res["one_organism_0.3pct"]["roofline"]), "streaming_kernel_alone_ms": ms0, "speedup_over_streaming_kernel": ms0 / ms}}
    del out
    for kk, dd in ((12, 256), (10, 1635)):
        free, _ = t.cuda.mem_get_info(R.dev)
        need = ((4 ** kk) // 2) * dd * 8 * 1.05 + windows / 512.0 * dd * 8 * 1.1 + 6e9
        if free < need:
            dims[str(dd)] = {"k": kk, "skipped": "needs %.0f GB of free HBM, %.1f GB free" % (need / 1e9, free / 1e9)}
            continue
        tw = R.kpop.Twister.synth(TWISTER_SEED, kk, dd)
        try:
            outd = t.zeros(n, dd, dtype=t.float64, device=R.dev)
            calld = lambda: api.dev_count_twist(tw, bases.data_ptr(), offs.data_ptr(), n, n * L, L, outd.data_ptr(), stream=R.sp)
            msd, alld = _event_ms(R, calld, 3, 1)
            api.debug_counters(16)
            api.tune("dbg", 32 << 24)
            calld()
            cnt = api.debug_counters(16)
            api.tune("dbg", 0)
            fl = 2.0 * 64 * dd * cnt[15]
            api.tune("dense", 0)
            try:
                ms0d, _ = _event_ms(R, calld, 1, 1)
            finally:
                api.tune("dense", 2)
            dims[str(dd)] = {
                "k": kk, "ms": msd, "ms_all": alld, "twister_GB": tw.info()["device_bytes"] / 1e9,
                "roofline": {"kernel": "count_twist_tile_pipe_kernel", "bound": "mfma", "achieved": fl / (msd * 1e-3) / 1e12, "peak": MFMA_F64_PEAK_TFLOPS, "unit": "TFLOP/s",
                             "frac": fl / (msd * 1e-3) / 1e12 / MFMA_F64_PEAK_TFLOPS, "traffic": None, "algorithmic_flops_per_launch": fl, "avg_launch_ms": msd,
                             "chunks_taken": cnt[14], "set_rows_multiplied": cnt[15],
                             "residual_rows_GB": (windows * 0.036) * dd * 8 / 1e9,
                             "note": "flops as for 64 dimensions (2 x 64 x D x set rows multiplied), over the whole call. At 0.3 % substitutions 3.6 % of the windows "
                                     "are private to their sequence: their rows (residual_rows_GB, D x 8 bytes each) come from HBM whatever the scheme -- 0.095 B per "
                                     "flop of the consensus where the machine has 0.064 (5 TB/s of gathered rows : 78.6 TFLOP/s): the HBM gather, not the matrix "
                                     "pipe, bounds this route at 0.6-0.67 of the matrix peak"},
                "streaming_kernel_alone_ms": ms0d, "speedup_over_streaming_kernel": ms0d / msd}
            del outd
        finally:
            tw.free()
        t.cuda.empty_cache()
    res["one_organism_dims"] = dims
    return res


def config3_distances_leg(R):
    """the distance step at the reference's own size (README.md:1054-1060: 650 K samples x 1,636 classes x 1,635 dimensions; :1101: 300
    neighbours in a database of 650 K x 1,635): kpop_dev_distance_rowwise of 100,000 x 1,636 x 1,635 and kpop_dev_distance_summary of
    256 x 650,000 x 1,635 on the f64 matrix cores (distance_mfma.hip), beside the vector pipe (kpop_tune(..., 0))"""
    t, api = R.torch, R.api
    free, _ = t.cuda.mem_get_info(R.dev)
    if free < 40e9:
        return {"skipped": "needs 40 GB of free HBM, %.1f GB free" % (free / 1e9)}
    g = t.Generator(device=R.dev)
    g.manual_seed(1)
    d, r1, r2 = 1635, 1636, 100000
    res = {"workload": "synthetic rows (standard normal), D = %d, euclidean, distance-normalised, metric random" % d}
    metric = t.rand(d, dtype=t.float64, device=R.dev, generator=g) + 0.1
    metric /= metric.sum()
    m1 = t.randn(r1, d, dtype=t.float64, device=R.dev, generator=g)
    m2 = t.randn(r2, d, dtype=t.float64, device=R.dev, generator=g)
    work = t.empty(api.dev_distance_workspace_bytes(r1, r2, d), dtype=t.uint8, device=R.dev)
    out = t.empty(r2, r1, dtype=t.float64, device=R.dev)
    call = lambda: api.dev_distance_rowwise(m1.data_ptr(), r1, m2.data_ptr(), r2, d, metric.data_ptr(), work.data_ptr(), out.data_ptr(), stream=R.sp)
    ms, all_ms = _event_ms(R, call, 3, 1)
    got = out.clone()
    api.tune("distance_mfma", 0)
    try:
        ms0, _ = _event_ms(R, call, 1, 1)
    finally:
        api.tune("distance_mfma", 1)
    rel = float(((got - out).abs() / out.abs().clamp_min(1e-300)).max())
    fl = 2.0 * r1 * r2 * d
    res["distance_rowwise_100k_x_1636"] = {
        "ms": ms, "ms_all": all_ms, "value": r2 / (ms * 1e-3), "unit": "sequences/sec",
        "roofline": {"kernel": "distance_gemm_mfma_kernel", "bound": "mfma", "achieved": fl / (ms * 1e-3) / 1e12, "peak": MFMA_F64_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": fl / (ms * 1e-3) / 1e12 / MFMA_F64_PEAK_TFLOPS, "traffic": None, "algorithmic_flops_per_launch": fl, "avg_launch_ms": ms,
                     "note": "flops = 2 x rows x rows x D of the contraction a . (b m), over the whole call (both operands' norms and divisions, the contraction, "
                             "square roots, the pairs that cancel recomputed with the reference's chain)"},
        "vector_pipe_ms": ms0, "speedup_over_vector_pipe": ms0 / ms, "max_relative_difference_from_the_vector_pipe": rel}
    del m2, out, got, work
    t.cuda.empty_cache()
    r1s, q = 650000, 256
    db = t.randn(r1s, d, dtype=t.float64, device=R.dev, generator=g)
    qs = db[t.randperm(r1s, device=R.dev, generator=g)[:q]].clone()
    work = t.empty(api.dev_distance_workspace_bytes(r1s, q, d), dtype=t.uint8, device=R.dev)
    K = 304
    stats = t.zeros(q, 4, dtype=t.float64, device=R.dev)
    nn = t.zeros(q, dtype=t.int32, device=R.dev)
    idx = t.zeros(q, K, dtype=t.int32, device=R.dev)
    dd = t.zeros(q, K, dtype=t.float64, device=R.dev)
    zz = t.zeros_like(dd)
    calls = lambda: api.dev_distance_summary(db.data_ptr(), r1s, qs.data_ptr(), q, d, metric.data_ptr(), work.data_ptr(), stats.data_ptr(), nn.data_ptr(),
                                             idx.data_ptr(), dd.data_ptr(), zz.data_ptr(), keep_at_most=300, max_neighbours=K, stream=R.sp)
    mss, alls = _event_ms(R, calls, 3, 1)
    a = [x.clone() for x in (stats, nn, idx, dd)]
    api.tune("summary_mfma", 0)
    try:
        mss0, _ = _event_ms(R, calls, 1, 1)
    finally:
        api.tune("summary_mfma", 1)
    same = bool(t.equal(a[0][:, 2:], stats[:, 2:]) and t.equal(a[1], nn) and t.equal(a[2][:, :300], idx[:, :300]) and t.equal(a[3][:, :300], dd[:, :300]))
    fls = 2.0 * r1s * q * d
    res["distance_summary_256_x_650k"] = {
        "ms": mss, "ms_all": alls, "value": q / (mss * 1e-3), "unit": "query rows/sec", "keep_at_most": 300,
        "roofline": {"kernel": "distance_gemm_mfma_kernel", "bound": "mfma", "achieved": fls / (mss * 1e-3) / 1e12, "peak": MFMA_F64_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": fls / (mss * 1e-3) / 1e12 / MFMA_F64_PEAK_TFLOPS, "traffic": None, "algorithmic_flops_per_launch": fls, "avg_launch_ms": mss,
                     "note": "the whole call (norms, contraction, the summary's selection on the approximate rows, the exact refinement) over the contraction's flops"},
        "vector_pipe_ms": mss0, "speedup_over_vector_pipe": mss0 / mss, "medians_mads_neighbours_same_bits_as_the_vector_pipe": same}
    # ... and with enough query rows for the call's fixed costs (the reference set divided by its norms: 17 GB moved) to be shared:
    # what one call of the reference's own job (650 K query rows, README.md:1101) does chunk after chunk
    try:
        q2 = 4096
        qs2 = db[t.randperm(r1s, device=R.dev, generator=g)[:q2]].clone()
        work2 = t.empty(api.dev_distance_workspace_bytes(r1s, q2, d), dtype=t.uint8, device=R.dev)
        st2, n2 = t.zeros(q2, 4, dtype=t.float64, device=R.dev), t.zeros(q2, dtype=t.int32, device=R.dev)
        i2, d2, z2 = t.zeros(q2, K, dtype=t.int32, device=R.dev), t.zeros(q2, K, dtype=t.float64, device=R.dev), t.zeros(q2, K, dtype=t.float64, device=R.dev)
        call2 = lambda: api.dev_distance_summary(db.data_ptr(), r1s, qs2.data_ptr(), q2, d, metric.data_ptr(), work2.data_ptr(), st2.data_ptr(), n2.data_ptr(),
                                                 i2.data_ptr(), d2.data_ptr(), z2.data_ptr(), keep_at_most=300, max_neighbours=K, stream=R.sp)
        ms2, all2 = _event_ms(R, call2, 2, 1)
        fl2 = 2.0 * r1s * q2 * d
        res["distance_summary_4096_x_650k"] = {"ms": ms2, "ms_all": all2, "value": q2 / (ms2 * 1e-3), "unit": "query rows/sec", "keep_at_most": 300,
                                               "roofline": {"kernel": "distance_gemm_mfma_kernel", "bound": "mfma", "achieved": fl2 / (ms2 * 1e-3) / 1e12, "peak": MFMA_F64_PEAK_TFLOPS,
                                                            "unit": "TFLOP/s", "frac": fl2 / (ms2 * 1e-3) / 1e12 / MFMA_F64_PEAK_TFLOPS, "traffic": None,
                                                            "algorithmic_flops_per_launch": fl2, "avg_launch_ms": ms2},
                                               "all_650k_query_rows_at_this_rate_s": 650000 / (q2 / (ms2 * 1e-3))}
    except Exception as e:
        res["distance_summary_4096_x_650k"] = {"skipped": "%r" % (e,)}
    # ... and a database laid out lineage by lineage (what 650 K genomes sorted by lineage are: clusters of near-identical rows, cluster after
    # cluster -- neighbouring entries of a distance row are near-copies of each other): the same call beside the same database shuffled.  The
    # summary's brackets come from a sample of the reference ROWS, so the layout must not matter (it did until late in round 6: x 3.6)
    try:
        del db, work
        t.cuda.empty_cache()
        dl, r1l, ql, members = 64, 1000000, 256, 100
        ml = t.rand(dl, dtype=t.float64, device=R.dev, generator=g) + 0.1
        ml /= ml.sum()
        centres = t.randn((r1l + members - 1) // members, dl, dtype=t.float64, device=R.dev, generator=g)
        dbl = centres.repeat_interleave(members, dim=0)[:r1l].clone()
        dbl += 1e-3 * t.randn(r1l, dl, dtype=t.float64, device=R.dev, generator=g)
        perm = t.randperm(r1l, device=R.dev, generator=g)
        ql_rows = dbl[perm[:ql]].clone()
        dbs = dbl[perm].clone()  # the same rows, shuffled
        workl = t.empty(api.dev_distance_workspace_bytes(r1l, ql, dl), dtype=t.uint8, device=R.dev)
        stl, nl = t.zeros(ql, 4, dtype=t.float64, device=R.dev), t.zeros(ql, dtype=t.int32, device=R.dev)
        il, dl_, zl = t.zeros(ql, K, dtype=t.int32, device=R.dev), t.zeros(ql, K, dtype=t.float64, device=R.dev), t.zeros(ql, K, dtype=t.float64, device=R.dev)
        out_l = {}
        for name, ref in (("sorted_by_lineage", dbl), ("shuffled", dbs)):
            calll = lambda: api.dev_distance_summary(ref.data_ptr(), r1l, ql_rows.data_ptr(), ql, dl, ml.data_ptr(), workl.data_ptr(), stl.data_ptr(), nl.data_ptr(),
                                                     il.data_ptr(), dl_.data_ptr(), zl.data_ptr(), keep_at_most=300, max_neighbours=K, stream=R.sp)
            api.tune("summary_audit", 1)
            api.summary_fallbacks()
            msl, _ = _event_ms(R, calll, 3, 1)
            left = api.summary_fallbacks() // 4
            api.tune("summary_audit", 0)
            msl, alll = _event_ms(R, calll, 3, 1)
            out_l[name] = {"ms": msl, "ms_all": alll, "rows_through_a_slow_path_per_call": left, "medians": stl[:, 2].clone()}
        same_medians = bool(t.equal(out_l["sorted_by_lineage"].pop("medians"), out_l["shuffled"].pop("medians")))
        res["distance_summary_256_x_1M_lineages"] = {
            "workload": "%d reference rows x %d dimensions in clusters of %d near-identical rows (noise 1e-3), %d query rows out of them, 300 neighbours" % (r1l, dl, members, ql),
            "sorted_by_lineage": out_l["sorted_by_lineage"], "shuffled": out_l["shuffled"], "same_medians_either_layout": same_medians,
            "note": "tests/test_gpu_distance.py::test_distance_summary_against_a_database_laid_out_lineage_by_lineage; profiles/r06_summary_lanes.txt 10."}
    except Exception as e:
        res["distance_summary_256_x_1M_lineages"] = {"skipped": "%r" % (e,)}
    return res


def dims_sweep(R):
    """SURVEY 8(d)'s sweep of the headline reads kernel over D in {9, 64, 256, 1635}: 100,000 x 150 bp, the fused count->twist alone
    (k = 12 up to 256 dimensions; k = 10 at 1,635, where a k = 12 twister would be 110 GB -- 6.9 GB is still 27 Infinity Caches)"""
    t, api = R.torch, R.api
    n, L = 100000, 150
    bases = t.empty(n * L, dtype=t.uint8, device=R.dev)
    offs = t.empty(n + 1, dtype=t.int64, device=R.dev)
    api.dev_synth_reads(READ_SEED, n, L, bases.data_ptr(), offs.data_ptr(), stream=R.sp)
    res = {"workload": "%d reads x %d bp, the fused count->twist kernel alone, by the twister's dimensions" % (n, L)}
    for k, d in ((12, 9), (12, 64), (12, 256), (10, 1635)):
        tw = R.kpop.Twister.synth(TWISTER_SEED, k, d)
        try:
            out = t.zeros(n, d, dtype=t.float64, device=R.dev)
            ms, _ = _event_ms(R, lambda: api.dev_count_twist(tw, bases.data_ptr(), offs.data_ptr(), n, n * L, L, out.data_ptr(), stream=R.sp), 10, 2)
            w = L - k + 1
            alg = n * (L + w * d * 8 + d * 8)
            line_b = max(128, ((d + 15) // 16) * 128)  # what a row costs the memory system: whole 128-byte lines of its d_pad x 8 bytes
            res[str(d)] = {"k": k, "ms": ms, "twister_GB": tw.info()["device_bytes"] / 1e9,
                           "roofline": {"kernel": "count_twist_wave_kernel", "bound": "hbm", "achieved": alg / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                        "frac": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes_per_launch": alg, "avg_launch_ms": ms,
                                        "line_bytes_per_launch": n * w * line_b}}
            del out
        finally:
            tw.free()
        t.cuda.empty_cache()
    return res


def config5_leg(R):
    """BASELINE config 5: k = 15 (536,870,912 canonical 15-mers), D = 16, 10,000 reads x 150 bp: the whole 69 GB twister on this GPU"""
    t, api, kp = R.torch, R.api, R.kpop
    k, d, n, L = 15, 16, 10000, 150
    free, _ = t.cuda.mem_get_info(R.dev)
    if free < 150e9:
        return {"skipped": "needs 150 GB of free HBM for the 69 GB twister and its synthesis (and 137 GB more for its rows at their hashes, built when they fit), %.1f GB free" % (free / 1e9)}
    t0 = time.perf_counter()
    tw = kp.Twister.synth(TWISTER_SEED, k, d)
    t.cuda.synchronize()
    synth_s = time.perf_counter() - t0
    try:
        bases = t.empty(n * L, dtype=t.uint8, device=R.dev)
        offs = t.empty(n + 1, dtype=t.int64, device=R.dev)
        api.dev_synth_reads(READ_SEED, n, L, bases.data_ptr(), offs.data_ptr(), stream=R.sp)
        out = t.zeros(n, d, dtype=t.float64, device=R.dev)
        ms, all_ms = _event_ms(R, lambda: api.dev_count_twist(tw, bases.data_ptr(), offs.data_ptr(), n, n * L, L, out.data_ptr(), stream=R.sp), 20, 3)
        w = L - k + 1
        alg = n * (L + w * d * 8 + d * 8)
        finite = bool(t.isfinite(out).all().item()) and bool((out.abs() < 1.0).all().item())
        return {
            "workload": "%d reads x %d bp, k=%d DNA-ds, D=%d; twister %d rows, %.1f GB resident%s (synthesised on the device in %.1f s, outside the timing)"
                        % (n, L, k, d, tw.info()["n_cols"], tw.info()["device_bytes"] / 1e9,
                           (" -- %.1f GB of it the rows once more at their hashes: no name -> row look-up" % (tw.info()["direct_bytes"] / 1e9)) if tw.info()["direct_bytes"] else "",
                           synth_s),
            "rows_at_their_hashes": bool(tw.info()["direct_bytes"]),
            "value": n / (ms * 1e-3), "unit": "sequences/sec", "ms_per_step": ms,
            "rows_finite_and_inside_the_coefficient_range": finite,
            "roofline": {"kernel": "count_twist_wave_kernel", "bound": "hbm", "achieved": alg / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes_per_launch": alg, "avg_launch_ms": ms,
                         "note": "128-byte rows (16 dims) gathered at random, every one an HBM miss. A complete twister of this shape keeps its rows a second "
                                 "time at the address the hash names (137 GB beside the 69 GB in rank order; twister.h `direct`), so a window costs ONE miss and one "
                                 "cache line: through the name -> row index (143 MB of 64-byte blocks) every window also moved a 128-byte line of index -- 1.96 x the "
                                 "algorithmic bytes, measured -- and waited for two dependent misses. 2,000 / 10,000 / 30,000 / 100,000 reads: 0.015 / 0.042 / 0.107 / "
                                 "0.321 ms with the rows at their hashes (5.4 TB/s of rows at 100,000), 0.017 / 0.059 / 0.157 / 0.500 ms through the index "
                                 "(tools/probes/r05_direct_rows.py, profiles/r05_direct_rows.txt); 10,000 wavefronts are 1.2 rounds of the chip"}}
    finally:
        tw.free()


def leg_traffic(R, roofline, **kw):
    """a leg's HBM bytes per launch, measured NOW (measure_traffic: two --pmc child passes on the leg's shape), into its roofline"""
    if R.args.no_children:
        roofline["traffic_source"] = "not measured: --no-children"
        return
    R.torch.cuda.empty_cache()  # (the child needs the HBM this process has cached)
    live = measure_traffic(R, **kw)
    if not live:
        roofline["traffic_source"] = "not measured: a rocprofv3 --pmc child pass failed or timed out"
        return
    roofline["traffic"] = live["hbm_bytes_per_launch"]
    roofline["traffic_source"] = live["source"]
    roofline["traffic_counters"] = {"FETCH_SIZE_KiB": live["FETCH_SIZE_KiB"], "WRITE_SIZE_KiB": live["WRITE_SIZE_KiB"]}
    alg = roofline.get("algorithmic_bytes_per_launch")
    if alg:
        roofline["traffic_over_algorithmic"] = live["hbm_bytes_per_launch"] / alg


def extra_configs(R):
    legs = {}
    for name, fn in (("config2_on_this_gpu", config2_leg), ("config3_on_this_gpu", config3_leg), ("config3_distances", config3_distances_leg), ("dims_sweep", dims_sweep), ("config5_on_this_gpu", config5_leg)):
        t0 = time.perf_counter()
        try:
            legs[name] = fn(R)
        except Exception as e:  # a leg is a report, never a reason to lose the headline line
            legs[name] = {"skipped": "leg failed: %r" % (e,)}
        R.torch.cuda.synchronize()
        R.torch.cuda.empty_cache()
        legs[name]["leg_seconds"] = time.perf_counter() - t0
    # the legs' HBM traffic, measured now (counter passes of child processes, after every timed leg: see the headline's)
    t0 = time.perf_counter()
    want = (("config2_on_this_gpu", None, dict(n_reads=10000, k=10, dims=64, read_len=150)),
            ("config3_on_this_gpu", "unrelated_genomes", dict(n_reads=50000, k=R.args.k, dims=R.args.dims, read_len=30000, kernel="count_twist_stream_kernel", launches=2, timeout=400)),
            ("config3_on_this_gpu", "one_organism_0.3pct", dict(n_reads=50000, k=R.args.k, dims=R.args.dims, read_len=30000, kernel="count_twist_tile_pipe_kernel", launches=2,
                                                                mutants=0.003, timeout=400)),
            ("config3_on_this_gpu", ("one_organism_dims", "256"), dict(n_reads=50000, k=12, dims=256, read_len=30000, kernel="count_twist_tile_pipe_kernel", launches=2,
                                                                        mutants=0.003, timeout=400)),
            ("config3_on_this_gpu", ("one_organism_dims", "1635"), dict(n_reads=50000, k=10, dims=1635, read_len=30000, kernel="count_twist_tile_pipe_kernel", launches=2,
                                                                         mutants=0.003, timeout=400)),
            ("dims_sweep", "9", dict(n_reads=100000, k=12, dims=9, read_len=150)),
            ("dims_sweep", "256", dict(n_reads=100000, k=12, dims=256, read_len=150)),
            ("dims_sweep", "1635", dict(n_reads=100000, k=10, dims=1635, read_len=150)),
            ("config5_on_this_gpu", None, dict(n_reads=10000, k=15, dims=16, read_len=150)))
    for name, sub, kw in want:
        leg = legs.get(name, {})
        for key in ((sub,) if isinstance(sub, str) else (sub or ())):
            leg = leg.get(key, {}) if isinstance(leg, dict) else {}
        if isinstance(leg.get("roofline"), dict):
            try:
                leg_traffic(R, leg["roofline"], **kw)
            except Exception as e:
                leg["roofline"]["traffic_source"] = "not measured: %r" % (e,)
    legs["traffic_passes_seconds"] = time.perf_counter() - t0
    try:  # (the 64-dimension entry of the sweep IS the one-organism leg: its traffic with it)
        c3 = legs["config3_on_this_gpu"]
        for key in ("traffic", "traffic_source", "traffic_counters"):
            if key in c3["one_organism_0.3pct"]["roofline"]:
                c3["one_organism_dims"]["64"]["roofline"][key] = c3["one_organism_0.3pct"]["roofline"][key]
    except Exception:
        pass
    return legs


def cpu_baseline(R, n_reads_gpu, gpu_twisted, gpu_dist, classes_host):
    """The oracle (CPU restatement of the reference's algorithm, kind "port") timed on this box's host
    cores on a bounded sample of the same workload; also used to check the GPU results of that sample."""
    import numpy as np
    from oracle import oracle as O
    args = R.args
    threads = os.cpu_count() or 1
    k, d, L = args.k, args.dims, args.read_len
    t0 = time.time()
    cols = O.enumerate_kmers(k)
    T = O.synth_twister(TWISTER_SEED, d, cols)  # the reference's dims-major layout
    setup_s = time.time() - t0
    calib = 2000
    bases, offs = O.synth_reads(READ_SEED, calib, L)
    _, _, secs = O.pipeline(bases, offs, k, T, cols, classes_host, R.metric_host, threads=threads)
    rate = calib / max(secs, 1e-9)
    n = int(min(n_reads_gpu, max(calib, rate * args.cpu_seconds)))
    bases, offs = O.synth_reads(READ_SEED, n, L)
    tw, di, secs = O.pipeline(bases, offs, k, T, cols, classes_host, R.metric_host, threads=threads)
    out = {"value": n / secs, "unit": "sequences/sec", "cores": threads, "kind": "port",
           "sample": "first %d of the %d synthetic reads, count->twist->distance in oracle/kpop_oracle.c, in memory "
                     "(OpenMP over reads; twister in the reference's dims-major layout, generated in %.1f s outside the "
                     "timed region).  The reference itself also writes the spectra as text (bin/KPopCount.ml:46), pipes "
                     "them and parses them back in a serial producer (lib/Twister.ml:91-145); the port skips both, so "
                     "the real OCaml path is slower than this figure" % (n, n_reads_gpu, setup_s)}
    m = min(n, gpu_twisted.shape[0])
    parity = {
        "reads_checked": m,
        "twisted_max_abs_err": float(np.max(np.abs(gpu_twisted[:m] - tw[:m]))),
        "twisted_bit_exact": bool(np.array_equal(gpu_twisted[:m], tw[:m])),
        "distance_max_rel_err": float(np.max(np.abs(gpu_dist[:m] - di[:m]) / np.maximum(di[:m], 1e-300))),
    }
    return out, parity


def bus_rates(lib, nbytes=256 << 20, reps=3):
    """page-locked H2D and D2H rates of this box (one large copy each way, best of `reps`), GB/s"""
    import ctypes as C
    hp, dp = C.c_void_p(), C.c_void_p()
    if lib.kpop_host_alloc(C.byref(hp), nbytes) or lib.kpop_dev_malloc(C.byref(dp), nbytes):
        raise RuntimeError(lib.kpop_last_error().decode())
    C.memset(hp, 1, nbytes)
    up, down = [], []
    try:
        for _ in range(reps + 1):
            t0 = time.perf_counter()
            lib.kpop_memcpy_h2d(dp, hp, nbytes)
            t1 = time.perf_counter()
            lib.kpop_memcpy_d2h(hp, dp, nbytes)
            t2 = time.perf_counter()
            up.append(t1 - t0)
            down.append(t2 - t1)
    finally:
        lib.kpop_dev_free(dp)
        lib.kpop_host_free(hp)
    return nbytes / min(up[1:]) / 1e9, nbytes / min(down[1:]) / 1e9


def pcie_inclusive(R, n, reps=5, stream_batches=10):
    """The same step from host memory to host memory through the streaming pipeline (kpop_pipeline_*): reads start in
    the caller's page-locked buffers, results end there; H2D of chunk c+1, kernels of chunk c and D2H of chunk c-1
    overlap on three streams.  Reported: the steady state (several batches in flight, what a streaming caller gets) and
    the single call (fill and drain included), for both outputs and for the outputs `-d` / `-s` need; each against
    this box's measured page-locked bus rates."""
    import ctypes as C
    import numpy as np
    from kpop_amd import _lib
    import kpop_amd
    a = R.args
    lib = _lib.load()
    L, d, Cn = a.read_len, a.dims, a.classes
    bases_d, offs_d = R.synth_reads(n, 0)
    R.torch.cuda.synchronize()
    bases = kpop_amd.host_empty(n * L, np.uint8)
    bases[:] = bases_d.cpu().numpy()
    offs = kpop_amd.host_empty(n + 1, np.uint64)
    offs[:] = offs_d.cpu().numpy().astype(np.uint64)
    classes = R.classes.cpu().numpy()
    h2d, d2h = bus_rates(lib)
    out = {"bus": {"h2d_GBps": h2d, "d2h_GBps": d2h, "note": "one 256 MiB page-locked copy each way, best of 3, measured in this run"}}

    def leg(outputs, label):
        pl = kpop_amd.Pipeline(R.tw, classes, R.metric_host, outputs=outputs, keep_at_most=2, max_neighbours=8)
        o = pl.alloc_outputs(n)
        pl.run(bases, offs, o)  # sizes the ring
        pl.run(bases, offs, o)
        single = []
        for _ in range(reps):
            t0 = time.perf_counter()
            pl.collect(pl.submit(bases, offs, o))
            single.append(time.perf_counter() - t0)
        best_stream = None
        for _ in range(3):
            t0 = time.perf_counter()
            tickets = [pl.submit(bases, offs, o) for _ in range(stream_batches)]
            pl.collect(tickets[-1])
            dt = (time.perf_counter() - t0) / stream_batches
            best_stream = dt if best_stream is None else min(best_stream, dt)
        st = pl.stats()
        up = int(bases.nbytes + offs.nbytes)
        down = int(sum(v.nbytes for v in o.values()))
        bound = max(up / (h2d * 1e9), down / (d2h * 1e9))
        res = {"outputs": label, "value": n / best_stream, "unit": "sequences/sec", "ms_per_batch": best_stream * 1e3,
               "single_call": {"value": n / min(single), "ms": min(single) * 1e3},
               "bytes_up": up, "bytes_down": down, "chunks": st["chunks"], "ring_depth": st["depth"], "pinned": st["pinned"],
               "bus_bound_ms": bound * 1e3, "pcie_roofline": bound / best_stream,
               "pcie_roofline_single_call": bound / min(single)}
        pl.close()
        return res, o

    both, o_both = leg(kpop_amd.OUT_TWISTED | kpop_amd.OUT_DISTANCES, "twisted rows + distances")
    out.update(both)
    out["note"] = ("kpop_pipeline_submit/collect from page-locked host buffers, %d batches of %d reads in flight (best of 3); "
                   "single_call = one submit + collect, best of %d" % (stream_batches, n, reps))
    out["distances_only"], o_d = leg(kpop_amd.OUT_DISTANCES, "distances (what -d needs)")
    out["summary_only"], _ = leg(kpop_amd.OUT_SUMMARY, "per-read summary (what -s needs)")
    out["pcie_roofline_note"] = ("bus_bound_ms = max(bytes_up / h2d, bytes_down / d2h) at the rates measured above (PCIe is full duplex); "
                                 "pcie_roofline = bus_bound_ms / ms_per_batch.  distances_only and summary_only are kernel-bound "
                                 "(the kernels take ms_per_step), so their fraction of the BUS is low by construction")
    # the pipeline's rows are the device-resident path's rows
    out["matches_device_resident"] = bool(np.array_equal(o_both["distances"], o_d["distances"]))
    # round 2's route for comparison: two separate host entry points from pageable memory, nothing overlapped
    twisted = np.zeros((n, d))
    dist = np.zeros((n, Cn))
    pb, po = np.array(bases), np.array(offs)
    p = lambda arr, ty: arr.ctypes.data_as(C.POINTER(ty))

    def once():
        rc = lib.kpop_count_twist(R.tw.handle, p(pb, C.c_uint8), p(po, C.c_uint64), n, 0, 1, p(twisted, C.c_double))
        rc = rc or lib.kpop_distance_rowwise(p(classes, C.c_double), Cn, p(twisted, C.c_double), n, d,
                                             p(R.metric_host, C.c_double), 0, 2.0, 1, p(dist, C.c_double))
        if rc:
            raise RuntimeError(lib.kpop_last_error().decode())
    once()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        once()
        ts.append(time.perf_counter() - t0)
    out["separate_calls_pageable"] = {"value": n / min(ts), "unit": "sequences/sec", "ms_per_step": min(ts) * 1e3,
                                      "note": "kpop_count_twist + kpop_distance_rowwise from pageable buffers, serial (round 2's pcie_inclusive)"}
    out["matches_separate_calls"] = bool(np.array_equal(o_both["twisted"], twisted) and np.array_equal(o_both["distances"], dist))
    return out


def pcie_inclusive_config3(R, n=5000, L=30000, reps=5):
    """BASELINE config 3's kind of batch (assemblies of one organism, 0.3 % substitutions; 5,000 x 30 kb = 150 MB: a tenth of it) from
    host memory to host memory through the streaming pipeline, one byte a base against the packed form (packed.hip: 2.25 bits a base):
    a host caller of assemblies is bound by the BUS on ASCII -- the kernels take a third of the time the bytes take to arrive"""
    import numpy as np
    from kpop_amd import _lib
    import kpop_amd
    t, api = R.torch, R.api
    lib = _lib.load()
    bases_d, offs_d = _mutants_on_device(R, n, L, 0.003, 0x0123)
    t.cuda.synchronize()
    bases = kpop_amd.host_empty(n * L, np.uint8)
    bases[:] = bases_d.cpu().numpy()
    offs = kpop_amd.host_empty(n + 1, np.uint64)
    offs[:] = offs_d.cpu().numpy().astype(np.uint64)
    out_d = t.zeros(n, R.args.dims, dtype=t.float64, device=R.dev)
    ms_kernel, _ = _event_ms(R, lambda: api.dev_count_twist(R.tw, bases_d.data_ptr(), offs_d.data_ptr(), n, n * L, L, out_d.data_ptr(), stream=R.sp), 5, 2)
    # the packed form (a host that keeps its sequences packed does this once, not per call: timed, reported, not part of the step)
    cw, mw = int(lib.kpop_packed_code_words(n * L)), int(lib.kpop_packed_mask_words(n * L))
    codes, invalid = kpop_amd.host_empty(cw, np.uint32), kpop_amd.host_empty(mw, np.uint32)
    pack_s = []
    for th in (1, 0):
        t0 = time.perf_counter()
        kpop_amd.check(lib.kpop_pack_bases(bases.ctypes.data, n * L, codes.ctypes.data, invalid.ctypes.data, th))
        pack_s.append(time.perf_counter() - t0)
    dc, dm = t.from_numpy(np.array(codes).view(np.int32)).to(R.dev), t.from_numpy(np.array(invalid).view(np.int32)).to(R.dev)
    spread = t.empty(n * L, dtype=t.uint8, device=R.dev)
    ms_unpack, _ = _event_ms(R, lambda: api.dev_unpack_bases(dc.data_ptr(), dm.data_ptr(), n * L, spread.data_ptr(), stream=R.sp), 5, 2)
    same_letters = bool(t.equal(spread, bases_d))
    h2d, d2h = bus_rates(lib)
    pl = kpop_amd.Pipeline(R.tw, outputs=kpop_amd.OUT_TWISTED)
    res = {}
    rows = {}
    for name, submit, up in (("ascii", lambda o: pl.submit(bases, offs, o), int(bases.nbytes + offs.nbytes)),
                             ("packed", lambda o: pl.submit_packed(codes, invalid, offs, o), int(codes.nbytes + invalid.nbytes + offs.nbytes))):
        o = pl.alloc_outputs(n)
        pl.collect(submit(o))
        pl.collect(submit(o))
        single = []
        for _ in range(reps):
            t0 = time.perf_counter()
            pl.collect(submit(o))
            single.append(time.perf_counter() - t0)
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            tickets = [submit(o) for _ in range(6)]
            pl.collect(tickets[-1])
            dt = (time.perf_counter() - t0) / 6
            best = dt if best is None else min(best, dt)
        down = int(o["twisted"].nbytes)
        bus = max(up / (h2d * 1e9), down / (d2h * 1e9))
        bound = max(bus, ms_kernel * 1e-3)
        res[name] = {"value": n / best, "unit": "sequences/sec", "ms_per_batch": best * 1e3, "single_call_ms": min(single) * 1e3, "bytes_up": up, "bytes_down": down,
                     "bus_bound_ms": bus * 1e3, "kernel_bound_ms": ms_kernel, "bound": "bus" if bus > ms_kernel * 1e-3 else "kernels",
                     "fraction_of_the_larger_bound": bound / best, "chunks": pl.stats()["chunks"]}
        rows[name] = np.array(o["twisted"])
    pl.close()
    res["workload"] = "%d assemblies x %d bp of one organism (0.3 %% substitutions), k=%d, D=%d, twisted rows back; six batches in flight, best of 3" % (n, L, R.args.k, R.args.dims)
    res["speedup_packed_over_ascii"] = res["ascii"]["ms_per_batch"] / res["packed"]["ms_per_batch"]
    res["same_rows_bit_for_bit"] = bool(np.array_equal(rows["ascii"], rows["packed"]))
    res["unpack_on_device"] = {"ms": ms_unpack, "GBps_of_bytes_written": n * L / (ms_unpack * 1e-3) / 1e9, "letters_equal_the_ascii_batch": same_letters}
    res["host_packing"] = {"one_thread_GBps": n * L / pack_s[0] / 1e9, "library_chosen_threads_GBps": n * L / pack_s[1] / 1e9,
                           "note": "kpop_pack_bases on this host (input bytes per second); not part of the step: a caller of the packed entry points keeps its sequences packed"}
    res["bus"] = {"h2d_GBps": h2d, "d2h_GBps": d2h}
    # the headline batch from its packed words, device-resident: reads of up to 512 windows are twisted straight from them (no bytes made)
    try:
        nr, Lr = 100000, R.args.read_len
        rb, ro = R.synth_reads(nr, 0)
        t.cuda.synchronize()
        hb = rb.cpu().numpy()
        cw2, mw2 = int(lib.kpop_packed_code_words(nr * Lr)), int(lib.kpop_packed_mask_words(nr * Lr))
        hc, hm = np.zeros(cw2, np.uint32), np.zeros(mw2, np.uint32)
        kpop_amd.check(lib.kpop_pack_bases(hb.ctypes.data, nr * Lr, hc.ctypes.data, hm.ctypes.data, 0))
        dcr, dmr = t.from_numpy(hc.view(np.int32)).to(R.dev), t.from_numpy(hm.view(np.int32)).to(R.dev)
        o1, o2 = t.zeros(nr, R.args.dims, dtype=t.float64, device=R.dev), t.zeros(nr, R.args.dims, dtype=t.float64, device=R.dev)
        ms_a, _ = _event_ms(R, lambda: api.dev_count_twist(R.tw, rb.data_ptr(), ro.data_ptr(), nr, nr * Lr, Lr, o1.data_ptr(), stream=R.sp), 10, 2)
        ms_p, _ = _event_ms(R, lambda: api.dev_count_twist_packed(R.tw, dcr.data_ptr(), dmr.data_ptr(), ro.data_ptr(), nr, nr * Lr, Lr, o2.data_ptr(), stream=R.sp), 10, 2)
        res["headline_reads_from_packed_words"] = {"ms_bytes": ms_a, "ms_packed": ms_p, "same_rows_bit_for_bit": bool(t.equal(o1, o2)),
                                                   "note": "count_twist_wave_kernel<..., PACKED>: a read's codes staged from the batch's words (15 words a 150-base read instead of 150 bytes)"}
    except Exception as e:
        res["headline_reads_from_packed_words"] = {"skipped": "%r" % (e,)}
    return res


def measure_traffic(R, n_reads, k=None, dims=None, read_len=None, kernel="count_twist_wave_kernel", launches=3, mutants=0.0, timeout=240):
    """HBM bytes per launch of the fused kernel, measured NOW: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; counters
    never share a run with a trace summary) of a child process that launches the same kernel on the same shape
    (tools/pmc_workload.py), corrected as MI355X_MICROARCH.md's HBM section prescribes (unit KiB; FETCH_SIZE reads one
    half on gfx950 -- checked here on the child's own calibration stream of a known byte count).  None when rocprofv3 is not
    there or a pass fails (the line then falls back to profiles/traffic.json and says so)."""
    import csv
    import glob
    import shutil
    import tempfile
    a = R.args
    if not shutil.which("rocprofv3"):
        return None
    got = {}
    calib = {}
    calib_rows = 1 << 20  # 512 MiB each way: beyond the Infinity Cache
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = tempfile.mkdtemp(prefix="kpop_pmc_", dir="/tmp")
            cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", out, "--",
                   sys.executable, os.path.join(ROOT, "tools", "pmc_workload.py"), "--reads", str(n_reads), "--read-len", str(read_len or a.read_len),
                   "-k", str(k or a.k), "--dims", str(dims or a.dims), "--launches", str(launches), "--calib-rows", str(calib_rows)]
            if mutants > 0.0:
                cmd += ["--mutants", repr(mutants)]
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
            if r.returncode != 0:
                return None
            vals, cal = [], []
            for f in glob.glob(os.path.join(out, "**", "*_counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if row["Counter_Name"] != counter:
                        continue
                    if kernel in row["Kernel_Name"]:
                        vals.append(float(row["Counter_Value"]))
                    elif "row_norms_kernel" in row["Kernel_Name"]:
                        cal.append(float(row["Counter_Value"]))
            shutil.rmtree(out, ignore_errors=True)
            if not vals:
                return None
            got[counter] = sorted(vals)[len(vals) // 2]
            calib[counter] = max(cal) if cal else None
    except Exception:
        return None
    known_kib = calib_rows * 64 * 8 / 1024.0  # the calibration kernel reads exactly this much
    factor = known_kib / calib["FETCH_SIZE"] if calib.get("FETCH_SIZE") else 2.0
    if not (1.8 <= factor <= 2.2):  # the documented gfx950 correction; anything else means the counter is not what we think
        return None
    return {"hbm_bytes_per_launch": (2.0 * got["FETCH_SIZE"] + got["WRITE_SIZE"]) * 1024.0, "FETCH_SIZE_KiB": got["FETCH_SIZE"],
            "WRITE_SIZE_KiB": got["WRITE_SIZE"], "fetch_correction_checked": factor,
            "source": "measured in this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes (separate) of a child process launching "
                      "the same kernel on the same shape (tools/pmc_workload.py); FETCH_SIZE x 2 (gfx950), unit KiB; the correction "
                      "checked on the child's 512 MiB calibration stream (%.3f)" % factor}


def file_to_file(R):
    """FASTA file -> .KPopTwisted / summary through the drop-in binaries, as README.md:606,656 chain them."""
    tool = os.path.join(ROOT, "tools", "file_to_file.py")
    if not os.path.exists(tool):
        return {"value": None, "note": "tools/file_to_file.py is missing"}
    try:
        out = subprocess.run([sys.executable, tool, "--reads", str(R.args.f2f_reads), "-k", str(R.args.k), "--json"],
                             stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
        if out.returncode != 0:
            return {"value": None, "note": "tools/file_to_file.py failed: " + out.stderr.decode("utf-8", "replace")[-400:]}
        return json.loads(out.stdout.decode().strip().splitlines()[-1])
    except Exception as e:  # the leg is a report, never a reason to lose the headline line
        return {"value": None, "note": "file-to-file leg failed: %r" % (e,)}


def in_process(args):
    """`--gpus N --in-process`: BASELINE config 4 (strong scaling) with every GPU driven from THIS process through the C
    ABI -- what an OCaml host gets: kpop_init_devices, one kpop_sharded job, per step kpop_sharded_resident_step (twist in
    chunks, hipMemcpyPeerAsync pushes of every finished chunk to all peers, distances to the classes).  No torch, no
    RCCL.  KPOP_BENCH_SHARE_GPU=1 aliases every slot to GPU 0 (a rig for one-GPU boxes, not a scaling number)."""
    import ctypes as C
    import numpy as np
    import kpop_amd
    from kpop_amd import _lib
    t_start = time.perf_counter()
    lib = _lib.load()
    shared = os.environ.get("KPOP_BENCH_SHARE_GPU") == "1"
    N = args.gpus
    kpop_amd.init_devices([0] * N if shared else list(range(N)))
    k, d, L, Cn = args.k, args.dims, args.read_len, args.classes
    reads = args.reads or 1000000
    tw = kpop_amd.Twister.synth(TWISTER_SEED, k, d)
    # class vectors: twist of C synthetic genomes, on slot 0
    vp = C.c_void_p

    def dmalloc(nbytes):
        h = vp()
        kpop_amd.check(lib.kpop_dev_malloc(C.byref(h), int(max(nbytes, 8))))
        return h
    cb, co, cl = dmalloc(Cn * args.class_len), dmalloc((Cn + 1) * 8), dmalloc(Cn * d * 8)
    kpop_amd.check(lib.kpop_dev_synth_reads(CLASS_SEED, Cn, args.class_len, 0, cb, co, None))
    kpop_amd.check(lib.kpop_dev_count_twist(tw.handle, cb, co, Cn, Cn * args.class_len, args.class_len, 0, 1, cl, None))
    classes = np.zeros((Cn, d))
    kpop_amd.check(lib.kpop_memcpy_d2h(classes.ctypes.data, cl, classes.nbytes))
    for h in (cb, co, cl):
        lib.kpop_dev_free(h)
    w = np.exp2(-np.arange(d, dtype=np.float64) / 8.0)
    metric = kpop_amd.metric_compute(w / w.sum())
    sh = kpop_amd.Sharded(tw, classes, metric, outputs=kpop_amd.OUT_TWISTED | kpop_amd.OUT_DISTANCES)
    d_bases, d_offs, n_reads, n_bases = [], [], [], []
    for s in range(N):
        lo, hi = kpop_amd.shard_bounds(reads, s, N)
        kpop_amd.use_device(s)
        pb, po = dmalloc((hi - lo) * L), dmalloc((hi - lo + 1) * 8)
        kpop_amd.check(lib.kpop_dev_synth_reads(READ_SEED, hi - lo, L, lo, pb, po, None))
        kpop_amd.check(lib.kpop_synchronize(None))
        d_bases.append(pb.value)
        d_offs.append(po.value)
        n_reads.append(hi - lo)
        n_bases.append((hi - lo) * L)
    kpop_amd.use_device(0)
    chunks = args.ag_chunks if args.ag_chunks > 0 else max(4, N)
    gather = N > 1 or args.force_dist
    startup_s = time.perf_counter() - t_start

    def step():
        sh.resident_step(d_bases, d_offs, n_reads, n_bases, L, chunks=chunks if gather else 1, gather=gather)
    for _ in range(args.warmup):
        step()
    t0 = time.perf_counter()
    tm, ck = [], []
    for _ in range(args.steps):
        step()  # returns when every device is done and every copy of the matrix is complete
        tm.append([sh.timings(s) for s in range(N)])
        ck.append([sh.chunk_timings(s) for s in range(N)])
    elapsed = time.perf_counter() - t0
    # the same steps again with nothing between them (no timing calls): what the host side costs a step = wall - the slowest
    # slot's own time from its rendezvous to its last push having landed
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    bare = (time.perf_counter() - t0) / args.steps * 1e3
    exposed = [float(np.mean([t[s]["ms_exposed_comm"] for t in tm])) for s in range(N)]
    compute = [float(np.mean([t[s]["ms_compute"] for t in tm])) for s in range(N)]
    # is every copy of the gathered matrix the union of the shards?  order-free checksum of the f64 bit patterns
    ok = None
    if gather:
        sums = []
        for s in sorted({0, N - 1}):
            full, first, rows, _ = sh.resident_buffers(s)
            kpop_amd.use_device(s)
            host = np.zeros((reads, d))
            kpop_amd.check(lib.kpop_memcpy_d2h(host.ctypes.data, full, host.nbytes))
            sums.append((int(host.view(np.int64).sum()), [int(host[a:b].view(np.int64).sum()) for a, b in
                                                          (kpop_amd.shard_bounds(reads, r, N) for r in range(N))]))
        kpop_amd.use_device(0)
        ok = all(x == sums[0] for x in sums)
    q = max(1, args.queries // N)
    t0 = time.perf_counter()
    qid, stats, nn, idx, dd, z = sh.all_vs_all_summary(queries_per_slot=q, keep_at_most=2, max_neighbours=8) if gather or N == 1 else (None,) * 6
    ava = time.perf_counter() - t0
    own = bool(qid is not None and all(dd[j, 0] == 0.0 and int(qid[j]) in idx[j, :min(int(nn[j]), 8)].tolist() for j in range(len(qid))))
    windows = max(L - k + 1, 0)
    per_read = L + windows * d * 8 + d * 8
    chunk_rows = -(-max(n_reads) // (chunks if gather else 1))
    all_chunks = [x for c in ck for sl in c for x in sl if x > 0]
    avg_chunk_ms = float(np.mean(all_chunks)) if all_chunks else None
    line = {
        "metric": "sequences/sec end-to-end count->twist->all-gather->distance, k=%d, %dk x %dbp in total" % (k, reads // 1000, L),
        "value": reads * args.steps / elapsed, "unit": "sequences/sec", "n_gpus": N, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "value_is": "device-resident; the all-gather of the twisted vectors is inside the timed region",
        "launcher": "in-process: one process, kpop_init_devices + kpop_sharded_resident_step (one host thread per device, "
                    "hipMemcpyPeerAsync pushes on one stream per destination); no torch.distributed, no RCCL",
        "config": {"read_len": L, "k": k, "n_dims": d, "n_classes": Cn, "class_len": args.class_len,
                   "workload": "BASELINE config 4: %d reads x %d bp in total, k=%d DNA-ds, sharded over %d GPU(s); twist in %d chunks, "
                               "every finished chunk pushed to all peers under the next; distances of the shard's rows to %d class vectors"
                               % (reads, L, k, N, chunks if gather else 1, Cn),
                   "reads_per_gpu": max(n_reads),
                   "sharding": "reads in contiguous shards (kpop_shard_bounds); twister, classes and metric replicated; ONE exchange: "
                               "all-gather of twisted vectors by peer copies over xGMI"},
        "roofline": {"kernel": "count_twist_wave_kernel", "bound": "hbm", "achieved": chunk_rows * per_read / (avg_chunk_ms * 1e-3) / 1e9 if avg_chunk_ms else None,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": chunk_rows * per_read / (avg_chunk_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if avg_chunk_ms else None,
                     "traffic": None, "algorithmic_bytes_per_launch": chunk_rows * per_read, "avg_launch_ms": avg_chunk_ms,
                     "note": "one launch per chunk of %d reads; HIP events on every slot's compute stream around each launch "
                             "(kpop_sharded_chunk_timings), averaged over slots, chunks and steps" % chunk_rows},
        "per_device_ms": {"compute_until_kernels_done": compute, "exposed_comm": exposed,
                          "count_twist_kernels_per_step": [float(np.mean([sum(c[s]) for c in ck])) for s in range(N)]},
        "host_overhead_ms_per_step": bare - max(float(np.mean([t[s]["ms_compute"] + t[s]["ms_exposed_comm"] for t in tm])) for s in range(N)),
        "host_overhead_is": "wall per step of %d back-to-back steps (%.3f ms) minus the slowest slot's own time from the step's rendezvous to its last "
                            "push having landed: job hand-over to the persistent slot threads, buffer checks, the one rendezvous, the return" % (args.steps, bare),
        "startup_s": startup_s,
        "gather_checksum_ok": ok,
        "all_vs_all": {"queries_total": 0 if qid is None else int(len(qid)), "against": reads, "seconds": ava,
                       "every_query_finds_itself_at_distance_0": own},
    }
    if shared:
        line["config"]["rig"] = "KPOP_BENCH_SHARE_GPU=1: every device slot is GPU 0; not a scaling number"
    line["n_ranks_seen"] = int(sum(1 for n in n_reads if n > 0))  # (device slots that twisted a shard and pushed it to the others)
    if os.environ.get("KPOP_BENCH_FALLBACK_REASON"):
        line["fallback"] = "in-process after RCCL failure: " + os.environ["KPOP_BENCH_FALLBACK_REASON"]
    sh.close()
    print(json.dumps(line))


def _config5_shape(args):
    """BASELINE config 5's shape unless the command line says otherwise (-k 12 / --dims 64 are the other workloads' defaults)"""
    k = 15 if args.k == 12 else args.k
    d = 16 if args.dims == 64 else args.dims
    return k, d, args.reads or 10000, args.read_len


def _config5_line(args, N, k, d, n, L, elapsed, launcher, sharding, extra):
    line = {"metric": "sequences/sec count->twist, k=%d, D=%d, %d x %d bp, the twister's k-mer rows over %d GPU(s)" % (k, d, n, L, N),
            "value": n * args.steps / elapsed, "unit": "sequences/sec", "n_gpus": N, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "value_is": "device-resident; the all-reduce of the partial rows is inside the timed region",
            "launcher": launcher,
            "config": {"workload": "BASELINE config 5: %d reads x %d bp, k=%d DNA-ds (%d canonical k-mers), D=%d" % (n, L, k, ((4 ** k) + (2 ** k if k % 2 == 0 else 0)) // 2, d),
                       "read_len": L, "k": k, "n_dims": d, "sharding": sharding}}
    line.update(extra)
    return line


def run_config5(R):
    """BASELINE config 5 for any number of ranks.  N = 1: the whole twister on the GPU (rows also at their hashes when they fit), the fused
    count->twist.  N > 1: rank r keeps the k-mer rows of hash range r (equal cuts of the hash space, kpop_amd.shard.kmer_slice_bounds;
    its rows ALSO at their hashes: a window that is another rank's costs nothing, one of its own ONE miss) with the all-ones column
    that sums `acc`; every rank twists ALL reads against its rows, un-normalised; ONE all-reduce of [n x (D + 1)] f64 (RCCL; through
    the host under KPOP_BENCH_SHARE_GPU=1), then the division by the reduced acc (lib/Twister.ml:158,177-183)."""
    t, api, kp, args = R.torch, R.api, R.kpop, R.args
    from kpop_amd.shard import kmer_slice_bounds
    k, d, n, L = _config5_shape(args)
    N = R.world
    sharded = N > 1 or args.force_dist
    if R.shared_gpu:
        api.tune("direct", 0)  # (every rank's tables on ONE GPU: the rows at their hashes would not fit twice)
    t0 = time.perf_counter()
    tw = kp.Twister.synth(TWISTER_SEED, k, d, hash_range=kmer_slice_bounds(k, R.rank, N), acc_dim=True) if sharded else kp.Twister.synth(TWISTER_SEED, k, d)
    t.cuda.synchronize()
    synth_s = time.perf_counter() - t0
    info = tw.info()
    bases = t.empty(n * L, dtype=t.uint8, device=R.dev)
    offs = t.empty(n + 1, dtype=t.int64, device=R.dev)
    api.dev_synth_reads(READ_SEED, n, L, bases.data_ptr(), offs.data_ptr(), stream=R.sp)
    cols = d + 1 if sharded else d
    part = t.zeros(n, cols, dtype=t.float64, device=R.dev)
    rows = t.zeros(n, d, dtype=t.float64, device=R.dev)
    e0, e1 = t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
    kms = []

    def step(_i):
        e0.record(R.stream)
        api.dev_count_twist(tw, bases.data_ptr(), offs.data_ptr(), n, n * L, L, part.data_ptr(), normalize=not sharded, stream=R.sp)
        e1.record(R.stream)
        if not sharded:
            return
        if R.shared_gpu:  # gloo: through the host
            h = part.cpu()
            R.dist.all_reduce(h)
            tot = h.to(R.dev)
        else:
            tot = part.clone()
            R.dist.all_reduce(tot)  # RCCL
        acc = tot[:, d:]
        t.where(acc != 0, tot[:, :d] / t.where(acc != 0, acc, t.ones_like(acc)), tot[:, :d], out=rows)
        t.cuda.synchronize()
        kms.append(e0.elapsed_time(e1))
    elapsed = R.timed(step, args.steps, args.warmup)
    if not sharded:
        t.cuda.synchronize()
        rows.copy_(part)
        ms_k, _ = _event_ms(R, lambda: api.dev_count_twist(tw, bases.data_ptr(), offs.data_ptr(), n, n * L, L, part.data_ptr(), stream=R.sp), 10, 1)
        own_windows = float(n * (L - k + 1))
    else:
        ms_k = float(sorted(kms[-args.steps:])[len(kms[-args.steps:]) // 2]) if kms else None
        own_windows = float(part[:, d].sum().item())  # the windows whose k-mer is one of this rank's rows (its share of acc)
    row_b = info["n_dims"] * 8
    alg = n * L + own_windows * row_b + n * cols * 8
    ranks_windows = R.gather_floats(own_windows)
    ranks_ms = R.gather_floats(ms_k or 0.0)
    ranks_bytes = R.gather_floats(float(info["device_bytes"]))
    ranks_direct = R.gather_floats(float(info.get("direct_bytes", 0)))
    finite = bool(t.isfinite(rows).all().item()) and bool((rows.abs() < 1.0).all().item())
    digest = {"sum": float(rows.sum().item()), "sum_of_squares": float((rows * rows).sum().item()), "row0": [float(x) for x in rows[0, :4].tolist()],
              "row_last": [float(x) for x in rows[n - 1, :4].tolist()]}
    line = None
    if R.rank == 0:
        line = _config5_line(args, N, k, d, n, L, elapsed,
                             "torch.distributed, one rank per GPU (%s)" % ("gloo through the host: KPOP_BENCH_SHARE_GPU=1, every rank on GPU 0 -- not a scaling number" if R.shared_gpu else "RCCL") if R.use_dist else "one process, no collective",
                             ("k-mer rows in %d equal cuts of the hash space, every rank twists ALL reads against its rows + the all-ones column; ONE exchange: all-reduce of "
                              "[%d x %d] f64 partial rows (%.2f MB a rank)" % (N, n, cols, n * cols * 8 / 1e6)) if sharded else "none: the whole twister on one GPU",
                             {"n_ranks_seen": int(sum(1 for w in ranks_windows if w > 0)) if sharded else 1,
                              "roofline": {"kernel": "count_twist_wave_kernel", "bound": "hbm", "achieved": alg / (ms_k * 1e-3) / 1e9 if ms_k else None, "peak": HBM_PEAK_GBS,
                                           "unit": "GB/s", "frac": alg / (ms_k * 1e-3) / 1e9 / HBM_PEAK_GBS if ms_k else None, "traffic": None,
                                           "algorithmic_bytes_per_launch": alg, "avg_launch_ms": ms_k,
                                           "note": "rank 0's launch: the reads' bytes + one row (%d B) per window whose k-mer is one of the rank's + the partial rows written" % row_b},
                              "per_rank": {"windows_of_own_kmers": ranks_windows, "count_twist_ms": ranks_ms, "twister_bytes": ranks_bytes, "rows_at_their_hashes_bytes": ranks_direct},
                              "twister_synth_s": synth_s, "startup_s": R.startup_s,
                              "rows_finite_and_inside_the_coefficient_range": finite, "rows_digest": digest})
    tw.free()
    return line


def in_process_config5(args):
    """`--gpus N --in-process --workload config5`: the same job from ONE process through the C ABI (kpop_init_devices, a slice of the
    twister a device slot, kpop_dev_count_twist on every slot's stream), the partial rows added on the host in slot order -- the
    launcher's fall-back when RCCL fails, and what a host without torch.distributed does.  No torch."""
    import ctypes as C
    import numpy as np
    import kpop_amd
    from kpop_amd import _lib
    from kpop_amd.shard import kmer_slice_bounds
    t_start = time.perf_counter()
    lib = _lib.load()
    shared = os.environ.get("KPOP_BENCH_SHARE_GPU") == "1"
    N = args.gpus
    kpop_amd.init_devices([0] * N if shared else list(range(N)))
    if shared:
        kpop_amd.api.tune("direct", 0)
    k, d, n, L = _config5_shape(args)
    sharded = N > 1 or args.force_dist
    cols = d + 1 if sharded else d
    vp = C.c_void_p

    def dmalloc(nbytes):
        h = vp()
        kpop_amd.check(lib.kpop_dev_malloc(C.byref(h), int(max(nbytes, 8))))
        return h
    tws, bufs = [], []
    for s_ in range(N):
        kpop_amd.use_device(s_)
        tws.append(kpop_amd.Twister.synth(TWISTER_SEED, k, d, hash_range=kmer_slice_bounds(k, s_, N), acc_dim=True) if sharded else kpop_amd.Twister.synth(TWISTER_SEED, k, d))
        pb, po, pp = dmalloc(n * L), dmalloc((n + 1) * 8), dmalloc(n * cols * 8)
        kpop_amd.check(lib.kpop_dev_synth_reads(READ_SEED, n, L, 0, pb, po, None))
        kpop_amd.check(lib.kpop_synchronize(None))
        bufs.append((pb, po, pp))
    host = [np.zeros((n, cols)) for _ in range(N)]
    rows = np.zeros((n, d))
    startup_s = time.perf_counter() - t_start

    def step():
        for s_ in range(N):  # (launches return at once: the slots' kernels run side by side)
            kpop_amd.use_device(s_)
            pb, po, pp = bufs[s_]
            kpop_amd.check(lib.kpop_dev_count_twist(tws[s_].handle, pb, po, n, n * L, L, 0, 0 if sharded else 1, pp, None))
        for s_ in range(N):
            kpop_amd.use_device(s_)
            kpop_amd.check(lib.kpop_memcpy_d2h(host[s_].ctypes.data, bufs[s_][2], host[s_].nbytes))
        tot = host[0].copy()
        for s_ in range(1, N):
            tot += host[s_]
        if sharded:
            acc = tot[:, d:]
            np.divide(tot[:, :d], np.where(acc != 0, acc, 1.0), out=rows)
        else:
            rows[:] = tot
    for _ in range(args.warmup):
        step()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    elapsed = time.perf_counter() - t0
    own = [float(h[:, d].sum()) if sharded else float(n * (L - k + 1)) for h in host]
    digest = {"sum": float(rows.sum()), "sum_of_squares": float((rows * rows).sum()), "row0": [float(x) for x in rows[0, :4]], "row_last": [float(x) for x in rows[n - 1, :4]]}
    line = _config5_line(args, N, k, d, n, L, elapsed,
                         "in-process: one process, kpop_init_devices + kpop_dev_count_twist on every slot, the partial rows added on the host in slot order; no torch.distributed, no RCCL",
                         ("k-mer rows in %d equal cuts of the hash space; ONE exchange: the [%d x %d] f64 partial rows to the host and their sum" % (N, n, cols)) if sharded else "none",
                         {"n_ranks_seen": int(sum(1 for w in own if w > 0)) if sharded else 1, "per_rank": {"windows_of_own_kmers": own},
                          "roofline": {"kernel": "count_twist_wave_kernel", "bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                                       "note": "the in-process form times whole steps only (launches on N devices + N downloads + the host's sum)"},
                          "startup_s": startup_s, "rows_finite_and_inside_the_coefficient_range": bool(np.isfinite(rows).all() and (np.abs(rows) < 1.0).all()), "rows_digest": digest})
    if shared:
        line["config"]["rig"] = "KPOP_BENCH_SHARE_GPU=1: every device slot is GPU 0; not a scaling number"
    if os.environ.get("KPOP_BENCH_FALLBACK_REASON"):
        line["fallback"] = "in-process after RCCL failure: " + os.environ["KPOP_BENCH_FALLBACK_REASON"]
    for tw in tws:
        tw.free()
    print(json.dumps(line))


def main():
    args = parse_args()
    if args.gpus < 1:
        sys.exit("--gpus must be positive")
    if args.in_process:
        if args.workload == "config5":
            in_process_config5(args)
        else:
            in_process(args)
        return
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.spawn):
        self_launch(args)  # never returns
    if "WORLD_SIZE" in os.environ:
        rank_preamble()
    workload = args.workload if args.workload != "auto" else ("headline" if args.gpus == 1 else "config4")
    scaling = args.scaling if args.scaling != "auto" else ("weak" if workload == "headline" else "strong")
    reads = args.reads or (100000 if workload == "headline" else 1000000)
    R = Rank(args)
    if workload == "config5":
        R.finish(run_config5(R))
        return
    k, d, L, C = args.k, args.dims, args.read_len, args.classes
    common_cfg = {"read_len": L, "k": k, "n_dims": d, "n_classes": C, "class_len": args.class_len,
                  "twister": "%d canonical k-mers x %d dims, f64, synthetic, replicated on every GPU" % (R.tw.info()["n_cols"], d)}
    if R.shared_gpu:
        common_cfg["rig"] = "KPOP_BENCH_SHARE_GPU=1: all ranks on one GPU, collectives through gloo + host staging; not a scaling number"

    if workload == "headline":
        if scaling == "weak":
            n_local, first, n_total = reads, R.rank * reads, reads * R.world
        else:
            from kpop_amd.shard import shard_bounds
            lo, hi = shard_bounds(reads, R.rank, R.world)
            n_local, first, n_total = hi - lo, lo, reads
        res = R.run_headline(n_local, first, args.steps, args.warmup, want_outputs=20000 if R.world == 1 else 0)
        line = None
        if R.rank == 0:
            line = {
                "metric": "sequences/sec end-to-end count->twist->distance, k=%d, %dk x %dbp" % (k, reads // 1000, L),
                "value": n_total * args.steps / res["elapsed"],
                "unit": "sequences/sec", "n_gpus": R.world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": res["elapsed"] / args.steps * 1e3, "higher_is_better": True, "scaling": scaling,
                "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                "value_is": "device-resident: reads already in HBM when the timed region starts, results left in HBM "
                            "(see pcie_inclusive and file_to_file for what a host caller and a shell user get)",
                "config": dict(common_cfg, workload="%d reads x %d bp%s, k=%d DNA-ds, %d class vectors, euclidean, metric powers(1,1,2)"
                               % (reads, L, " per GPU" if scaling == "weak" else " in total", k, C),
                               reads_per_gpu=n_local,
                               sharding="reads sharded across ranks, twister and classes replicated, no collective"),
                "roofline": R.roofline(n_local, res["ms_fused"]),
                "kernels_ms": {"count_twist": res["ms_fused"], "distance_rowwise(+norms)": res["ms_dist"]},
            }
        if R.world == 1 and not args.no_cpu_baseline:
            cb, parity = cpu_baseline(R, n_local, res["twisted"], res["dmat"], R.classes.cpu().numpy())
            line["cpu_baseline"] = cb
            line["parity_check"] = parity
        if R.world == 1 and not args.no_extras:
            # (the timed legs first, the counter passes after them: with the rocprofv3 --pmc children run BEFORE it, the
            # pipeline leg of the same process came out at 38 M sequences/s instead of 50, four runs out of five --
            # tools/probes/pcie_leg_repeat.py, the leg alone: 50.4-50.5, eight times out of eight)
            line["pcie_inclusive"] = pcie_inclusive(R, n_local)
            try:
                line["pcie_inclusive"]["config3"] = pcie_inclusive_config3(R)
            except Exception as e:  # (a report, never a reason to lose the line)
                line["pcie_inclusive"]["config3"] = {"skipped": "leg failed: %r" % (e,)}
            c4 = R.run_config4(1000000, max(3, min(args.steps, 10)), 2)
            live = None if args.no_children else measure_traffic(R, n_local)
            if live:
                line["roofline"]["traffic"] = live["hbm_bytes_per_launch"]
                line["roofline"]["traffic_source"] = live["source"]
                line["roofline"]["traffic_counters"] = {"FETCH_SIZE_KiB": live["FETCH_SIZE_KiB"], "WRITE_SIZE_KiB": live["WRITE_SIZE_KiB"]}
            line["config4_on_this_gpu"] = {
                "value": 1000000 * max(3, min(args.steps, 10)) / c4["elapsed"], "unit": "sequences/sec",
                "ms_per_step": c4["elapsed"] / max(3, min(args.steps, 10)) * 1e3,
                "kernels_ms": {"count_twist": c4["ms_twist_step"], "distance_rowwise(+norms)": c4["ms_dist"]},
                "all_vs_all": c4["all_vs_all"],
                "note": "BASELINE config 4 (1M x %d bp in total) on one GPU: the N = 1 point of the strong-scaling curve "
                        "`bench.py --gpus N` reports for N > 1" % L}
            line["file_to_file"] = {"value": None, "note": "--no-children"} if args.no_children else file_to_file(R)
            line.update(extra_configs(R))
            # the legs' headline figures as flat scalars as well (a record that keeps only the names of nested objects still has these)
            def _get(d, *path):
                for key in path:
                    d = d.get(key) if isinstance(d, dict) else None
                return d
            flat = {"config2_ms": _get(line, "config2_on_this_gpu", "ms_per_step"), "config2_frac": _get(line, "config2_on_this_gpu", "roofline", "frac"),
                    "config3_unrelated_ms": _get(line, "config3_on_this_gpu", "unrelated_genomes", "ms_per_step"),
                    "config3_unrelated_frac": _get(line, "config3_on_this_gpu", "unrelated_genomes", "roofline", "frac"),
                    "config3_one_organism_ms": _get(line, "config3_on_this_gpu", "one_organism_0.3pct", "ms_per_step"),
                    "config3_one_organism_frac": _get(line, "config3_on_this_gpu", "one_organism_0.3pct", "roofline", "frac"),
                    "config3_d256_ms": _get(line, "config3_on_this_gpu", "one_organism_dims", "256", "ms"),
                    "config3_d256_frac": _get(line, "config3_on_this_gpu", "one_organism_dims", "256", "roofline", "frac"),
                    "config3_d1635_ms": _get(line, "config3_on_this_gpu", "one_organism_dims", "1635", "ms"),
                    "config3_d1635_frac": _get(line, "config3_on_this_gpu", "one_organism_dims", "1635", "roofline", "frac"),
                    "config3_distances_d1635_ms": _get(line, "config3_distances", "distance_rowwise_100k_x_1636", "ms"),
                    "config3_distances_d1635_frac": _get(line, "config3_distances", "distance_rowwise_100k_x_1636", "roofline", "frac"),
                    "config4_n1_ms": _get(line, "config4_on_this_gpu", "ms_per_step"),
                    "config4_all_vs_all_s": _get(line, "config4_on_this_gpu", "all_vs_all", "seconds"),
                    "config5_ms": _get(line, "config5_on_this_gpu", "ms_per_step"), "config5_frac": _get(line, "config5_on_this_gpu", "roofline", "frac")}
            head = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step")
            line = dict([(key, line[key]) for key in head if key in line] + list(flat.items()) + [(key, v) for key, v in line.items() if key not in head])
        R.finish(line)
        return

    # config 4
    res = R.run_config4(reads, args.steps, args.warmup)
    line = None
    if R.rank == 0:
        line = {
            "metric": "sequences/sec end-to-end count->twist->all-gather->distance, k=%d, %dk x %dbp in total" % (k, reads // 1000, L),
            "value": reads * args.steps / res["elapsed"],
            "unit": "sequences/sec", "n_gpus": R.world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": res["elapsed"] / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "value_is": "device-resident; the all-gather of the twisted vectors is inside the timed region",
            "config": dict(common_cfg, workload="BASELINE config 4: %d reads x %d bp in total, k=%d DNA-ds, sharded over %d GPU(s); "
                           "twist in %d chunks with the all-gather of each chunk on its own stream; distances of the "
                           "rank's rows to %d class vectors" % (reads, L, k, R.world, res["chunks"], C),
                           reads_per_gpu=res["n_local"],
                           sharding="reads in contiguous shards; twister, classes and metric replicated; ONE exchange: "
                                    "all-gather of twisted vectors (RCCL over xGMI)",
                           n1_reference="the 1-GPU line's config4_on_this_gpu.value is the same job on one GPU"),
            "roofline": R.roofline(res["chunk_rows"], res["ms_twist_chunk"],
                                   "one launch per chunk of %d reads, overlapped with the exchange of the previous chunk" % res["chunk_rows"]),
            "kernels_ms": {"count_twist_per_step": res["ms_twist_step"], "distance_rowwise(+norms)": res["ms_dist"],
                           "allgather_on_its_stream_per_step": res.get("ms_allgather_step_on_its_stream")},
            "n_ranks_seen": res.get("n_ranks_seen"),
            "ms_compute_per_rank": res.get("ms_compute_per_rank"),
            "exposed_comm_ms_per_rank": res.get("ms_exposed_comm_per_rank"),
            "startup_s_per_rank": res.get("startup_s_per_rank"),
            "collective": res.get("allgather"),
            "gather_checksum_ok": res.get("gather_checksum_ok"),
            "all_vs_all": res["all_vs_all"],
        }
    R.finish(line)


if __name__ == "__main__":
    main()
