#!/usr/bin/env python3
"""The kernels the three drop-in CLIs reach, one section at a time, for rocprofv3 (kernel trace and PMC passes).

    python3 tools/cli_kernels_workload.py --only <section> [--algo-json out.json]

Each section runs ONE kernel configuration a few times on device-resident data (host-API sections: the transfers are
outside the kernels and do not show in a kernel trace) and records the ALGORITHMIC bytes (SURVEY.md 8d's per-unit figures
times the units of one launch) of the kernels it exists for, keyed by a substring of the kernel name; tools/
cli_kernels_report.py joins that with the profiler's per-kernel averages.

  count_L      KPopCount -L, reads          count_wave_kernel                 100k x 150 bp, k = 12
  twist_reads  KPopTwistDB -k, read spectra twist_csr_kernel<double>          100k spectra of <= 139 lines, D = 64
  twist_genomes KPopTwistDB -k, genomes     twist_csr_kernel<double>          2,000 spectra of ~29.7k lines (wuhan mutants)
  summary_65   KPopTwistDB -s vs classes    distance_summary_batch_kernel         r1 = 65, r2 = 100k, D = 64
  summary_1M   relatedness engine           summary1_pass_kernel (+ sample, finish, rowwise)  r1 = 1M, r2 = 256, keep 300
  merged_hist  KPopCount -l                 read_hist / window_hist + compaction   100k reads; 5,000 x 30 kb genomes
  merged_hist_mutants  KPopCount -l, one organism   window_hist_combine_kernel (LDS (hash, count) tables)   5,000 wuhan mutants, k = 12
  merged_hist_k7       KPopCount -l, small k        window_hist_lds_kernel (private LDS tables)             5,000 wuhan mutants, k = 7
  merged_sort  KPopCount -l, k > 13 or hist off   window_keys + radix passes    the same inputs, kpop_tune("hist", 0)
  genomes_L    KPopCount -L, genomes        window_keys + radix passes        2,000 x 30 kb genomes, per-sequence spectra
  merged_hist_genomes  KPopCount -l, unrelated genomes  hist_partition_kernel (+ sizes, bucket count)   5,000 x 30 kb random genomes
  fused_genomes KPopTwistDB on a reads stream of assemblies  count_twist_tile_kernel (consensus on the matrix cores)   5,000 wuhan mutants (0.1 %)
  fused_genomes_1pct   the same at 1 % divergence
  fused_genomes_stream the same batch with kpop_tune("dense", 0)   count_twist_stream_kernel (rows from L2: latency-bound)
  fused_genomes_unrelated   5,000 x 30 kb random genomes, default dispatch   count_twist_stream_kernel (the HBM gather)
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SECTIONS = ["count_L", "twist_reads", "twist_genomes", "summary_65", "summary_1M", "merged_hist", "merged_hist_genomes", "merged_hist_mutants", "merged_hist_k7",
            "merged_sort", "genomes_L", "fused_genomes", "fused_genomes_1pct", "fused_genomes_stream", "fused_genomes_unrelated"]


def mutants(n, rate=0.001, seed=5):
    """n copies of tests/golden/wuhan.fasta with point substitutions at `rate`, made in this process: the profiled
    program must not start children (the profiler's preload has initialised the GPU before main() runs)."""
    from tools.realistic_inputs import read_fasta
    ref = np.frombuffer(read_fasta(open(os.path.join(ROOT, "tests", "golden", "wuhan.fasta")).read())[0].encode(), dtype=np.uint8)
    rng = np.random.RandomState(seed)
    out = np.tile(ref, n)
    hit = np.flatnonzero(rng.rand(out.size) < rate)
    out[hit] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.randint(0, 4, size=len(hit))]
    return out, np.arange(n + 1, dtype=np.uint64) * len(ref)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", required=True, choices=SECTIONS)
    ap.add_argument("--algo-json", default=None)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    import torch

    import kpop_amd
    from kpop_amd import api
    from oracle import oracle as O  # synthetic-input generators only
    kpop_amd.init(0)
    dev = torch.device("cuda", 0)
    sp = torch.cuda.current_stream().cuda_stream
    k, d, n, L = 12, 64, 100000, 150
    algo = {}
    sec = a.only

    def synth_reads_dev(n, L, seed=0x4B506F70):
        b = torch.empty(n * L, dtype=torch.uint8, device=dev)
        o = torch.empty(n + 1, dtype=torch.int64, device=dev)
        api.dev_synth_reads(seed, n, L, b.data_ptr(), o.data_ptr(), stream=sp)
        return b, o

    if sec in ("count_L", "twist_reads"):
        b, o = synth_reads_dev(n, L)
        w = L - k + 1
        scratch = torch.empty(api.dev_count_reads_scratch_bytes(n, L, k), dtype=torch.uint8, device=dev)
        oh = torch.empty(n * w, dtype=torch.int64, device=dev)
        oc = torch.empty(n * w, dtype=torch.int32, device=dev)
        oo = torch.empty(n + 1, dtype=torch.int64, device=dev)
        reps = a.reps if sec == "count_L" else 1
        for _ in range(reps):
            api.dev_count_reads(b.data_ptr(), o.data_ptr(), n, L, k, scratch.data_ptr(), oh.data_ptr(), oc.data_ptr(), oo.data_ptr(), stream=sp)
        torch.cuda.synchronize()
        nnz = int(oo[-1].item())
        algo["count_wave_kernel"] = {"bytes": n * L + nnz * 12 + (n + 1) * 8, "bound": "valu",
                                     "note": "read L B per read, write (hash u64, count u32) per distinct k-mer + offsets; instruction-bound: ~700 vector instructions a read, half of them the 256-slot sort network"}
        if sec == "twist_reads":
            tw = kpop_amd.Twister.synth(0x5EED, k, d)
            val = oc[:nnz].to(torch.float64)
            out = torch.zeros(n, d, dtype=torch.float64, device=dev)
            for _ in range(a.reps):
                api.dev_twist(tw, oh.data_ptr(), val.data_ptr(), oo.data_ptr(), n, w, out.data_ptr(), stream=sp)
            torch.cuda.synchronize()
            algo = {"twist_csr_kernel": {"bytes": nnz * 16 + nnz * d * 8 + n * d * 8, "note": "lines (hash, value) + one twister row per line + the twisted row"}}
    elif sec in ("twist_genomes", "fused_genomes", "fused_genomes_1pct", "fused_genomes_stream", "fused_genomes_unrelated"):
        ng = 2000 if sec == "twist_genomes" else 5000
        if sec == "fused_genomes_unrelated":
            mb, mo = O.synth_reads(0xC1A55, ng, 30000)
            mo = mo.astype(np.int64)
        else:
            mb, mo = mutants(ng, rate=0.01 if sec == "fused_genomes_1pct" else 0.001)
        tw = kpop_amd.Twister.synth(0x5EED, k, d)
        if sec == "twist_genomes":
            h, c, o = kpop_amd.count_reads(mb, mo.astype(np.uint64), k)
            dh, dv, do = torch.from_numpy(h.view(np.int64)).to(dev), torch.from_numpy(c.astype(np.float64)).to(dev), torch.from_numpy(o.view(np.int64)).to(dev)
            out = torch.zeros(ng, d, dtype=torch.float64, device=dev)
            for _ in range(a.reps):
                api.dev_twist(tw, dh.data_ptr(), dv.data_ptr(), do.data_ptr(), ng, int(np.diff(o.astype(np.int64)).max()), out.data_ptr(), stream=sp)
            torch.cuda.synchronize()
            algo["twist_csr_kernel"] = {"bytes": len(h) * 16 + len(h) * d * 8 + ng * d * 8, "bound": "l2",
                                        "note": "lines + one twister row per line + the twisted row; mutants of one genome: the same ~30k rows (15 MB) for every spectrum, served by the L2s and the Infinity Cache -- HBM traffic is a tenth of these bytes"}
        else:
            db, do = torch.from_numpy(np.ascontiguousarray(mb)).to(dev), torch.from_numpy(np.asarray(mo, dtype=np.int64)).to(dev)
            out = torch.zeros(ng, d, dtype=torch.float64, device=dev)
            if sec == "fused_genomes_stream":
                api.tune("dense", 0)
            for _ in range(a.reps):
                api.dev_count_twist(tw, db.data_ptr(), do.data_ptr(), ng, db.numel(), int(np.diff(mo).max()), out.data_ptr(), stream=sp)
            torch.cuda.synchronize()
            lens = np.diff(mo)
            gather = int(lens.sum() + np.maximum(lens - k + 1, 0).sum() * d * 8)
            if sec == "fused_genomes_unrelated":
                algo["count_twist_stream_kernel"] = {"bytes": gather, "bound": "hbm", "note": "bases + one twister row per window (SURVEY 8d): every row a random 512 B row of a 4.3 GB table"}
            elif sec == "fused_genomes_stream":
                algo["count_twist_stream_kernel"] = {"bytes": gather, "bound": "l2", "note": "kpop_tune(\"dense\", 0) on assemblies of one organism: the rows are shared between sequences and come from L2 "
                                                                                        "(PMC traffic ~0.02 x these bytes); ~100 SIMD-cycles a window of hashing, look-up and a dependent row load: latency, not a bandwidth"}
            else:
                # the matrix cores' work, counted by the kernel itself in one more (profiled but separate) launch
                api.debug_counters(16)
                api.tune("dbg", 32 << 24)
                api.dev_count_twist(tw, db.data_ptr(), do.data_ptr(), ng, db.numel(), int(np.diff(mo).max()), out.data_ptr(), stream=sp)
                cnt = api.debug_counters(16)
                api.tune("dbg", 0)
                algo["count_twist_tile_kernel"] = {"flops": 2.0 * 64 * d * cnt[15], "bound": "mfma",
                                                   "note": "2 x 64 sequences x D x the rows of every chunk's consensus set as multiplied (%d chunks, %d rows); the kernel also finds every "
                                                           "window's row, builds the set and gathers the residual rows: its matrix phase is a third of it" % (cnt[14], cnt[15])}
    elif sec in ("summary_65", "summary_1M"):
        r1, r2, keep = (65, 100000, 2) if sec == "summary_65" else (1000000, 256, 300)
        g = torch.Generator(device=dev)
        g.manual_seed(1)
        m1 = torch.randn(r1, d, dtype=torch.float64, device=dev, generator=g)
        m2 = torch.randn(r2, d, dtype=torch.float64, device=dev, generator=g)
        metric = torch.from_numpy(kpop_amd.metric_compute(O.synth_inertia(d))).to(dev)
        work = torch.empty(api.dev_distance_workspace_bytes(r1, r2, d), dtype=torch.uint8, device=dev)
        mx = 8 if keep == 2 else 512
        stats = torch.zeros(r2, 4, dtype=torch.float64, device=dev)
        nn = torch.zeros(r2, dtype=torch.int32, device=dev)
        idx = torch.zeros(r2, mx, dtype=torch.int32, device=dev)
        dd = torch.zeros(r2, mx, dtype=torch.float64, device=dev)
        zz = torch.zeros(r2, mx, dtype=torch.float64, device=dev)
        for _ in range(a.reps if sec == "summary_65" else 2):
            api.dev_distance_summary(m1.data_ptr(), r1, m2.data_ptr(), r2, d, metric.data_ptr(), work.data_ptr(), stats.data_ptr(), nn.data_ptr(),
                                     idx.data_ptr(), dd.data_ptr(), zz.data_ptr(), keep_at_most=keep, max_neighbours=mx, stream=sp)
        torch.cuda.synchronize()
        if sec == "summary_65":
            algo["distance_summary"] = {"bytes": (r1 + r2) * d * 8 + r2 * (32 + 4 + keep * 20), "flops": 4.0 * r1 * r2 * d, "bound": "valu",
                                               "note": "both operands once + one summary row; instruction-bound: 4 unfusable f64 ops per pair and dimension, then a sort of the row's (distance, column) pairs"}
        else:
            algo["summary1_pass_kernel"] = {"bytes": r2 * r1 * 8, "note": "one pass over the r1 distances of each query row is the algorithmic minimum: this kernel is that pass; fused_sample_kernel (a 6 % sample, several selections over it) and fused_finish_kernel (the ~13 % candidates, five passes) add theirs; summary_large_kernel is the fallback, idle here"}
            algo["distance_rowwise_kernel"] = {"bytes": (r1 + r2) * d * 8 + r1 * r2 * 8, "flops": 4.0 * r1 * r2 * d, "note": "chunk rows written for the summary kernel"}
    elif sec in ("merged_hist_mutants", "merged_hist_k7"):
        kk = 12 if sec == "merged_hist_mutants" else 7
        mb, mo = mutants(5000)
        cap = (4 ** kk + 2 ** kk) // 2 + 1
        for _ in range(3):
            kpop_amd.count_reads(mb, mo, kk, per_read=False, capacity=cap)
        lens = np.diff(mo.astype(np.int64))
        wg = int(np.maximum(lens - kk + 1, 0).sum())
        name = "window_hist_combine_kernel" if kk == 12 else "window_hist_lds_kernel"
        algo[name] = {"bytes": int(lens.sum()) + wg * 8, "note": "SURVEY 8d: L B read + one 8-byte atomic read-modify-write per window (5,000 mutants of one 29.9 kb genome, k = %d)" % kk}
    elif sec in ("merged_hist", "merged_hist_genomes", "merged_sort", "genomes_L"):
        if sec == "genomes_L":
            gb, go = O.synth_reads(0xC1A55, 2000, 30000)
            for _ in range(3):
                h, c, o = kpop_amd.count_reads(gb, go, k)
            win = 2000 * (30000 - k + 1)
            algo["window_keys_kernel"] = {"bytes": 2000 * 30000 + win * 8, "note": "bases in, one 8-byte key per window out"}
            algo["radix_scatter_kernel"] = {"bytes": win * 16, "note": "per pass: keys read and written once"}
            algo["radix_count_kernel"] = {"bytes": win * 8, "note": "per pass: keys read once"}
        else:
            api.tune("hist", 0 if sec == "merged_sort" else 1)
            rb, ro = O.synth_reads(0x4B506F70, n, L)
            gb, go = O.synth_reads(0xC1A55, 5000, 30000)
            cap = (4 ** k + 2 ** k) // 2 + 1
            if sec != "merged_hist_genomes":
                for _ in range(3):
                    kpop_amd.count_reads(rb, ro, k, per_read=False, capacity=cap)
            if sec != "merged_hist":
                for _ in range(3):
                    kpop_amd.count_reads(gb, go, k, per_read=False, capacity=cap)
            wr, wg = n * (L - k + 1), 5000 * (30000 - k + 1)
            if sec in ("merged_hist", "merged_hist_genomes"):
                nbases, w = (n * L, wr) if sec == "merged_hist" else (5000 * 30000, wg)
                what = "100k x 150 bp random reads" if sec == "merged_hist" else "5,000 unrelated 30 kb genomes"
                algo["hist_part_sizes_kernel"] = {"bytes": nbases, "bound": "valu", "note": what + ": the bases read, every window hashed, 512 LDS counters a block"}
                algo["hist_partition_kernel"] = {"bytes": nbases + w * 2, "bound": "valu", "note": "the bases read again, one u16 entry written per window (scattered 2-byte stores, a run per bucket and round)"}
                algo["hist_bucket_count_kernel"] = {"bytes": w * 2 + 4 ** k * 4, "bound": "hbm", "note": "the entries read once, the whole table written once (no memset, no global atomic)"}
                algo["scan_apply_kernel"] = {"bytes": 4 ** k * 4, "bound": "hbm", "note": "compaction: the table read once (plus the spectrum written)"}
            else:
                algo["window_keys_kernel"] = {"bytes": "mixed", "note": "two input sizes in one section: see per-dispatch rows"}
    if a.algo_json:
        os.makedirs(os.path.dirname(os.path.abspath(a.algo_json)), exist_ok=True)
        json.dump({"section": sec, "kernels": algo}, open(a.algo_json, "w"), indent=1)
    print("section %s done" % sec)


if __name__ == "__main__":
    main()
