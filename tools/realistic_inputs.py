#!/usr/bin/env python3
"""Count->twist on inputs with the structure real data has, against the uniformly random worst case bench.py times.

  reads    100k x 150 bp sampled (either strand, 0.5 % substitutions) from 65 class genomes of 30 kb  -- what a
           classifier sees: every read's k-mers are among the ~1.9 M k-mers of the classes;
  genomes  N copies of tests/golden/wuhan.fasta (the reference's own test genome, 29,903 bp) with 0.1 % substitutions
           -- BASELINE config 3, "SARS-CoV-2-scale assemblies": ~30 k distinct 12-mers in all.

For each: the kernel time (HIP events, median of 9), the algorithmic bytes of SURVEY.md 8d per second, for the
row-load policies kpop_tune("nt", 0|1|2) and -- genomes -- the segment sizes kpop_tune("seg", ...).  Outputs must be
bit-identical across policies.  Also the table-size sweep behind the automatic policy (nt = 2).
"""
import argparse
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SYNTH = os.path.join(ROOT, "kpop_amd", "bin", "kpop_synth")


def read_fasta(text):
    seqs, cur = [], []
    for line in text.split("\n"):
        if line.startswith(">"):
            if cur:
                seqs.append("".join(cur))
            cur = []
        elif line:
            cur.append(line)
    if cur:
        seqs.append("".join(cur))
    return seqs


def to_arrays(seqs):
    bases = np.frombuffer("".join(seqs).encode(), dtype=np.uint8).copy()
    offs = np.zeros(len(seqs) + 1, dtype=np.int64)
    offs[1:] = np.cumsum([len(s) for s in seqs])
    return bases, offs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genomes", type=int, default=5000)
    ap.add_argument("--reads", type=int, default=100000)
    ap.add_argument("--dims", type=int, default=64)
    args = ap.parse_args()
    import torch

    import kpop_amd
    from kpop_amd import api
    kpop_amd.init(0)
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream()
    k, d = 12, args.dims

    def upload(bases, offs):
        return torch.from_numpy(bases).to(dev), torch.from_numpy(offs).to(dev)

    def timed(tw, b, o, n, max_len, reps=9):
        out = torch.zeros(n, tw.info()["n_dims"], dtype=torch.float64, device=dev)
        ms = []
        for i in range(reps + 2):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            api.dev_count_twist(tw, b.data_ptr(), o.data_ptr(), n, b.numel(), max_len, out.data_ptr(), stream=st.cuda_stream)
            e1.record(st)
            torch.cuda.synchronize()
            if i >= 2:
                ms.append(e0.elapsed_time(e1))
        return float(np.median(ms)), out.cpu().numpy()

    def alg_bytes(offs, dd):
        lens = np.diff(offs)
        w = np.maximum(lens - k + 1, 0)
        return float(lens.sum() + (w.sum() + len(lens)) * dd * 8)

    def run(label, tw, bases, offs, settings):
        n = len(offs) - 1
        b, o = upload(bases, offs)
        max_len = int(np.diff(offs).max())
        ab = alg_bytes(offs, tw.info()["n_dims"])
        ref = None
        for name, knobs in settings:
            for key, val in knobs.items():
                api.tune(key, val)
            ms, out = timed(tw, b, o, n, max_len)
            same = "" if ref is None else ("  bit-identical" if np.array_equal(out, ref) else "  max rel diff %.1e" % (np.max(np.abs(out - ref)) / np.max(np.abs(ref))))
            if ref is None:
                ref = out
            print("  %-58s %-24s %8.3f ms  %7.0f GB/s algorithmic (%.2f of 8 TB/s)%s" % (label, name, ms, ab / ms / 1e6, ab / ms / 1e6 / 8000, same), flush=True)
        api.tune("nt", 2)
        api.tune("seg", 0)

    NT = [("nt=1 (non-temporal)", {"nt": 1}), ("nt=0 (plain)", {"nt": 0}), ("nt=2 (automatic)", {"nt": 2})]
    full = kpop_amd.Twister.synth(0x5EED, k, d)
    print("== reads, k=12, D=%d, full synthetic twister (8,390,656 rows, %.1f GB)" % (d, full.info()["device_bytes"] / 1e9))
    from oracle import oracle as O  # noqa: E402  (tooling: the same generator bench.py uses)
    bases, offs = O.synth_reads(0x4B506F70, args.reads, 150)
    run("uniformly random reads (bench.py's worst case)", full, bases, offs.astype(np.int64), NT)
    classes = subprocess.run([SYNTH, "genomes", "--n", "65", "--len", "30000", "--seed", "12648430"], stdout=subprocess.PIPE, check=True).stdout
    open("/tmp/kpop_classes.fa", "wb").write(classes)
    reads = subprocess.run([SYNTH, "reads", "--from", "/tmp/kpop_classes.fa", "--n", str(args.reads), "--len", "150", "--mutate", "0.005"],
                           stdout=subprocess.PIPE, check=True).stdout.decode()
    rb, ro = to_arrays(read_fasta(reads))
    run("reads sampled from 65 x 30 kb genomes", full, rb, ro, NT)
    # a trained twister only holds the k-mers of its classes
    cb, co = to_arrays(read_fasta(classes.decode()))
    h, c, o = kpop_amd.count_reads(cb, co.astype(np.uint64), k, per_read=False)
    rng = np.random.RandomState(1)
    T = rng.uniform(-1, 1, size=(d, len(h)))
    trained = kpop_amd.Twister.load(T, h, k)
    del T
    print("== reads, trained-like twister: the %d k-mers of the class genomes (%.2f GB of rows)" % (len(h), trained.info()["device_bytes"] / 1e9))
    run("reads sampled from the class genomes", trained, rb, ro, NT)
    trained.free()

    print("== genomes (streaming kernel), k=12, D=%d, full synthetic twister" % d)
    SEG = [("nt=1 seg=16384 (round 1)", {"nt": 1, "seg": 16384}), ("nt=1 seg=auto", {"nt": 1, "seg": 0}), ("nt=0 seg=16384", {"nt": 0, "seg": 16384}),
           ("nt=0 seg=auto", {"nt": 0, "seg": 0}), ("nt=0 seg=1024", {"nt": 0, "seg": 1024}), ("nt=2 seg=auto (defaults)", {"nt": 2, "seg": 0})]
    gb, go = O.synth_reads(0xC1A55, args.genomes, 30000)
    run("%d unrelated random genomes of 30 kb" % args.genomes, full, gb, go.astype(np.int64), SEG)
    mut = subprocess.run([SYNTH, "mutants", "--from", os.path.join(ROOT, "tests", "golden", "wuhan.fasta"), "--n", str(args.genomes), "--mutate", "0.001"],
                         stdout=subprocess.PIPE, check=True).stdout.decode()
    mb, mo = to_arrays(read_fasta(mut))
    run("%d copies of wuhan.fasta, 0.1 %% substitutions" % args.genomes, full, mb, mo, SEG)
    wh, wc, wo = kpop_amd.count_reads(mb[:int(mo[1])], mo[:2].astype(np.uint64), k, per_read=False)
    print("   (one such genome holds %d distinct 12-mers: %.1f MB of rows)" % (len(wh), len(wh) * 512 / 1e6))
    full.free()

    print("== table-size sweep: uniformly random reads, all canonical k-mers, D=%d" % d)
    for kk in (8, 9, 10, 11, 12):
        k = kk
        tw = kpop_amd.Twister.synth(0x5EED, kk, d)
        run("k=%d  rows %.0f MB" % (kk, tw.info()["n_cols"] * 512 / 1e6), tw, bases, offs.astype(np.int64), NT[:2])
        tw.free()


if __name__ == "__main__":
    main()
