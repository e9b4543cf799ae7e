#!/usr/bin/env python3
"""A/B of the fused count->twist kernel's tuning knobs: variants interleaved over several rounds in ONE
process (cdna_hip_programming.md 5.4 rule 24), median and min of HIP-event times per variant."""
import argparse
import itertools
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=100000)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("-k", type=int, default=12)
    ap.add_argument("--dims", type=int, default=64)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    import torch

    import kpop_amd
    from kpop_amd import api
    kpop_amd.init(0)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream()
    sp = stream.cuda_stream
    tw = kpop_amd.Twister.synth(0x5EED, a.k, a.dims)
    n, L = a.reads, a.read_len
    bases = torch.empty(n * L, dtype=torch.uint8, device=dev)
    offs = torch.empty(n + 1, dtype=torch.int64, device=dev)
    api.dev_synth_reads(0x4B506F70, n, L, bases.data_ptr(), offs.data_ptr(), stream=sp)
    out = torch.zeros(n, a.dims, dtype=torch.float64, device=dev)
    variants = [dict(nt=t, unroll=u) for t, u in itertools.product((0, 1), (8, 16))]
    ref = None
    times = {str(v): [] for v in variants}
    for rnd in range(a.rounds):
        for v in variants:
            for key, val in v.items():
                api.tune(key, val)
            api.dev_count_twist(tw, bases.data_ptr(), offs.data_ptr(), n, n * L, L, out.data_ptr(), stream=sp)  # warm
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(a.reps):
                api.dev_count_twist(tw, bases.data_ptr(), offs.data_ptr(), n, n * L, L, out.data_ptr(), stream=sp)
            e1.record(stream)
            torch.cuda.synchronize()
            times[str(v)].append(e0.elapsed_time(e1) / a.reps)
            if rnd == 0:
                got = out.cpu().numpy().copy()
                if ref is None:
                    ref = got
                assert np.array_equal(ref, got), "variant %s changes the result" % v
    windows = L - a.k + 1
    gb = n * (L + windows * a.dims * 8 + a.dims * 8) / 1e9
    print("algorithmic GB per launch: %.3f" % gb)
    for v in variants:
        t = np.array(times[str(v)])
        print("%-40s median %.4f ms  min %.4f ms  -> %.0f GB/s (%.3f of 8 TB/s)" % (v, np.median(t), t.min(), gb / np.median(t) * 1e3,
                                                                                  gb / np.median(t) * 1e3 / 8000))


if __name__ == "__main__":
    main()
