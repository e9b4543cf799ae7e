#!/usr/bin/env python3
"""The merged (-l) histogram staged through LDS against round 2's direct global atomics, kpop_tune("histlds", 0|1|2):
5,000 mutants of one 30 kb genome (one organism: BASELINE config 3's kind of batch) and 5,000 unrelated genomes, k = 7
(private LDS tables) and k = 12 (sorted chunks).  Wall time includes the upload of 150 MB of bases; the kernels' own
times and HBM traffic come from running this under rocprofv3 (--kernel-trace --stats / --pmc FETCH_SIZE WRITE_SIZE).
    python tools/probes/ab_hist_lds.py [one|unrelated] [k] [histlds]   (no arguments: everything)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def mutants(n, L, rate, seed):
    rng = np.random.RandomState(seed)
    ref = rng.randint(0, 4, size=L).astype(np.uint8)
    out = np.tile(ref, n)
    hit = np.flatnonzero(rng.rand(n * L) < rate)
    out[hit] = rng.randint(0, 4, size=len(hit))
    return np.frombuffer(b"ACGT", dtype=np.uint8)[out], np.arange(n + 1, dtype=np.uint64) * L


def main():
    import kpop_amd
    from kpop_amd import api
    from oracle import oracle as O
    kpop_amd.init(0)
    n, L = int(os.environ.get("AB_GENOMES", "5000")), 30000
    which = sys.argv[1] if len(sys.argv) > 1 else None
    ks = [int(sys.argv[2])] if len(sys.argv) > 2 else [7, 12]
    modes = [int(sys.argv[3])] if len(sys.argv) > 3 else [0, 1, 2]
    work = []
    if which in (None, "one"):
        work.append(("%d mutants of one 30 kb genome" % n, mutants(n, L, 0.001, 5)))
    if which in (None, "unrelated"):
        work.append(("%d unrelated 30 kb genomes" % n, O.synth_reads(0xC1A55, n, L)))
    for label, (bases, offs) in work:
        for k in ks:
            ref = None
            for mode in modes:
                api.tune("histlds", mode)
                best, out = 1e9, None
                for _ in range(3):
                    t0 = time.perf_counter()
                    out = kpop_amd.count_reads(bases, offs, k, per_read=False, capacity=(4 ** k + 2 ** k) // 2 + 1)
                    best = min(best, time.perf_counter() - t0)
                same = ref is None or all(np.array_equal(a, b) for a, b in zip(ref, out))
                ref = ref or out
                print("%-34s k=%-2d histlds=%d  %9.2f ms wall  %8d distinct  identical: %s" % (label, k, mode, best * 1e3, len(out[0]), same), flush=True)
    api.tune("histlds", 1)


if __name__ == "__main__":
    main()
