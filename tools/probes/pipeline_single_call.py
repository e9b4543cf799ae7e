#!/usr/bin/env python3
"""one submit + collect of 100,000 reads at a time (a synchronous caller), both outputs / distances only: ms per call, best of 9"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import kpop_amd  # noqa: E402


def main():
    kpop_amd.init(0)
    n, L, d, C, k = 100000, 150, 64, 65, 12
    tw = kpop_amd.Twister.synth(0x5EED, k, d)
    rng = np.random.RandomState(1)
    bases = kpop_amd.host_empty(n * L, np.uint8)
    bases[:] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.randint(0, 4, size=n * L)]
    offs = kpop_amd.host_empty(n + 1, np.uint64)
    offs[:] = np.arange(n + 1, dtype=np.uint64) * L
    classes = rng.randn(C, d)
    w = np.exp2(-np.arange(d) / 8.0)
    metric = kpop_amd.metric_compute(w / w.sum())
    for outputs, label in ((3, "twisted rows + distances"), (2, "distances only")):
        pl = kpop_amd.Pipeline(tw, classes, metric, outputs=outputs)
        o = pl.alloc_outputs(n)
        pl.run(bases, offs, o)
        pl.run(bases, offs, o)
        ts = []
        for _ in range(9):
            t0 = time.perf_counter()
            pl.collect(pl.submit(bases, offs, o))
            ts.append(time.perf_counter() - t0)
        print("%s: %.3f ms a call (%d chunks)" % (label, min(ts) * 1e3, pl.stats()["chunks"]), flush=True)
        pl.close()


if __name__ == "__main__":
    main()
