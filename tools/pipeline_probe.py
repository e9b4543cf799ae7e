#!/usr/bin/env python3
"""Sweep of the streaming pipeline (kpop_pipeline_*) at the headline shape: batches in flight x ring depth x chunk size
x outputs -> ms per batch; the numbers behind DESIGN.md's pipeline section.  `--trace` runs ONE configuration a few
times so that `rocprofv3 --kernel-trace --memory-copy-trace` of this script shows the three-way overlap.

    python tools/pipeline_probe.py [--reads 100000] [--trace]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import kpop_amd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=100000)
    ap.add_argument("-k", type=int, default=12)
    ap.add_argument("--dims", type=int, default=64)
    ap.add_argument("--classes", type=int, default=65)
    ap.add_argument("--trace", action="store_true")
    ap.add_argument("--timeline", action="store_true", help="print the device-side timeline of one batch (kpop_pipeline_timeline)")
    ap.add_argument("--outputs", type=int, default=3)
    ap.add_argument("--depth", type=int, default=0)
    ap.add_argument("--chunk", type=int, default=0)
    ap.add_argument("--inflight", type=int, default=10)
    a = ap.parse_args()
    kpop_amd.init(0)
    n, L, d, C = a.reads, 150, a.dims, a.classes
    tw = kpop_amd.Twister.synth(0x5EED, a.k, d)
    rng = np.random.RandomState(1)
    bases = kpop_amd.host_empty(n * L, np.uint8)
    bases[:] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.randint(0, 4, size=n * L)]
    offs = kpop_amd.host_empty(n + 1, np.uint64)
    offs[:] = np.arange(n + 1, dtype=np.uint64) * L
    classes = rng.randn(C, d)
    w = np.exp2(-np.arange(d) / 8.0)
    metric = kpop_amd.metric_compute(w / w.sum())

    def measure(outputs, depth, chunk, inflight, reps=3):
        pl = kpop_amd.Pipeline(tw, classes, metric, outputs=outputs, chunk_reads=chunk, depth=depth)
        o = pl.alloc_outputs(n)
        pl.run(bases, offs, o)
        pl.run(bases, offs, o)
        best, sub = None, None
        for _ in range(reps):
            t0 = time.perf_counter()
            tickets = [pl.submit(bases, offs, o) for _ in range(inflight)]
            t1 = time.perf_counter()
            pl.collect(tickets[-1])
            dt = (time.perf_counter() - t0) / inflight
            if best is None or dt < best:
                best, sub = dt, (t1 - t0) / inflight
        st = pl.stats()
        pl.close()
        return {"outputs": outputs, "depth": st["depth"], "chunks": st["chunks"], "inflight": inflight, "ms_per_batch": best * 1e3,
                "ms_submit_per_batch": sub * 1e3, "Mseq_s": n / best / 1e6}

    if a.timeline:
        for outputs, label in ((3, "twisted rows + distances"), (2, "distances only")):
            pl = kpop_amd.Pipeline(tw, classes, metric, outputs=outputs, chunk_reads=a.chunk, depth=a.depth, record_timeline=True)
            o = pl.alloc_outputs(n)
            pl.run(bases, offs, o)
            pl.run(bases, offs, o)
            t0 = time.perf_counter()
            pl.collect(pl.submit(bases, offs, o))
            wall = (time.perf_counter() - t0) * 1e3
            tl = pl.timeline()
            print("one batch of %d reads, outputs: %s -- %d chunks, host wall %.3f ms" % (n, label, len(tl), wall))
            print("  chunk   upload [start, end]    kernels [start, end]   download [start, end]   (ms from the first upload)")
            for c, r in enumerate(tl):
                print("  %5d   %8.3f %8.3f      %8.3f %8.3f      %8.3f %8.3f" % ((c,) + tuple(r)))
            span = tl[:, 5].max() - tl[:, 0].min()
            up, kern, down = (tl[:, 1] - tl[:, 0]).sum(), (tl[:, 3] - tl[:, 2]).sum(), (tl[:, 5] - tl[:, 4]).sum()

            def overlap(a0, a1, b0, b1):
                tot = 0.0
                for x0, x1 in zip(a0, a1):
                    for y0, y1 in zip(b0, b1):
                        tot += max(0.0, min(x1, y1) - max(x0, y0))
                return tot
            dk = overlap(tl[:, 4], tl[:, 5], tl[:, 2], tl[:, 3])
            uk = overlap(tl[:, 0], tl[:, 1], tl[:, 2], tl[:, 3])
            print("  span %.3f ms; busy: upload %.3f, kernels %.3f, download %.3f (sum %.3f = %.2f x the span); download beside kernels %.3f ms (%.0f %% of it), "
                  "upload beside kernels %.3f ms (%.0f %%)\n" % (span, up, kern, down, up + kern + down, (up + kern + down) / span, dk, 100 * dk / max(down, 1e-9),
                                                                   uk, 100 * uk / max(up, 1e-9)))
            pl.close()
        return
    if a.trace:
        print(json.dumps(measure(a.outputs, a.depth, a.chunk, a.inflight, reps=2)))
        return
    for outputs in (3, 2, 1):
        for inflight in (1, 4, 10):
            for depth, chunk in ((4, 0), (4, 25600), (3, 25600), (8, 25600), (8, 12800), (2, 51200)):
                print(json.dumps(measure(outputs, depth, chunk, inflight)), flush=True)


if __name__ == "__main__":
    main()
