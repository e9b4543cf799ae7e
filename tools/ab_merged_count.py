#!/usr/bin/env python3
"""A/B of the merged (-l) spectrum, bin/KPopCount.ml:60: atomic histogram vs device-wide radix sort.

Wall time of kpop_count_reads(per_read=0) from host buffers (H2D + kernels + D2H of the spectrum), best of 5, for
100k x 150 bp reads and N x 30 kb genomes at k = 12 and 13; outputs must be identical.  Kernel times: run under
`rocprofv3 --kernel-trace --stats -- python3 tools/ab_merged_count.py` (window_hist / read_hist vs window_keys + radix_*)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import kpop_amd
    from kpop_amd import api
    from oracle import oracle as O  # the synthetic-read generator only
    kpop_amd.init(0)
    n_genomes = int(os.environ.get("AB_GENOMES", "50000"))
    work = [("100k x 150 bp reads", O.synth_reads(0x4B506F70, 100000, 150)),
            ("%d x 30 kb genomes" % n_genomes, O.synth_reads(0xC1A55, n_genomes, 30000))]
    for label, (bases, offs) in work:
        windows = int(np.maximum(np.diff(offs.astype(np.int64)) - 11, 0).sum())
        for k in (12, 13):
            res = {}
            for hist in (1, 0):
                api.tune("hist", hist)
                best, out = 1e9, None
                for _ in range(3 if len(bases) > 1e9 else 5):
                    t0 = time.perf_counter()
                    out = kpop_amd.count_reads(bases, offs, k, per_read=False, capacity=min(len(bases) + 1, (4 ** k + 2 ** k) // 2 + 1))
                    best = min(best, time.perf_counter() - t0)
                res[hist] = (best, out)
            same = all(np.array_equal(a, b) for a, b in zip(res[1][1], res[0][1]))
            print("%-22s k=%d  %11d windows -> %8d distinct:  histogram %8.2f ms   sort %8.2f ms   (%.1fx)  identical: %s"
                  % (label, k, windows, len(res[1][1][0]), res[1][0] * 1e3, res[0][0] * 1e3, res[0][0] / res[1][0], same), flush=True)
    api.tune("hist", 1)


if __name__ == "__main__":
    main()
