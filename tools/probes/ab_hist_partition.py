#!/usr/bin/env python3
"""The merged (-l) histogram on inputs that do not repeat: kpop_tune("histlds", AB_MODE) -- 0 direct global atomics, 1 the default
choice, 3 always partition-then-count -- on 100k x 150 bp random reads, 5,000 unrelated 30 kb genomes and 5,000 mutants of one
genome, k = AB_K (12).  Host wall time of kpop_count_reads(per_read = 0), best of 3; kernel times: run under
rocprofv3 --kernel-trace --stats (tools/probes/r04_hist.sh does, once per mode)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import kpop_amd
    from kpop_amd import api
    from oracle import oracle as O  # the synthetic-read generator only
    from tools.cli_kernels_workload import mutants
    kpop_amd.init(0)
    mode, k = int(os.environ.get("AB_MODE", "1")), int(os.environ.get("AB_K", "12"))
    n = int(os.environ.get("AB_GENOMES", "5000"))
    api.tune("histlds", mode)
    work = [("100k x 150 bp reads", O.synth_reads(0x4B506F70, 100000, 150)),
            ("%d unrelated 30 kb genomes" % n, O.synth_reads(0xC1A55, n, 30000)),
            ("%d mutants of one genome" % n, mutants(n))]
    ref = {}
    for label, (bases, offs) in work:
        best, out = 1e9, None
        for _ in range(3):
            t0 = time.perf_counter()
            out = kpop_amd.count_reads(bases, offs, k, per_read=False, capacity=min(len(bases) + 1, (4 ** k + 2 ** k) // 2 + 1))
            best = min(best, time.perf_counter() - t0)
        chk = (int(out[0].sum() % (1 << 61)), int(out[1].astype(np.int64).sum()), len(out[0]))
        print("histlds %d  %-28s k=%d  %9d distinct, %11d windows counted, host wall %8.2f ms  checksum %s" % (mode, label, k, chk[2], chk[1], best * 1e3, chk[0]), flush=True)


if __name__ == "__main__":
    main()
