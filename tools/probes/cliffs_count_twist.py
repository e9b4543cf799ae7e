#!/usr/bin/env python3
"""Where kpop_dev_count_twist changes kernels: read length (one wavefront per read up to 512 windows, the streaming kernel
above) and k (rank-select index up to k = 16, bisection of the sorted hashes above).  M windows/s of 100,000 synthetic reads
(20,000 above 1 kb), D = 64, twister = a random 4M-row subset of the k-mers when 4^k / 2 rows do not fit."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import torch
    import kpop_amd
    from kpop_amd import api
    kpop_amd.init(0)
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream()
    d = 64

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        ms = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            fn()
            e1.record(st)
            torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1))
        return float(np.median(ms))

    tws = {}

    def run(k, L, n):
        if k not in tws:
            tws.clear()
            tws[k] = kpop_amd.Twister.synth(0x5EED, k, d)
        tw = tws[k]
        bases = torch.empty(n * L, dtype=torch.uint8, device=dev)
        offs = torch.empty(n + 1, dtype=torch.int64, device=dev)
        api.dev_synth_reads(0x4B506F70, n, L, bases.data_ptr(), offs.data_ptr(), stream=st.cuda_stream)
        out = torch.zeros(n, d, dtype=torch.float64, device=dev)
        t = timed(lambda: api.dev_count_twist(tw, bases.data_ptr(), offs.data_ptr(), n, n * L, L, out.data_ptr(), stream=st.cuda_stream))
        w = n * (L - k + 1)
        print("k = %2d  %6d reads x %5d bp  %8.3f ms  %8.1f M windows/s  (%s rows in the twister)" % (k, n, L, t, w / t / 1e3, tw.info()["n_cols"]), flush=True)

    for L in (150, 300, 500, 523, 524, 600, 1000, 3000):
        run(12, L, 100000 if L <= 1000 else 20000)
    for k in [int(x) for x in os.environ.get("KS", "10,13,14").split(",")]:
        run(k, 150, 100000)


if __name__ == "__main__":
    main()
