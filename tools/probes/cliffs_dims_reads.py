#!/usr/bin/env python3
"""kpop_dev_count_twist on 100,000 x 150 bp reads, k = 12, over numbers of dimensions: ms and the gathered bytes per second
(139 rows of D x 8 bytes per read) -- is the time the bytes', or the passes' (one per 64 dimensions)?"""
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
    k, n, L = 12, 100000, 150
    bases = torch.empty(n * L, dtype=torch.uint8, device=dev)
    offs = torch.empty(n + 1, dtype=torch.int64, device=dev)
    api.dev_synth_reads(0x4B506F70, n, L, bases.data_ptr(), offs.data_ptr(), stream=st.cuda_stream)
    for d in [int(x) for x in os.environ.get("DIMS", "8,16,32,33,48,64,65,96,100,128,129,200,256").split(",")]:
        tw = kpop_amd.Twister.synth(0x5EED, k, d)
        out = torch.zeros(n, d, dtype=torch.float64, device=dev)
        f = lambda: api.dev_count_twist(tw, bases.data_ptr(), offs.data_ptr(), n, n * L, L, out.data_ptr(), stream=st.cuda_stream)
        f()
        torch.cuda.synchronize()
        ms = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            f()
            e1.record(st)
            torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1))
        t = float(np.median(ms))
        print("D = %3d  %7.3f ms  %6.2f TB/s of rows gathered" % (d, t, n * (L - k + 1) * d * 8 / t / 1e9), flush=True)
        del tw, out


if __name__ == "__main__":
    main()
