#!/usr/bin/env python3
"""Does the VALU-bound distance stage of batch i hide under the HBM-bound count->twist of batch i+1 when the two run on
separate HIP streams (double-buffered twisted/distance rows)?  Headline shape; prints ms per step both ways."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch

    import kpop_amd
    from kpop_amd import api
    kpop_amd.init(0)
    dev = torch.device("cuda", 0)
    n, L, k, d, C = 100000, 150, 12, 64, 65
    s0 = torch.cuda.current_stream()
    s1 = torch.cuda.Stream()
    tw = kpop_amd.Twister.synth(0x7457, k, d)
    bases = torch.empty(n * L, dtype=torch.uint8, device=dev)
    offs = torch.empty(n + 1, dtype=torch.int64, device=dev)
    api.dev_synth_reads(0x4B506F70, n, L, bases.data_ptr(), offs.data_ptr(), stream=s0.cuda_stream)
    classes = torch.randn(C, d, dtype=torch.float64, device=dev)
    metric = torch.full((d,), 1.0 / d, dtype=torch.float64, device=dev)
    twisted = [torch.zeros(n, d, dtype=torch.float64, device=dev) for _ in range(2)]
    dmat = [torch.zeros(n, C, dtype=torch.float64, device=dev) for _ in range(2)]
    work = [torch.empty(api.dev_distance_workspace_bytes(C, n, d), dtype=torch.uint8, device=dev) for _ in range(2)]
    torch.cuda.synchronize()

    def sequential(steps):
        for i in range(steps):
            api.dev_count_twist(tw, bases.data_ptr(), offs.data_ptr(), n, n * L, L, twisted[0].data_ptr(), stream=s0.cuda_stream)
            api.dev_distance_rowwise(classes.data_ptr(), C, twisted[0].data_ptr(), n, d, metric.data_ptr(), work[0].data_ptr(),
                                     dmat[0].data_ptr(), stream=s0.cuda_stream)

    def pipelined(steps):
        done_dist = [None, None]
        for i in range(steps):
            b = i & 1
            if done_dist[b] is not None:
                s0.wait_event(done_dist[b])       # twisted[b] is free again
            api.dev_count_twist(tw, bases.data_ptr(), offs.data_ptr(), n, n * L, L, twisted[b].data_ptr(), stream=s0.cuda_stream)
            ev = torch.cuda.Event()
            ev.record(s0)
            s1.wait_event(ev)
            api.dev_distance_rowwise(classes.data_ptr(), C, twisted[b].data_ptr(), n, d, metric.data_ptr(), work[b].data_ptr(),
                                     dmat[b].data_ptr(), stream=s1.cuda_stream)
            done_dist[b] = torch.cuda.Event()
            done_dist[b].record(s1)

    # "ldspad": extra dynamic LDS per block of the fused kernel, i.e. fewer of its blocks per CU (17.6 KB each: 8 fit 160 KB),
    # which leaves LDS and wave slots for the distance kernel's blocks (50 KB each) on the second stream
    for pad in (0, 6144, 10240, 16384):
        api.tune("ldspad", pad)
        for name, fn in (("sequential", sequential), ("two streams", pipelined), ("sequential", sequential), ("two streams", pipelined)):
            fn(5)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn(40)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 40 * 1e3
            print("ldspad %5d  %-12s %.4f ms per step  (%.1f M sequences/s)" % (pad, name, ms, n / ms / 1e3), flush=True)
    api.tune("ldspad", 0)
    a, b = dmat[0].clone(), dmat[1].clone()
    sequential(1)
    torch.cuda.synchronize()
    print("results identical:", bool(torch.equal(a, dmat[0]) and torch.equal(b, dmat[0])))


if __name__ == "__main__":
    main()
