#!/usr/bin/env python3
"""HIP-event timing of single stages of the path on device-resident data (development aid)."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timeit(torch, stream, fn, reps=20):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(5):
        e0.record(stream)
        for _ in range(reps):
            fn()
        e1.record(stream)
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps)
    return float(np.median(ts))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="65x100000x64,64x100000x64,9x100000x9,256x20000x256,1000x1000x64,4096x4096x64")
    a = ap.parse_args()
    import torch

    import kpop_amd
    from kpop_amd import api
    kpop_amd.init(0)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream()
    sp = stream.cuda_stream
    # -L counting, device resident (count_wave_kernel + scan + compaction)
    from oracle import oracle as O
    for n, L, k in ((100000, 150, 12), (100000, 150, 21), (1000000, 150, 12), (100000, 500, 12)):
        bases = torch.empty(n * L, dtype=torch.uint8, device=dev)
        offs = torch.empty(n + 1, dtype=torch.int64, device=dev)
        api.dev_synth_reads(0x4B506F70, n, L, bases.data_ptr(), offs.data_ptr(), stream=sp)
        w = L - k + 1
        scratch = torch.empty(api.dev_count_reads_scratch_bytes(n, L, k), dtype=torch.uint8, device=dev)
        oh = torch.empty(n * w, dtype=torch.int64, device=dev)
        oc = torch.empty(n * w, dtype=torch.int32, device=dev)
        oo = torch.empty(n + 1, dtype=torch.int64, device=dev)
        t = timeit(torch, stream, lambda: api.dev_count_reads(bases.data_ptr(), offs.data_ptr(), n, L, k, scratch.data_ptr(),
                                                               oh.data_ptr(), oc.data_ptr(), oo.data_ptr(), stream=sp))
        gb = n * (L + w * 8) / 1e9  # read L bytes, write (L-k+1)*8 bytes per read
        if n == 100000 and L == 150:  # parity of the device-resident CSR against the oracle
            hb, ho = O.synth_reads(0x4B506F70, n, L)
            h, c, o = O.count_reads(hb, ho, k)
            tot = int(oo[-1].item())
            assert tot == len(h) and np.array_equal(oh[:tot].cpu().numpy().view(np.uint64), h)
            assert np.array_equal(oc[:tot].cpu().numpy().view(np.uint32), c) and np.array_equal(oo.cpu().numpy().view(np.uint64), o)
        print("count_reads -L n=%d L=%d k=%d: %.4f ms  %.1f M reads/s  algorithmic %.0f GB/s (%.3f of 8 TB/s)"
              % (n, L, k, t, n / t / 1e3, gb / t * 1e3, gb / t * 1e3 / 8000))
    for shp in a.shapes.split(","):
        r1, r2, d = (int(x) for x in shp.split("x"))
        m1 = torch.randn(r1, d, dtype=torch.float64, device=dev)
        m2 = torch.randn(r2, d, dtype=torch.float64, device=dev)
        metric = torch.full((d,), 1.0 / d, dtype=torch.float64, device=dev)
        work = torch.empty(api.dev_distance_workspace_bytes(r1, r2, d), dtype=torch.uint8, device=dev)
        out = torch.empty(r2, r1, dtype=torch.float64, device=dev)
        for norm in (True, False):
            t = timeit(torch, stream, lambda: api.dev_distance_rowwise(m1.data_ptr(), r1, m2.data_ptr(), r2, d, metric.data_ptr(),
                                                                     work.data_ptr(), out.data_ptr(), normalize=norm, stream=sp))
            ops = 4.0 * r1 * r2 * d
            print("distance_rowwise r1=%d r2=%d D=%d normalize=%d: %.4f ms  %.2f Tops/s f64 (non-FMA peak 39.3)" % (r1, r2, d, norm, t, ops / t / 1e9))


if __name__ == "__main__":
    main()
