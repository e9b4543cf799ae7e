#!/usr/bin/env python3
"""HIP-event timing of the k-mer database kernels (csrc/counter.hip) on device-resident storage, with the
algorithmic bandwidth of each (4 B per count read, 8 B per f64 written) against the 8 TB/s HBM peak."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.time_stage import timeit  # noqa: E402


def main():
    import ctypes as C

    import torch

    import kpop_amd
    from kpop_amd import _lib
    kpop_amd.init(0)
    L = _lib.load()
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream()
    sp = stream.cuda_stream
    for n_rows, n_cols in ((8_390_656, 64), (8_390_656, 500), (524_800, 1636)):
        ld = L.kpop_dev_counter_ld(n_rows)
        storage = torch.randint(0, 40, (n_cols, ld), dtype=torch.int32, device=dev)
        ws = torch.empty(L.kpop_dev_counter_workspace_bytes(n_cols, n_rows), dtype=torch.uint8, device=dev)
        cs = torch.empty(n_cols, 4, dtype=torch.float64, device=dev)
        rs = torch.empty(n_rows, 4, dtype=torch.float64, device=dev)
        out = torch.empty(n_rows, dtype=torch.int32, device=dev)
        nrm = torch.empty(1, dtype=torch.float64, device=dev)
        sel = torch.arange(n_cols, dtype=torch.int32, device=dev)
        gb = n_rows * n_cols * 4 / 1e9

        def chk(rc):
            assert rc == 0, L.kpop_last_error()
        t = timeit(torch, stream, lambda: chk(L.kpop_dev_counter_stats(storage.data_ptr(), ld, n_cols, n_rows, 1.0, 1.0, ws.data_ptr(),
                                                                       cs.data_ptr(), None, sp)), reps=5)
        print("%9d k-mers x %4d spectra: col stats      %8.3f ms  %6.0f GB/s (%.2f of 8 TB/s)" % (n_rows, n_cols, t, gb / t * 1e3, gb / t / 8))
        t = timeit(torch, stream, lambda: chk(L.kpop_dev_counter_stats(storage.data_ptr(), ld, n_cols, n_rows, 1.0, 1.0, ws.data_ptr(),
                                                                       None, rs.data_ptr(), sp)), reps=5)
        g2 = gb + n_rows * 32 / 1e9
        print("%9d k-mers x %4d spectra: row stats      %8.3f ms  %6.0f GB/s (%.2f)" % (n_rows, n_cols, t, g2 / t * 1e3, g2 / t / 8))
        norm = cs[:, 2].contiguous()
        mx = float(norm.max().item())
        for crit, name in ((0, "combine mean  "), (1, "combine median")):
            t = timeit(torch, stream, lambda: chk(L.kpop_dev_counter_combine(storage.data_ptr(), ld, n_rows, sel.data_ptr(), norm.data_ptr(),
                                                                             n_cols, n_cols, mx, crit, ws.data_ptr(), out.data_ptr(),
                                                                             nrm.data_ptr(), sp)), reps=3)
            print("%9d k-mers x %4d spectra: %s %8.3f ms  %6.0f GB/s (%.2f)" % (n_rows, n_cols, name, t, gb / t * 1e3, gb / t / 8))
        if n_rows * n_cols * 8 < 40e9:
            tab = torch.empty(n_rows * n_cols, dtype=torch.float64, device=dev)
            for km, name in ((0, "transform [c][r]"), (1, "transform [r][c]")):
                t = timeit(torch, stream, lambda: chk(L.kpop_dev_counter_transform(storage.data_ptr(), ld, n_cols, n_rows, 1, 1.0, 1.0,
                                                                                   cs.data_ptr(), km, tab.data_ptr(), sp)), reps=3)
                g3 = gb * 3
                print("%9d k-mers x %4d spectra: %s %6.3f ms  %6.0f GB/s (%.2f)" % (n_rows, n_cols, name, t, g3 / t * 1e3, g3 / t / 8))
            del tab
        del storage, rs
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
