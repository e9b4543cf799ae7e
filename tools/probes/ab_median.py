#!/usr/bin/env python3
"""combine median (HIP events), and the same launch with the selection taken out (kpop_tune dbg 256): what the staging costs."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tools.time_stage import timeit  # noqa: E402


def main():
    import torch

    import kpop_amd
    from kpop_amd import _lib, api
    kpop_amd.init(0)
    L = _lib.load()
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream()
    sp = stream.cuda_stream
    for n_rows, n_cols, sparse in ((8_390_656, 12, 0), (8_390_656, 33, 0), (8_390_656, 64, 0), (8_390_656, 64, 1), (8_390_656, 100, 0), (8_390_656, 500, 0), (8_390_656, 500, 1), (8_390_656, 500, 2), (2_000_000, 1000, 0), (524_800, 1636, 0), (524_800, 1636, 1)):
        ld = L.kpop_dev_counter_ld(n_rows)
        if sparse:  # most k-mers absent from most spectra
            storage = (torch.rand((n_cols, ld), device=dev) < (0.3 if sparse == 1 else 0.01)).to(torch.int32) * torch.randint(1, 40, (n_cols, ld), dtype=torch.int32, device=dev)
        else:
            storage = torch.randint(0, 40, (n_cols, ld), dtype=torch.int32, device=dev)
        ws = torch.empty(L.kpop_dev_counter_workspace_bytes(n_cols, n_rows), dtype=torch.uint8, device=dev)
        out = [torch.empty(n_rows, dtype=torch.int32, device=dev) for _ in range(2)]
        nrm = torch.empty(1, dtype=torch.float64, device=dev)
        sel = torch.arange(n_cols, dtype=torch.int32, device=dev)
        norm = storage[:, :n_rows].sum(dim=1).to(torch.float64).contiguous()
        mx = float(norm.max().item())
        gb = n_rows * n_cols * 4 / 1e9
        res = []
        for which, dbg in ((0, 0), (1, 256)):
            api.tune("dbg", dbg)

            def run():
                assert L.kpop_dev_counter_combine(storage.data_ptr(), ld, n_rows, sel.data_ptr(), norm.data_ptr(), n_cols, n_cols, mx, 1, ws.data_ptr(),
                                                  out[which].data_ptr(), nrm.data_ptr(), sp) == 0, L.kpop_last_error()
            t = timeit(torch, stream, run, reps=3)
            res.append(t)

        api.tune("dbg", 0)

        print("%9d k-mers x %4d spectra%s: median %8.3f ms (%.2f of 8 TB/s)   staging and rescaling alone %8.3f ms" % (
            n_rows, n_cols, (" (70 % zeros)" if sparse == 1 else " (99 % zeros)" if sparse else ""), res[0], gb / res[0] / 8, res[1]), flush=True)
        del storage
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
