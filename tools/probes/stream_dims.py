#!/usr/bin/env python3
"""count_twist_stream_kernel over numbers of dimensions: 2,000 mutants of wuhan.fasta and 2,000 unrelated 30 kb genomes,
k = 12, D = 64 / 100 / 128 / 200 / 256 / 300; ms per kpop_dev_count_twist call."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import torch
    import kpop_amd
    from kpop_amd import api
    from oracle import oracle as O
    from tools.cli_kernels_workload import mutants
    kpop_amd.init(0)
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream()
    k, n = 12, 2000
    sets = (("mutants", mutants(n)), ("unrelated", O.synth_reads(0xC1A55, n, 30000)))
    for d in [int(x) for x in os.environ.get("DIMS", "64,100,128,200,256,300").split(",")]:
        tw = kpop_amd.Twister.synth(0x5EED, k, d)
        for label, (b, o) in sets:
            db, do = torch.from_numpy(np.ascontiguousarray(b)).to(dev), torch.from_numpy(o.astype(np.int64)).to(dev)
            L = int(np.diff(o.astype(np.int64)).max())
            out = torch.zeros(n, d, dtype=torch.float64, device=dev)
            f = lambda: api.dev_count_twist(tw, db.data_ptr(), do.data_ptr(), n, db.numel(), L, out.data_ptr(), stream=st.cuda_stream)
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
            print("D = %3d  %-10s %8.3f ms  (checksum %.12e)" % (d, label, float(np.median(ms)), float(out.sum().item())), flush=True)
        del tw


if __name__ == "__main__":
    main()
