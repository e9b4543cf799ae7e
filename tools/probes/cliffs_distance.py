#!/usr/bin/env python3
"""kpop_dev_distance_rowwise over first-operand sizes (second operand 100,000 rows, or fewer so that the pairs stay at
<= 4e8), D = 64, normalised: ms and G pair-dimensions/s -- looking for sizes where the tile choice falls off."""
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
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    metric = torch.rand(d, dtype=torch.float64, device=dev, generator=g) + 0.1
    for r1 in (1, 8, 33, 64, 65, 96, 127, 128, 129, 200, 256, 257, 1000, 4096, 4097, 20000, 65535, 65536, 200000):
        r2 = int(min(100000, max(64, 4e8 // r1)))
        m1 = torch.randn(r1, d, dtype=torch.float64, device=dev, generator=g)
        m2 = torch.randn(r2, d, dtype=torch.float64, device=dev, generator=g)
        work = torch.empty(api.dev_distance_workspace_bytes(r1, r2, d), dtype=torch.uint8, device=dev)
        out = torch.empty(r2, r1, dtype=torch.float64, device=dev)
        f = lambda: api.dev_distance_rowwise(m1.data_ptr(), r1, m2.data_ptr(), r2, d, metric.data_ptr(), work.data_ptr(), out.data_ptr(), stream=st.cuda_stream)
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
        print("r1 = %6d  r2 = %6d  %8.3f ms  %7.1f G pair-dims/s  (%.1f Tops/s f64)" % (r1, r2, t, r1 * r2 * d / t / 1e6, 4 * r1 * r2 * d / t / 1e9), flush=True)


if __name__ == "__main__":
    main()
