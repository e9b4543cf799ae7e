#!/usr/bin/env python3
"""kpop_dev_distance_summary of 256 query rows against 1,000,000 twisted vectors (64 dimensions, keep_at_most 2) under
kpop_tune("summary2", 1 | 3 | 2): one pass over distance rows (default), two passes, no distance rows; ms per call (HIP events,
median of 5), results compared."""
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
    r1, r2, d = int(os.environ.get("AB_REFS", "1000000")), int(os.environ.get("AB_QUERIES", "256")), 64
    keep = int(os.environ.get("AB_KEEP", "2"))
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    m1 = torch.randn(r1, d, dtype=torch.float64, device=dev, generator=g)
    m2 = torch.randn(r2, d, dtype=torch.float64, device=dev, generator=g)
    metric = torch.rand(d, dtype=torch.float64, device=dev, generator=g) + 0.1
    work = torch.empty(api.dev_distance_workspace_bytes(r1, r2, d), dtype=torch.uint8, device=dev)
    cap = 8
    outs = {}
    for mode in [int(x) for x in os.environ.get("AB_MODES", "1,3,2").split(",")]:
        api.tune("summary2", mode)
        stats = torch.zeros(r2, 4, dtype=torch.float64, device=dev)
        nn = torch.zeros(r2, dtype=torch.int32, device=dev)
        idx = torch.zeros(r2, cap, dtype=torch.int32, device=dev)
        dd = torch.zeros(r2, cap, dtype=torch.float64, device=dev)
        z = torch.zeros(r2, cap, dtype=torch.float64, device=dev)

        def run():
            api.dev_distance_summary(m1.data_ptr(), r1, m2.data_ptr(), r2, d, metric.data_ptr(), work.data_ptr(), stats.data_ptr(), nn.data_ptr(),
                                     idx.data_ptr(), dd.data_ptr(), z.data_ptr(), keep_at_most=keep, max_neighbours=cap, stream=st.cuda_stream)

        run()
        torch.cuda.synchronize()
        ms = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            run()
            e1.record(st)
            torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1))
        outs[mode] = (stats.cpu().numpy(), nn.cpu().numpy(), idx.cpu().numpy(), dd.cpu().numpy())
        print("summary2 = %d: %8.3f ms a call (%d x %d x %d, keep_at_most %d; the normalisation of both operands included)"
              % (mode, float(np.median(ms)), r2, r1, d, keep), flush=True)
    api.tune("summary2", 1)
    if 1 in outs and len(outs) > 1:
      for other in [m for m in outs if m != 1]:
        a, b = outs[other], outs[1]
        print("summary2 = %d against 1:" % other, end=" ")
        print("median / MAD identical: %s; neighbours identical: %s; mean / sd max rel diff %.2e"
              % (np.array_equal(a[0][:, 2:], b[0][:, 2:]), np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3]),
                 float(np.max(np.abs(a[0][:, :2] - b[0][:, :2]) / np.abs(b[0][:, :2])))))


if __name__ == "__main__":
    main()
