#!/usr/bin/env python3
"""Distance stage of the headline bench (65 class vectors x 100k twisted rows x 64 dims), a few launches, for a
`rocprofv3 --pmc ... --kernel-trace` pass (development aid)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch

    import kpop_amd
    from kpop_amd import api
    kpop_amd.init(0)
    dev = torch.device("cuda", 0)
    sp = torch.cuda.current_stream().cuda_stream
    r1, r2, d = 65, 100000, 64
    m1 = torch.randn(r1, d, dtype=torch.float64, device=dev)
    m2 = torch.randn(r2, d, dtype=torch.float64, device=dev)
    metric = torch.full((d,), 1.0 / d, dtype=torch.float64, device=dev)
    work = torch.empty(api.dev_distance_workspace_bytes(r1, r2, d), dtype=torch.uint8, device=dev)
    out = torch.empty(r2, r1, dtype=torch.float64, device=dev)
    for _ in range(5):
        api.dev_distance_rowwise(m1.data_ptr(), r1, m2.data_ptr(), r2, d, metric.data_ptr(), work.data_ptr(), out.data_ptr(), stream=sp)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
