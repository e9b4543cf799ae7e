#!/usr/bin/env python3
"""Genome-shaped workload (BASELINE config 3 scaled to 5,000 x 30 kb, k = 12, D = 64) for a `rocprofv3 --pmc` pass:
the streaming fused kernel count_twist_stream_kernel, three launches (development aid)."""
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
    n, L, k, d = 5000, 30000, 12, 64
    tw = kpop_amd.Twister.synth(0x7457, k, d)
    bases = torch.empty(n * L, dtype=torch.uint8, device=dev)
    offs = torch.empty(n + 1, dtype=torch.int64, device=dev)
    api.dev_synth_reads(0xC0FFEE, n, L, bases.data_ptr(), offs.data_ptr(), stream=sp)
    out = torch.zeros(n, d, dtype=torch.float64, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    api.dev_count_twist(tw, bases.data_ptr(), offs.data_ptr(), n, n * L, L, out.data_ptr(), stream=sp)
    e0.record()
    for _ in range(3):
        api.dev_count_twist(tw, bases.data_ptr(), offs.data_ptr(), n, n * L, L, out.data_ptr(), stream=sp)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    w = L - k + 1
    alg = n * (L + w * d * 8 + d * 8)
    print("genomes %d x %d: %.3f ms per launch, algorithmic %.2f GB -> %.0f GB/s" % (n, L, ms, alg / 1e9, alg / ms / 1e6))


if __name__ == "__main__":
    main()
