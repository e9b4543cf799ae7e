#!/usr/bin/env python3
"""Workload run under `rocprofv3 --pmc ...` (one counter set per pass, kernel-trace only).

Launches, on the default stream:
  1. calibration: kpop_dev_distance_rowwise(normalize=1) on a 1 GiB operand -- its
     normalise_rows_kernel streams exactly r2*D*8 bytes in and out with 8-byte-per-lane coalesced
     accesses (the access width of the twist gather), far beyond the 256 MiB Infinity Cache, so
     FETCH_SIZE / WRITE_SIZE can be calibrated on a known byte count (MI355X_MICROARCH.md, HBM section);
  2. the bench workload's fused count->twist kernel, `--launches` times.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=100000)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("-k", type=int, default=12)
    ap.add_argument("--dims", type=int, default=64)
    ap.add_argument("--launches", type=int, default=5)
    ap.add_argument("--calib-rows", type=int, default=2 * 1024 * 1024)
    ap.add_argument("--mutants", type=float, default=0.0, help="the sequences are copies of ONE synthetic genome with point substitutions at this rate (assemblies of one organism)")
    a = ap.parse_args()
    import torch

    import kpop_amd
    from kpop_amd import api
    kpop_amd.init(0)
    dev = torch.device("cuda", 0)
    sp = torch.cuda.current_stream().cuda_stream
    # 1. calibration
    r2, d = a.calib_rows, 64
    m1 = torch.ones(1, d, dtype=torch.float64, device=dev)
    m2 = torch.rand(r2, d, dtype=torch.float64, device=dev)
    metric = torch.full((d,), 1.0 / d, dtype=torch.float64, device=dev)
    work = torch.empty(api.dev_distance_workspace_bytes(1, r2, d), dtype=torch.uint8, device=dev)
    out = torch.empty(r2, 1, dtype=torch.float64, device=dev)
    api.dev_distance_rowwise(m1.data_ptr(), 1, m2.data_ptr(), r2, d, metric.data_ptr(), work.data_ptr(),
                             out.data_ptr(), stream=sp)
    torch.cuda.synchronize()
    # 2. the fused kernel
    tw = kpop_amd.Twister.synth(0x5EED, a.k, a.dims)
    n, L = a.reads, a.read_len
    if a.mutants > 0.0:  # as bench.py's _mutants_on_device
        ref = torch.empty(L, dtype=torch.uint8, device=dev)
        ro = torch.empty(2, dtype=torch.int64, device=dev)
        api.dev_synth_reads(0x0123, 1, L, ref.data_ptr(), ro.data_ptr(), stream=sp)
        torch.cuda.synchronize()
        bases = ref.repeat(n)
        acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
        g = torch.Generator(device=dev)
        g.manual_seed(0x0123)
        step = 1 << 27
        for lo in range(0, n * L, step):
            hi = min(n * L, lo + step)
            hit = torch.rand(hi - lo, device=dev, generator=g) < a.mutants
            sub = acgt[torch.randint(0, 4, (hi - lo,), device=dev, generator=g)]
            bases[lo:hi] = torch.where(hit, sub, bases[lo:hi])
        offs = torch.arange(n + 1, dtype=torch.int64, device=dev) * L
    else:
        bases = torch.empty(n * L, dtype=torch.uint8, device=dev)
        offs = torch.empty(n + 1, dtype=torch.int64, device=dev)
        api.dev_synth_reads(0x4B506F70 if L <= 1000 else 0xC1A55, n, L, bases.data_ptr(), offs.data_ptr(), stream=sp)
    tout = torch.zeros(n, a.dims, dtype=torch.float64, device=dev)
    for _ in range(a.launches):
        api.dev_count_twist(tw, bases.data_ptr(), offs.data_ptr(), n, n * L, L, tout.data_ptr(), stream=sp)
    torch.cuda.synchronize()
    print("calibration bytes per direction: %d; fused launches: %d" % (r2 * d * 8, a.launches))


if __name__ == "__main__":
    main()
