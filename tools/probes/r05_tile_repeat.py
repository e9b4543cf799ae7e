#!/usr/bin/env python3
"""BASELINE config 3's one-organism batch (N x 30 kb at 0.3 %) through the default dispatch REP times: the same bits every time?
Prints the rows that differ between calls (a race shows here long before it shows in a tolerance)."""
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
    if os.environ.get("DBG"):
        api.tune("dbg", int(os.environ["DBG"]) << 24)
    k, d, n, L = int(os.environ.get("K", "12")), int(os.environ.get("D", "64")), int(os.environ.get("N", "50000")), int(os.environ.get("L", "30000"))
    dev = torch.device("cuda", 0)
    sp = torch.cuda.current_stream().cuda_stream
    tw = kpop_amd.Twister.synth(0x7457, k, d)
    ref = torch.empty(L, dtype=torch.uint8, device=dev)
    ro = torch.empty(2, dtype=torch.int64, device=dev)
    api.dev_synth_reads(0xBEEF, 1, L, ref.data_ptr(), ro.data_ptr(), stream=sp)
    torch.cuda.synchronize()
    bases = ref.repeat(n)
    acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    step = 1 << 27
    for lo in range(0, n * L, step):
        hi = min(n * L, lo + step)
        hit = torch.rand(hi - lo, device=dev, generator=g) < float(os.environ.get("RATE", "0.003"))
        sub = acgt[torch.randint(0, 4, (hi - lo,), device=dev, generator=g)]
        bases[lo:hi] = torch.where(hit, sub, bases[lo:hi])
    offs = torch.arange(n + 1, dtype=torch.int64, device=dev) * L
    out = torch.zeros(n, d, dtype=torch.float64, device=dev)
    first = None
    for rep in range(int(os.environ.get("REP", "5"))):
        out.zero_()
        api.dev_count_twist(tw, bases.data_ptr(), offs.data_ptr(), n, n * L, L, out.data_ptr(), stream=sp)
        torch.cuda.synchronize()
        if first is None:
            first = out.clone()
            continue
        bad = (first != out).any(dim=1).nonzero().flatten()
        if len(bad) and os.environ.get("COLS"):
            r0 = int(bad[0])
            print("   row %d columns that differ: %s" % (r0, (first[r0] != out[r0]).nonzero().flatten()[:40].tolist()), flush=True)
        rel = float(((first - out).abs().max() / first.abs().max()).item())
        print("call %d: %d rows differ from the first call's (max relative difference %.2e)%s" % (rep, len(bad), rel, (": rows " + str(bad[:12].tolist())) if len(bad) else ""), flush=True)


if __name__ == "__main__":
    main()
