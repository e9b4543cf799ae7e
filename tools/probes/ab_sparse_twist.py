#!/usr/bin/env python3
"""ms of the sparse twist (kpop_dev_twist) on genome spectra and on read spectra, for the library named by KPOP_HIP_LIB
(default: this tree's): run it once per build to compare two builds on the same box.
    python tools/probes/ab_sparse_twist.py; KPOP_HIP_LIB=kpop_amd/bin/libkpop_hip_r02.so python tools/probes/ab_sparse_twist.py"""
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
    kpop_amd.init(0)
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream()
    d = 64

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        ms = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            fn()
            e1.record(st)
            torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1))
        return float(np.median(ms))

    def case(label, tw, h, c, o, max_lines):
        n = len(o) - 1
        dh = torch.from_numpy(h.view(np.int64)).to(dev)
        dv = torch.from_numpy(c.astype(np.float64)).to(dev)
        do = torch.from_numpy(o.view(np.int64)).to(dev)
        out = torch.zeros(n, d, dtype=torch.float64, device=dev)
        for ml in sorted({0, max_lines}):
            t = timed(lambda: api.dev_twist(tw, dh.data_ptr(), dv.data_ptr(), do.data_ptr(), n, ml, out.data_ptr(), stream=st.cuda_stream))
            print("%-36s max_lines=%-6d %9.3f ms   [%s]" % (label, ml, t, os.environ.get("KPOP_HIP_LIB", "this tree")), flush=True)

    gb, go = O.synth_reads(0xC1A55, 4096, 30000)
    for k in (7, 8):
        tw = kpop_amd.Twister.synth(0x5EED, k, d)
        h, c, o = kpop_amd.count_reads(gb, go, k)
        case("4096 genomes 30 kb, k=%d" % k, tw, h, c, o, 0)
        tw.free()
    k = 12
    tw = kpop_amd.Twister.synth(0x5EED, k, d)
    rb, ro = O.synth_reads(0x4B506F70, 100000, 150)
    h, c, o = kpop_amd.count_reads(rb, ro, k)
    case("100,000 read spectra, k=12", tw, h, c, o, int(np.diff(o.astype(np.int64)).max()))


if __name__ == "__main__":
    main()
