#!/usr/bin/env python3
"""BASELINE config 5's shape (k = 15, D = 16, every canonical 15-mer) with and without the rows at their hashes (twister.h `direct`):
ms per launch of the fused count->twist for a few batch sizes, the same bits both ways."""
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
    sp = st.cuda_stream
    k, d, L = int(os.environ.get("AB_K", "15")), int(os.environ.get("AB_D", "16")), 150
    outs = {}
    for mode in (0, 2):
        api.tune("direct", mode)
        tw = kpop_amd.Twister.synth(0x5EED, k, d)
        info = tw.info()
        for n in (2000, 10000, 30000, 100000):
            bases = torch.empty(n * L, dtype=torch.uint8, device=dev)
            offs = torch.empty(n + 1, dtype=torch.int64, device=dev)
            api.dev_synth_reads(0x4B506F70, n, L, bases.data_ptr(), offs.data_ptr(), stream=sp)
            out = torch.zeros(n, d, dtype=torch.float64, device=dev)
            line = "direct %d (%.1f GB of %.1f)  k=%2d D=%2d n=%6d:" % (mode, info["direct_bytes"] / 1e9, info["device_bytes"] / 1e9, k, d, n)
            for u in (8, 16):
                api.tune("unroll", u)
                f = lambda: api.dev_count_twist(tw, bases.data_ptr(), offs.data_ptr(), n, n * L, L, out.data_ptr(), stream=sp)
                f()
                torch.cuda.synchronize()
                ms = []
                for _ in range(7):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(st)
                    for _ in range(10):
                        f()
                    e1.record(st)
                    torch.cuda.synchronize()
                    ms.append(e0.elapsed_time(e1) / 10)
                m = float(np.median(ms))
                w = L - k + 1
                line += "   unroll %2d: %.4f ms = %.2f TB/s of rows" % (u, m, n * w * d * 8 / (m * 1e-3) / 1e12)
            api.tune("unroll", 8)
            if mode == 0:
                outs[n] = out.clone()
            else:
                line += "   same bits as direct 0: %s" % torch.equal(outs[n], out)
            print(line, flush=True)
        tw.free()
    api.tune("direct", 2)


if __name__ == "__main__":
    main()
