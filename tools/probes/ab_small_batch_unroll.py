#!/usr/bin/env python3
"""Small batches through the fused count->twist (BASELINE configs 2 and 5: 10,000 reads, one or two rounds of wavefronts): the
launch is one gather chain per wavefront, so its length is rows / (row loads in flight).  ms per launch with kpop_tune("unroll", 8 | 16)
for a few batch sizes; k = 10, D = 64 (rows from the caches) and -- AB_K15=1 -- k = 15, D = 16 against the 69 GB twister."""
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
    cases = [(15, 16)] if os.environ.get("AB_K15") else [(10, 64), (12, 64)]
    if os.environ.get("AB_CASES"):  # "k:d,k:d"
        cases = [tuple(int(x) for x in c.split(":")) for c in os.environ["AB_CASES"].split(",")]
    for k, d in cases:
        tw = kpop_amd.Twister.synth(0x5EED, k, d)
        for n in (2000, 10000, 30000, 100000):
            L = 150
            bases = torch.empty(n * L, dtype=torch.uint8, device=dev)
            offs = torch.empty(n + 1, dtype=torch.int64, device=dev)
            api.dev_synth_reads(0x4B506F70, n, L, bases.data_ptr(), offs.data_ptr(), stream=sp)
            out = torch.zeros(n, d, dtype=torch.float64, device=dev)
            res = {}
            line = "k=%2d D=%2d n=%6d:" % (k, d, n)
            for u in [int(x) for x in os.environ.get("AB_UNROLLS", "8,16,0").split(",")]:
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
                res[u] = out.clone()
                line += "   unroll %2d: %.4f ms" % (u, float(np.median(ms)))
            api.tune("unroll", 8)
            keys = list(res)
            same = all(torch.equal(res[keys[0]], res[x]) for x in keys[1:])
            print(line + "   same bits: %s" % same, flush=True)
        tw.free()


if __name__ == "__main__":
    main()
