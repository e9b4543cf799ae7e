#!/usr/bin/env python3
"""kpop_dev_count_twist on 100,000 reads of 150 bp, alone and with ONE longer sequence in the batch (300 bp, 500 bp, 30 kb):
the wavefront kernel's slot count is chosen from the longest read of the batch.  k = 12, D = 64."""
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
    k, d, n, L = 12, 64, 100000, 150
    tw = kpop_amd.Twister.synth(0x5EED, k, d)
    rng = np.random.RandomState(1)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    base = acgt[rng.randint(0, 4, size=n * L)]
    for extra in [int(x) for x in os.environ.get("EXTRA", "0,300,500,30000").split(",")]:
        lens = np.full(n, L, dtype=np.int64)
        b = base
        if extra:
            lens = np.concatenate([lens, [extra]])
            b = np.concatenate([base, acgt[rng.randint(0, 4, size=extra)]])
        offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        nn = len(lens)
        db, do = torch.from_numpy(b).to(dev), torch.from_numpy(offs).to(dev)
        out = torch.zeros(nn, d, dtype=torch.float64, device=dev)
        f = lambda: api.dev_count_twist(tw, db.data_ptr(), do.data_ptr(), nn, db.numel(), int(lens.max()), out.data_ptr(), stream=st.cuda_stream)
        f()
        torch.cuda.synchronize()
        ms = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            f()
            e1.record(st)
            torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1))
        print("100,000 x 150 bp%s: %.3f ms" % ((" + one sequence of %d bp" % extra) if extra else "", float(np.median(ms))), flush=True)


if __name__ == "__main__":
    main()
