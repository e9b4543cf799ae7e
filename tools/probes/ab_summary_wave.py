#!/usr/bin/env python3
"""distance_summary_wave_kernel on the classifier's shape (65 class vectors x 100,000 twisted reads x 64 dimensions): ms per
kpop_dev_distance_summary call (HIP events, median of 9), and -- AB_DBG=1 -- the phase ablation of the kernel
(kpop_tune("dbg", bits << 16): 1 no sort of the (distance, column) pairs, 2 no sequential chains, 4 no sort for the MAD,
8 no distances; results are wrong under them)."""
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
    r2, d = int(os.environ.get("AB_READS", "100000")), 64
    rng = np.random.default_rng(5)
    for r1 in [int(x) for x in os.environ.get("AB_CLASSES", "65").split(",")]:
        m1 = torch.from_numpy(rng.standard_normal((r1, d))).to(dev)
        m2 = torch.from_numpy(rng.standard_normal((r2, d))).to(dev)
        metric = torch.from_numpy(rng.random(d) + 0.1).to(dev)
        work = torch.empty(api.dev_distance_workspace_bytes(r1, r2, d) // 8 + 16, dtype=torch.float64, device=dev)
        stats = torch.empty(r2, 4, dtype=torch.float64, device=dev)
        nn = torch.empty(r2, dtype=torch.int32, device=dev)
        idx = torch.empty(r2, 8, dtype=torch.int32, device=dev)
        dd = torch.empty(r2, 8, dtype=torch.float64, device=dev)
        z = torch.empty(r2, 8, dtype=torch.float64, device=dev)

        def run():
            api.dev_distance_summary(m1.data_ptr(), r1, m2.data_ptr(), r2, d, metric.data_ptr(), work.data_ptr(), stats.data_ptr(),
                                     nn.data_ptr(), idx.data_ptr(), dd.data_ptr(), z.data_ptr(), keep_at_most=2, max_neighbours=8,
                                     stream=st.cuda_stream)

        def timed():
            run()
            torch.cuda.synchronize()
            ms = []
            for _ in range(9):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st)
                run()
                e1.record(st)
                torch.cuda.synchronize()
                ms.append(e0.elapsed_time(e1))
            return float(np.median(ms))

        print("%d classes x %d reads x %d: %.3f ms a call (normalisation of the rows included)" % (r1, r2, d, timed()), flush=True)
        if os.environ.get("AB_DBG"):
            for bits in (1, 2, 4, 8, 15):
                api.tune("dbg", bits << 16)
                print("   dbg bits %2d   %.3f ms" % (bits, timed()), flush=True)
            api.tune("dbg", 0)


if __name__ == "__main__":
    main()
