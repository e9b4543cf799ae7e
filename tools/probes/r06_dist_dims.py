#!/usr/bin/env python3
"""round 6: distances on the f64 matrix cores at any number of dimensions (distance_mfma.hip, the tiled contraction).
  (a) kpop_dev_distance_rowwise (-d) R2 x R1 x D (default 100,000 samples x 1,636 classes x 1,635 dimensions: the reference's own job,
      README.md:1054-1060), matrix cores against kpop_tune("distance_mfma", 0), the largest relative difference between the two;
  (b) kpop_dev_distance_summary Q x RS x D (256 x 650,000 x 1,635: README.md:1101), matrix cores against kpop_tune("summary_mfma", 0).
ms per call (HIP events); flops = 2 x rows x rows x D of the contraction, over the whole call."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
PEAK = 78.6


def main():
    import torch
    import kpop_amd
    from kpop_amd import api
    kpop_amd.init(0)
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream()
    g = torch.Generator(device=dev)
    g.manual_seed(1)

    for kv in [x for x in os.environ.get("R06_TUNE", "").split(",") if x]:  # e.g. R06_TUNE=summary_lanes=1
        key, val = kv.split("=")
        api.tune(key, int(val))

    def timed(fn, reps=3):
        fn()
        torch.cuda.synchronize()
        ms = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            fn()
            e1.record(st)
            torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1))
        return float(np.median(ms))

    for case in os.environ.get("R06_D_CASES", "1636:100000:1635,1636:100000:256,4096:100000:64").split(","):
        if not case:
            continue
        r1, r2, d = (int(x) for x in case.split(":"))
        m1 = torch.randn(r1, d, dtype=torch.float64, device=dev, generator=g)
        m2 = torch.randn(r2, d, dtype=torch.float64, device=dev, generator=g)
        metric = torch.rand(d, dtype=torch.float64, device=dev, generator=g) + 0.1
        metric /= metric.sum()
        work = torch.empty(api.dev_distance_workspace_bytes(r1, r2, d), dtype=torch.uint8, device=dev)
        out = torch.empty(r2, r1, dtype=torch.float64, device=dev)
        res = {}
        for mode in (1, 0):
            api.tune("distance_mfma", mode)
            ms = timed(lambda: api.dev_distance_rowwise(m1.data_ptr(), r1, m2.data_ptr(), r2, d, metric.data_ptr(), work.data_ptr(), out.data_ptr(), stream=st.cuda_stream), 3 if mode else 1)
            res[mode] = (ms, out.clone())
        api.tune("distance_mfma", 1)
        fl = 2.0 * r1 * r2 * d
        rel = float(((res[1][1] - res[0][1]).abs() / res[0][1].abs().clamp_min(1e-300)).max())
        print("-d  %d x %d x %d: matrix cores %9.3f ms = %.2f TFLOP/s = %.3f of %.1f   vector pipe %9.3f ms (%.2fx)   max rel diff %.1e"
              % (r2, r1, d, res[1][0], fl / res[1][0] / 1e9, fl / res[1][0] / 1e9 / PEAK, PEAK, res[0][0], res[0][0] / res[1][0], rel), flush=True)
        del m1, m2, out, res, work
        torch.cuda.empty_cache()
    for case in os.environ.get("R06_S_CASES", "650000:256:1635,650000:1024:1635,1000000:256:64").split(","):
        if not case:
            continue
        r1, r2, d = (int(x) for x in case.split(":"))
        m1 = torch.randn(r1, d, dtype=torch.float64, device=dev, generator=g)
        data = os.environ.get("R06_S_DATA", "random")
        if data.startswith("clusters"):  # clusters:<members>:<noise>:<copies> -- a database of near-identical genomes (lineages), some of them exact copies
            _, members, noise, copies = data.split(":")
            members, noise, copies = int(members), float(noise), float(copies)
            centres = torch.randn((r1 + members - 1) // members, d, dtype=torch.float64, device=dev, generator=g)
            m1 = centres.repeat_interleave(members, dim=0)[:r1].clone()
            keep = torch.rand(r1, device=dev, generator=g) < copies  # these stay exact copies of their centre
            m1 += torch.where(keep[:, None], torch.zeros_like(m1), noise * torch.randn(r1, d, dtype=torch.float64, device=dev, generator=g))
        m2 = m1[torch.randperm(r1, device=dev)[:r2]].clone()
        metric = torch.rand(d, dtype=torch.float64, device=dev, generator=g) + 0.1
        metric /= metric.sum()
        work = torch.empty(api.dev_distance_workspace_bytes(r1, r2, d), dtype=torch.uint8, device=dev)
        K = int(os.environ.get("R06_S_K", "304"))
        stats = torch.zeros(r2, 4, dtype=torch.float64, device=dev)
        n = torch.zeros(r2, dtype=torch.int32, device=dev)
        idx = torch.zeros(r2, K, dtype=torch.int32, device=dev)
        dd = torch.zeros(r2, K, dtype=torch.float64, device=dev)
        z = torch.zeros_like(dd)
        keep = {}
        for mode in [int(x) for x in os.environ.get("R06_S_MODES", "1,0").split(",")]:  # 1: approximate rows then the summary's passes (default), 2: the pass inside the contraction (up to 128 dimensions), 0: vector pipe
            api.tune("summary_mfma", mode)
            api.tune("summary_audit", 1)
            api.summary_fallbacks()
            ms = timed(lambda: api.dev_distance_summary(m1.data_ptr(), r1, m2.data_ptr(), r2, d, metric.data_ptr(), work.data_ptr(), stats.data_ptr(), n.data_ptr(),
                                                        idx.data_ptr(), dd.data_ptr(), z.data_ptr(), keep_at_most=min(300, K - 4), max_neighbours=K, stream=st.cuda_stream), 3 if mode else 1)
            left = api.summary_fallbacks()
            api.tune("summary_audit", 0)
            ms = timed(lambda: api.dev_distance_summary(m1.data_ptr(), r1, m2.data_ptr(), r2, d, metric.data_ptr(), work.data_ptr(), stats.data_ptr(), n.data_ptr(),
                                                        idx.data_ptr(), dd.data_ptr(), z.data_ptr(), keep_at_most=min(300, K - 4), max_neighbours=K, stream=st.cuda_stream), 3 if mode else 1)
            keep[mode] = (ms, [x.cpu().numpy().copy() for x in (stats, n, idx, dd)])
            print("      summary_mfma %d: %9.3f ms   (rows left to the exact fall-back, per call: %d of %d)" % (mode, ms, left // (4 if mode else 2), r2), flush=True)
        api.tune("summary_mfma", 1)
        m_on = max(keep)
        keep[1] = keep[m_on]
        a, b = keep[1][1], keep[min(keep)][1]
        fl = 2.0 * r1 * r2 * d
        same = np.array_equal(a[0][:, 2:], b[0][:, 2:]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2][:, :min(300, K - 4)], b[2][:, :min(300, K - 4)]) and np.array_equal(a[3][:, :min(300, K - 4)], b[3][:, :min(300, K - 4)])
        print("-s  %d x %d x %d, 300 neighbours: matrix cores %9.3f ms = %.3f of %.1f TFLOP/s on the contraction's flops   vector pipe %9.3f ms (%.2fx)   medians, MADs, neighbours the same bits: %s"
              % (r2, r1, d, keep[1][0], fl / keep[1][0] / 1e9 / PEAK, PEAK, keep[min(keep)][0], keep[min(keep)][0] / keep[1][0], same), flush=True)
        del m1, m2, work
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
