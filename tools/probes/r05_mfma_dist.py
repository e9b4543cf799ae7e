"""times kpop_dev_distance_summary 256 x 1M and 1024 x 1M (64 dimensions) with and without the matrix-core path"""
import sys, time
import numpy as np, torch
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from kpop_amd import api, _lib
lib = _lib.load()
api.init(0)
dev = torch.device("cuda", 0)
d, r1 = 64, 1_000_000
g = torch.Generator(device=dev); g.manual_seed(1)
if os.environ.get("DATA") == "twisted":  # what bench.py's config 4 summarises: twisted rows of 1M random reads (k = 12), metric powers(1,1,2) of w_d ~ 2^(-d/8)
    import kpop_amd
    sp = torch.cuda.current_stream().cuda_stream
    tw = kpop_amd.Twister.synth(0x5EED, 12, d)
    bases = torch.empty(r1 * 150, dtype=torch.uint8, device=dev)
    offs = torch.empty(r1 + 1, dtype=torch.int64, device=dev)
    api.dev_synth_reads(0x4B506F70, r1, 150, bases.data_ptr(), offs.data_ptr(), stream=sp)
    m1 = torch.zeros(r1, d, dtype=torch.float64, device=dev)
    api.dev_count_twist(tw, bases.data_ptr(), offs.data_ptr(), r1, r1 * 150, 150, m1.data_ptr(), stream=sp)
    torch.cuda.synchronize()
    w = 2.0 ** (-np.arange(d) / 8.0)
    metric = torch.from_numpy(kpop_amd.metric_compute(w / w.sum())).to(dev)
else:
    m1 = torch.randn(r1, d, dtype=torch.float64, device=dev, generator=g)
    metric = torch.rand(d, dtype=torch.float64, device=dev, generator=g) + 0.1
for r2 in (256, 1024):
    m2 = m1[int(os.environ.get("Q0", "0")):int(os.environ.get("Q0", "0")) + r2].clone() if os.environ.get("DATA") == "twisted" else m1[torch.randperm(r1, device=dev)[:r2]].clone()
    work = torch.empty(api.dev_distance_workspace_bytes(r1, r2, d), dtype=torch.uint8, device=dev)
    K = 8
    stats = torch.zeros(r2, 4, dtype=torch.float64, device=dev); n = torch.zeros(r2, dtype=torch.int32, device=dev)
    idx = torch.zeros(r2, K, dtype=torch.int32, device=dev); dd = torch.zeros(r2, K, dtype=torch.float64, device=dev); z = torch.zeros_like(dd)
    keep = {}
    for mode in (1, 0, 1):
        api.tune("summary_mfma", mode)
        ts = []
        for it in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            api.dev_distance_summary(m1.data_ptr(), r1, m2.data_ptr(), r2, d, metric.data_ptr(), work.data_ptr(), stats.data_ptr(), n.data_ptr(),
                                     idx.data_ptr(), dd.data_ptr(), z.data_ptr(), keep_at_most=2, max_neighbours=K)
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        print(r2, "mfma" if mode else "vector", ["%.3f" % t for t in ts], flush=True)
        keep[mode] = [x.cpu().numpy().copy() for x in (stats, n, idx, dd, z)]
    a, b = keep[1], keep[0]
    print("   median/MAD equal", np.array_equal(a[0][:, 2:], b[0][:, 2:]), "n equal", np.array_equal(a[1], b[1]), "idx equal", np.array_equal(a[2][:, :2], b[2][:, :2]),
          "dist equal", np.array_equal(a[3][:, :2], b[3][:, :2]), "mean/sd rel", np.max(np.abs(a[0][:, :2] - b[0][:, :2]) / np.abs(b[0][:, :2])),
          "z rel", np.nanmax(np.abs(a[4][:, 1] - b[4][:, 1]) / np.abs(b[4][:, 1])))
