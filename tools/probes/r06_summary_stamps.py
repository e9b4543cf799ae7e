import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, kpop_amd
from kpop_amd import api, _lib
kpop_amd.init(0)
dev = torch.device("cuda", 0); st = torch.cuda.current_stream()
g = torch.Generator(device=dev); g.manual_seed(1)
r1, r2, d = 1000000, 256, 64
m1 = torch.randn(r1, d, dtype=torch.float64, device=dev, generator=g)
data = os.environ.get("R06_S_DATA", "random")
if data.startswith("clusters"):  # clusters:<members>:<noise>:<copies>, as tools/probes/r06_dist_dims.py
    _, members, noise, copies = data.split(":")
    members, noise, copies = int(members), float(noise), float(copies)
    centres = torch.randn((r1 + members - 1) // members, d, dtype=torch.float64, device=dev, generator=g)
    m1 = centres.repeat_interleave(members, dim=0)[:r1].clone()
    keep = torch.rand(r1, device=dev, generator=g) < copies
    m1 += torch.where(keep[:, None], torch.zeros_like(m1), noise * torch.randn(r1, d, dtype=torch.float64, device=dev, generator=g))
m2 = m1[torch.randperm(r1, device=dev)[:r2]].clone()
metric = torch.rand(d, dtype=torch.float64, device=dev, generator=g) + 0.1; metric /= metric.sum()
work = torch.empty(api.dev_distance_workspace_bytes(r1, r2, d), dtype=torch.uint8, device=dev)
K = 304
stats = torch.zeros(r2, 4, dtype=torch.float64, device=dev); n = torch.zeros(r2, dtype=torch.int32, device=dev)
idx = torch.zeros(r2, K, dtype=torch.int32, device=dev); dd = torch.zeros(r2, K, dtype=torch.float64, device=dev); z = torch.zeros_like(dd)
api.tune('summary_mfma', int(os.environ.get('R06_MODE', '1')))
for _ in range(2):
    api.dev_distance_summary(m1.data_ptr(), r1, m2.data_ptr(), r2, d, metric.data_ptr(), work.data_ptr(), stats.data_ptr(), n.data_ptr(), idx.data_ptr(), dd.data_ptr(), z.data_ptr(), keep_at_most=300, max_neighbours=K, stream=st.cuda_stream)
torch.cuda.synchronize()
out = (C.c_ulonglong * 32)()
L = _lib.load()
print("rc", L.kpop_debug_summary_stamps(out))
v = list(out)
print("finish: phases (cycles):", [v[i + 1] - v[i] for i in range(6)], " n_c", v[8], "n_nb", v[9], "nmed", v[10])
print("sample: phases:", [v[i + 1] - v[i] for i in range(16, 23)])
print("rows left to the fall-back by the finish kernel, by reason 1..6 (cumulative over the calls):", v[25:31], "last such row", v[31])
