import sys, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
import torch, kpop_amd
from kpop_amd import api
kpop_amd.init(0)
dev = torch.device('cuda', 0); st = torch.cuda.current_stream(); sp = st.cuda_stream
n, L, k = 100000, 150, 12
bases = torch.empty(n * L, dtype=torch.uint8, device=dev); offs = torch.empty(n + 1, dtype=torch.int64, device=dev)
api.dev_synth_reads(0x4B506F70, n, L, bases.data_ptr(), offs.data_ptr(), stream=sp)
w = L - k + 1
scratch = torch.empty(api.dev_count_reads_scratch_bytes(n, L, k), dtype=torch.uint8, device=dev)
oh = torch.empty(n * 256, dtype=torch.int64, device=dev); oc = torch.empty(n * 256, dtype=torch.int32, device=dev); oo = torch.empty(n + 1, dtype=torch.int64, device=dev)
for dbg in (0, 8, 16, 48, 1, 3):  # 0 two-level look-back, 8 one level (round 2), 16.. naps, 1 none, 3 none and no ticket
    api.tune("dbg", dbg)
    f = lambda: api.dev_count_reads(bases.data_ptr(), offs.data_ptr(), n, L, k, scratch.data_ptr(), oh.data_ptr(), oc.data_ptr(), oo.data_ptr(), stream=sp)
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(5):
        e0.record(st)
        for _ in range(20): f()
        e1.record(st); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 20)
    print("dbg=%d  %.4f ms" % (dbg, np.median(ts)))
