#!/usr/bin/env python3
"""count->twist of assemblies, kpop_tune("dense", 0 | 2): the streaming kernel alone against the default route --
count_twist_tile_kernel (consensus on the matrix cores) + tile_residual_kernel (private rows) + the streaming kernel for what
they leave --, on 5,000 mutants of tests/golden/wuhan.fasta at AB_RATES divergences (one organism: BASELINE config 3's kind
of batch) and on 5,000 unrelated 30 kb genomes, k = 12, D = 64.  ms per kpop_dev_count_twist call (HIP events, median of 7)."""
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
    from tools.cli_kernels_workload import mutants
    kpop_amd.init(0)
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream()
    k, d, n = 12, 64, int(os.environ.get("AB_GENOMES", "5000"))
    tw = kpop_amd.Twister.synth(0x5EED, k, d)
    if os.environ.get("AB_TILEG"):
        api.tune("tileg", int(os.environ["AB_TILEG"]))

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

    rates = [float(x) for x in os.environ.get("AB_RATES", "0.001,0.003,0.01,0.03").split(",") if x]
    cases = [("%d mutants of wuhan.fasta at %.1f %%" % (n, 100 * r), mutants(n, rate=r)) for r in rates]
    if not os.environ.get("AB_NO_UNRELATED"):
        cases.append(("%d unrelated 30 kb genomes" % n, O.synth_reads(0xC1A55, n, 30000)))
    if os.environ.get("AB_DBG"):  # phase ablation of the tile kernel (results are wrong): kpop_tune("dbg", bits << 24)
        b, o = mutants(n, rate=float(os.environ.get("AB_DBG_RATE", "0.001")))
        db, do = torch.from_numpy(np.ascontiguousarray(b)).to(dev), torch.from_numpy(o.astype(np.int64)).to(dev)
        L = int(np.diff(o.astype(np.int64)).max())
        out = torch.zeros(n, d, dtype=torch.float64, device=dev)
        api.tune("dense", 2)
        for bits in [int(x) for x in os.environ["AB_DBG"].split(",")]:
            api.tune("dbg", bits << 24)
            t = timed(lambda: api.dev_count_twist(tw, db.data_ptr(), do.data_ptr(), n, db.numel(), L, out.data_ptr(), stream=st.cuda_stream))
            print("dbg bits %2d   %8.3f ms" % (bits, t), flush=True)
            if bits & 16:
                c = api.debug_counters(8)
                tot = float(sum(c)) or 1.0
                names = ["bases staged", "rows found", "set built", "X cleared + set numbered", "windows vs set", "X counted / listed", "matrix cores", "sums out"]
                print("   phase clocks, share of the blocks' time: " + "   ".join("%s %.3f" % (nm, x / tot) for nm, x in zip(names, c)), flush=True)
        api.tune("dbg", 0)
        api.tune("dense", 2)
        return
    for label, (b, o) in cases:
        db, do = torch.from_numpy(np.ascontiguousarray(b)).to(dev), torch.from_numpy(o.astype(np.int64)).to(dev)
        L = int(np.diff(o.astype(np.int64)).max())
        outs = {}
        for mode in (0, 2):
            api.tune("dense", mode)
            out = torch.zeros(n, d, dtype=torch.float64, device=dev)
            t = timed(lambda: api.dev_count_twist(tw, db.data_ptr(), do.data_ptr(), n, db.numel(), L, out.data_ptr(), stream=st.cuda_stream))
            outs[mode] = (t, out.cpu().numpy())
        api.tune("dense", 2)
        rel = float(np.max(np.abs(outs[0][1] - outs[2][1])) / np.max(np.abs(outs[0][1])))
        print("%-44s dense 0: streaming kernel %8.3f ms   default: tile + residual + streaming %8.3f ms  (%.2fx)  max rel diff %.1e"
              % (label, outs[0][0], outs[2][0], outs[0][0] / outs[2][0], rel), flush=True)


if __name__ == "__main__":
    main()
