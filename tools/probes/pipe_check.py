#!/usr/bin/env python3
"""count_twist_tile_pipe_kernel (tile_pipe.h) against the streaming kernel and round 4's tile kernel on small batches of wuhan
mutants: max relative difference, bit-stability over repeated runs, then (PIPE_TIME=1) ms per call on 5,000 of them.
Run under `timeout`: a protocol error between the kernel's halves would spin for ever."""
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
    k, d = 12, int(os.environ.get("PIPE_D", "64"))
    tw = kpop_amd.Twister.synth(0x5EED, k, d)

    def run(db, do, n, L, mode, pipe):
        api.tune("dense", mode)
        api.tune("tilepipe", pipe)
        out = torch.zeros(n, d, dtype=torch.float64, device=dev)
        api.dev_count_twist(tw, db.data_ptr(), do.data_ptr(), n, db.numel(), L, out.data_ptr(), stream=st.cuda_stream,
                            normalize=not os.environ.get("PIPE_RAW"))
        torch.cuda.synchronize()
        return out.cpu().numpy()

    cases = [(int(n), float(r)) for n, r in (c.split(":") for c in os.environ.get("PIPE_CASES", "64:0.001,100:0.01,333:0.003,1000:0.03").split(","))]
    for n, rate in cases:
        b, o = mutants(n, rate=rate)
        if os.environ.get("PIPE_NS"):  # a few Ns and lowercase stretches
            rng = np.random.RandomState(7)
            b = b.copy()
            b[rng.randint(0, b.size, size=b.size // 5000)] = ord("N")
        db, do = torch.from_numpy(np.ascontiguousarray(b)).to(dev), torch.from_numpy(o.astype(np.int64)).to(dev)
        L = int(np.diff(o.astype(np.int64)).max())
        ref = run(db, do, n, L, 0, 0)
        old = run(db, do, n, L, 2, 0)
        if os.environ.get("PIPE_DBG"):
            api.tune("dbg", int(os.environ["PIPE_DBG"]) << 24)
        if os.environ.get("PIPE_COUNTS"):
            api.tune("dbg", (32 | int(os.environ.get("PIPE_COUNTS_DBG", "0"))) << 24)
            api.debug_counters(16)
            run(db, do, n, L, 2, 1)
            c = api.debug_counters(16)
            print("   chunks seen %d, count past 255 %d, reference unusable %d, share too little %d, rebuilt %d; taken %d, members multiplied %d"
                  % (c[8], c[9], c[10], c[11], c[12], c[14], c[15]))
        new = [run(db, do, n, L, 2, 1) for _ in range(3)]
        api.tune("dbg", 0)
        sc = np.max(np.abs(ref))
        if os.environ.get("PIPE_ROWS"):
            err = np.max(np.abs(new[0] - ref), axis=1) / sc
            print("   per-row error (first 72 rows): " + " ".join("%.0e" % e for e in err[:72]))
            r = int(np.argmax(err))
            print("   worst row %d: pipe %s" % (r, np.array2string(new[0][r, :6], precision=4)))
            print("   worst row %d: ref  %s" % (r, np.array2string(ref[r, :6], precision=4)))
            print("   row 0: pipe %s  ref %s" % (np.array2string(new[0][0, :4], precision=5), np.array2string(ref[0, :4], precision=5)))
        print("%5d mutants at %.1f %%: old tile vs streaming %.1e   pipe vs streaming %.1e   pipe run to run identical: %s"
              % (n, 100 * rate, np.max(np.abs(old - ref)) / sc, np.max(np.abs(new[0] - ref)) / sc,
                 all(np.array_equal(new[0], x) for x in new[1:])), flush=True)
    if os.environ.get("PIPE_TIME"):
        n = int(os.environ.get("PIPE_TIME_N", "5000"))
        for rate in [float(x) for x in os.environ.get("PIPE_RATES", "0.001,0.01").split(",")]:
            b, o = mutants(n, rate=rate)
            db, do = torch.from_numpy(np.ascontiguousarray(b)).to(dev), torch.from_numpy(o.astype(np.int64)).to(dev)
            L = int(np.diff(o.astype(np.int64)).max())
            out = torch.zeros(n, d, dtype=torch.float64, device=dev)
            res = {}
            only = os.environ.get("PIPE_ONLY")
            for name, mode, pipe in (("streaming", 0, 0), ("tile r4", 2, 0), ("pipe", 2, 1)):
                if only and name != only:
                    continue
                api.tune("dense", mode)
                api.tune("tilepipe", pipe)
                ms = []
                for it in range(6):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(st)
                    api.dev_count_twist(tw, db.data_ptr(), do.data_ptr(), n, db.numel(), L, out.data_ptr(), stream=st.cuda_stream)
                    e1.record(st)
                    torch.cuda.synchronize()
                    ms.append(e0.elapsed_time(e1))
                res[name] = float(np.median(ms[1:]))
            print("%d mutants at %.1f %%: " % (n, 100 * rate) + "   ".join("%s %.3f ms" % kv for kv in res.items()), flush=True)
        if os.environ.get("PIPE_ABLATE"):  # results are wrong: timing only
            api.tune("dense", 2)
            api.tune("tilepipe", 1)
            for bits in [int(x) for x in os.environ["PIPE_ABLATE"].split(",")]:
                api.tune("dbg", bits << 24)
                ms = []
                for it in range(5):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(st)
                    api.dev_count_twist(tw, db.data_ptr(), do.data_ptr(), n, db.numel(), L, out.data_ptr(), stream=st.cuda_stream)
                    e1.record(st)
                    torch.cuda.synchronize()
                    ms.append(e0.elapsed_time(e1))
                print("   ablation bits %2d (1 no MFMA, 2 no X adds, 4 no gather, 8 no pass B): %.3f ms" % (bits, float(np.median(ms[1:]))), flush=True)
            api.tune("dbg", 0)
        if os.environ.get("PIPE_STAMPS"):
            api.tune("dense", 2)
            api.tune("tilepipe", 1)
            api.tune("dbg", 16 << 24)
            api.debug_counters(16)
            api.dev_count_twist(tw, db.data_ptr(), do.data_ptr(), n, db.numel(), L, out.data_ptr(), stream=st.cuda_stream)
            torch.cuda.synchronize()
            c = api.debug_counters(16)
            api.tune("dbg", 0)
            pn = {0: "stage+clear", 1: "seeds+set", 2: "rows+number+X clear", 3: "windows", 4: "misses listed", 5: "their rows+publish", 6: "IN BARRIERS", 7: "WAIT FOR EMPTY", 13: "seeds hashed", 14: "next chunk's bases asked for", 15: "windows: pass A (of windows)"}
            cn = {8: "WAIT FOR A CHUNK", 9: "mfma loop", 10: "sums out", 11: "IN BARRIERS", 12: "gather tail"}
            ptot = float(sum(c[i] for i in pn)) or 1.0
            ctot = float(sum(c[i] for i in cn)) or 1.0
            print("producer wavefront 8 (%.0f ticks): " % ptot + "   ".join("%s %.3f" % (pn[i], c[i] / ptot) for i in sorted(pn)))
            print("consumer wavefront 0 (%.0f ticks): " % ctot + "   ".join("%s %.3f" % (cn[i], c[i] / ctot) for i in sorted(cn)))
    api.tune("dense", 2)
    api.tune("tilepipe", 1)


if __name__ == "__main__":
    main()
