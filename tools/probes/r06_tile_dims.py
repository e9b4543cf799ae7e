#!/usr/bin/env python3
"""count->twist of assemblies of ONE organism through the pipelined tile kernel at several numbers of dimensions (round 6: beyond 64
the kernel takes the twister's columns in slabs of 64, tile_pipe.h WIDE).  N mutants (0.3 % substitutions by default) of one
synthetic 30 kb genome, made on the device; per (k, D): ms per kpop_dev_count_twist call (HIP events, median), the flops of the
consensus contraction (2 x 64 x D x the set rows multiplied, counted by the kernel in an extra call) as a fraction of the f64 matrix
peak, the streaming kernel alone (kpop_tune("dense", 0)) and, with R06_CLOCKS=1, the kernel's phase clocks.

  R06_N=5000 R06_CASES=12:64,12:256,10:1635 R06_RATE=0.003 python tools/probes/r06_tile_dims.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
PEAK = 78.6  # TFLOP/s, dense f64 MFMA (MI355X_MICROARCH.md)


def main():
    import torch
    import kpop_amd
    from kpop_amd import api
    kpop_amd.init(0)
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream()
    n, L = int(os.environ.get("R06_N", "5000")), int(os.environ.get("R06_L", "30000"))
    rate = float(os.environ.get("R06_RATE", "0.003"))
    cases = [tuple(int(x) for x in c.split(":")) for c in os.environ.get("R06_CASES", "12:64,12:256,10:1635").split(",")]
    reps = int(os.environ.get("R06_REPS", "5"))
    ref = torch.empty(L, dtype=torch.uint8, device=dev)
    ro = torch.empty(2, dtype=torch.int64, device=dev)
    api.dev_synth_reads(0x0123, 1, L, ref.data_ptr(), ro.data_ptr(), stream=st.cuda_stream)
    torch.cuda.synchronize()
    bases = ref.repeat(n)
    acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    g = torch.Generator(device=dev)
    g.manual_seed(0x0123)
    step = 1 << 27
    for lo in range(0, n * L, step):
        hi = min(n * L, lo + step)
        hit = torch.rand(hi - lo, device=dev, generator=g) < rate
        sub = acgt[torch.randint(0, 4, (hi - lo,), device=dev, generator=g)]
        bases[lo:hi] = torch.where(hit, sub, bases[lo:hi])
    offs = torch.arange(n + 1, dtype=torch.int64, device=dev) * L

    def timed(fn, reps):
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

    for key in ("pipeprio", "tilewide"):
        if os.environ.get("R06_" + key.upper()):
            api.tune(key, int(os.environ["R06_" + key.upper()]))
    print("%d mutants of one %d-base genome at %.2f %% substitutions" % (n, L, 100 * rate), flush=True)
    for k, d in cases:
        tw = kpop_amd.Twister.synth(0x5EED, k, d)
        out = torch.zeros(n, d, dtype=torch.float64, device=dev)
        call = lambda: api.dev_count_twist(tw, bases.data_ptr(), offs.data_ptr(), n, n * L, L, out.data_ptr(), stream=st.cuda_stream)
        ms = timed(call, reps)
        got = out.clone()
        for bits in [int(x) for x in os.environ.get("R06_DBG", "").split(",") if x]:  # ablations (results are wrong under them): 1 no MFMAs, 4 no residual gather, 64 one issue priority
            api.tune("dbg", bits << 24)
            print("      dbg %2d: %9.3f ms" % (bits, timed(call, 3)), flush=True)
            api.tune("dbg", 0)
        api.debug_counters(16)
        api.tune("dbg", 32 << 24)
        call()
        cnt = api.debug_counters(16)
        api.tune("dbg", 0)
        flops = 2.0 * 64 * d * cnt[15]
        line = "k=%2d D=%4d   %9.3f ms   chunks %d  set rows %d   %.2f TFLOP/s = %.3f of %.1f" % (k, d, ms, cnt[14], cnt[15], flops / ms / 1e9, flops / ms / 1e9 / PEAK, PEAK)
        if not os.environ.get("R06_NO_STREAM"):
            api.tune("dense", 0)
            ms0 = timed(call, 2)
            api.tune("dense", 2)
            rel = float((out - got).abs().max() / out.abs().max())
            line += "   streaming kernel alone %9.3f ms (%.2fx)  max rel diff %.1e" % (ms0, ms0 / ms, rel)
        print(line, flush=True)
        if os.environ.get("R06_CLOCKS"):
            api.tune("dbg", 16 << 24)
            call()
            c = api.debug_counters(16)
            api.tune("dbg", 0)
            pn = ["staged", "set built", "rows+X set", "windows", "misses listed", "misses' rows", "p barriers", "wait empty"]
            ptot = float(sum(c[:8])) or 1.0
            print("   producers: " + "  ".join("%s %.3f" % (nm, x / ptot) for nm, x in zip(pn, c[:8])), flush=True)
            cn = ["wait chunk", "matrix cores(+gather)", "sums out", "c barriers", "list tail (D<=64)"]
            ctot = float(sum(c[8:13])) or 1.0
            print("   consumers: " + "  ".join("%s %.3f" % (nm, x / ctot) for nm, x in zip(cn, c[8:13])), flush=True)
        tw.free()
        del out, got
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
