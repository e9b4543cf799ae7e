#!/usr/bin/env python3
"""Few dimensions (few classes): what the reads stream of KPopTwistDB runs (kpop_spectra_twist: count + CSR twist for
n_dims <= 32) against the fused kernel (kpop_count_twist), wall time of the host entry points on 1M reads, k = 12."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import kpop_amd
    from oracle import oracle as O
    kpop_amd.init(0)
    k, n, L = 12, 1000000, 150
    b, o = O.synth_reads(0x4B506F70, n, L)
    for d in (9, 16, 32, 33, 64):
        tw = kpop_amd.Twister.synth(0x5EED, k, d)
        res = {}
        for name, fn in (("spectra_twist", lambda: tw.spectra_twist(b, o, k)), ("count_twist (fused)", lambda: tw.count_twist(b, o))):
            fn()
            ts = []
            for _ in range(3):
                t0 = time.time()
                r = fn()
                ts.append(time.time() - t0)
            res[name] = (min(ts), r)
        a, bb = res["spectra_twist"][1], res["count_twist (fused)"][1]
        print("D = %2d: spectra_twist %.3f s, fused %.3f s per 1M reads; max |difference| %.2e (bit-identical: %s)" % (
            d, res["spectra_twist"][0], res["count_twist (fused)"][0], float(np.max(np.abs(a - bb))), bool(np.array_equal(a, bb))), flush=True)
        tw.free()


if __name__ == "__main__":
    main()
