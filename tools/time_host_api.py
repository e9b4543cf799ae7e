#!/usr/bin/env python3
"""Wall time of the host-buffer entry points (PCIe included): what a host-language binding sees."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def best(fn, reps=5):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return min(ts)


def main():
    import kpop_amd
    from oracle import oracle as O
    kpop_amd.init(0)
    k, d, n, L, C = 12, 64, 100000, 150, 65
    tw = kpop_amd.Twister.synth(0x5EED, k, d)
    bases, offs = O.synth_reads(0x4B506F70, n, L)
    cb, co = O.synth_reads(0xC1A55, C, 500)
    classes = tw.count_twist(cb, co)
    metric = kpop_amd.metric_compute(O.synth_inertia(d))
    t = best(lambda: tw.count_twist(bases, offs))
    print("kpop_count_twist   100k x 150 bp: %.2f ms  (%.1f M reads/s; %.0f MB up, %.0f MB down)" % (t * 1e3, n / t / 1e6, bases.nbytes / 1e6, n * d * 8 / 1e6))
    twisted = tw.count_twist(bases, offs)
    t = best(lambda: kpop_amd.distance_rowwise(classes, twisted, metric))
    print("kpop_distance_rowwise 65 x 100k:  %.2f ms  (%.0f MB up, %.0f MB down)" % (t * 1e3, twisted.nbytes / 1e6, n * C * 8 / 1e6))
    # straight through the C ABI with caller-owned, already-touched buffers (what a C/OCaml caller does)
    import ctypes as C
    from kpop_amd import _lib
    L_ = _lib.load()
    cap = n * (L - k + 1)
    oh, oc, oo = np.zeros(cap, np.uint64), np.zeros(cap, np.uint32), np.zeros(n + 1, np.uint64)
    p = lambda a, ty: a.ctypes.data_as(C.POINTER(ty))
    def count(per_read):
        rc = L_.kpop_count_reads(p(bases, C.c_uint8), p(offs, C.c_uint64), n, k, 0, per_read, p(oh, C.c_uint64), p(oc, C.c_uint32),
                                 p(oo, C.c_uint64), cap)
        assert rc == 0
    t = best(lambda: count(1))
    print("kpop_count_reads -L 100k x 150:   %.2f ms  (%.1f M reads/s; %.0f MB down)" % (t * 1e3, n / t / 1e6, int(oo[n]) * 12 / 1e6))
    t = best(lambda: count(0))
    print("kpop_count_reads -l 100k x 150:   %.2f ms" % (t * 1e3))
    out = np.zeros((n, d))
    def ct():
        assert L_.kpop_count_twist(tw.handle, p(bases, C.c_uint8), p(offs, C.c_uint64), n, 0, 1, p(out, C.c_double)) == 0
    t = best(ct)
    print("kpop_count_twist (reused output): %.2f ms  (%.1f M reads/s)" % (t * 1e3, n / t / 1e6))


if __name__ == "__main__":
    main()
