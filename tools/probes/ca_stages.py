#!/usr/bin/env python3
"""kpop_ca at 524,800 x 1,636 with KPOP_TIMING=1: where the wall time goes (stderr lines of the library)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["KPOP_TIMING"] = "1"


def main():
    import kpop_amd
    kpop_amd.init(0)
    I, J = 524800, 1636
    rng = np.random.RandomState(J)
    N = rng.poisson(3.0, size=(I, J)).astype(np.float64) * rng.lognormal(0, 0.5, size=(I, 1))
    N = np.floor(N) + 1.0
    for rep in range(int(os.environ.get("REPS", "2"))):
        t0 = time.time()
        nd = J - 1
        outs = (np.zeros((J, nd)), np.zeros(nd), np.zeros((nd, I)))
        t1 = time.time()
        import ctypes as C
        from kpop_amd import _lib
        n_out = C.c_uint32()
        _lib.load().kpop_ca(N.ctypes.data_as(C.POINTER(C.c_double)), I, J, 1, C.byref(n_out), *[o.ctypes.data_as(C.POINTER(C.c_double)) for o in outs])
        t2 = time.time()
        print("rep %d: outputs allocated in %.3f s, kpop_ca %.3f s wall" % (rep, t1 - t0, t2 - t1), file=sys.stderr, flush=True)
        if rep == int(os.environ.get("REPS", "2")) - 1:
            t1 = time.time()
            _lib.load().kpop_ca(N.ctypes.data_as(C.POINTER(C.c_double)), I, J, 1, C.byref(n_out), *[o.ctypes.data_as(C.POINTER(C.c_double)) for o in outs])
            print("again into the same (touched) outputs: kpop_ca %.3f s wall" % (time.time() - t1), file=sys.stderr, flush=True)
        del outs
    import torch
    from kpop_amd import api
    dev = torch.device("cuda", 0)
    nd = J - 1
    dN = torch.from_numpy(N).to(dev)
    work = torch.empty(api.dev_ca_workspace_bytes(I, J), dtype=torch.uint8, device=dev)
    d_tw = torch.zeros(J, nd, dtype=torch.float64, device=dev)
    d_in = torch.zeros(nd, dtype=torch.float64, device=dev)
    d_T = torch.zeros(nd, I, dtype=torch.float64, device=dev)
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.time()
        api.dev_ca(dN.data_ptr(), I, J, work.data_ptr(), d_tw.data_ptr(), d_in.data_ptr(), d_T.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        print("kpop_dev_ca rep %d: %.3f s wall" % (rep, time.time() - t0), file=sys.stderr, flush=True)


if __name__ == "__main__":
    main()
