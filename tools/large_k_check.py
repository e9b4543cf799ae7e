#!/usr/bin/env python3
"""Config 5 of BASELINE.json: k=15 (537 M canonical k-mers).  Builds the synthetic twister on the device
(537 M rows x 16 dims = 69 GB), twists 10k reads and a few genomes, checks a sample against the oracle
(twister restricted to the sample's k-mers) and times the fused kernel."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch

    import kpop_amd
    from kpop_amd import api
    from oracle import oracle as O
    k, d = int(sys.argv[1]) if len(sys.argv) > 1 else 15, int(sys.argv[2]) if len(sys.argv) > 2 else 16
    kpop_amd.init(0)
    t0 = time.time()
    tw = kpop_amd.Twister.synth(0x5EED, k, d)
    info = tw.info()
    print("twister k=%d D=%d: %d columns, %.1f GB on device, built in %.1f s" % (k, d, info["n_cols"], info["device_bytes"] / 1e9, time.time() - t0))
    n, L = 10000, 150
    bases, offs = O.synth_reads(0x4B506F70, n, L)
    gb, go = O.synth_reads(0xABCDE, 3, 30000)
    allb = np.concatenate([bases, gb])
    allo = np.concatenate([offs, go[1:] + offs[-1]])
    got = tw.count_twist(allb, allo)
    pick = list(range(0, n, 500)) + [n, n + 1, n + 2]
    sb = np.concatenate([allb[int(allo[r]):int(allo[r + 1])] for r in pick])
    so = np.zeros(len(pick) + 1, dtype=np.uint64)
    so[1:] = np.cumsum([int(allo[r + 1] - allo[r]) for r in pick])
    h, c, o = O.count_reads(sb, so, k)
    cols = np.unique(h)
    T = O.synth_twister(0x5EED, d, cols)
    want = O.twist(T, cols, h, c.astype(np.float64), o)
    err = np.max(np.abs(got[pick] - want)) / np.max(np.abs(want))
    print("sample of %d sequences vs oracle: max rel err %.2e, short reads bit-exact: %s" % (len(pick), err, np.array_equal(got[pick[:-3]], want[:-3])))
    assert err < 1e-12
    # timing, device resident
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream()
    sp = stream.cuda_stream
    n2 = 100000
    db = torch.empty(n2 * L, dtype=torch.uint8, device=dev)
    do = torch.empty(n2 + 1, dtype=torch.int64, device=dev)
    api.dev_synth_reads(1, n2, L, db.data_ptr(), do.data_ptr(), stream=sp)
    out = torch.zeros(n2, d, dtype=torch.float64, device=dev)
    for _ in range(3):
        api.dev_count_twist(tw, db.data_ptr(), do.data_ptr(), n2, n2 * L, L, out.data_ptr(), stream=sp)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(10):
        api.dev_count_twist(tw, db.data_ptr(), do.data_ptr(), n2, n2 * L, L, out.data_ptr(), stream=sp)
    e1.record(stream)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    w = L - k + 1
    gbytes = n2 * (L + w * d * 8 + d * 8) / 1e9
    print("fused count->twist 100k x 150 bp: %.3f ms  (%.1f M reads/s, algorithmic %.0f GB/s)" % (ms, n2 / ms / 1e3, gbytes / ms * 1e3))


if __name__ == "__main__":
    main()
