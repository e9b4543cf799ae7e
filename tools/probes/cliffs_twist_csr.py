#!/usr/bin/env python3
"""kpop_dev_twist (CSR spectra, f64 values) over numbers of spectra and lines per spectrum: G lines/s -- looking for the sizes
where the launch choice (lines kept in registers up to 512; 8 or 32 row loads in flight at 32,768 spectra) falls off.
k = 12, D = 64, the lines of a spectrum sorted random k-mers."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import torch
    import kpop_amd
    from kpop_amd import api
    kpop_amd.init(0)
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream()
    k, d = 12, int(os.environ.get("CSR_DIMS", "64"))
    tw = kpop_amd.Twister.synth(0x5EED, k, d)
    n_kmers = 4 ** k
    g = torch.Generator(device=dev)
    g.manual_seed(2)
    sizes = os.environ.get("SIZES")
    todo = [tuple(int(x) for x in q.split("x")) for q in sizes.split(",")] if sizes else None
    for n, lines in todo or ((100000, 60), (100000, 64), (100000, 65), (100000, 139), (100000, 512), (100000, 513), (32768, 139), (32769, 139), (20000, 139),
                     (5000, 139), (5000, 3000), (2000, 29700), (40000, 3000), (200, 300000)):
        total = n * lines
        # random hashes, sorted within each spectrum; canonical or not does not matter for the timing (unknown ones are skipped)
        h = torch.randint(0, n_kmers, (n, lines), dtype=torch.int64, device=dev, generator=g)
        h, _ = torch.sort(h, dim=1)
        v = torch.ones(total, dtype=torch.float64, device=dev)
        offs = (torch.arange(n + 1, dtype=torch.int64, device=dev) * lines)
        out = torch.empty(n, d, dtype=torch.float64, device=dev)
        f = lambda: api.dev_twist(tw, h.data_ptr(), v.data_ptr(), offs.data_ptr(), n, lines, out.data_ptr(), stream=st.cuda_stream)
        f()
        torch.cuda.synchronize()
        ms = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            f()
            e1.record(st)
            torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1))
        t = float(np.median(ms))
        print("%7d spectra x %6d lines  %8.3f ms  %6.2f G lines/s" % (n, lines, t, total / t / 1e6), flush=True)
        del h, v, out


if __name__ == "__main__":
    main()
