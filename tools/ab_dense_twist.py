#!/usr/bin/env python3
"""A/B: the twist as the reference's sparse mat-vec (twist_csr_kernel, one gathered twister row per line) against the
dense contraction on the f64 matrix cores (kpop_dev_twist_dense: spectra laid out as X[batch x n_kmers], one pass over
the twister per batch tile), on spectra kept in HBM.  Per case: ms of each (HIP events, median of 5), the flops the dense
form performs against the 78.6 TFLOP/s f64 matrix peak, and the largest relative difference of the results.

  genomes at k = 7..10   N assemblies of 30 kb (unrelated random ones: every k-mer of the k is about equally likely)
  class spectra, k = 12  65 spectra of ~30k k-mers against a twister that holds exactly those k-mers
  reads, k = 12          100k x 150 bp (the case SURVEY.md F5 settles by arithmetic; here by measurement, 2,000 reads)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch

    import kpop_amd
    from kpop_amd import api
    from oracle import oracle as O  # synthetic inputs only
    kpop_amd.init(0)
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream()
    d = int(os.environ.get("AB_DIMS", "64"))

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        ms = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            fn()
            e1.record(st)
            torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1))
        return float(np.median(ms))

    def case(label, tw, h, c, o):
        n = len(o) - 1
        dh = torch.from_numpy(h.view(np.int64)).to(dev)
        dv = torch.from_numpy(c.astype(np.float64)).to(dev)
        do = torch.from_numpy(o.view(np.int64)).to(dev)
        out1 = torch.zeros(n, d, dtype=torch.float64, device=dev)
        out2 = torch.zeros(n, d, dtype=torch.float64, device=dev)
        work = torch.empty(api.dev_twist_dense_workspace_bytes(tw, n), dtype=torch.uint8, device=dev)
        t1 = timed(lambda: api.dev_twist(tw, dh.data_ptr(), dv.data_ptr(), do.data_ptr(), n, 0, out1.data_ptr(), stream=st.cuda_stream))
        t3 = timed(lambda: api.dev_twist_dense(tw, dh.data_ptr(), dv.data_ptr(), do.data_ptr(), n, work.data_ptr(), out2.data_ptr(), stream=st.cuda_stream))
        t2 = timed(lambda: api.dev_twist_dense_sorted(tw, dh.data_ptr(), dv.data_ptr(), do.data_ptr(), n, work.data_ptr(), out2.data_ptr(), stream=st.cuda_stream))
        a, b = out1.cpu().numpy(), out2.cpu().numpy()
        rel = float(np.max(np.abs(a - b)) / max(np.max(np.abs(a)), 1e-300))
        rows = tw.info()["n_cols"]
        flops = 2.0 * n * rows * d
        nnz = len(h)
        print("%-44s spectra %6d  lines/spectrum %7.0f of %9d columns (%.2f %%)  sparse %9.3f ms  dense(fused) %9.3f ms  (%5.2fx)  %5.1f TFLOP/s = %.2f of 78.6  | dense(image in HBM, round 2) %9.3f ms = %.2f  max rel diff %.1e"
              % (label, n, nnz / n, rows, 100.0 * nnz / n / rows, t1, t2, t1 / t2, flops / t2 / 1e9, flops / t2 / 1e9 / 78.6, t3, flops / t3 / 1e9 / 78.6, rel), flush=True)

    def from_sequences(label, tw, gb, go, k):
        """the whole stage from sequences: fused sparse kernels (kpop_dev_count_twist) against the dense u32 image + f64 MFMA"""
        n = len(go) - 1
        db, dof = torch.from_numpy(gb).to(dev), torch.from_numpy(go.view(np.int64)).to(dev)
        out1 = torch.zeros(n, d, dtype=torch.float64, device=dev)
        out2 = torch.zeros(n, d, dtype=torch.float64, device=dev)
        work = torch.empty(api.dev_count_twist_dense_workspace_bytes(tw, n), dtype=torch.uint8, device=dev)
        L = int(np.diff(go.astype(np.int64)).max())
        t1 = timed(lambda: api.dev_count_twist(tw, db.data_ptr(), dof.data_ptr(), n, db.numel(), L, out1.data_ptr(), stream=st.cuda_stream))
        t2 = timed(lambda: api.dev_count_twist_dense(tw, db.data_ptr(), dof.data_ptr(), n, work.data_ptr(), out2.data_ptr(), stream=st.cuda_stream))
        a, b = out1.cpu().numpy(), out2.cpu().numpy()
        rel = float(np.max(np.abs(a - b)) / max(np.max(np.abs(a)), 1e-300))
        rows = tw.info()["n_cols"]
        flops = 2.0 * n * rows * d
        print("%-44s sequences %5d -> twisted rows: sparse fused (count_twist_stream) %9.3f ms   dense image (count_dense + MFMA) %9.3f ms  (%5.2fx)  whole stage %5.1f TFLOP/s = %.2f of 78.6 (the contraction kernel alone: see the rocprofv3 stats below)  max rel diff %.1e"
              % (label + ", from the sequences", n, t1, t2, t1 / t2, flops / t2 / 1e9, flops / t2 / 1e9 / 78.6, rel), flush=True)

    n_g = int(os.environ.get("AB_GENOMES", "4096"))
    gb, go = O.synth_reads(0xC1A55, n_g, 30000)
    for k in [int(x) for x in os.environ.get("AB_KS", "7,8,9,10").split(",")]:
        tw = kpop_amd.Twister.synth(0x5EED, k, d)
        h, c, o = kpop_amd.count_reads(gb, go, k)
        case("genomes 30 kb, k=%d" % k, tw, h, c, o)
        if k <= 8:
            from_sequences("genomes 30 kb, k=%d" % k, tw, gb, go, k)
        tw.free()
    if os.environ.get("AB_ONLY_GENOMES"):
        return
    # class spectra against a trained-like twister
    k = 12
    cb, co = O.synth_reads(0xC1A55, 65, 30000)
    hm, cm, om = kpop_amd.count_reads(cb, co, k, per_read=False)
    rng = np.random.RandomState(1)
    tw = kpop_amd.Twister.load(rng.uniform(-1, 1, size=(d, len(hm))), hm, k)
    h, c, o = kpop_amd.count_reads(cb, co, k)
    case("65 class spectra, k=12, trained-like twister", tw, h, c, o)
    rb, ro = O.synth_reads(0x4B506F70, 2000, 150)
    h, c, o = kpop_amd.count_reads(rb, ro, k)
    case("2,000 reads of 150 bp, k=12, same twister", tw, h, c, o)
    tw.free()


if __name__ == "__main__":
    main()
