#!/usr/bin/env python3
"""Randomised batches through count -> twist against the oracle (a soak, not a test of the suite): mutants of one sequence,
unrelated sequences and short reads mixed, k 8..14, D 8..130, single / double strand, kpop_tune("dense", 0 | 2) for the
fused kernels (wave, streaming, tile) and the CSR twist of the counted spectra (twist_csr_kernel, few long spectra)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def concat(seqs):
    bases = np.frombuffer("".join(seqs).encode(), dtype=np.uint8).copy()
    offs = np.zeros(len(seqs) + 1, dtype=np.uint64)
    if seqs:
        offs[1:] = np.cumsum([len(s) for s in seqs])
    return bases, offs


def main():
    import kpop_amd as kpop
    from kpop_amd import api
    from oracle import oracle as O
    kpop.init(0)
    rng = np.random.RandomState(int(os.environ.get("SEED", "1")))
    for it in range(int(os.environ.get("N", "30"))):
        k = int(rng.randint(8, 16))
        d = int(rng.choice([8, 9, 16, 24, 33, 64, 65, 100, 130, 257]))
        ref_len = int(rng.randint(2000, 12000))
        ref = rng.choice(list("ACGT"), size=ref_len)
        rate = float(rng.choice([0.0, 0.0005, 0.002, 0.01, 0.03, 0.05]))
        content = int(rng.choice([0, 0, 1]))  # DNA-ds mostly, single strand now and then
        seqs = []
        for i in range(int(rng.choice([3, 15, 16, 17, 63, 64, 65, 129, int(rng.randint(70, 220))]))):
            m = ref.copy()
            if rng.rand() < 0.05:
                a = int(rng.randint(0, ref_len - 700))
                m[a:a + int(rng.randint(100, 700))] = "N"  # a masked stretch
            hit = rng.rand(len(m)) < rate
            m[hit] = rng.choice(list("ACGTN"), size=int(hit.sum()), p=[.24, .24, .24, .24, .04])
            if rng.rand() < 0.15:
                cut = int(rng.randint(50, ref_len - 50))
                m = np.concatenate([m[:cut], m[cut + int(rng.randint(1, 9)):]])
            seqs.append("".join(m[: len(m) - int(rng.randint(0, ref_len // 8))]))
        seqs += ["".join(rng.choice(list("ACGT"), size=int(rng.randint(600, 9000)))) for _ in range(int(rng.randint(0, 80)))]
        seqs += ["".join(rng.choice(list("ACGT"), size=int(rng.randint(1, 400)))) for _ in range(int(rng.randint(0, 60)))] + ["", "ACG"]
        if rng.rand() < 0.5:
            order = rng.permutation(len(seqs))
            seqs = [seqs[i] for i in order]
        bases, offs = concat(seqs)
        h, c, o = O.count_reads(bases, offs, k, content)
        cols = np.unique(h)
        if k <= 10 and rng.rand() < 0.5:
            cols = O.enumerate_kmers(k, content)
        else:
            cols = cols[rng.rand(len(cols)) < 0.9]
        T = O.synth_twister(3 + it, d, cols)
        # (up to 32 dimensions and k <= 13: every other time with the rows also kept at their hashes -- twister.h `direct`, forced)
        direct = d <= 32 and k <= 13 and it % 2 == 0
        api.tune("direct", 1 if direct else 2)
        tw = kpop.Twister.load(T, cols, k)
        api.tune("direct", 2)
        assert (tw.info()["direct_bytes"] > 0) == direct
        normalize = bool(rng.rand() < 0.5)
        want = O.twist(T, cols, h, c.astype(np.float64), o, normalize=normalize)
        scale = max(np.max(np.abs(want)), 1.0)
        tag = "it=%d k=%d d=%d n=%d rate=%g content=%d normalize=%s direct=%s" % (it, k, d, len(seqs), rate, content, normalize, direct)
        for mode in (0, 2, 2):
            api.tune("dense", mode)
            api.tune("tileg", 64 if it % 3 else 32)
            got = tw.count_twist(bases, offs, content=content, normalize=normalize)
            err = np.max(np.abs(got - want))
            assert err <= 1e-12 * scale, (tag, mode, err)
            if mode == 2:
                again = tw.count_twist(bases, offs, content=content, normalize=normalize)
                assert np.array_equal(got, again), (tag, "not the same bits twice")
        api.tune("tileg", 64)
        api.tune("dense", 0)
        got = tw.twist(h, c.astype(np.float64), o, normalize=normalize)  # the counted spectra through the CSR twist
        assert np.max(np.abs(got - want)) <= 1e-12 * scale, (tag, "csr", np.max(np.abs(got - want)))
        got = tw.twist(h, c.astype(np.uint32), o, normalize=normalize) if hasattr(c, "astype") else got
        assert np.max(np.abs(got - want)) <= 1e-12 * scale, (tag, "csr u32")
        if it % 5 == 0:
            print("ok", tag, flush=True)
    api.tune("dense", 2)
    print("all agree")


if __name__ == "__main__":
    main()
