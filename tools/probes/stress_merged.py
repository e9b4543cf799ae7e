#!/usr/bin/env python3
"""Randomised batches through the merged (-l) count against the oracle (a soak, not a test of the suite): reads, genomes, both
mixed, Ns, empty and tiny sequences, DNA single / double strand and protein, k over the histogram's whole range, every
kpop_tune("histlds") mode (0 direct atomics, 1 the default choice, 2 combine, 3 partition) and the sort path."""
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
    for it in range(int(os.environ.get("N", "40"))):
        protein = rng.rand() < 0.2
        if protein:
            k, content, alpha, p = int(rng.randint(1, 6)), kpop.PROTEIN, list("ACDEFGHIKLMNPQRSTVWYX"), [0.0495] * 20 + [0.01]
        else:
            k, content, alpha, p = int(rng.randint(5, 14)), int(rng.choice([kpop.DNA_DS, kpop.DNA_SS])), list("ACGTN"), [0.2475] * 4 + [0.01]
        kind = rng.randint(0, 4)
        seqs = []
        if kind in (0, 2):  # reads
            seqs += ["".join(rng.choice(alpha, size=int(n), p=p)) for n in rng.randint(0, 400, size=int(rng.randint(10, 9000)))]
        if kind in (1, 2):  # long sequences, unrelated
            seqs += ["".join(rng.choice(alpha, size=int(n), p=p)) for n in rng.randint(3000, 60000, size=int(rng.randint(1, 40)))]
        if kind == 3:  # assemblies of one organism
            ref = rng.choice(alpha[:-1], size=int(rng.randint(5000, 20000)))
            for _ in range(int(rng.randint(16, 90))):
                m = ref.copy()
                hit = rng.rand(len(m)) < 0.003
                m[hit] = rng.choice(alpha, size=int(hit.sum()), p=p)
                seqs.append("".join(m))
        seqs += ["", alpha[0] * 3]
        if rng.rand() < 0.5:
            seqs = [seqs[i] for i in rng.permutation(len(seqs))]
        bases, offs = concat(seqs)
        want = O.count_reads(bases, offs, k, content, per_read=False)
        tag = "it=%d k=%d content=%d kind=%d n=%d bases=%d" % (it, k, content, kind, len(seqs), len(bases))
        for hist, lds in ((1, 0), (1, 1), (1, 2), (1, 3), (0, 1)):
            api.tune("hist", hist)
            api.tune("histlds", lds)
            got = kpop.count_reads(bases, offs, k, content, per_read=False)
            assert all(np.array_equal(a, b) for a, b in zip(got, want)), (tag, hist, lds)
        if it % 5 == 0:
            print("ok", tag, "distinct", len(want[0]), flush=True)
    api.tune("hist", 1)
    api.tune("histlds", 1)
    print("all agree")


if __name__ == "__main__":
    main()
