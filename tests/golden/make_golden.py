#!/usr/bin/env python3
"""Generates the committed golden fixtures under tests/golden/.

Run from the repo root:  python tests/golden/make_golden.py

Sources of truth
  * readme_kat.json   -- copied DATA from the reference's documentation
                         (README.md:620-624 twisted row, :645-649 distance row,
                         :660 summary row of sample "121"): the only numeric
                         known-answer test the reference holds for this path.
  * wuhan.fasta       -- the reference's own test data file (test/wuhan.fasta,
                         MN908947.3, 29,903 bp), copied verbatim as INPUT data.
  * everything else   -- small inputs pushed through BOTH restatements
                         (oracle/kpop_oracle.c and oracle/pyref.py); the script
                         refuses to write a fixture the two disagree on.  The
                         reference itself cannot run here (OCaml, and its
                         BiOCamLib submodule is empty), so these vectors pin the
                         restatements against each other and against later
                         regressions, not against an OCaml run.
Floats are stored as C99 hex strings so they round-trip bit for bit.
"""
import hashlib
import json
import os
import shutil
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from oracle import pyref as P  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference"


def fh(x):
    return float(x).hex()


def fhl(a):
    return [fh(x) for x in np.asarray(a, dtype=np.float64).ravel()]


def dump(name, obj):
    with open(os.path.join(OUT, name), "w") as f:
        json.dump(obj, f, sort_keys=True, separators=(",", ":"))
        f.write("\n")
    print("wrote", name)


# ---------------------------------------------------------------- README KAT
def readme_kat():
    kat = {
        "source": "README.md:620-624,645-649,660 of PaoloRibeca/KPop (documentation data)",
        "twisted_header": ["Dim1", "Dim2", "Dim3", "Dim4", "Dim5", "Dim6", "Dim7", "Dim8", "Dim9"],
        "twisted_row_label": "121",
        "twisted_row_text": ["0.461489766036905", "0.568113255163704", "-0.882699288338637", "-0.0272667586657692",
                             "0.0581302393092316", "-0.0189776114363531", "0.00161058296863769",
                             "0.0225299502172142", "-0.0278201311947722"],
        "distance_header": ["10", "1", "2", "3", "4", "5", "6", "7", "8", "9"],
        "distance_row_text": ["3.298234166382", "3.33901585931896", "1.85388698190156", "3.35586331250312",
                              "3.35677089495605", "3.35069768834933", "3.31043491848486", "3.30792669036762",
                              "3.32894004017435", "3.31694544810727"],
        "keep_at_most": 2,
        "summary_line": "\"121\"\t3.18187160005451\t0.467075454492746\t3.32894004017435\t0.0217576481749768\t"
                        "\"2\"\t1.85388698190156\t-2.84319076367473\t\"10\"\t3.298234166382\t0.249130124925674",
    }
    dump("readme_kat.json", kat)


# ---------------------------------------------------------------- count
EDGE_READS = [
    ("empty", ""),
    ("shorter_than_k", "ACG"),
    ("exactly_k5", "ACGTA"),
    ("homopolymer", "AAAAAAAAAAAAAAAAAAAA"),
    ("palindrome", "ACGTACGTACGTACGTACGT"),
    ("with_N", "ACGTACGTNACGTTTGACCANNACGTACGTAGCATCGACTAGCTAGCATCGACTACG"),
    ("lowercase", "acgtacgttgcaacgttgcatgca"),
    ("mixed_case_iupac", "ACGTRYACGTTGCAkmACGTTGCATGCAGGCTA"),
    ("dash_and_star", "ACGTAC-GTACGT*ACGTACGGTCA"),
    ("all_N", "NNNNNNNNNNNNNNNN"),
    ("poly_T", "TTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTT"),
    ("repeat", "ACACACACACACACACACACACACACACACACACAC"),
]


def concat(seqs):
    bases = np.frombuffer("".join(seqs).encode(), dtype=np.uint8).copy()
    offs = np.zeros(len(seqs) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(s) for s in seqs])
    return bases, offs


def count_fixture():
    rng = np.random.RandomState(20250303)
    reads = list(EDGE_READS)
    for i in range(8):
        L = int(rng.randint(20, 200))
        s = "".join("ACGT"[x] for x in rng.randint(0, 4, size=L))
        reads.append(("rand%d" % i, s))
    seqs = [s for _, s in reads]
    bases, offs = concat(seqs)
    cases = []
    for k in (1, 2, 5, 8, 12, 15, 16, 17, 21, 30):
        for content, ds in ((O.DNA_DS, True), (O.DNA_SS, False)):
            h, c, o = O.count_reads(bases, offs, k, content, per_read=True)
            spectra = []
            for r, s in enumerate(seqs):
                got = {int(a): int(b) for a, b in zip(h[int(o[r]):int(o[r + 1])], c[int(o[r]):int(o[r + 1])])}
                want = P.count_read(s, k, ds)
                if got != want:
                    raise SystemExit("count: restatements disagree on read %s k=%d" % (reads[r][0], k))
                spectra.append([[O.to_hex(a, k), int(got[a])] for a in sorted(got)])
                for a in got:
                    assert O.to_hex(a, k) == P.to_hex(a, k)
            hm, cm, om = O.count_reads(bases, offs, k, content, per_read=False)
            merged = {}
            for s in seqs:
                for a, b in P.count_read(s, k, ds).items():
                    merged[a] = merged.get(a, 0) + b
            if {int(a): int(b) for a, b in zip(hm, cm)} != merged:
                raise SystemExit("count: merged spectra disagree k=%d" % k)
            cases.append({"k": k, "content": "DNA-ds" if ds else "DNA-ss", "spectra": spectra,
                          "merged": [[O.to_hex(a, k), int(merged[a])] for a in sorted(merged)]})
    dump("count_small.json", {"reads": [[n, s] for n, s in reads], "cases": cases})


def protein_fixture():
    """KPopCount -C protein under the encoding declared in kpop_amd/csrc/kmer.h (5 bits per residue, the 20 standard
    amino acids in alphabetical order; anything else breaks the window): both restatements, k on both sides of the
    32/64-bit key boundary (5k <= 30 bits up to k = 6) and at the maximum of 12."""
    rng = np.random.RandomState(12)
    aa = "ACDEFGHIKLMNPQRSTVWY"
    seqs = [("empty", ""), ("short", "MK"), ("with_x", "MKTAYIAKQRXQISFVKSHFSRQ*LEERLG"), ("lower", "mktayiakqrqisf"),
            ("repeat", "GAGAGAGAGAGAGAGAGAGAGAGA"), ("ambiguous", "BZJOUX")]
    for i in range(4):
        seqs.append(("rand%d" % i, "".join(aa[x] for x in rng.randint(0, 20, size=int(rng.randint(15, 120))))))
    bases, offs = concat([s for _, s in seqs])
    cases = []
    for k in (1, 2, 3, 6, 7, 12):
        h, c, o = O.count_reads(bases, offs, k, O.PROTEIN, per_read=True)
        spectra, merged = [], {}
        for r, (_, s) in enumerate(seqs):
            got = {int(a): int(b) for a, b in zip(h[int(o[r]):int(o[r + 1])], c[int(o[r]):int(o[r + 1])])}
            want = P.count_read_protein(s, k)
            if got != want:
                raise SystemExit("protein count: restatements disagree on %s k=%d" % (seqs[r][0], k))
            for a, b in want.items():
                merged[a] = merged.get(a, 0) + b
                assert O.to_hex(a, k, O.PROTEIN) == P.to_hex_protein(a, k)
            spectra.append([[P.to_hex_protein(a, k), int(got[a])] for a in sorted(got)])
        hm, cm, om = O.count_reads(bases, offs, k, O.PROTEIN, per_read=False)
        if {int(a): int(b) for a, b in zip(hm, cm)} != merged:
            raise SystemExit("protein count: merged spectra disagree k=%d" % k)
        cases.append({"k": k, "spectra": spectra, "merged": [[P.to_hex_protein(a, k), int(merged[a])] for a in sorted(merged)]})
    dump("count_protein.json", {"reads": [[n, s] for n, s in seqs], "cases": cases})


# ---------------------------------------------------------------- twist
def twist_fixture():
    rng = np.random.RandomState(7)
    k = 5
    all_k = O.enumerate_kmers(k, O.DNA_DS)
    # twister knows only 2/3 of the k-mers, in a shuffled column order
    cols = all_k[rng.permutation(len(all_k))[: (2 * len(all_k)) // 3]]
    n_dims = 7
    T = rng.uniform(-1, 1, size=(n_dims, len(cols)))
    seqs = [s for _, s in EDGE_READS] + ["".join("ACGT"[x] for x in rng.randint(0, 4, size=120)) for _ in range(6)]
    bases, offs = concat(seqs)
    h, c, o = O.count_reads(bases, offs, k, O.DNA_DS, per_read=True)
    v = c.astype(np.float64)
    out = {}
    for normalize in (True, False):
        tw = O.twist(T, cols, h, v, o, normalize)
        col_names = [P.to_hex(int(x), k) for x in cols]
        for r in range(len(seqs)):
            lines = [(P.to_hex(int(a), k), "%d" % b) for a, b in zip(h[int(o[r]):int(o[r + 1])], c[int(o[r]):int(o[r + 1])])]
            want = P.twist(T.tolist(), col_names, lines, normalize)
            if not np.array_equal(np.asarray(want), tw[r]):
                raise SystemExit("twist: restatements disagree on read %d" % r)
        out["twisted_normalize_%s" % str(normalize).lower()] = fhl(tw)
    # a spectrum with duplicate and unknown lines and non-integer values (text spectra, lib/Twister.ml:155,160-163)
    dup_h = np.array([cols[3], cols[5], cols[3], all_k[0] if all_k[0] not in set(cols.tolist()) else cols[1], cols[8],
                      cols[5]], dtype=np.uint64)
    dup_v = np.array([1.5, 2.25, 0.125, 9.0, 3.0, 1e-3])
    dup_o = np.array([0, len(dup_h)], dtype=np.uint64)
    twd = O.twist(T, cols, dup_h, dup_v, dup_o, True)
    want = P.twist(T.tolist(), [P.to_hex(int(x), k) for x in cols],
                   [(P.to_hex(int(a), k), repr(float(b))) for a, b in zip(dup_h, dup_v)], True)
    if not np.array_equal(np.asarray(want), twd[0]):
        raise SystemExit("twist: restatements disagree on the duplicate-line spectrum")
    dump("twist_small.json", {
        "k": k, "n_dims": n_dims, "col_hash": [int(x) for x in cols], "twister_dims_major": fhl(T),
        "reads": seqs, **out,
        "dup_spectrum": {"hash": [int(x) for x in dup_h], "value": fhl(dup_v), "twisted": fhl(twd)},
    })


# ---------------------------------------------------------------- distance
def distance_fixture():
    rng = np.random.RandomState(11)
    d = 6
    m1 = rng.normal(size=(5, d))
    m2 = rng.normal(size=(7, d))
    m2[2] = 0.0          # zero row: norm 0 -> 1 (lib/Matrix.ml:67)
    m2[4] = m1[1]        # identical point: distance 0
    m2[5] = m2[3]        # tie
    inertia = O.synth_inertia(d)
    metric = O.metric_powers(inertia, 1.0, 1.0, 2.0)
    if not np.array_equal(np.asarray(P.metric_powers(inertia.tolist())), metric):
        raise SystemExit("metric: restatements disagree")
    cases = []
    for name, kind, p in (("euclidean", O.EUCLIDEAN, 2.0), ("cosine", O.COSINE, 2.0), ("minkowski(1)", O.MINKOWSKI, 1.0),
                          ("minkowski(3)", O.MINKOWSKI, 3.0)):
        for normalize in (True, False):
            dm = O.distance_rowwise(m1, m2, metric, kind, p, normalize)
            want = P.distance_rowwise(m1.tolist(), m2.tolist(), metric.tolist(), name.split("(")[0], p, normalize)
            if not np.array_equal(np.asarray(want), dm):
                raise SystemExit("distance: restatements disagree for %s" % name)
            summ = []
            for keep in (1, 2, 0):
                st, offs, idx, dist, z = O.distance_summary(m1, m2, metric, kind, p, normalize, keep)
                for j in range(m2.shape[0]):
                    ps, pn = P.summarize_row(dm[j].tolist(), keep if keep else m1.shape[0])
                    a, b = int(offs[j]), int(offs[j + 1])
                    if list(ps) != list(st[j]) or [x[0] for x in pn] != idx[a:b].tolist() \
                            or [x[1] for x in pn] != dist[a:b].tolist():
                        raise SystemExit("summary: restatements disagree for %s row %d" % (name, j))
                summ.append({"keep_at_most": keep, "stats": fhl(st), "offsets": [int(x) for x in offs],
                             "idx": [int(x) for x in idx], "dist": fhl(dist), "z": fhl(z)})
            cases.append({"distance": name, "kind": kind, "p": p, "normalize": normalize, "dmatrix": fhl(dm),
                          "summaries": summ})
    dump("distance_small.json", {"n_dims": d, "m1": fhl(m1), "m1_rows": 5, "m2": fhl(m2), "m2_rows": 7,
                                 "inertia": fhl(inertia), "metric_powers_1_1_2": fhl(metric),
                                 "metric_flat": fhl(O.metric_flat(d)), "cases": cases})


# ---------------------------------------------------------------- wuhan
def wuhan_fixture():
    src = os.path.join(REF, "test", "wuhan.fasta")
    dst = os.path.join(OUT, "wuhan.fasta")
    if os.path.exists(src):
        shutil.copyfile(src, dst)
    seq = "".join(l.strip() for l in open(dst) if not l.startswith(">"))
    bases, offs = concat([seq])
    res = {"length": len(seq), "cases": []}
    for k in (5, 10, 12):
        h, c, o = O.count_reads(bases, offs, k, O.DNA_DS, per_read=True)
        want = P.count_read(seq, k, True)
        if {int(a): int(b) for a, b in zip(h, c)} != want:
            raise SystemExit("wuhan: restatements disagree at k=%d" % k)
        text = P.spectrum_text("MN908947.3", want, k)
        top = sorted(want.items(), key=lambda t: (-t[1], t[0]))[:5]
        res["cases"].append({"k": k, "n_distinct": len(want), "total": int(sum(want.values())),
                             "spectrum_text_sha256": hashlib.sha256(text.encode()).hexdigest(),
                             "top5": [[P.to_hex(a, k), b] for a, b in top]})
    dump("wuhan_counts.json", res)


# ---------------------------------------------------------------- k-mer database
def counter_fixture():
    """lib/KMerDB.ml statistics, transformations and class combination on a small database with the edge cases the
    code paths branch on: an all-zero spectrum (norm 0, skipped by :693), a k-mer absent everywhere, counts below
    absolute and relative thresholds, an even and an odd number of combined spectra."""
    rng = np.random.default_rng(0x4B506F70)
    n_rows, n_cols = 37, 7
    cols = [rng.poisson(lam, n_rows).astype(np.int32) for lam in (0.7, 3.0, 12.0, 40.0, 1.5, 0.2)]
    cols.insert(3, np.zeros(n_rows, dtype=np.int32))      # spectrum 3: empty
    for v in cols:
        v[11] = 0                                          # k-mer 11 occurs nowhere
    cols[2][5] = 2_000_000_000                             # large count: rescaled sums overflow int32 and wrap
    names = {0: "binary", 1: "power", 2: "clr", 3: "pseudocounts"}
    res = {"n_rows": n_rows, "n_cols": n_cols, "columns": [[int(x) for x in v] for v in cols], "stats": [],
           "transforms": [], "combines": []}
    for thr, pw in ((1.0, 1.0), (3.0, 1.0), (0.01, 1.0), (1.0, 0.5), (0.0, 2.0)):
        cs, rs = O.counter_stats(cols, thr, pw)
        for c in range(n_cols):
            want = P.counter_vector_stats([int(x) for x in cols[c]], thr, pw)
            if not np.array_equal(np.array(want), cs[c], equal_nan=True):
                raise SystemExit("counter stats: restatements disagree (col %d, thr %g, pow %g)" % (c, thr, pw))
        for r in range(n_rows):
            want = P.counter_vector_stats([int(v[r]) for v in cols], thr, pw)
            if not np.array_equal(np.array(want), rs[r], equal_nan=True):
                raise SystemExit("counter stats: restatements disagree (row %d)" % r)
        res["stats"].append({"threshold": thr, "power": pw, "col_stats": fhl(cs), "row_stats": fhl(rs)})
        for which in range(4):
            if which == 2 and thr == 0.0:
                continue  # sum_log = -inf: every entry is nan/inf, nothing to pin
            t = O.counter_transform(cols, cs, which, thr, pw, kmer_major=True)
            for c in range(n_cols):
                for r in range(n_rows):
                    w = P.counter_transform_one(names[which], thr, pw, tuple(cs[c]), int(cols[c][r]))
                    if not (w == t[r, c] or (w != w and t[r, c] != t[r, c])):
                        raise SystemExit("counter transform: restatements disagree (%s, %d, %d)" % (names[which], r, c))
            res["transforms"].append({"which": names[which], "threshold": thr, "power": pw, "table": fhl(t)})
    lin, _ = O.counter_stats(cols, 1.0, 1.0)
    col_sum = lin[:, 2].copy()
    for sel in ([0, 1, 2], [6, 5, 4, 3, 2, 1, 0], [3], [1, 3, 4, 5], [2, 2], []):
        for crit, cname in ((0, "mean"), (1, "median")):
            out, norm = O.counter_combine(cols, sel, col_sum, crit)
            wout, wnorm = P.counter_combine([[int(x) for x in v] for v in cols], sel, list(col_sum), cname)
            if [int(x) for x in out] != wout or norm != wnorm:
                raise SystemExit("counter combine: restatements disagree (%s, %s)" % (sel, cname))
            res["combines"].append({"sel": sel, "criterion": cname, "out": [int(x) for x in out], "norm": fh(norm)})
    dump("counter_small.json", res)


# ---------------------------------------------------------------- the reference's own Distance.Iterator test
def distance_iterator_fixture():
    """test/DistanceIterator.ml + its expected output test/DistanceIterator.txt: the one executable test the reference
    ships.  Distance.Iterator (lib/Space.ml:231-420, used by no CLI) walks the pairs of a 12-point line in order of
    increasing Minkowski(1) component up to 0.3.  The DATA kept here: the input points (test/DistanceIterator.ml:7),
    the bound (:9) and the "(i, j): component" lines of the expected output -- a known answer for the component
    arithmetic of lib/Space.ml:150-158,192-200 and for the order of the pairs."""
    import re
    src = os.path.join(REF, "test", "DistanceIterator.txt")
    dst = os.path.join(OUT, "distance_iterator.json")
    if not os.path.exists(src):
        print("kept", os.path.basename(dst))
        return
    pairs = []
    for line in open(src):
        m = re.match(r"\((\d+), (\d+)\): (\S+)$", line.strip())
        if m:
            pairs.append([int(m.group(1)), int(m.group(2)), m.group(3)])
    dump("distance_iterator.json", {
        "source": "test/DistanceIterator.ml:5-9 (inputs) and test/DistanceIterator.txt (expected output) of PaoloRibeca/KPop",
        "distance": "minkowski(1)", "metric_weight": 1.0, "max_distance_component": 0.3,
        "points": [0.1, 0.1, 0.2, 0.2, 0.2, 0.7, 0.5, 0.99, 0.999, 0.05, 0.4, 0.05],
        "pairs": pairs})


if __name__ == "__main__":
    distance_iterator_fixture()
    counter_fixture()
    readme_kat()
    count_fixture()
    protein_fixture()
    twist_fixture()
    distance_fixture()
    wuhan_fixture()
