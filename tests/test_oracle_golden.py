"""CPU tests: the oracle against the reference's known answers and the committed golden vectors."""
import hashlib

import numpy as np
import pytest

from conftest import GOLDEN, concat, load_golden, unhex


def test_readme_known_answer_summary(oracle):
    """README.md:649 (distance row) -> README.md:660 (summary row): the only numeric KAT on the path.
    The README prints %.15g, so inputs carry 15 digits: agreement is to ~1e-13, not bitwise."""
    kat = load_golden("readme_kat.json")
    row = [float(x) for x in kat["distance_row_text"]]
    names = ['"%s"' % n for n in kat["distance_header"]]
    st, idx, d, z = oracle.summarize_row(row, kat["keep_at_most"])
    want = kat["summary_line"].split("\t")
    assert want[0] == '"121"'
    np.testing.assert_allclose(st, [float(x) for x in want[1:5]], rtol=1e-12)
    assert [names[int(i)] for i in idx] == [want[5], want[8]]
    np.testing.assert_allclose(d, [float(want[6]), float(want[9])], rtol=0, atol=0)
    np.testing.assert_allclose(z, [float(want[7]), float(want[10])], rtol=1e-11)
    line = oracle.format_summary_line('"121"', st, names, idx, d, z)
    got = line.rstrip("\n").split("\t")
    for g, w in zip(got, want):  # text level: equal to 12 significant digits
        if g.startswith('"'):
            assert g == w
        else:
            assert abs(float(g) - float(w)) <= 1e-12 * max(1.0, abs(float(w)))


def test_readme_known_answer_pyref(pyref):
    kat = load_golden("readme_kat.json")
    row = [float(x) for x in kat["distance_row_text"]]
    st, neigh = pyref.summarize_row(row, 2)
    want = kat["summary_line"].split("\t")
    np.testing.assert_allclose(st, [float(x) for x in want[1:5]], rtol=1e-12)
    assert [kat["distance_header"][c] for c, _, _ in neigh] == ["2", "10"]


def test_readme_twisted_row_format():
    """README.md:624: %.15g round trip of a twisted row (format fixture)."""
    kat = load_golden("readme_kat.json")
    for txt in kat["twisted_row_text"] + kat["distance_row_text"]:
        assert "%.15g" % float(txt) == txt


def test_structural_pin_canonical_kmers(oracle):
    """README.md:106: 5127 lines = 10 headers + ~512 rows per spectrum at k=5 => DNA-ds keeps one key
    per k-mer / reverse-complement pair: 4^5/2 = 512."""
    assert len(oracle.enumerate_kmers(5, oracle.DNA_DS)) == 512
    assert len(oracle.enumerate_kmers(5, oracle.DNA_SS)) == 1024
    assert len(oracle.enumerate_kmers(6, oracle.DNA_DS)) == (4 ** 6 + 4 ** 3) // 2  # palindromes at even k


def test_count_golden(oracle):
    g = load_golden("count_small.json")
    seqs = [s for _, s in g["reads"]]
    bases, offs = concat(seqs)
    for case in g["cases"]:
        k = case["k"]
        content = oracle.DNA_DS if case["content"] == "DNA-ds" else oracle.DNA_SS
        h, c, o = oracle.count_reads(bases, offs, k, content, per_read=True)
        for r in range(len(seqs)):
            got = [[oracle.to_hex(a, k), int(b)] for a, b in zip(h[int(o[r]):int(o[r + 1])], c[int(o[r]):int(o[r + 1])])]
            assert got == case["spectra"][r], (g["reads"][r][0], k)
        hm, cm, om = oracle.count_reads(bases, offs, k, content, per_read=False)
        assert [[oracle.to_hex(a, k), int(b)] for a, b in zip(hm, cm)] == case["merged"]


def test_count_pyref_agrees(oracle, pyref):
    rng = np.random.RandomState(3)
    for k in (3, 7, 13, 18, 30):
        for _ in range(5):
            L = int(rng.randint(0, 90))
            s = "".join("ACGTN"[x] for x in rng.choice(5, size=L, p=[.24, .24, .24, .24, .04]))
            bases, offs = concat([s])
            for content, ds in ((oracle.DNA_DS, True), (oracle.DNA_SS, False)):
                h, c, _ = oracle.count_reads(bases, offs, k, content)
                assert {int(a): int(b) for a, b in zip(h, c)} == pyref.count_read(s, k, ds)


def test_wuhan_golden(oracle, pyref):
    g = load_golden("wuhan_counts.json")
    seq = "".join(l.strip() for l in open(GOLDEN + "/wuhan.fasta") if not l.startswith(">"))
    assert len(seq) == g["length"] == 29903
    bases, offs = concat([seq])
    for case in g["cases"]:
        k = case["k"]
        h, c, _ = oracle.count_reads(bases, offs, k, oracle.DNA_DS)
        assert len(h) == case["n_distinct"]
        assert int(c.sum()) == case["total"] == len(seq) - k + 1 - 0
        table = {int(a): int(b) for a, b in zip(h, c)}
        text = pyref.spectrum_text("MN908947.3", table, k)
        assert hashlib.sha256(text.encode()).hexdigest() == case["spectrum_text_sha256"]


def test_twist_golden(oracle):
    g = load_golden("twist_small.json")
    k, d = g["k"], g["n_dims"]
    cols = np.array(g["col_hash"], dtype=np.uint64)
    T = unhex(g["twister_dims_major"], (d, len(cols)))
    bases, offs = concat(g["reads"])
    h, c, o = oracle.count_reads(bases, offs, k, oracle.DNA_DS)
    for normalize in (True, False):
        tw = oracle.twist(T, cols, h, c.astype(np.float64), o, normalize)
        want = unhex(g["twisted_normalize_%s" % str(normalize).lower()], tw.shape)
        assert np.array_equal(tw, want)
    ds = g["dup_spectrum"]
    tw = oracle.twist(T, cols, np.array(ds["hash"], dtype=np.uint64), unhex(ds["value"]),
                      np.array([0, len(ds["hash"])], dtype=np.uint64), True)
    assert np.array_equal(tw, unhex(ds["twisted"], tw.shape))


def test_twist_properties(oracle):
    """Unknown k-mers are dropped AND excluded from the normaliser (lib/Twister.ml:158,167-169)."""
    k = 4
    cols = oracle.enumerate_kmers(k)[::2]
    T = oracle.synth_twister(5, 3, cols)
    h = np.array([cols[0], cols[1], 0xFFFF, cols[0]], dtype=np.uint64)  # 0xFFFF: not a 4-mer of the twister
    v = np.array([2.0, 1.0, 100.0, 1.0])
    o = np.array([0, 4], dtype=np.uint64)
    tw = oracle.twist(T, cols, h, v, o, True)
    want = T[:, 0] * (3.0 / 4.0) + T[:, 1] * (1.0 / 4.0)
    np.testing.assert_allclose(tw[0], want, rtol=1e-15)
    # acc = 0 -> no normalisation, empty spectrum -> zeros
    tw0 = oracle.twist(T, cols, np.array([0xFFFF], dtype=np.uint64), np.array([5.0]), np.array([0, 1], dtype=np.uint64))
    assert np.array_equal(tw0, np.zeros((1, 3)))


def test_distance_golden(oracle):
    g = load_golden("distance_small.json")
    d = g["n_dims"]
    m1 = unhex(g["m1"], (g["m1_rows"], d))
    m2 = unhex(g["m2"], (g["m2_rows"], d))
    inertia = unhex(g["inertia"])
    metric = oracle.metric_powers(inertia, 1.0, 1.0, 2.0)
    assert np.array_equal(metric, unhex(g["metric_powers_1_1_2"]))
    assert np.array_equal(oracle.metric_flat(d), unhex(g["metric_flat"]))
    for case in g["cases"]:
        dm = oracle.distance_rowwise(m1, m2, metric, case["kind"], case["p"], case["normalize"])
        assert np.array_equal(dm, unhex(case["dmatrix"], dm.shape)), case["distance"]
        for s in case["summaries"]:
            st, offs, idx, dist, z = oracle.distance_summary(m1, m2, metric, case["kind"], case["p"],
                                                            case["normalize"], s["keep_at_most"])
            assert np.array_equal(st, unhex(s["stats"], st.shape))
            assert offs.tolist() == s["offsets"] and idx.tolist() == s["idx"]
            assert np.array_equal(dist, unhex(s["dist"]))
            assert np.array_equal(z, unhex(s["z"]), equal_nan=True)


def test_metric_default_is_squared_inertia(oracle):
    """powers(1,1,2) (bin/KPopTwistDB.ml:92) under the declared semantics: m_d = w_d^2 / sum w^2."""
    w = oracle.synth_inertia(9)
    m = oracle.metric_powers(w, 1.0, 1.0, 2.0)
    np.testing.assert_allclose(m, w ** 2 / np.sum(w ** 2), rtol=1e-14)
    assert abs(m.sum() - 1.0) < 1e-14
    # threshold keeps the leading elements only
    m_half = oracle.metric_powers(w, 1.0, 0.5, 1.0)
    kept = np.nonzero(m_half)[0]
    assert kept.tolist() == list(range(len(kept))) and 0 < len(kept) < 9


def test_summary_ties_extend(oracle, pyref):
    """lib/Matrix.ml:648-649: a whole tie group is kept even past keep_at_most."""
    row = [2.0, 1.0, 1.0, 1.0, 3.0]
    st, idx, d, z = oracle.summarize_row(row, 2)
    assert idx.tolist() == [1, 2, 3] and d.tolist() == [1.0, 1.0, 1.0]
    ps, pn = pyref.summarize_row(row, 2)
    assert [c for c, _, _ in pn] == [1, 2, 3]
    assert st[2] == 1.0  # upper median: sorted[5//2]
    # sd = 0 -> z is nan/inf, unguarded as in the reference (:688)
    st, idx, d, z = oracle.summarize_row([1.0, 1.0], 1)
    assert st[1] == 0.0 and np.isnan(z).all()


def test_synth_generators(oracle):
    """SplitMix64 reference values (seed 0x4B506F70) and counter addressing."""
    L = oracle.lib()
    s = 0x4B506F70
    seq = [L.kpo_splitmix_at(s, i) for i in range(3)]
    state = s
    for i in range(3):  # the sequential form of SplitMix64
        state = (state + 0x9E3779B97F4A7C15) & (2 ** 64 - 1)
        assert L.kpo_mix64(state) == seq[i]
    bases, offs = oracle.synth_reads(s, 4, 10)
    assert bytes(bases[:10]).decode() == "".join("ACGT"[L.kpo_splitmix_at(s, i) >> 62] for i in range(10))
    w = oracle.synth_inertia(64)
    assert abs(w.sum() - 1) < 1e-15 and np.all(np.diff(w) < 0)  # strictly decreasing (lib/Space.ml:98-100)
    c = L.kpo_synth_twister_coeff(1, 3, 77)
    assert -1.0 <= c < 1.0


def test_counter_golden(oracle, pyref):
    """lib/KMerDB.ml statistics / transformations / class combination: C restatement against the committed fixture and
    the independent Python restatement."""
    g = load_golden("counter_small.json")
    cols = [np.array(v, dtype=np.int32) for v in g["columns"]]
    n_rows, n_cols = g["n_rows"], g["n_cols"]
    names = {"binary": 0, "power": 1, "clr": 2, "pseudocounts": 3}
    for st in g["stats"]:
        cs, rs = oracle.counter_stats(cols, st["threshold"], st["power"])
        assert np.array_equal(cs, unhex(st["col_stats"], (n_cols, 4)), equal_nan=True)
        assert np.array_equal(rs, unhex(st["row_stats"], (n_rows, 4)), equal_nan=True)
    for tr in g["transforms"][:8]:
        st = next(s for s in g["stats"] if s["threshold"] == tr["threshold"] and s["power"] == tr["power"])
        cs = unhex(st["col_stats"], (n_cols, 4))
        got = oracle.counter_transform(cols, cs, names[tr["which"]], tr["threshold"], tr["power"])
        assert np.array_equal(got, unhex(tr["table"], (n_rows, n_cols)), equal_nan=True)
    lin = unhex(g["stats"][0]["col_stats"], (n_cols, 4))[:, 2]
    for cb in g["combines"]:
        out, norm = oracle.counter_combine(cols, cb["sel"], lin, 0 if cb["criterion"] == "mean" else 1)
        assert out.tolist() == cb["out"] and norm == float.fromhex(cb["norm"])
        pout, pnorm = pyref.counter_combine([[int(x) for x in v] for v in cols], cb["sel"], list(lin), cb["criterion"])
        assert pout == cb["out"] and pnorm == norm
    # wrap-around of Int32.of_float is exercised (a rescaled sum beyond 2^31)
    assert any(x < 0 for cb in g["combines"] for x in cb["out"])


def test_protein_count_golden(oracle, pyref):
    g = load_golden("count_protein.json")
    seqs = [s for _, s in g["reads"]]
    bases = np.frombuffer("".join(seqs).encode(), dtype=np.uint8)
    offs = np.concatenate([[0], np.cumsum([len(s) for s in seqs])]).astype(np.uint64)
    for case in g["cases"]:
        k = case["k"]
        h, c, o = oracle.count_reads(bases, offs, k, oracle.PROTEIN)
        for r, s in enumerate(seqs):
            got = [[oracle.to_hex(a, k, oracle.PROTEIN), int(b)] for a, b in zip(h[int(o[r]):int(o[r + 1])], c[int(o[r]):int(o[r + 1])])]
            assert got == case["spectra"][r]
            assert {pyref.to_hex_protein(a, k): b for a, b in pyref.count_read_protein(s, k).items()} == dict(map(tuple, case["spectra"][r]))
    assert g["cases"][-1]["k"] == 12 and len(g["cases"][-1]["merged"][0][0]) == 15    # 60 bits = 15 hex digits


def test_reference_distance_iterator_known_answer(oracle, pyref):
    """The reference's own executable test (test/DistanceIterator.ml -> test/DistanceIterator.txt): every pair of the 12
    points whose Minkowski(1) component is <= 0.3, in order of increasing component, printed %.15g.  The oracle's
    distance between the 1-dimensional vectors [a_i] and [a_j] must print the same digits, list the same pairs and
    respect the same order."""
    g = load_golden("distance_iterator.json")
    pts = np.array(g["points"], dtype=np.float64).reshape(-1, 1)
    d = oracle.distance_rowwise(pts, pts, np.array([g["metric_weight"]]), oracle.MINKOWSKI, 1.0, normalize=False)
    for i, j, text in g["pairs"]:
        assert "%.15g" % d[j, i] == text and "%.15g" % d[i, j] == text
        assert "%.15g" % pyref.distance("minkowski", 1.0, [g["metric_weight"]], [pts[i, 0]], 1.0, [pts[j, 0]], 1.0) == text
    n = len(pts)
    want = {(i, j) for i in range(n) for j in range(i + 1, n) if d[j, i] <= g["max_distance_component"]}
    assert {(min(i, j), max(i, j)) for i, j, _ in g["pairs"]} == want and len(g["pairs"]) == len(want)
    vals = [d[j, i] for i, j, _ in g["pairs"]]
    assert vals == sorted(vals)


def test_splits_golden(pyref):
    """the splits fixture is what the restatement of lib/Matrix.ml:524-612 gives today (regeneration would show a drift)"""
    g = load_golden("splits_small.json")
    for c in g["cases"]:
        emb = [[float.fromhex(x) for x in row] for row in c["emb"]]
        assert [[a.hex(), m] for a, m in pyref.splits_gaps(emb, c["keep"])] == c["gaps"]
        assert [[a.hex(), m] for a, m in pyref.splits_centroids(emb)] == c["centroids"]
        # structure: the largest gap first; a centroids run ends in one singleton split per leaf
        assert all(float.fromhex(x[0]) >= float.fromhex(y[0]) for x, y in zip(c["gaps"], c["gaps"][1:]))
        assert {m[0] for w, m in c["centroids"] if len(m) == 1 and float.fromhex(w) == 0.0} == set(range(len(emb)))
