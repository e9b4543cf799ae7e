"""GPU tests of the drop-in CLIs: README-style shell pipelines (README.md:91-94,606,641,656) through
kpop_amd/bin/KPopCount and KPopTwistDB, compared byte for byte with text built from the oracle."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from test_cli import BIN, COUNT, TWISTDB, run, write_table

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not os.path.exists(TWISTDB), reason="host CLIs not built")]

READS = [
    ("r1 first read", "ACGTACGTTGCAACGTTGCATGCANNACGTACGTAGCATCGACTAGC"),
    ("r2", "acgtacgttgcaacgttgcatgca"),
    ("r3-dashes", "ACGTAC-GTACGT--ACGTACGGTCAGGATC"),
    ("r4|short", "ACG"),
    ("r5", "TTTTTTTTTTTTTTTTTTTTTTTTTTTTTT"),
    ("r6 multi", "ACGATCGATCGATCGATTAGCTAGCTAGCTAGCATCGATCGATGCATGCATGCATGCATGCATCGATCGATCGATCGTAGCTAGCTAGCTAGC"),
]


def lint(seq):
    return seq.upper().replace("-", "")


def write_fasta(path, reads, width=17):
    with open(path, "w") as f:
        for tag, seq in reads:
            f.write(">%s\n" % tag)
            for i in range(0, len(seq), width):
                f.write(seq[i:i + width] + "\n")


def write_fastq(path, reads):
    with open(path, "w") as f:
        for tag, seq in reads:
            f.write("@%s\n%s\n+\n%s\n" % (tag, seq, "I" * len(seq)))


def expected_spectra(pyref, reads, k, ds=True):
    return "".join(pyref.spectrum_text(tag, pyref.count_read(lint(seq), k, ds), k) for tag, seq in reads)


def test_kpopcount_per_sequence_fasta(tmp_path, pyref):
    fa = tmp_path / "x.fa"
    write_fasta(fa, READS)
    for k in (3, 5, 12):
        r = run([COUNT, "-k", str(k), "-L", "-f", str(fa)])
        assert r.returncode == 0, r.stderr
        assert r.stdout == expected_spectra(pyref, READS, k)
    r = run([COUNT, "-k", "5", "-L", "-C", "DNA-ss", "-f", str(fa), "-o", str(tmp_path / "out")])
    assert r.returncode == 0 and r.stdout == ""
    assert (tmp_path / "out.KPopSpectra.txt").read_text() == expected_spectra(pyref, READS, 5, ds=False)


def test_kpopcount_label_mode_fastq_and_paired(tmp_path, pyref):
    fq1, fq2 = tmp_path / "a_1.fastq", tmp_path / "a_2.fastq"
    write_fastq(fq1, READS[:3])
    write_fastq(fq2, READS[3:])
    k = 4
    merged = {}
    for _, seq in READS:
        for h, c in pyref.count_read(lint(seq), k).items():
            merged[h] = merged.get(h, 0) + c
    want = pyref.spectrum_text("sample A", merged, k)
    r = run([COUNT, "-k", str(k), "-l", "sample A", "-p", str(fq1), str(fq2)])
    assert r.returncode == 0, r.stderr
    assert r.stdout == want
    r = run([COUNT, "-k", str(k), "-l", '"sample A"', "-s", str(fq1), "-s", str(fq2)])  # quotes stripped (:169)
    assert r.stdout == want
    # -L with mates: every mate is its own spectrum, in segment order (bin/KPopCount.ml:39-50)
    r = run([COUNT, "-k", str(k), "-L", "-p", str(fq1), str(fq2)])
    inter = [READS[0], READS[3], READS[1], READS[4], READS[2], READS[5]]
    assert r.stdout == expected_spectra(pyref, inter, k)


def make_twister(tmp_path, oracle, k, d, seed=3, keep=0.8):
    rng = np.random.RandomState(seed)
    cols = oracle.enumerate_kmers(k)
    cols = cols[rng.rand(len(cols)) < keep]
    T = oracle.synth_twister(seed, d, cols)
    # what the tool sees is the %.15g text
    T = np.array([[float("%.15g" % x) for x in row] for row in T])
    dims = ["Dim%d" % (i + 1) for i in range(d)]
    w = np.array([float("%.15g" % x) for x in oracle.synth_inertia(d)])
    write_table(tmp_path / "Classes.KPopTwister.txt", [oracle.to_hex(h, k) for h in cols], dims, T)
    write_table(tmp_path / "Classes.KPopInertia.txt", dims, ["inertia"], [w])
    return cols, T, dims, w


def twisted_text(dims, labels, rows):
    order = sorted(range(len(labels)), key=lambda i: labels[i].encode())
    s = '""' + "".join('\t"%s"' % c for c in dims) + "\n"
    for i in order:
        s += '"%s"' % labels[i] + "".join("\t%.15g" % v for v in rows[i]) + "\n"
    return s


def test_summary_keep_at_most_all_against_a_large_reference_set(tmp_path, oracle):
    """KPopTwistDB -I d .. --summary-keep-at-most all -S ..: 5,000 reference columns (more than the summary kernels list by
    themselves), every one of them listed per row, in (distance, column) order -- lib/Matrix.ml:723-726,641-650"""
    rng = np.random.RandomState(5)
    r1, r2 = 5000, 3
    dm = np.round(rng.rand(r2, r1) * 10, 2)  # two decimals: ties by the dozen
    cols, rows = ["c%d" % i for i in range(r1)], ["q%d" % j for j in range(r2)]
    text = '""' + "".join('\t"%s"' % c for c in cols) + "\n"
    for j in range(r2):
        text += '"%s"' % rows[j] + "".join("\t%.15g" % v for v in dm[j]) + "\n"
    (tmp_path / "D.KPopDMatrix.txt").write_text(text)
    r = run([TWISTDB, "-I", "d", str(tmp_path / "D"), "--summary-keep-at-most", "all", "-S", str(tmp_path / "All")])
    assert r.returncode == 0, r.stderr
    got = (tmp_path / "All.KPopSummary.txt").read_text().splitlines()
    assert len(got) == r2
    for j in range(r2):
        st, idx, dist, z = oracle.summarize_row(dm[j], r1)
        assert len(idx) == r1
        g, w = got[j].split("\t"), oracle.format_summary_line(rows[j], st, cols, idx, dist, z).rstrip("\n").split("\t")
        assert len(g) == len(w) == 5 + 3 * r1 and g[0] == w[0]
        # (against more than 4,096 columns the statistics are summed in another order than the reference's: 1e-10, and the z-scores with them)
        np.testing.assert_allclose([float(v) for v in g[1:5]], [float(v) for v in w[1:5]], rtol=1e-10)
        assert g[5::3] == w[5::3] and g[6::3] == w[6::3]  # names and distances, to the character
        np.testing.assert_allclose([float(v) for v in g[7::3]], [float(v) for v in w[7::3]], rtol=1e-8, atol=1e-9)


def test_count_twist_distance_summary_pipeline(tmp_path, oracle, pyref):
    """KPopCount -L | KPopTwistDB -I T .. -k /dev/stdin -O t ..; then -d / -s / -S against the class vectors."""
    k, d = 5, 6
    cols, T, dims, w = make_twister(tmp_path, oracle, k, d)
    fa, cfa = tmp_path / "test.fa", tmp_path / "classes.fa"
    rng = np.random.RandomState(9)
    test_reads = READS + [("z%d" % i, "".join(rng.choice(list("ACGT"), size=80))) for i in range(6)]
    class_reads = [("c%d" % i, "".join(rng.choice(list("ACGT"), size=400))) for i in range(5)]
    write_fasta(fa, test_reads)
    write_fasta(cfa, class_reads)

    def twist_cli(fasta, prefix):
        p1 = subprocess.Popen([COUNT, "-k", str(k), "-L", "-f", str(fasta)], stdout=subprocess.PIPE)
        p2 = subprocess.run([TWISTDB, "-I", "T", str(tmp_path / "Classes"), "-k", "/dev/stdin", "-O", "t", str(tmp_path / prefix)],
                            stdin=p1.stdout, capture_output=True, text=True, timeout=120)
        p1.wait()
        assert p1.returncode == 0 and p2.returncode == 0, p2.stderr

    def twist_oracle(reads):
        rows = []
        for tag, seq in reads:
            table = pyref.count_read(lint(seq), k)
            h = np.array(sorted(table), dtype=np.uint64)
            v = np.array([table[int(x)] for x in h], dtype=np.float64)
            rows.append(oracle.twist(T, cols, h, v, np.array([0, len(h)], dtype=np.uint64))[0])
        return np.array(rows)

    twist_cli(fa, "Test")
    twist_cli(cfa, "ClassVecs")
    t_test, t_cls = twist_oracle(test_reads), twist_oracle(class_reads)
    got = (tmp_path / "Test.KPopTwisted.txt").read_text()
    want = twisted_text(dims, [t for t, _ in test_reads], t_test)
    assert got.splitlines()[0] == want.splitlines()[0]
    for g, x in zip(got.splitlines()[1:], want.splitlines()[1:]):
        gf, xf = g.split("\t"), x.split("\t")
        assert gf[0] == xf[0]
        np.testing.assert_allclose([float(v) for v in gf[1:]], [float(v) for v in xf[1:]], rtol=1e-12, atol=1e-15)
    # from here on both sides start from the SAME %.15g tables the CLI wrote
    def load(prefix):
        lines = (tmp_path / (prefix + ".KPopTwisted.txt")).read_text().splitlines()
        names = [l.split("\t")[0].strip('"') for l in lines[1:]]
        return names, np.array([[float(v) for v in l.split("\t")[1:]] for l in lines[1:]])
    cls_names, m1 = load("ClassVecs")
    test_names, m2 = load("Test")
    metric = oracle.metric_powers(w, 1.0, 1.0, 2.0)
    base = [TWISTDB, "-I", "T", str(tmp_path / "Classes"), "-I", "t", str(tmp_path / "ClassVecs")]
    r = run(base + ["-d", str(tmp_path / "Test"), "-O", "d", str(tmp_path / "TvC")])
    assert r.returncode == 0, r.stderr
    dm = oracle.distance_rowwise(m1, m2, metric)
    want = '""' + "".join('\t"%s"' % c for c in cls_names) + "\n"
    for j, nm in enumerate(test_names):
        want += '"%s"' % nm + "".join("\t%.15g" % v for v in dm[j]) + "\n"
    assert (tmp_path / "TvC.KPopDMatrix.txt").read_text() == want  # rows = the -d operand (lib/Matrix.ml:264-266)
    # -s: summary straight from the twisted vectors; -S: from the distance register
    r = run(base + ["-s", str(tmp_path / "Test"), str(tmp_path / "S1"), "-I", "d", str(tmp_path / "TvC"), "-S", str(tmp_path / "S2"),
                    "--summary-keep-at-most", "all", "-S", str(tmp_path / "S3")])
    assert r.returncode == 0, r.stderr
    st, offs, idx, dist, z = oracle.distance_summary(m1, m2, metric, keep_at_most=2)
    want = "".join(oracle.format_summary_line(test_names[j], st[j], cls_names, idx[int(offs[j]):int(offs[j + 1])],
                                              dist[int(offs[j]):int(offs[j + 1])], z[int(offs[j]):int(offs[j + 1])])
                   for j in range(len(test_names)))
    assert (tmp_path / "S1.KPopSummary.txt").read_text() == want
    # S2 starts from the %.15g distance table
    dm_rt = np.array([[float("%.15g" % v) for v in row] for row in dm])
    want2 = ""
    for j in range(len(test_names)):
        s, i2, d2, z2 = oracle.summarize_row(dm_rt[j], 2)
        want2 += oracle.format_summary_line(test_names[j], s, cls_names, i2, d2, z2)
    assert (tmp_path / "S2.KPopSummary.txt").read_text() == want2
    assert len((tmp_path / "S3.KPopSummary.txt").read_text().splitlines()[0].split("\t")) == 5 + 3 * len(cls_names)
    # other distances and switches reach the kernels
    r = run(base + ["--distance", "minkowski(1)", "--distance-normalize", "false", "-m", "flat", "-d", str(tmp_path / "Test"),
                    "-O", "d", "/dev/stdout"])
    dm1 = oracle.distance_rowwise(m1, m2, oracle.metric_flat(d), oracle.MINKOWSKI, 1.0, False)
    got = np.array([[float(v) for v in l.split("\t")[1:]] for l in r.stdout.splitlines()[1:]])
    np.testing.assert_allclose(got, dm1, rtol=1e-11)


def test_duplicate_label_and_unknown_kmers(tmp_path, oracle):
    k, d = 4, 3
    make_twister(tmp_path, oracle, k, d, keep=1.0)
    sp = tmp_path / "dup.KPopSpectra.txt"
    sp.write_text("\ta\n0a\t3\nzz\t9\n0a0\t4\n\tb\n01\t1\n\ta\n02\t2\n")
    r = run([TWISTDB, "-I", "T", str(tmp_path / "Classes"), "-k", str(sp)])
    assert r.returncode == 1 and 'Duplicate_label("a")' in r.stderr  # lib/Twister.ml:195
    sp.write_text("0a\t3\n")
    r = run([TWISTDB, "-I", "T", str(tmp_path / "Classes"), "-k", str(sp)])
    assert r.returncode == 1 and "Header_expected" in r.stderr       # :106-107
    sp.write_text("\ta\n0a\t3\textra\n")
    r = run([TWISTDB, "-I", "T", str(tmp_path / "Classes"), "-k", str(sp)])
    assert r.returncode == 1 and "Wrong_number_of_columns" in r.stderr  # :103-104
    sp.write_text("\ta\n0a\tthree\n")
    r = run([TWISTDB, "-I", "T", str(tmp_path / "Classes"), "-k", str(sp)])
    assert r.returncode == 1 and "Float_expected" in r.stderr        # :155-157
    # names that are not k-mers of this twister are dropped and do not enter the normaliser (:167-169)
    sp.write_text('\t"q"\n0a\t3\nzz\t9\n0a0\t4\n0a\t1\n')
    r = run([TWISTDB, "-I", "T", str(tmp_path / "Classes"), "-k", str(sp), "-O", "t", "/dev/stdout"])
    assert r.returncode == 0, r.stderr
    row = r.stdout.splitlines()[1].split("\t")
    assert row[0] == '"q"'
    tw = np.array([[float(v) for v in l.split("\t")[1:]] for l in (tmp_path / "Classes.KPopTwister.txt").read_text().splitlines()[1:]])
    names = (tmp_path / "Classes.KPopTwister.txt").read_text().splitlines()[0].split("\t")[1:]
    c = names.index('"0a"')
    np.testing.assert_allclose([float(v) for v in row[1:]], tw[:, c], rtol=1e-14)  # (3+1)/4 of that column


def test_readme_commands_verbatim_with_binary_registers(tmp_path, oracle, pyref):
    """README.md:606,641,656 as written there (binary -i/-o/-d/-s forms):
         KPopCount ... | KPopTwistDB -i T Classes -k /dev/stdin -o t Test
         KPopTwistDB -i t Classes -i T Classes -d Test -O d Test-vs-Classes -o d Test-vs-Classes
         KPopTwistDB -i T Classes -i t Classes -s Test Test-vs-Classes
       and the same through the table forms must give the same summary text."""
    k, d = 5, 6
    make_twister(tmp_path, oracle, k, d)
    rng = np.random.RandomState(4)
    test_reads = [("s%d" % i, "".join(rng.choice(list("ACGT"), size=150))) for i in range(20)]
    class_reads = [("%d" % (i + 1), "".join(rng.choice(list("ACGT"), size=1000))) for i in range(10)]
    write_fasta(tmp_path / "test.fa", test_reads)
    write_fasta(tmp_path / "classes.fa", class_reads)
    cwd = str(tmp_path)
    penv = dict(os.environ, PATH=BIN + ":" + os.environ.get("PATH", ""))
    sh = lambda cmd: subprocess.run(cmd, shell=True, cwd=cwd, capture_output=True, text=True, timeout=120, env=penv)
    env = ""
    r = sh(env + "KPopTwistDB -I T Classes -o T Classes")                      # stands in for KPopTwist (R)
    assert r.returncode == 0, r.stderr
    r = sh(env + "KPopCount -k 5 -L -f classes.fa | KPopTwistDB -i T Classes -k /dev/stdin -o t Classes -O t Classes")
    assert r.returncode == 0, r.stderr
    r = sh(env + "KPopCount -k 5 -L -f test.fa | KPopTwistDB -i T Classes -k /dev/stdin -o t Test -v")
    assert r.returncode == 0, r.stderr
    r = sh(env + "KPopTwistDB -i t Test -O t Test")
    assert r.returncode == 0, r.stderr
    r = sh(env + "KPopTwistDB -i t Classes -i T Classes -d Test -O d Test-vs-Classes -o d Test-vs-Classes")
    assert r.returncode == 0, r.stderr
    r = sh(env + "KPopTwistDB -i T Classes -i t Classes -s Test Test-vs-Classes")
    assert r.returncode == 0, r.stderr
    summary = (tmp_path / "Test-vs-Classes.KPopSummary.txt").read_text()
    assert len(summary.splitlines()) == 20 and all(len(l.split("\t")) >= 11 for l in summary.splitlines())
    r = sh(env + "KPopTwistDB -i T Classes -i d Test-vs-Classes -S FromD")
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "FromD.KPopSummary.txt").read_text() == summary  # binary distances keep every bit
    # binary twisted == what the oracle computes from the same %.15g twister
    dm_lines = (tmp_path / "Test-vs-Classes.KPopDMatrix.txt").read_text().splitlines()
    assert dm_lines[0].split("\t")[1:] == ['"%s"' % n for n in sorted((t for t, _ in class_reads), key=lambda s: s.encode())]
    assert [l.split("\t")[0] for l in dm_lines[1:]] == ['"%s"' % n for n in sorted((t for t, _ in test_reads), key=lambda s: s.encode())]


def test_training_without_r_then_classification(tmp_path, oracle, pyref):
    """The README quick-start shape end to end with no R: class spectra table (what KPopCountDB -t exports,
    src/KPopTwist:38-44) -> KPopTwistCA (replaces the Rscript stage, :49-119) -> KPopTwistDB -I/-o as the wrapper
    does (:122-127) -> classify test reads with the README commands (:606,656)."""
    from oracle import ca_ref
    k, n_classes, glen = 5, 6, 3000
    rng = np.random.RandomState(8)
    genomes = ["".join(rng.choice(list("ACGT"), size=glen)) for _ in range(n_classes)]
    kmers = sorted({h for g in genomes for h in pyref.count_read(g, k)})
    N = np.array([[pyref.count_read(g, k).get(h, 0) for g in genomes] for h in kmers], dtype=np.float64)
    with open(tmp_path / "TABLE.KPopCounter.txt", "w") as f:
        f.write("\t".join("C%d" % (j + 1) for j in range(n_classes)) + "\n")
        for row in N:
            f.write("\t".join("%.15g" % v for v in row) + "\n")
    (tmp_path / "NAMES.KPopCounter.txt").write_text("".join(pyref.to_hex(h, k) + "\n" for h in kmers))
    CA = os.path.join(BIN, "KPopTwistCA")
    r = run([CA, str(tmp_path / "TABLE.KPopCounter.txt"), str(tmp_path / "NAMES.KPopCounter.txt"), str(tmp_path / "Classes"),
             str(tmp_path / "ClassKmers"), "", "1.", "TRUE", "0", "8", "FALSE", "TRUE"])
    assert r.returncode == 0, r.stderr
    # the three tables against the numpy restatement of R's ca (sign-aligned per dimension)
    tw_o, in_o, T_o = ca_ref.ca(N)
    def table(path):
        lines = open(path).read().splitlines()
        return lines[0].split("\t"), [l.split("\t")[0] for l in lines[1:]], np.array([[float(v) for v in l.split("\t")[1:]] for l in lines[1:]])
    hdr, rows, tw = table(tmp_path / "Classes.KPopTwisted.txt")
    assert hdr == ['"rn"'] + ['"Dim%d"' % (d + 1) for d in range(n_classes - 1)] and rows == ['"C%d"' % (j + 1) for j in range(n_classes)]
    assert np.max(np.abs(ca_ref.align_signs(tw, tw_o, 1) - tw_o)) <= 1e-9 * np.max(np.abs(tw_o))
    hdr, rows, inertia = table(tmp_path / "Classes.KPopInertia.txt")
    assert rows == ['"inertia"'] and np.allclose(inertia[0], in_o, rtol=1e-9)
    hdr, rows, T = table(tmp_path / "Classes.KPopTwister.txt")
    assert hdr == ['""'] + ['"%s"' % pyref.to_hex(h, k) for h in kmers] and rows == ['"Dim%d"' % (d + 1) for d in range(n_classes - 1)]
    assert np.max(np.abs(ca_ref.align_signs(T, T_o, 0) - T_o)) <= 1e-9 * np.max(np.abs(T_o))
    hdr, rows, F = table(tmp_path / "ClassKmers.KPopTwisted.txt")
    assert F.shape == (len(kmers), n_classes - 1)
    # encode as the wrapper does, then classify
    penv = dict(os.environ, PATH=BIN + ":" + os.environ.get("PATH", ""))
    sh = lambda cmd: subprocess.run(cmd, shell=True, cwd=str(tmp_path), capture_output=True, text=True, timeout=120, env=penv)
    assert sh("KPopTwistDB -I t Classes -o t Classes").returncode == 0
    assert sh("KPopTwistDB -I T Classes -o T Classes").returncode == 0
    reads, truth = [], []
    for j, g in enumerate(genomes):
        for i in range(5):
            s = int(rng.randint(0, glen - 500))
            reads.append(("t%d_%d" % (j + 1, i), g[s:s + 500]))
            truth.append("C%d" % (j + 1))
    write_fasta(tmp_path / "test.fa", reads)
    r = sh("KPopCount -k 5 -L -f test.fa | KPopTwistDB -i T Classes -k /dev/stdin -o t Test")
    assert r.returncode == 0, r.stderr
    r = sh("KPopTwistDB -i T Classes -i t Classes -s Test Test-vs-Classes")
    assert r.returncode == 0, r.stderr
    got = {l.split("\t")[0]: l.split("\t")[5] for l in (tmp_path / "Test-vs-Classes.KPopSummary.txt").read_text().splitlines()}
    correct = sum(got[name] == cls for (name, _), cls in zip(reads, truth))
    assert correct >= 0.9 * len(reads), (correct, len(reads))


# ---------------------------------------------------------------- KPopCountDB / KPopTwist
COUNTDB = os.path.join(BIN, "KPopCountDB")
TWIST = os.path.join(BIN, "KPopTwist")


def fmt(v, precision=15):
    return "%.*g" % (precision, v)


def class_fixture(pyref, rng, k, n_classes, per_class, glen):
    """n_classes random genomes, per_class mutated copies each; -> {class: [(name, seq)]}, spectra per sequence"""
    classes = {}
    for c in range(n_classes):
        g = rng.choice(list("ACGT"), size=glen)
        members = []
        for i in range(per_class):
            s = g.copy()
            pos = rng.choice(glen, size=glen // 50, replace=False)
            s[pos] = rng.choice(list("ACGT"), size=pos.size)
            members.append(("S%d-C%d" % (i, c + 1), "".join(s[: glen - 37 * i])))   # unequal lengths: unequal norms
        classes["C%d" % (c + 1)] = members
    return classes


def expected_combination(oracle, pyref, members, k, criterion):
    """add_combined_selected on the spectra of `members` as KPopCountDB sees them: rows in order of first appearance,
    spectra visited in descending label order (lib/KMerDB.ml:650-660)."""
    spectra = [(name, pyref.count_read(seq, k)) for name, seq in members]
    rows = []
    for _, sp in spectra:                     # KPopCount prints ascending hashes; new k-mers are appended as they come
        for h in sorted(sp):
            if h not in rows:
                rows.append(h)
    seen, order = set(), []
    for h in rows:
        if h not in seen:
            seen.add(h)
            order.append(h)
    cols = [np.array([sp.get(h, 0) for h in order], dtype=np.int32) for _, sp in spectra]
    lin, _ = oracle.counter_stats(cols, 1.0, 1.0)
    labels = [name for name, _ in spectra]
    sel = [labels.index(l) for l in sorted(labels, reverse=True)]
    out, _ = oracle.counter_combine(cols, sel, lin[:, 2], criterion)
    return order, out


@pytest.mark.parametrize("criterion", ["mean", "median"])
def test_readme_training_flow(tmp_path, oracle, pyref, criterion):
    """README.md:91-92 verbatim (the quick-start training loop), then :606/:656 classification -- no OCaml, no R."""
    from oracle import ca_ref
    k, n_classes, per_class, glen = 5, 5, 4, 2500
    rng = np.random.default_rng(91)
    classes = class_fixture(pyref, rng, k, n_classes, per_class, glen)
    with open(tmp_path / "clusters-small.fasta", "w") as f:
        for members in zip(*classes.values()):             # interleaved, as a real file would be
            for name, seq in members:
                f.write(">%s\n%s\n" % (name, seq))
    penv = dict(os.environ, PATH=BIN + ":" + os.environ.get("PATH", ""))
    sh = lambda cmd: subprocess.run(["bash", "-c", cmd], cwd=str(tmp_path), capture_output=True, text=True, timeout=300, env=penv)
    crit = "" if criterion == "mean" else "--combination-criterion median "
    loop = ("K=%d; for CLASS in %s; do cat clusters-small.fasta | awk -v CLASS=$CLASS '{nr=(NR-1)%%2; ok=(nr==0?$0~(\"-\"CLASS\"$\"):nr==1&&ok); "
            "if (ok) print}' | KPopCount -k $K -L -f /dev/stdin | KPopCountDB -k /dev/stdin %s-R \"~.\" -A $CLASS -L $CLASS -N -D "
            "-t /dev/stdout; done") % (k, " ".join(classes), crit)
    r = sh(loop + " | cat > combined.txt")  # a pipe, as in the README: opening /dev/stdout truncates a redirected file
    assert r.returncode == 0, r.stderr
    # every class representative, byte for byte
    want_text, reps = "", {}
    for cname, members in classes.items():
        order, out = expected_combination(oracle, pyref, members, k, 0 if criterion == "mean" else 1)
        reps[cname] = dict(zip(order, out))
        want_text += "\t%s\n" % cname + "".join("%s\t%s\n" % (pyref.to_hex(h, k), fmt(float(v))) for h, v in zip(order, out))
    got = (tmp_path / "combined.txt").read_text()
    # the table drops k-mers whose row is all zero (lib/KMerDB.ml:1033-1036)
    want_text = "".join(l + "\n" for l in want_text.splitlines() if not l.endswith("\t0"))
    assert got == want_text
    r = sh("cat combined.txt | KPopCountDB -k /dev/stdin -o Classes.%d -v" % k)
    assert r.returncode == 0 and "Read %d spectra" % n_classes in r.stderr, r.stderr
    r = sh("KPopTwist -i Classes.%d -o Classes.%d -v" % (k, k))
    assert r.returncode == 0, r.stderr
    assert sorted(p.name for p in tmp_path.iterdir() if p.name.startswith("Classes")) == [
        "Classes.%d.KPopCounter" % k, "Classes.%d.KPopTwisted" % k, "Classes.%d.KPopTwister" % k]
    # the twister against the numpy restatement of R's ca on the same class table
    r = sh("KPopTwistDB -i T Classes.%d -O T /dev/stdout" % k)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.splitlines()
    lines = lines[:1 + (n_classes - 1)]      # -O T writes the twister, then the inertia table, to the same path
    kmers_out = [c.strip('"') for c in lines[0].split("\t")[1:]]
    T = np.array([[float(v) for v in l.split("\t")[1:]] for l in lines[1:]])
    row_order = []
    for cname in classes:
        for h, v in reps[cname].items():
            if v != 0 and h not in row_order:
                row_order.append(h)
    assert kmers_out == [pyref.to_hex(h, k) for h in row_order]
    N = np.array([[float(reps[c].get(h, 0)) for c in classes] for h in row_order])
    tw_o, in_o, T_o = ca_ref.ca(N)
    assert T.shape == T_o.shape
    assert np.max(np.abs(ca_ref.align_signs(T, T_o, 0) - T_o)) <= 1e-8 * np.max(np.abs(T_o))
    # classification of fragments of the training genomes (README.md:606,656)
    reads, truth = [], []
    for cname, members in classes.items():
        for i in range(4):
            name, seq = members[i % per_class]
            s = int(rng.integers(0, len(seq) - 600))
            reads.append(("q%s_%d" % (cname, i), seq[s:s + 600]))
            truth.append(cname)
    write_fasta(tmp_path / "test.fa", reads)
    r = sh("KPopCount -k %d -L -f test.fa | KPopTwistDB -i T Classes.%d -k /dev/stdin -o t Test" % (k, k))
    assert r.returncode == 0, r.stderr
    r = sh("KPopTwistDB -i T Classes.%d -i t Classes.%d -s Test Test-vs-Classes" % (k, k))
    assert r.returncode == 0, r.stderr
    got = {l.split("\t")[0]: l.split("\t")[5] for l in (tmp_path / "Test-vs-Classes.KPopSummary.txt").read_text().splitlines()}
    correct = sum(got[name] == cls for (name, _), cls in zip(reads, truth))
    assert correct >= 0.9 * len(reads), (correct, len(reads))
    # README.md:93 / :164 verbatim: the twisted register travels as an OCaml-Marshal stream through the pipe
    r = sh("K=%d; cat test.fa | KPopCount -k $K -L -f /dev/stdin | KPopTwistDB -i T Classes.$K -k /dev/stdin -o t /dev/stdout | "
           "KPopTwistDB -i T Classes.$K -i t Classes.$K -s /dev/stdin Test_prediction.$K -v" % k)
    assert r.returncode == 0, r.stderr
    assert (tmp_path / ("Test_prediction.%d.KPopSummary.txt" % k)).read_bytes() == (tmp_path / "Test-vs-Classes.KPopSummary.txt").read_bytes()
    # README.md:182: ... -o t /dev/stdout -v | KPopTwistDB -i t /dev/stdin -O t Test.$K
    r = sh("K=%d; cat test.fa | KPopCount -k $K -L -f /dev/stdin | KPopTwistDB -i T Classes.$K -k /dev/stdin -o t /dev/stdout -v | "
           "KPopTwistDB -i t /dev/stdin -O t Test.$K && KPopTwistDB -i t Test -O t /dev/stdout | cmp - Test.$K.KPopTwisted.txt" % k)
    assert r.returncode == 0, r.stderr
    # README.md:768: --keep-at-most (the README's spelling of --summary-keep-at-most)
    r = sh("KPopTwistDB -i T Classes.%d -i t Classes.%d --keep-at-most 3 -s Test Top3 && awk -F'\t' '{print NF}' Top3.KPopSummary.txt | sort -u" % (k, k))
    assert r.returncode == 0 and r.stdout.split() == ["14"], (r.stdout, r.stderr)   # 5 statistics + 3 x (name, distance, z)


def test_countdb_tables_spectra_split_and_distances(tmp_path, oracle, pyref):
    """-c (combine by class through metadata), -t with every layout switch, -s, -F and --distances against the oracle."""
    k = 4
    rng = np.random.default_rng(5)
    seqs = [("s%d" % i, "".join(rng.choice(list("ACGT"), size=int(rng.integers(150, 400))))) for i in range(6)]
    write_fasta(tmp_path / "x.fa", seqs)
    (tmp_path / "meta.txt").write_text("label\tclass\tsite\n" + "".join("s%d\t%s\t%s\n" % (i, "ab"[i % 2], "north") for i in range(6)))
    penv = dict(os.environ, PATH=BIN + ":" + os.environ.get("PATH", ""))
    sh = lambda cmd: subprocess.run(["bash", "-c", cmd], cwd=str(tmp_path), capture_output=True, text=True, timeout=300, env=penv)
    assert sh("KPopCount -k %d -L -f x.fa | KPopCountDB -k /dev/stdin -m meta.txt -o db" % k).returncode == 0
    spectra = [pyref.count_read(s, k) for _, s in seqs]
    rows = []
    for sp in spectra:
        rows += [h for h in sorted(sp) if h not in rows]
    cols = [np.array([sp.get(h, 0) for h in rows], dtype=np.int32) for sp in spectra]
    names = [pyref.to_hex(h, k) for h in rows]
    labels = [n for n, _ in seqs]

    def table_text(cols_idx, cs, which, thr, pw, precision=15, transpose=False, row_names=True, col_names=True, meta=None, rs=None,
                   zero_rows=False):
        vals = oracle.counter_transform(cols, cs, which, thr, pw, kmer_major=True)
        keep = [r for r in range(len(rows)) if zero_rows or rs[r, 2] > 0]
        out = ""
        if not transpose:
            if col_names:
                out += "".join(("\t" if (i > 0 or row_names) else "") + labels[c] for i, c in enumerate(cols_idx)) + "\n"
            for mname, mvals in (meta or []):
                out += (mname if row_names else "") + "".join(("\t" if (i > 0 or row_names) else "") + mvals[c] for i, c in enumerate(cols_idx)) + "\n"
            for r in keep:
                out += (names[r] if row_names else "") + "".join(("\t" if (i > 0 or row_names) else "") + fmt(vals[r, c], precision)
                                                                for i, c in enumerate(cols_idx)) + "\n"
        else:
            if col_names:
                cells = [m for m, _ in (meta or [])] + [names[r] for r in keep]
                out += "".join(("\t" if (i > 0 or row_names) else "") + c for i, c in enumerate(cells)) + "\n"
            for c in cols_idx:
                cells = [mv[c] for _, mv in (meta or [])] + [fmt(vals[r, c], precision) for r in keep]
                out += (labels[c] if row_names else "") + "".join(("\t" if (i > 0 or row_names) else "") + x for i, x in enumerate(cells)) + "\n"
        return out

    meta = [("class", ["ab"[i % 2] for i in range(6)]), ("site", ["north"] * 6)]
    cs, rs = oracle.counter_stats(cols, 1.0, 1.0)
    r = sh("KPopCountDB -i db -t /dev/stdout")
    assert r.returncode == 0 and r.stdout == table_text(range(6), cs, 1, 1.0, 1.0, rs=rs), r.stderr
    r = sh("KPopCountDB -i db --table-output-metadata true --table-transpose true --counts-precision 4 -t /dev/stdout")
    assert r.stdout == table_text(range(6), cs, 1, 1.0, 1.0, precision=4, transpose=True, meta=meta, rs=rs)
    r = sh("KPopCountDB -i db --table-output-row-names false --table-output-metadata true -L s1,s4 -F -t /dev/stdout")
    assert r.stdout == table_text([0, 2, 3, 5], cs, 1, 1.0, 1.0, row_names=False, meta=meta, rs=rs)
    cs2, rs2 = oracle.counter_stats(cols, 2.0, 1.0)
    r = sh("KPopCountDB -i db --counts-threshold 2 --counts-transform binary --table-output-col-names false -t /dev/stdout")
    assert r.stdout == table_text(range(6), cs2, 0, 2.0, 1.0, col_names=False, rs=rs2)
    r = sh("KPopCountDB -i db --counts-threshold 2 --counts-output-zero-kmers true --counts-transform pseudocounts -t out && cat out.KPopCounter.txt")
    want = table_text(range(6), cs2, 3, 2.0, 1.0, rs=rs2, zero_rows=True)
    assert r.stdout == want
    cs3, rs3 = oracle.counter_stats(cols, 1.0, 0.5)
    r = sh("KPopCountDB -i db --counts-power 0.5 --counts-transform clr --counts-precision 9 -t /dev/stdout")
    got_rows = [l.split("\t") for l in r.stdout.splitlines()]
    want_rows = [l.split("\t") for l in table_text(range(6), cs3, 2, 1.0, 0.5, precision=9, rs=rs3).splitlines()]
    assert [g[0] for g in got_rows] == [w[0] for w in want_rows] and got_rows[0] == want_rows[0]
    np.testing.assert_allclose([[float(v) for v in g[1:]] for g in got_rows[1:]], [[float(v) for v in w[1:]] for w in want_rows[1:]], rtol=1e-8)
    # -s: spectra, only positive values
    r = sh("KPopCountDB -i db --counts-threshold 2 -s /dev/stdout")
    vals = oracle.counter_transform(cols, cs2, 1, 2.0, 1.0, kmer_major=True)
    want = "".join("\t%s\n" % labels[c] + "".join("%s\t%s\n" % (names[i], fmt(vals[i, c])) for i in range(len(rows)) if rs2[i, 2] > 0 and vals[i, c] > 0)
                   for c in range(6))
    assert r.stdout == want
    # the tables and spectra are produced block by block: tiny blocks must give the same bytes
    tiny = "KPOP_HOST_BLOCK=50 "
    for cmd in ("KPopCountDB -i db -t /dev/stdout", "KPopCountDB -i db --table-output-metadata true --table-transpose true -t /dev/stdout",
                "KPopCountDB -i db --counts-threshold 2 -s /dev/stdout", "KPopCountDB -i db -L s1,s4 -F --counts-transform clr -t /dev/stdout"):
        a, b = sh(cmd), sh(tiny + cmd)
        assert a.returncode == 0 and b.returncode == 0 and a.stdout == b.stdout and len(a.stdout) > 1000, cmd
    # -c: classes "a" (s0,s2,s4) and "b" (s1,s3,s5), originals removed, metadata inherited where unanimous
    r = sh("KPopCountDB -i db -c class --table-output-metadata true -t /dev/stdout")
    assert r.returncode == 0, r.stderr
    lin = cs[:, 2]
    comb = [oracle.counter_combine(cols, sel, lin, 0)[0] for sel in ([4, 2, 0], [5, 3, 1])]
    want = "\ta\tb\nclass\ta\tb\nsite\tnorth\tnorth\n" + "".join("%s\t%s\t%s\n" % (names[i], fmt(float(comb[0][i])), fmt(float(comb[1][i])))
                                                                 for i in range(len(rows)) if comb[0][i] + comb[1][i] > 0)
    assert r.stdout == want
    r = sh("KPopCountDB -i db -L a -A a -c class -t /dev/null")
    assert r.returncode == 1 and "Class_label_is_also_spectrum_name" in r.stderr
    r = sh("KPopCountDB -i db -c nothere")
    assert r.returncode == 1 and "Classes_label_not_found" in r.stderr
    # --distances between the two classes' members, spectra normalised by their sums, flat metric over k-mers
    r = sh("KPopCountDB -i db --distance 'minkowski(1.5)' --distances 'class~a' 'class~b' D && KPopTwistDB -i d D -O d /dev/stdout")
    assert r.returncode == 0, r.stderr
    m = np.array([c / c.sum() for c in cols], dtype=np.float64)
    want = oracle.distance_rowwise(m[[0, 2, 4]], m[[1, 3, 5]], np.ones(len(rows)), oracle.MINKOWSKI, 1.5, True)
    lines = r.stdout.splitlines()
    assert lines[0].split("\t") == ['""', '"s0"', '"s2"', '"s4"'] and [l.split("\t")[0] for l in lines[1:]] == ['"s1"', '"s3"', '"s5"']
    got = np.array([[float(v) for v in l.split("\t")[1:]] for l in lines[1:]])
    np.testing.assert_allclose(got, want, rtol=1e-12)


def test_embeddings_register(tmp_path, oracle):
    """KPopTwistDB -e: twisted -> embeddings ('e' register, .KPopVectors[.txt]), metric from the twister's inertia."""
    d, k = 6, 4
    cols = oracle.enumerate_kmers(k)
    T = oracle.synth_twister(5, d, cols)
    inertia = oracle.synth_inertia(d)
    dims = ["Dim%d" % (i + 1) for i in range(d)]
    write_table(tmp_path / "X.KPopTwister.txt", [oracle.to_hex(h, k) for h in cols], dims, T)
    write_table(tmp_path / "X.KPopInertia.txt", dims, ["inertia"], [inertia])
    rng = np.random.RandomState(3)
    tw = rng.standard_normal((7, d))
    names = ["s%d" % i for i in range(7)]
    write_table(tmp_path / "tw.KPopTwisted.txt", dims, names, tw)
    tw = np.array([[float("%.15g" % v) for v in row] for row in tw])
    inertia_rt = np.array([float("%.15g" % v) for v in inertia])
    metric = oracle.metric_powers(inertia_rt)
    for dist, kind, p in (("euclidean", 0, 2.0), ("cosine", 1, 2.0), ("minkowski(1.5)", 2, 1.5)):
        r = run([TWISTDB, "-I", "T", str(tmp_path / "X"), "-I", "t", str(tmp_path / "tw"), "--distance", dist, "-e", "-O", "e", "/dev/stdout",
                 "-o", "e", str(tmp_path / "emb")])
        assert r.returncode == 0, r.stderr
        want = oracle.embeddings(tw, metric, kind, p, True)
        lines = r.stdout.splitlines()
        assert lines[0] == '""\t' + "\t".join('"%s"' % x for x in dims) and [l.split("\t")[0] for l in lines[1:]] == ['"%s"' % n for n in names]
        got = np.array([[float(v) for v in l.split("\t")[1:]] for l in lines[1:]])
        np.testing.assert_allclose(got, want, rtol=1e-13)
    # binary round trip through the embeddings register, and the type check of the archive
    r = run([TWISTDB, "-i", "e", str(tmp_path / "emb"), "-O", "e", "/dev/stdout"])
    assert r.returncode == 0 and r.stdout.splitlines()[1].startswith('"s0"\t')
    (tmp_path / "emb.KPopTwisted").write_bytes((tmp_path / "emb.KPopVectors").read_bytes())
    r = run([TWISTDB, "-i", "t", str(tmp_path / "emb")])
    assert r.returncode == 1 and "Unexpected_type" in r.stderr
    r = run([TWISTDB, "-I", "t", str(tmp_path / "tw"), "-e"])
    assert r.returncode == 1 and "require a twister" in r.stderr


def test_parallel_text_paths_are_chunking_invariant(tmp_path, oracle, pyref):
    """KPopCount's spectra writer and KPopTwistDB's spectra parser cut their work over host threads: forcing 32 tiny
    chunks must not change a byte, and errors keep the sequential parser's line numbers."""
    k, d = 5, 7
    rng = np.random.RandomState(17)
    reads = [("read %d" % i, "".join(rng.choice(list("ACGTN"), p=[.24, .24, .24, .24, .04], size=rng.randint(0, 90)))) for i in range(400)]
    write_fasta(tmp_path / "x.fa", reads)
    tiny = dict(os.environ, KPOP_HOST_THREADS="32", KPOP_HOST_CHUNK="64")
    one = dict(os.environ, KPOP_HOST_THREADS="1")
    outs = [subprocess.run([COUNT, "-k", str(k), "-L", "-f", str(tmp_path / "x.fa")], capture_output=True, text=True, env=e) for e in (tiny, one)]
    assert outs[0].returncode == 0 and outs[0].stdout == outs[1].stdout == expected_spectra(pyref, reads, k)
    (tmp_path / "x.KPopSpectra.txt").write_text(outs[0].stdout)
    cols = make_twister(tmp_path, oracle, k, d)[0]
    col = oracle.to_hex(int(cols[3]), k)  # a name that IS a column: its value is read (lib/Twister.ml:153-157)
    res = []
    for name, e in (("a", tiny), ("b", one)):
        r = subprocess.run([TWISTDB, "-I", "T", str(tmp_path / "Classes"), "-k", str(tmp_path / "x.KPopSpectra.txt"), "-O", "t", "/dev/stdout"],
                           capture_output=True, text=True, env=e)
        assert r.returncode == 0, r.stderr
        res.append(r.stdout)
    assert res[0] == res[1] and res[0].count("\n") == len(reads) + 1
    lines = outs[0].stdout.splitlines(keepends=True)
    for bad_at, bad, what in ((700, "aaa\t1\t2\n", "Wrong_number_of_columns(701, 3, 2)"), (1500, col + "\tx1\n", "Float_expected(\"x1\")"),
                              (0, "aaa\t3\n", "Header_expected(\"aaa\t3\")"), (900, "\tq\"uote\n", "Quotes_in_name")):
        (tmp_path / "bad.txt").write_text("".join(lines[:bad_at]) + bad + "".join(lines[bad_at:]))
        for e in (tiny, one):
            r = subprocess.run([TWISTDB, "-I", "T", str(tmp_path / "Classes"), "-k", str(tmp_path / "bad.txt")], capture_output=True, text=True, env=e)
            assert r.returncode == 1 and what in r.stderr, (what, r.stderr)
    # "aaa" is no 5-mer: the line is dropped before its value is looked at (:167-169)
    (tmp_path / "bad.txt").write_text("".join(lines[:1500]) + "aaa\tx1\n" + "".join(lines[1500:]))
    r = subprocess.run([TWISTDB, "-I", "T", str(tmp_path / "Classes"), "-k", str(tmp_path / "bad.txt"), "-O", "t", "/dev/stdout"], capture_output=True, text=True, env=tiny)
    assert r.returncode == 0 and r.stdout == res[0], r.stderr


def test_protein_content_through_the_clis(tmp_path, oracle, pyref):
    """KPopCount -C protein (bin/KPopCount.ml:246-248) then KPopTwistDB -k on the protein spectra: the twist side only
    sees column names, so a twister whose columns are protein k-mers works unchanged."""
    k, d = 3, 6
    rng = np.random.RandomState(5)
    aa = "ACDEFGHIKLMNPQRSTVWY"
    prots = [("p%d some description" % i, "".join(aa[x] for x in rng.randint(0, 20, size=rng.randint(5, 300)))) for i in range(40)]
    prots.append(("with-gaps", "MKT-AYI AKQ\tRXQISFVKSHFSRQ*LEERLG"))
    prots.append(("lower", "mktayiakqrqisfvkshf"))
    with open(tmp_path / "p.fa", "w") as f:
        for tag, s in prots:
            f.write(">%s\n%s\n" % (tag, s))

    def lint(s):
        return "".join(ch for ch in s.upper() if ch not in "- \t")
    want = "".join("\t%s\n" % tag + "".join("%s\t%d\n" % (pyref.to_hex_protein(h, k), c) for h, c in sorted(pyref.count_read_protein(lint(s), k).items()))
                   for tag, s in prots)
    r = run([COUNT, "-k", str(k), "-C", "protein", "-L", "-f", str(tmp_path / "p.fa")])
    assert r.returncode == 0 and r.stdout == want, r.stderr
    merged = {}
    for _, s in prots:
        for h, c in pyref.count_read_protein(lint(s), k).items():
            merged[h] = merged.get(h, 0) + c
    r = run([COUNT, "-k", str(k), "-C", "prot", "-l", "all", "-f", str(tmp_path / "p.fa")])
    assert r.stdout == "\tall\n" + "".join("%s\t%d\n" % (pyref.to_hex_protein(h, k), c) for h, c in sorted(merged.items()))
    r = run([COUNT, "-k", "13", "-C", "protein", "-L", "-f", str(tmp_path / "p.fa")])
    assert r.returncode == 1 and "<= 12 for protein" in r.stderr
    # a twister over the protein 3-mers that occur, then README.md:606 with -C protein
    cols = np.array(sorted(merged), dtype=np.uint64)
    T = np.array([[float("%.15g" % x) for x in row] for row in oracle.synth_twister(7, d, cols)])
    dims = ["Dim%d" % (i + 1) for i in range(d)]
    write_table(tmp_path / "P.KPopTwister.txt", [pyref.to_hex_protein(int(h), k) for h in cols], dims, T)
    write_table(tmp_path / "P.KPopInertia.txt", dims, ["inertia"], [oracle.synth_inertia(d)])
    penv = dict(os.environ, PATH=BIN + ":" + os.environ.get("PATH", ""))
    r = subprocess.run(["bash", "-c", "KPopCount -k %d -C protein -L -f p.fa | KPopTwistDB -I T P -k /dev/stdin -O t /dev/stdout" % k],
                       cwd=str(tmp_path), capture_output=True, text=True, env=penv)
    assert r.returncode == 0, r.stderr
    pb = np.frombuffer("".join(lint(s) for _, s in prots).encode(), dtype=np.uint8)
    po = np.concatenate([[0], np.cumsum([len(lint(s)) for _, s in prots])]).astype(np.uint64)
    h, c, o = oracle.count_reads(pb, po, k, oracle.PROTEIN)
    rows = oracle.twist(T, cols, h, c.astype(np.float64), o)
    assert r.stdout == twisted_text(dims, [t for t, _ in prots], rows)
    # more than 32 dimensions: the reads stream's blocks of short proteins take the fused kernel (five bits a residue), a block
    # with a long one (more than 512 windows) the count + line-by-line twist; the same text either way
    d2 = 40
    T2 = np.array([[float("%.15g" % x) for x in row] for row in oracle.synth_twister(8, d2, cols)])
    dims2 = ["Dim%d" % (i + 1) for i in range(d2)]
    write_table(tmp_path / "Q.KPopTwister.txt", [pyref.to_hex_protein(int(h), k) for h in cols], dims2, T2)
    write_table(tmp_path / "Q.KPopInertia.txt", dims2, ["inertia"], [oracle.synth_inertia(d2)])
    for extra, name in (([], "p.fa"), ([("titin-like", "".join(aa[x] for x in rng.randint(0, 20, size=1400)))], "p2.fa")):
        ps = prots + extra
        with open(tmp_path / name, "w") as f:
            for tag, s_ in ps:
                f.write(">%s\n%s\n" % (tag, s_))
        r = subprocess.run(["bash", "-c", "KPopCount -k %d -C protein -L -f %s | KPopTwistDB -I T Q -k /dev/stdin -O t /dev/stdout" % (k, name)],
                           cwd=str(tmp_path), capture_output=True, text=True, env=penv)
        assert r.returncode == 0, r.stderr
        pb = np.frombuffer("".join(lint(s_) for _, s_ in ps).encode(), dtype=np.uint8)
        po = np.concatenate([[0], np.cumsum([len(lint(s_)) for _, s_ in ps])]).astype(np.uint64)
        h, c, o = oracle.count_reads(pb, po, k, oracle.PROTEIN)
        assert r.stdout == twisted_text(dims2, [t for t, _ in ps], oracle.twist(T2, cols, h, c.astype(np.float64), o))


def test_reference_wrapper_steps_match_one_process_kpoptwist(tmp_path, oracle, pyref):
    """The steps of the reference's bash wrapper (src/KPopTwist:19-131: KPopTwist_ echo, two KPopCountDB exports, the
    R stage -- here KPopTwistCA with the same positional arguments -- and three KPopTwistDB encodings) give the twister
    that the one-process KPopTwist writes."""
    k = 4
    rng = np.random.default_rng(44)
    seqs = [("c%d" % i, "".join(rng.choice(list("ACGT"), size=int(rng.integers(800, 1500))))) for i in range(7)]
    write_fasta(tmp_path / "x.fa", seqs)
    penv = dict(os.environ, PATH=BIN + ":" + os.environ.get("PATH", ""))
    sh = lambda cmd: subprocess.run(["bash", "-c", cmd], cwd=str(tmp_path), capture_output=True, text=True, timeout=300, env=penv)
    assert sh("KPopCount -k %d -L -f x.fa | KPopCountDB -k /dev/stdin -o Classes" % k).returncode == 0
    r = sh("KPopTwist -i Classes -o V -K Vk")
    assert r.returncode == 0, r.stderr
    script = r"""
set -e
PARAMETERS="$(KPopTwist_ -i Classes -o W -K Wk)"
IFS=$'\x01' read -r PREFIX_IN KMERS_KEEP KMERS_SAMPLE THRESHOLD_COUNTS POWER TRANSFORM NORMALIZE THRESHOLD_KMERS PREFIX_OUT PREFIX_OUT_KMERS THREADS TEMPORARIES VERBOSE <<< "$PARAMETERS"
mkdir T
KPopCountDB -T "$THREADS" -i "$PREFIX_IN" --counts-threshold "$THRESHOLD_COUNTS" --counts-power "$POWER" --counts-transform "$TRANSFORM" \
    --table-output-row-names false -t T/TABLE -R "~." -D \
    --counts-output-zero-kmers true --counts-threshold 1. --counts-power 1. --counts-transform power \
    --table-output-row-names true --table-output-metadata false -t /dev/stdout | tail -n +2 > T/NAMES.KPopCounter.txt
KPopTwistCA T/TABLE.KPopCounter.txt T/NAMES.KPopCounter.txt "$PREFIX_OUT" "$PREFIX_OUT_KMERS" "$KMERS_KEEP" "$KMERS_SAMPLE" "$NORMALIZE" "$THRESHOLD_KMERS" "$THREADS" "$TEMPORARIES" "$VERBOSE"
KPopTwistDB -T "$THREADS" -I t "$PREFIX_OUT" -o t "$PREFIX_OUT"
KPopTwistDB -T "$THREADS" -I t "$PREFIX_OUT_KMERS" -o t "$PREFIX_OUT_KMERS"
KPopTwistDB -T "$THREADS" -I T "$PREFIX_OUT" -o T "$PREFIX_OUT"
"""
    r = sh(script)
    assert r.returncode == 0, r.stderr

    def table(cmd):
        out = sh(cmd).stdout.splitlines()
        return out[0], [l.split("\t")[0] for l in out[1:]], np.array([[float(v) for v in l.split("\t")[1:]] for l in out[1:]])
    for reg, a, b, n_lines in (("T", "V", "W", 7), ("t", "V", "W", None), ("t", "Vk", "Wk", None)):
        ha, ra, da = table("KPopTwistDB -i %s %s -O %s /dev/stdout%s" % (reg, a, reg, " | head -%d" % n_lines if n_lines else ""))
        hb, rb, db = table("KPopTwistDB -i %s %s -O %s /dev/stdout%s" % (reg, b, reg, " | head -%d" % n_lines if n_lines else ""))
        assert ha == hb and ra == rb and da.shape == db.shape and da.size > 0
        np.testing.assert_allclose(da, db, rtol=1e-12, atol=1e-13)   # W went through %.15g text on the way


def test_splits_through_the_cli(tmp_path, kpop, oracle, pyref):
    """KPopTwistDB -e -p -O s (bin/KPopTwistDB.ml:494-506,534-535): embeddings from the twisted register, splits by both
    algorithms, the declared '.PhyloSplits.txt' text against the restatement run on the embeddings the tool itself wrote"""
    k, d = 5, 6
    make_twister(tmp_path, oracle, k, d)
    rng = np.random.RandomState(21)
    rows = np.round(rng.normal(size=(30, d)), 3)
    names = ["s%02d" % i for i in range(30)]
    write_table(tmp_path / "X.KPopTwisted.txt", ["Dim%d" % (i + 1) for i in range(d)], names, rows)
    base = [TWISTDB, "-I", "T", str(tmp_path / "Classes"), "-I", "t", str(tmp_path / "X"), "-e", "-O", "e", str(tmp_path / "X")]
    r = run(base + ["--splits-keep-at-most", "12", "-p", "-O", "s", str(tmp_path / "gaps"),
                    "--splits-algorithm", "centroids", "--precision-for-splits", "8", "-p", "-O", "s", str(tmp_path / "cent")])
    assert r.returncode == 0, r.stderr
    emb_lines = (tmp_path / "X.KPopVectors.txt").read_text().splitlines()[1:]
    # the tool computed on the full-precision embeddings; the restatement needs the same numbers, so go through the API
    import kpop_amd
    T = np.array([[float(x) for x in l.split("\t")[1:]] for l in (tmp_path / "Classes.KPopInertia.txt").read_text().splitlines()[1:]])
    emb = kpop_amd.embeddings(rows, kpop_amd.metric_compute(T[0]))
    assert [("%.15g" % x) for x in emb[0]] == emb_lines[0].split("\t")[1:]
    want_g = pyref.splits_text(names, pyref.splits_gaps(emb.tolist(), 12), 10)
    assert (tmp_path / "gaps.PhyloSplits.txt").read_text() == want_g
    want_c = pyref.splits_text(names, pyref.splits_centroids(emb.tolist()), 8)
    assert (tmp_path / "cent.PhyloSplits.txt").read_text() == want_c
    r = run([TWISTDB, "-I", "T", str(tmp_path / "Classes"), "-I", "t", str(tmp_path / "X"), "-e", "-p", "-o", "s", str(tmp_path / "bin")])
    assert r.returncode == 1 and "binary splits" in r.stderr


def test_kpoptwist_kmer_selection_device_path_equals_host_path(tmp_path):
    """KPopTwist keeps its table on the device (run_ca_device: row sums, keep list / sample / threshold, row gather and
    kpop_dev_ca in place); KPopTwistCA, the R stage of the reference's wrapper, goes through the host table (run_ca,
    kpop_ca).  Same k-mer selection options (src/KPopTwist:76-91), same twister: names equal, numbers to rounding."""
    k = 5
    rng = np.random.default_rng(77)
    seqs = [("c%d" % i, "".join(rng.choice(list("ACGT"), size=int(rng.integers(1500, 2500))))) for i in range(9)]
    write_fasta(tmp_path / "x.fa", seqs)
    penv = dict(os.environ, PATH=BIN + ":" + os.environ.get("PATH", ""))
    sh = lambda cmd: subprocess.run(["bash", "-c", cmd], cwd=str(tmp_path), capture_output=True, text=True, timeout=300, env=penv)
    assert sh("KPopCount -k %d -L -f x.fa | KPopCountDB -k /dev/stdin -o Classes" % k).returncode == 0
    # the table and the names as the wrapper exports them (src/KPopTwist:38-44)
    r = sh("mkdir T && KPopCountDB -i Classes --table-output-row-names false -t T/TABLE -R '~.' -D --counts-output-zero-kmers true "
           "--table-output-row-names true --table-output-metadata false -t /dev/stdout | tail -n +2 > T/NAMES.KPopCounter.txt")
    assert r.returncode == 0, r.stderr
    names = [l.split("\t")[0].strip('"') for l in (tmp_path / "T" / "NAMES.KPopCounter.txt").read_text().splitlines()]
    keep = [n for i, n in enumerate(names) if i % 3 != 1]
    (tmp_path / "keep.txt").write_text("".join(n + "\n" for n in reversed(keep)))  # in another order than the table's

    def table(cmd):
        out = sh(cmd).stdout.splitlines()
        return out[0], [l.split("\t")[0] for l in out[1:]], np.array([[float(v) for v in l.split("\t")[1:]] for l in out[1:]])
    for tag, opts, ca_args in (("plain", "", ("", "1", "0")), ("keep", "-k keep.txt", ("keep.txt", "1", "0")), ("sample", "-s 0.6", ("", "0.6", "0")),
                               ("thr", "--kmers-threshold 0.3", ("", "1", "0.3")), ("all3", "-k keep.txt -s 0.7 --kmers-threshold 0.2", ("keep.txt", "0.7", "0.2"))):
        r = sh("KPopTwist -i Classes -o D_%s %s" % (tag, opts))
        assert r.returncode == 0, (tag, r.stderr)
        r = sh("KPopTwistCA T/TABLE.KPopCounter.txt T/NAMES.KPopCounter.txt H_%s '' '%s' %s true %s 1 false false && "
               "KPopTwistDB -I T H_%s -o T H_%s && KPopTwistDB -I t H_%s -o t H_%s" % (tag, ca_args[0], ca_args[1], ca_args[2], tag, tag, tag, tag))
        assert r.returncode == 0, (tag, r.stderr)
        for reg in ("T", "t"):
            cut = " | head -%d" % len(seqs) if reg == "T" else ""  # (the twister's table is followed by the inertia's: 8 dimensions + header)
            ha, ra, da = table("KPopTwistDB -i %s D_%s -O %s /dev/stdout%s" % (reg, tag, reg, cut))
            hb, rb, db = table("KPopTwistDB -i %s H_%s -O %s /dev/stdout%s" % (reg, tag, reg, cut))
            assert ha == hb and ra == rb and da.shape == db.shape and da.size > 0, (tag, reg)
            np.testing.assert_allclose(da, db, rtol=1e-9, atol=1e-11, err_msg="%s %s" % (tag, reg))   # H went through %.15g text
        if tag != "plain":
            n_kmers = len(sh("KPopTwistDB -i T D_%s -O T /dev/stdout | head -1" % tag).stdout.split("\t")) - 1
            assert n_kmers < len(names), tag


def _pyref_rows(pyref, T, names, text, normalize=True):
    """lib/Twister.ml:91-188 in pure Python over the TEXT: parse, then one spectrum at a time by column NAME."""
    sp = pyref.parse_spectra(text)
    return [lab for lab, _ in sp], [pyref.twist(T.tolist(), names, lines, normalize) for _, lines in sp]


def test_a_twister_over_names_of_its_own(tmp_path, oracle, pyref):
    """lib/Twister.ml:71-76,151: spectra meet the twister's columns by NAME, and the names are any strings -- a reduced alphabet,
    another tool's k-mers.  Such a twister's columns get numbers through a dictionary; text spectra are twisted as the reference
    twists them (unknown names dropped before their value is read, :167-169; a repeated column name shadowed by its last, :73-76),
    and KPopCount's own reads stream meets it through the names KPopCount would have written."""
    d = 5
    names = ["AAB", "x|y", "0a", "0A", "a b", "0a1", "q", "AAB", "ffff", "1b"]  # "AAB" twice: the second one is the column
    rng = np.random.RandomState(21)
    T = np.array([[float("%.15g" % x) for x in row] for row in rng.randn(d, len(names))])
    dims = ["Dim%d" % (i + 1) for i in range(d)]
    write_table(tmp_path / "O.KPopTwister.txt", names, dims, T)
    write_table(tmp_path / "O.KPopInertia.txt", dims, ["inertia"], [oracle.synth_inertia(d)])
    text = ('\ts1\nAAB\t3\nx|y\t2.5\n0a\t1\n0A\t4\nnot-a-column\tthree\nAAB\t1e0\n'
            '\t"s2"\na b\t7\nq\t0\nzz\t9\n0a10\t4\n'
            '\tempty\n'
            '\tonly-unknown\nzz\t1\n'
            '\ts5\nffff\t2\n1b\t3\n0a1\t5\n')
    sp = tmp_path / "o.KPopSpectra.txt"
    sp.write_text(text)
    for extra in ([], ["--counts-normalize", "false"]):
        r = run([TWISTDB, "-I", "T", str(tmp_path / "O")] + extra + ["-k", str(sp), "-O", "t", "/dev/stdout"])
        assert r.returncode == 0, r.stderr
        labels, rows = _pyref_rows(pyref, T, names, text, normalize=not extra)
        want = twisted_text(dims, labels, rows)
        assert r.stdout.splitlines()[0] == want.splitlines()[0]
        for g, x in zip(r.stdout.splitlines()[1:], want.splitlines()[1:]):
            gf, xf = g.split("\t"), x.split("\t")
            assert gf[0] == xf[0]
            np.testing.assert_allclose([float(v) for v in gf[1:]], [float(v) for v in xf[1:]], rtol=1e-13, atol=1e-300)
    # many small blocks and threads: the same text
    r2 = run([TWISTDB, "-I", "T", str(tmp_path / "O"), "-k", str(sp), "-O", "t", "/dev/stdout"],
             env=dict(os.environ, KPOP_TEXT_BLOCK="16", KPOP_HOST_THREADS="5", KPOP_HOST_CHUNK="8"))
    r1 = run([TWISTDB, "-I", "T", str(tmp_path / "O"), "-k", str(sp), "-O", "t", "/dev/stdout"])
    assert r2.returncode == 0 and r2.stdout == r1.stdout
    # a value that is not a float is an error on a column's line only (:153-157)
    sp.write_text("\ta\nx|y\tthree\n")
    r = run([TWISTDB, "-I", "T", str(tmp_path / "O"), "-k", str(sp)])
    assert r.returncode == 1 and 'Float_expected("three")' in r.stderr
    # KPopCount | KPopTwistDB: the reads stream's k-mers by the names KPopCount writes for k=3 and 4 ("0a", "1b": two hex digits)
    write_fasta(tmp_path / "r.fa", READS)
    penv = dict(os.environ, PATH=BIN + ":" + os.environ.get("PATH", ""))
    for k in (4, 3, 7):
        r = subprocess.run(["bash", "-c", "KPopCount -k %d -L -f r.fa | KPopTwistDB -I T O -k /dev/stdin -O t /dev/stdout" % k],
                           cwd=str(tmp_path), capture_output=True, text=True, env=penv)
        assert r.returncode == 0, r.stderr
        labels, rows = _pyref_rows(pyref, T, names, expected_spectra(pyref, READS, k))
        if k != 7:
            assert any(any(v != 0 for v in row) for row in rows)
        want = twisted_text(dims, labels, rows)
        for g, x in zip(r.stdout.splitlines()[1:], want.splitlines()[1:]):
            gf, xf = g.split("\t"), x.split("\t")
            assert gf[0] == xf[0]
            np.testing.assert_allclose([float(v) for v in gf[1:]], [float(v) for v in xf[1:]], rtol=1e-13, atol=1e-300)
        assert len(r.stdout.splitlines()) == len(want.splitlines())


def test_names_are_strings_for_a_twister_of_kmer_hashes_too(tmp_path, oracle, pyref):
    """A twister KPopTwist wrote (lowercase hexadecimal names): an uppercase spelling is another name (Hashtbl over strings,
    lib/Twister.ml:151), and a bad value on a line whose name is no column of the twister is never read (:153-157,167-169)."""
    k, d = 4, 3
    cols, T, dims, w = make_twister(tmp_path, oracle, k, d, keep=0.7)
    names = [oracle.to_hex(h, k) for h in cols]
    inside = [n for n in names if any(c in "abcdef" for c in n)][:3]
    outside = [oracle.to_hex(h, k) for h in oracle.enumerate_kmers(k) if oracle.to_hex(h, k) not in set(names)][:2]
    assert len(inside) == 3 and len(outside) == 2
    text = ("\tu\n%s\t3\n%s\t5\n%s\t2\n%s\tthree\nzz\tfour\n%s\t1\n" % (inside[0], inside[1].upper(), inside[2], outside[0], outside[1]))
    sp = tmp_path / "u.KPopSpectra.txt"
    sp.write_text(text)
    r = run([TWISTDB, "-I", "T", str(tmp_path / "Classes"), "-k", str(sp), "-O", "t", "/dev/stdout"])
    assert r.returncode == 0, r.stderr
    labels, rows = _pyref_rows(pyref, T, names, text)
    want = twisted_text(dims, labels, rows)
    gf, xf = r.stdout.splitlines()[1].split("\t"), want.splitlines()[1].split("\t")
    assert gf[0] == xf[0] == '"u"'
    np.testing.assert_allclose([float(v) for v in gf[1:]], [float(v) for v in xf[1:]], rtol=1e-13)
    sp.write_text("\tu\n%s\tthree\n" % inside[0])
    r = run([TWISTDB, "-I", "T", str(tmp_path / "Classes"), "-k", str(sp)])
    assert r.returncode == 1 and 'Float_expected("three")' in r.stderr
