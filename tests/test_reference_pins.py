"""Files written by a REAL KPop (tools/pin_from_reference.md) against this repository.

The oracle half runs on CPU: spectra compared as {label: {k-mer name: count}} (the reference prints in Hashtbl order), the
twisted table, the metric vectors, distances and summary against the oracle's restatement.  The binary registers written by
OCaml go through the drop-in's Marshal reader (no GPU needed).  With no tests/golden/ref/ files -- the state of this
checkout: the reference cannot be built here -- every test SKIPS and says what stays unpinned."""
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, concat
from test_cli import TWISTDB

PIN = os.path.join(GOLDEN, "pin")
REF = os.environ.get("KPOP_REF_DIR", os.path.join(GOLDEN, "ref"))  # the override: the kit rehearsed with the drop-ins, below


def ref(name, what):
    p = os.path.join(REF, name)
    if not os.path.exists(p):
        pytest.skip("parity unpinned (%s): no reference-produced %s -- see tools/pin_from_reference.md" % (what, name))
    return p


def read_fasta(path):
    out, tag, seq = [], None, []
    for line in open(path):
        line = line.rstrip("\n")
        if line.startswith(">"):
            if tag is not None:
                out.append((tag, "".join(seq)))
            tag, seq = line[1:], []
        else:
            seq.append(line)
    if tag is not None:
        out.append((tag, "".join(seq)))
    return out


def parse_spectra(path):
    res, cur = {}, None
    for line in open(path):
        a, b = line.rstrip("\n").split("\t")
        if a == "":
            cur = res.setdefault(b.strip('"'), {})
        else:
            cur[a] = cur.get(a, 0) + int(b)
    return res


def lint(s):
    return s.upper().replace("-", "")


def read_table(path):
    lines = open(path).read().splitlines()
    cols = [c.strip('"') for c in lines[0].split("\t")[1:]]
    rows = [l.split("\t")[0].strip('"') for l in lines[1:]]
    data = np.array([[float(x) for x in l.split("\t")[1:]] for l in lines[1:]])
    return cols, rows, data


@pytest.mark.parametrize("fname,k,content", [("pin_reads_k4_L.KPopSpectra.txt", 4, "ds"), ("pin_reads_k4_L_ss.KPopSpectra.txt", 4, "ss")])
def test_per_sequence_spectra_encoding(pyref, fname, k, content):
    got = parse_spectra(ref(fname, "k-mer encoding, canonical form, linting"))
    for tag, seq in read_fasta(os.path.join(PIN, "pin_reads.fasta")):
        want = {pyref.to_hex(h, k): c for h, c in pyref.count_read(lint(seq), k, content == "ds").items()}
        assert got.get(tag, {}) == want, tag


@pytest.mark.parametrize("k", [5, 12, 17])
def test_merged_spectrum_of_the_references_own_genome(oracle, k):
    got = parse_spectra(ref("wuhan_k%d_l.KPopSpectra.txt" % k, "k-mer encoding at k=%d" % k))["wuhan"]
    seqs = [lint(s) for _, s in read_fasta(os.path.join(GOLDEN, "wuhan.fasta"))]
    bases, offs = concat(seqs)
    h, c, o = oracle.count_reads(bases, offs, k, per_read=False)
    assert got == {oracle.to_hex(int(a), k): int(b) for a, b in zip(h, c)}


def test_binary_twister_written_by_ocaml(tmp_path):
    path = ref("pin.KPopTwister", "Marshal layout of Matrix.Base.t")
    r = subprocess.run([TWISTDB, "-i", "T", path[:-len(".KPopTwister")], "-O", "T", str(tmp_path / "back")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    for ext in ("KPopTwister.txt", "KPopInertia.txt"):
        a, b = read_table(tmp_path / ("back." + ext)), read_table(os.path.join(PIN, "pin." + ext))
        assert a[0] == b[0] and a[1] == b[1] and np.array_equal(a[2], b[2])


@pytest.mark.gpu
def test_binary_counter_written_by_ocaml(tmp_path):
    """an OCaml-written .KPopCounter (Bigarray custom blocks) through the drop-in KPopCountDB: its spectra come back out"""
    from conftest import ROOT
    path = ref("pin_reads.KPopCounter", "Marshal layout of KMerDB.marshalled_t")
    countdb = os.path.join(ROOT, "kpop_amd", "bin", "KPopCountDB")
    r = subprocess.run([countdb, "-i", path[:-len(".KPopCounter")], "--counts-transform", "power", "-s", str(tmp_path / "back")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = parse_spectra_float(tmp_path / "back.KPopSpectra.txt")
    want = parse_spectra(ref("pin_reads_k4_L.KPopSpectra.txt", "k-mer encoding"))
    assert {l: v for l, v in got.items() if v} == {l: {k: float(c) for k, c in v.items()} for l, v in want.items() if v}


def parse_spectra_float(path):
    res, cur = {}, None
    for line in open(path):
        a, b = line.rstrip("\n").split("\t")
        if a == "":
            cur = res.setdefault(b.strip('"'), {})
        else:
            cur[a] = float(b)
    return res


def test_twisted_rows_order_of_additions(oracle):
    cols, rows, want = read_table(ref("pin_reads.KPopTwisted.txt", "sparse mat-vec addition order, found-only normaliser"))
    tcols, dims, T = read_table(os.path.join(PIN, "pin.KPopTwister.txt"))
    assert cols == dims
    k = 4
    reads = dict(read_fasta(os.path.join(PIN, "pin_reads.fasta")))
    col_hash = np.array([int(c, 16) for c in tcols], dtype=np.uint64)
    for i, tag in enumerate(rows):
        bases, offs = concat([lint(reads[tag])])
        h, c, o = oracle.count_reads(bases, offs, k)
        got = oracle.twist(T, col_hash, h, c.astype(np.float64), o)[0]
        assert ["%.15g" % x for x in got] == ["%.15g" % x for x in want[i]], tag
    assert rows == sorted(rows, key=lambda s: s.encode())  # bytewise label order (lib/Twister.ml:197-204)


@pytest.mark.parametrize("fname,args", [("pin_default.KPopMetrics.txt", (1.0, 1.0, 2.0)), ("pin_thresholded.KPopMetrics.txt", (2.0, 0.7, 1.0))])
def test_powers_metric(oracle, fname, args):
    _, _, want = read_table(ref(fname, "Numbers.Frequencies.Vector behind the powers metric"))
    got = oracle.metric_powers(np.array([0.5, 0.3, 0.2]), *args)
    assert ["%.15g" % x for x in got] == ["%.15g" % x for x in want[0]]


def test_distances_on_top(oracle):
    _, rows, tw = read_table(ref("pin_reads.KPopTwisted.txt", "twist"))
    cols, rows2, want = read_table(ref("pin_reads.KPopDMatrix.txt", "distance cross-check"))
    metric = oracle.metric_powers(np.array([0.5, 0.3, 0.2]))
    got = oracle.distance_rowwise(tw, tw, metric)
    assert cols == rows and rows2 == rows
    np.testing.assert_allclose(got, want, rtol=1e-13, atol=1e-15)  # the inputs are %.15g-rounded


@pytest.mark.gpu
def test_pin_kit_rehearsal_with_the_dropins(tmp_path):
    """The commands of tools/pin_from_reference.md, run verbatim with the drop-in binaries standing in for the real tools:
    the kit's commands are well-formed and every comparison above passes on their output (which pins nothing -- it is this
    repository agreeing with itself -- but shows the day-one behaviour of the kit)."""
    import re
    import sys
    from conftest import ROOT
    md = open(os.path.join(ROOT, "tools", "pin_from_reference.md")).read()
    script = re.search(r"```bash\n(.*?)```", md, re.S).group(1)
    script = script.replace("R=tests/golden/ref", "R=%s" % tmp_path)
    env = dict(os.environ, PATH=os.path.join(ROOT, "kpop_amd", "bin") + os.pathsep + os.environ["PATH"], KPOP_PIPE_FORMAT="text")
    r = subprocess.run(["bash", "-e", "-c", script], cwd=ROOT, env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-k", "not rehearsal", os.path.abspath(__file__)], cwd=ROOT,
                       env=dict(os.environ, KPOP_REF_DIR=str(tmp_path)), capture_output=True, text=True)
    assert r.returncode == 0 and "skipped" not in r.stdout.splitlines()[-1], r.stdout[-3000:]
