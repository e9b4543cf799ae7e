"""CPU tests of the host-side drop-ins (kpop_amd/host): text formats and argument handling of the two CLIs.
Only I/O actions run here; anything that computes needs the GPU (tests/test_gpu_cli.py)."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, load_golden

BIN = os.path.join(ROOT, "kpop_amd", "bin")
TWISTDB = os.path.join(BIN, "KPopTwistDB")
COUNT = os.path.join(BIN, "KPopCount")

pytestmark = pytest.mark.skipif(not os.path.exists(TWISTDB), reason="host CLIs not built (run __graft_entry__.build())")


def run(cmd, **kw):
    return subprocess.run(cmd, capture_output=True, text=True, timeout=120, **kw)


def write_table(path, cols, rows, data, quote=True, fmt="%.15g"):
    q = '"' if quote else ""
    with open(path, "w") as f:
        f.write('""' + "".join("\t%s%s%s" % (q, c, q) for c in cols) + "\n")
        for r, vals in zip(rows, data):
            f.write("%s%s%s" % (q, r, q) + "".join("\t" + fmt % v for v in vals) + "\n")


def test_twisted_table_round_trip_readme_row(tmp_path):
    """README.md:620-624: header '""  "Dim1" ...', rows '"121"  0.4614...' at %.15g."""
    kat = load_golden("readme_kat.json")
    src = tmp_path / "in.KPopTwisted.txt"
    src.write_text('""\t' + "\t".join('"%s"' % c for c in kat["twisted_header"]) + "\n" +
                   '"%s"\t' % kat["twisted_row_label"] + "\t".join(kat["twisted_row_text"]) + "\n")
    r = run([TWISTDB, "-I", "t", str(tmp_path / "in"), "-O", "t", str(tmp_path / "out")])
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "out.KPopTwisted.txt").read_text() == src.read_text()
    # precision option and /dev/stdout naming (lib/Matrix.ml:309-311)
    r = run([TWISTDB, "-I", "t", str(tmp_path / "in"), "--precision-for-tables", "3", "-O", "t", "/dev/stdout"])
    assert r.stdout.splitlines()[1].split("\t")[:3] == ['"121"', "0.461", "0.568"]


def test_add_tables_and_unquoted_input(tmp_path):
    write_table(tmp_path / "a.KPopTwisted.txt", ["Dim1", "Dim2"], ["x", "y"], [[1, 2], [3, 4]], quote=False)
    write_table(tmp_path / "b.KPopTwisted.txt", ["Dim1", "Dim2"], ["z"], [[5, 6.5]])
    r = run([TWISTDB, "-I", "t", str(tmp_path / "a"), "-A", "t", str(tmp_path / "b"), "-O", "t", "/dev/stdout"])
    assert r.returncode == 0, r.stderr
    assert r.stdout == '""\t"Dim1"\t"Dim2"\n"x"\t1\t2\n"y"\t3\t4\n"z"\t5\t6.5\n'
    write_table(tmp_path / "c.KPopTwisted.txt", ["Dim1", "DimX"], ["w"], [[0, 0]])
    r = run([TWISTDB, "-I", "t", str(tmp_path / "a"), "-A", "t", str(tmp_path / "c")])
    assert r.returncode == 1 and "Incompatible_geometries" in r.stderr
    r = run([TWISTDB, "-I", "t", str(tmp_path / "a"), "-z", "t", "-O", "t", "/dev/stdout"])
    assert r.stdout == '""\n'


def test_twister_tables_and_metric(tmp_path, oracle):
    """-I T reads .KPopTwister.txt + .KPopInertia.txt and checks them (lib/Twister.ml:32-51); -O m writes the
    metric induced by the inertia (lib/Twister.ml:208-217) -- O(D) host arithmetic, no GPU."""
    d, k = 4, 3
    cols = oracle.enumerate_kmers(k)
    T = oracle.synth_twister(1, d, cols)
    dims = ["Dim%d" % (i + 1) for i in range(d)]
    names = [oracle.to_hex(h, k) for h in cols]
    w = oracle.synth_inertia(d)
    write_table(tmp_path / "tw.KPopTwister.txt", names, dims, T)
    write_table(tmp_path / "tw.KPopInertia.txt", dims, ["inertia"], [w])
    r = run([TWISTDB, "-I", "T", str(tmp_path / "tw"), "-O", "T", str(tmp_path / "o"), "-O", "m", str(tmp_path / "o"),
             "-m", "flat", "-O", "m", "/dev/stdout"])
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "o.KPopTwister.txt").read_text() == (tmp_path / "tw.KPopTwister.txt").read_text()
    got = (tmp_path / "o.KPopMetrics.txt").read_text().splitlines()
    assert got[0].split("\t")[1:] == ['"%s"' % x for x in dims]
    m = oracle.metric_powers([float("%.15g" % x) for x in w], 1.0, 1.0, 2.0)  # the inertia went through %.15g text
    assert got[1] == '"metrics"' + "".join("\t%.15g" % x for x in m)
    assert r.stdout.splitlines()[1] == '"metrics"' + "\t0.25" * 4  # flat metric
    # mismatched files
    write_table(tmp_path / "bad.KPopTwister.txt", names, dims, T)
    write_table(tmp_path / "bad.KPopInertia.txt", dims[::-1], ["inertia"], [w])
    r = run([TWISTDB, "-I", "T", str(tmp_path / "bad")])
    assert r.returncode == 1 and "Mismatched_twister_files" in r.stderr


def test_argument_errors():
    r = run([TWISTDB, "-k", "x.txt"])
    assert r.returncode == 1 and "requires a twister" in r.stderr  # bin/KPopTwistDB.ml:373-377
    r = run([TWISTDB, "-d", "x"])
    assert r.returncode == 1 and "require a twister" in r.stderr   # :378-383
    r = run([TWISTDB, "-I", "m", "x"])
    assert r.returncode == 1 and "cannot load content into the metric" in r.stderr
    r = run([TWISTDB, "--distance", "manhattan"])
    assert r.returncode == 1 and "Unknown_distance" in r.stderr
    r = run([TWISTDB, "-m", "powers(1,2,1)"])
    assert r.returncode == 1 and "Invalid_threshold" in r.stderr
    r = run([TWISTDB, "--summary-keep-at-most", "0"])
    assert r.returncode == 1 and "Invalid_keep_at_most" in r.stderr
    r = run([TWISTDB, "-i", "t", "whatever"])
    assert r.returncode == 1 and "Marshal" in r.stderr
    assert run([TWISTDB]).returncode == 0          # empty program: usage, exit 0 (:362-365)
    assert run([TWISTDB, "-V"]).stdout.strip() == "38-hip"
    r = run([COUNT, "-f", "x.fa"])
    assert r.returncode == 1 and "mandatory" in r.stderr            # bin/KPopCount.ml:213-214
    r = run([COUNT, "-L", "-f", "a.fa", "-s", "b.fq"])
    assert r.returncode == 1 and "FASTA and FASTQ" in r.stderr      # :236
    r = run([COUNT, "-l", 'a"b', "-f", "a.fa"])
    assert r.returncode == 1 and "must not contain quotes" in r.stderr  # :171
    r = run([COUNT, "-L", "-k", "31", "-f", "a.fa"])
    assert r.returncode == 1
    assert run([COUNT, "-L"]).returncode == 0                        # no inputs: nothing to do (:218)


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_count_fails_loudly_without_gpu(tmp_path):
    fa = tmp_path / "x.fa"
    fa.write_text(">r1\nACGTACGT\n")
    r = run([COUNT, "-L", "-k", "3", "-f", str(fa)])
    assert r.returncode == 1 and "no HIP device" in r.stderr and r.stdout == ""
