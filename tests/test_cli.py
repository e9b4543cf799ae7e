"""CPU tests of the host-side drop-ins (kpop_amd/host): text formats and argument handling of the two CLIs.
Only I/O actions run here; anything that computes needs the GPU (tests/test_gpu_cli.py)."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, load_golden

BIN = os.environ.get("KPOP_TEST_BIN", os.path.join(ROOT, "kpop_amd", "bin"))  # e.g. an ASan build of kpop_amd/host
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
    r = run([TWISTDB, "-i", "t", "/nonexistent/whatever"])
    assert r.returncode == 1 and "cannot open" in r.stderr
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


# ---------------------------------------------------------------- binary registers (OCaml Marshal)
def be(v, n):
    return int(v).to_bytes(n, "big")


def marshal_header(data, n_obj, s32, s64):
    return be(0x8495A6BE, 4) + be(len(data), 4) + be(n_obj, 4) + be(s32, 4) + be(s64, 4)


def hand_marshalled_twisted():
    """{col_names=[|"Dim1";"Dim2"|]; row_names=[|"a";"b"|]; data=[|[|1.;2.|];[|3.;4.|]|]} preceded by
    "KPopTwisted" and "2022-04-03", assembled by hand from the OCaml runtime's extern.c rules."""
    import struct
    v1 = bytes([0x20 + 11]) + b"KPopTwisted"
    v2 = bytes([0x20 + 10]) + b"2022-04-03"
    rec = bytes([0xB0])                                                   # block tag 0, 3 fields
    rec += bytes([0xA0]) + bytes([0x24]) + b"Dim1" + bytes([0x24]) + b"Dim2"
    rec += bytes([0xA0]) + bytes([0x21]) + b"a" + bytes([0x21]) + b"b"
    rec += bytes([0xA0])
    rec += bytes([0x0E, 2]) + struct.pack("<2d", 1.0, 2.0)                  # CODE_DOUBLE_ARRAY8_LITTLE
    rec += bytes([0x0E, 2]) + struct.pack("<2d", 3.0, 4.0)
    return (marshal_header(v1, 1, 4, 3) + v1 + marshal_header(v2, 1, 4, 3) + v2 + marshal_header(rec, 10, 33, 27) + rec)


def test_binary_register_bytes_match_the_ocaml_format(tmp_path):
    write_table(tmp_path / "x.KPopTwisted.txt", ["Dim1", "Dim2"], ["a", "b"], [[1, 2], [3, 4]])
    r = run([TWISTDB, "-I", "t", str(tmp_path / "x"), "-o", "t", str(tmp_path / "x")])
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "x.KPopTwisted").read_bytes() == hand_marshalled_twisted()


def test_binary_register_reader_honours_sharing_and_wide_codes(tmp_path):
    """What OCaml may emit and our writer never does: shared strings (CODE_SHARED8), BLOCK32, STRING8,
    DOUBLE_ARRAY32 and the 32-byte big header."""
    import struct
    v1 = bytes([0x20 + 11]) + b"KPopTwisted"
    v2 = bytes([0x20 + 10]) + b"2022-04-03"
    rec = bytes([0xB0])
    rec += bytes([0x08]) + be((2 << 10) | 0, 4) + bytes([0x09, 4]) + b"Dim1" + bytes([0x24]) + b"Dim2"  # BLOCK32, STRING8
    # objects so far: record 0, column array 1, "Dim1" 2, "Dim2" 3, row array 4 -> intern_obj_table[5 - 3] is "Dim1"
    rec += bytes([0xA0]) + bytes([0x04, 3]) + bytes([0x21]) + b"b"
    rec += bytes([0xA0])
    rec += bytes([0x07]) + be(2, 4) + struct.pack("<2d", 1.5, -2.0)   # DOUBLE_ARRAY32_LITTLE
    rec += bytes([0x0D, 2]) + struct.pack(">2d", 3.0, 4.25)            # DOUBLE_ARRAY8_BIG
    big = be(0x8495A6BF, 4) + be(0, 4) + be(len(rec), 8) + be(9, 8) + be(25, 8)
    (tmp_path / "y.KPopTwisted").write_bytes(marshal_header(v1, 1, 4, 3) + v1 + marshal_header(v2, 1, 4, 3) + v2 + big + rec)
    r = run([TWISTDB, "-i", "t", str(tmp_path / "y"), "-O", "t", "/dev/stdout"])
    assert r.returncode == 0, r.stderr
    assert r.stdout == '""\t"Dim1"\t"Dim2"\n"Dim1"\t1.5\t-2\n"b"\t3\t4.25\n'


def test_binary_round_trips_and_checks(tmp_path, oracle):
    d, k = 5, 4
    cols = oracle.enumerate_kmers(k)
    T = oracle.synth_twister(2, d, cols)
    dims = ["Dim%d" % (i + 1) for i in range(d)]
    write_table(tmp_path / "c.KPopTwister.txt", [oracle.to_hex(h, k) for h in cols], dims, T)
    write_table(tmp_path / "c.KPopInertia.txt", dims, ["inertia"], [oracle.synth_inertia(d)])
    r = run([TWISTDB, "-I", "T", str(tmp_path / "c"), "-o", "T", str(tmp_path / "c"), "-z", "T", "-i", "T", str(tmp_path / "c"),
             "-O", "T", str(tmp_path / "c2")])
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "c2.KPopTwister.txt").read_text() == (tmp_path / "c.KPopTwister.txt").read_text()
    assert (tmp_path / "c2.KPopInertia.txt").read_text() == (tmp_path / "c.KPopInertia.txt").read_text()
    # -a merges rows; an empty register round-trips (three atoms)
    write_table(tmp_path / "p.KPopTwisted.txt", dims, ["s1"], [np.arange(d) / 7.0])
    write_table(tmp_path / "q.KPopTwisted.txt", dims, ["s2", "s3"], [np.ones(d), -np.ones(d)])
    for n in "pq":
        assert run([TWISTDB, "-I", "t", str(tmp_path / n), "-o", "t", str(tmp_path / n)]).returncode == 0
    r = run([TWISTDB, "-i", "t", str(tmp_path / "p"), "-a", "t", str(tmp_path / "q"), "-O", "t", "/dev/stdout"])
    assert [l.split("\t")[0] for l in r.stdout.splitlines()] == ['""', '"s1"', '"s2"', '"s3"']
    r = run([TWISTDB, "-z", "d", "-o", "d", str(tmp_path / "e"), "-i", "d", str(tmp_path / "e"), "-O", "d", "/dev/stdout"])
    assert r.returncode == 0 and r.stdout == '""\n'
    # type and version checks (lib/Matrix.ml:831-832,841-842)
    r = run([TWISTDB, "-i", "d", str(tmp_path / "p.KPopTwisted").replace(".KPopTwisted", "")])
    assert r.returncode == 1
    os.rename(tmp_path / "p.KPopTwisted", tmp_path / "p.KPopDMatrix")
    r = run([TWISTDB, "-i", "d", str(tmp_path / "p")])
    assert r.returncode == 1 and "Unexpected_type" in r.stderr
    raw = (tmp_path / "q.KPopTwisted").read_bytes().replace(b"2022-04-03", b"2021-01-01")
    (tmp_path / "q.KPopTwisted").write_bytes(raw)
    r = run([TWISTDB, "-i", "t", str(tmp_path / "q")])
    assert r.returncode == 1 and "Incompatible_archive_version" in r.stderr


# ---------------------------------------------------------------- KPopCountDB: host-only actions and '.KPopCounter'
COUNTDB = os.path.join(BIN, "KPopCountDB")
TWIST = os.path.join(BIN, "KPopTwist")


def hand_marshalled_counter():
    """"KPopCounter", "2022-04-03", then { n_cols=2; n_rows=2; n_meta=1; [|"A";"B"|]; [|"aa";"ac"|]; [|"class"|];
    [|[|"X"|];[|"Y"|]|]; [| int32 Bigarray [3;5]; [0;7] |] } assembled by hand from extern.c / bigarray.c:
    BLOCK32 for the 8-field record, CUSTOM_LEN + "_bigarr02" + sizes (20, 40) + num_dims, kind INT32 (6), a 2-byte
    dimension and big-endian elements for each spectrum."""
    v1 = bytes([0x20 + 11]) + b"KPopCounter"
    v2 = bytes([0x20 + 10]) + b"2022-04-03"
    rec = bytes([0x08]) + be((8 << 10) | 0, 4) + bytes([0x42, 0x42, 0x41])
    rec += bytes([0xA0, 0x21]) + b"A" + bytes([0x21]) + b"B"
    rec += bytes([0xA0, 0x22]) + b"aa" + bytes([0x22]) + b"ac"
    rec += bytes([0x90, 0x25]) + b"class"
    rec += bytes([0xA0, 0x90, 0x21]) + b"X" + bytes([0x90, 0x21]) + b"Y"
    rec += bytes([0xA0])
    for col in ((3, 5), (0, 7)):
        rec += bytes([0x18]) + b"_bigarr02\0" + be(20, 4) + be(40, 8) + be(1, 4) + be(6, 4) + be(2, 2)
        rec += b"".join(be(v, 4) for v in col)
    return marshal_header(v1, 1, 4, 3) + v1 + marshal_header(v2, 1, 4, 3) + v2 + marshal_header(rec, 17, 56, 55) + rec


def test_counter_archive_bytes_and_host_actions(tmp_path):
    (tmp_path / "s.KPopSpectra.txt").write_text("\tA\naa\t3\nac\t5\n\t\"B\"\nac\t2\nac\t0x5\n")
    (tmp_path / "meta.txt").write_text("label\tclass\nA\tX\n\"B\"\t\"Y\"\n")
    r = run([COUNTDB, "-k", str(tmp_path / "s"), "-m", str(tmp_path / "meta.txt"), "-o", str(tmp_path / "db"), "--summary", "-v"])
    assert r.returncode == 0, r.stderr
    assert "[Spectrum labels (2)]: 'A' 'B'\n[K-mer hashes (2)]: 'aa' 'ac'\n[Meta-data fields (1)]: 'class'\n" in r.stderr
    assert (tmp_path / "db.KPopCounter").read_bytes() == hand_marshalled_counter()
    # selection register: regexps on labels and metadata (anchored at the start only), negation, explicit labels
    r = run([COUNTDB, "-i", str(tmp_path / "db"), "-R", "~.", "-P", "-R", "class~X", "-P", "-N", "-P", "-L", "q,A", "-P", "-C", "-P",
             "-R", "~A\\|B,class~[^X]", "-P", "-R", "nofield~.", "-P"])
    assert r.returncode == 0, r.stderr
    sel = [l for l in r.stderr.splitlines() if l.startswith("Currently selected")]
    assert sel == ["Currently selected spectra = [ 'A' 'B' ].", "Currently selected spectra = [ 'A' ].",
                   "Currently selected spectra = [ 'B' ].", "Currently selected spectra = [ 'A' 'q' ].",
                   "Currently selected spectra = [ ].", "Currently selected spectra = [ 'B' ].", "Currently selected spectra = [ ]."]
    # -D and -e are host-only too
    r = run([COUNTDB, "-i", str(tmp_path / "db"), "-L", "A", "-D", "--summary", "-e", "--summary"])
    assert "[Spectrum labels (1)]: 'B'" in r.stderr and "[Spectrum labels (0)]:" in r.stderr


def test_counter_archive_reader_accepts_what_ocaml_may_emit(tmp_path):
    """Old-style Bigarray blocks (CODE_CUSTOM, "_bigarray", 4-byte dimensions), shared strings, name buffers longer than
    the declared sizes."""
    v1 = bytes([0x20 + 11]) + b"KPopCounter"
    v2 = bytes([0x20 + 10]) + b"2022-04-03"
    rec = bytes([0x08]) + be((8 << 10) | 0, 4) + bytes([0x41, 0x42, 0x41])
    rec += bytes([0xA0, 0x21]) + b"A" + bytes([0x20])                      # one spare (empty) label slot
    rec += bytes([0xA0, 0x22]) + b"aa" + bytes([0x22]) + b"ac"
    rec += bytes([0x90, 0x21]) + b"m"
    # objects so far: record 0, labels 1, "A" 2, "" 3, k-mers 4, "aa" 5, "ac" 6, fields 7, "m" 8, meta 9, its row 10
    rec += bytes([0x90, 0x90, 0x04, 9])                                     # -> the value shares the string "A" (11 - 9)
    rec += bytes([0x90, 0x12]) + b"_bigarray\0" + be(1, 4) + be(6, 4) + be(3, 4) + be(9, 4) + be(2**32 - 4, 4) + be(0, 4)
    (tmp_path / "old.KPopCounter").write_bytes(marshal_header(v1, 1, 4, 3) + v1 + marshal_header(v2, 1, 4, 3) + v2 +
                                                marshal_header(rec, 12, 0, 0) + rec)
    r = run([COUNTDB, "-i", str(tmp_path / "old"), "--summary", "-v", "-R", "m~^A$", "-P"])
    assert r.returncode == 0, r.stderr
    assert "[Spectrum labels (1)]: 'A'" in r.stderr and "[K-mer hashes (2)]: 'aa' 'ac'" in r.stderr
    assert "Currently selected spectra = [ 'A' ]." in r.stderr


def test_countdb_errors(tmp_path):
    def fails(args, what, code=1):
        r = run([COUNTDB] + args)
        assert r.returncode == code and what in r.stderr and r.stdout == "", (args, r.stderr)
    fails(["--nonsense"], "Unknown option")
    fails(["-R", "a~b~c"], "Wrong number of fields in list (expected 2, found 3)")
    fails(["--combination-criterion", "mode"], "Unknown_combination_criterion")
    fails(["--table-transpose", "maybe"], "is not a boolean")
    fails(["--counts-threshold", "-1"], "non-negative")
    fails(["-i", str(tmp_path / "missing")], "cannot open")
    (tmp_path / "bad1.KPopSpectra.txt").write_text("aa\t3\n")
    fails(["-k", str(tmp_path / "bad1")], "Header_expected")
    (tmp_path / "bad2.KPopSpectra.txt").write_text("\tA\naa\t3\t4\n")
    fails(["-k", str(tmp_path / "bad2")], "Wrong_number_of_columns(2, 3, 2)")
    (tmp_path / "bad3.KPopSpectra.txt").write_text("\tA\naa\t3.5\n")
    fails(["-k", str(tmp_path / "bad3")], "Wrong_format(2, \"3.5\")")
    (tmp_path / "bad4.KPopSpectra.txt").write_text("\tA\"B\naa\t3\n")
    fails(["-k", str(tmp_path / "bad4")], "Quotes_in_name")
    (tmp_path / "x.KPopCounter").write_bytes(b"\x84\x95\xa6\xbe" + b"\0" * 3)
    fails(["-i", str(tmp_path / "x")], "truncated")
    (tmp_path / "meta.txt").write_text("label\tclass\nA\n")
    fails(["-m", str(tmp_path / "meta.txt")], "Wrong_number_of_columns(2, 1, 2)")
    fails(["-e", "-d", "class", "out"], "not supported")
    r = run([COUNTDB])
    assert r.returncode == 0 and "Usage: KPopCountDB" in r.stdout
    r = run([TWIST, "-o", "x"])
    assert r.returncode == 1 and "Option '-i' is mandatory" in r.stderr
    r = run([TWIST, "-i", "x", "-o", "y", "--counts-transform", "sqrt"])
    assert r.returncode == 1 and "Unknown_transformation" in r.stderr


def test_kpoptwist_underscore_echoes_its_arguments():
    """bin/KPopTwist_.ml:136-140: one line of \\001-separated fields for the bash wrapper src/KPopTwist:19-27."""
    r = run([os.path.join(BIN, "KPopTwist_"), "-i", "In", "-o", "Out", "-s", "0.5", "--counts-transform", "clr", "-K", "km", "-T", "3",
             "--kmers-threshold", "0.25", "--counts-normalize", "false", "-v"])
    assert r.returncode == 0
    assert r.stdout == "\001".join(["In", "", "0.5", "1", "1", "clr", "false", "0.25", "Out", "km", "3", "false", "true"]) + "\n"
    r = run([os.path.join(BIN, "KPopTwist_"), "-i", "In"])
    assert r.returncode == 1 and "Option '-o' is mandatory" in r.stderr and r.stdout == ""


def test_large_archives_mapped_paths_equal_the_streamed_ones(tmp_path):
    """Archives of 16 MB and more are written through a mapping (rows laid down by the host threads) and read back from one;
    both must agree byte for byte with the general writer (a pipe) and reader (a pipe), and a twister archive too."""
    import numpy as np
    rng = np.random.RandomState(4)
    rows, cols = 36000, 64
    data = rng.randint(-40, 40, size=(rows, cols)) / 8.0
    names = ["row %d" % i for i in range(rows)]
    dims = ["Dim%d" % (i + 1) for i in range(cols)]
    write_table(tmp_path / "big.KPopTwisted.txt", dims, names, data)
    r = run([TWISTDB, "-I", "t", str(tmp_path / "big"), "-o", "t", str(tmp_path / "mapped")])
    assert r.returncode == 0, r.stderr
    piped = subprocess.run([TWISTDB, "-I", "t", str(tmp_path / "big"), "-o", "t", "/dev/stdout"], stdout=subprocess.PIPE)
    assert piped.returncode == 0
    blob = (tmp_path / "mapped.KPopTwisted").read_bytes()
    assert len(blob) > (16 << 20) and blob == piped.stdout
    # mapped reader vs the general one (which a pipe forces)
    r = run([TWISTDB, "-i", "t", str(tmp_path / "mapped"), "-O", "t", str(tmp_path / "back1")])
    assert r.returncode == 0, r.stderr
    r2 = subprocess.run([TWISTDB, "-i", "t", "/dev/stdin", "-O", "t", str(tmp_path / "back2")], input=blob)
    assert r2.returncode == 0
    t1 = (tmp_path / "back1.KPopTwisted.txt").read_bytes()
    assert t1 == (tmp_path / "back2.KPopTwisted.txt").read_bytes() == (tmp_path / "big.KPopTwisted.txt").read_bytes()
    # a twister: few rows, each millions of bytes (32-bit float-array prefix), inertia behind it
    n_k = 300000
    tw = rng.randint(-9, 9, size=(9, n_k)) / 4.0
    kd = ["Dim%d" % (i + 1) for i in range(9)]
    write_table(tmp_path / "T.KPopTwister.txt", ["%06x" % i for i in range(n_k)], kd, tw)
    write_table(tmp_path / "T.KPopInertia.txt", kd, ["inertia"], [np.arange(9, 0, -1) / 45.0])
    r = run([TWISTDB, "-I", "T", str(tmp_path / "T"), "-o", "T", str(tmp_path / "Tb")])
    assert r.returncode == 0, r.stderr
    assert os.path.getsize(tmp_path / "Tb.KPopTwister") > (16 << 20)
    r = run([TWISTDB, "-i", "T", str(tmp_path / "Tb"), "-O", "T", str(tmp_path / "Tc"), "-O", "m", str(tmp_path / "Tc")])
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "Tc.KPopTwister.txt").read_bytes() == (tmp_path / "T.KPopTwister.txt").read_bytes()
    assert (tmp_path / "Tc.KPopInertia.txt").read_bytes() == (tmp_path / "T.KPopInertia.txt").read_bytes()
    tb = (tmp_path / "Tb.KPopTwister").read_bytes()
    r3 = subprocess.run([TWISTDB, "-i", "T", "/dev/stdin", "-O", "T", str(tmp_path / "Td")], input=tb)
    assert r3.returncode == 0
    assert (tmp_path / "Td.KPopTwister.txt").read_bytes() == (tmp_path / "T.KPopTwister.txt").read_bytes()


def _decode_reads_stream(raw):
    """the reads stream of kpop_amd/host/fast_seq.h -> (k, content, [(tag, linted bases)])"""
    import struct
    assert raw[:8] == b"\0KPopRd1"
    k, content = struct.unpack_from("<II", raw, 8)
    at, recs = 16, []
    while True:
        n, zero = struct.unpack_from("<II", raw, at)
        at += 8
        if n == 0:
            break
        nb, nt = struct.unpack_from("<QQ", raw, at)
        at += 16
        lens = struct.unpack_from("<%dI" % n, raw, at)
        at += 4 * n
        tag_lens = struct.unpack_from("<%dI" % n, raw, at)
        at += 4 * n
        tags, bases = raw[at:at + nt], raw[at + nt:at + nt + nb]
        at += nt + nb
        tb = bb = 0
        for L, T in zip(lens, tag_lens):
            recs.append((tags[tb:tb + T], bases[bb:bb + L]))
            tb += T
            bb += L
    assert at == len(raw)
    return k, content, recs


def test_paired_end_through_the_block_reader(tmp_path):
    """KPopCount -p mates_1 mates_2 (bin/KPopCount.ml:36-54: segment 0, segment 1, ... alternate) reads both files block by
    block and deals their records alternately: the reads it hands on are those of the two files interleaved into one,
    whatever the block size; files with different numbers of reads are refused.  (No GPU: with KPOP_PIPE_FORMAT=reads
    KPopCount defers the counting and only lints.)"""
    import random
    rng = random.Random(4)
    n = 3000
    one, two, both = [], [], []
    for i in range(n):
        for m, dst in ((1, one), (2, two)):
            s = "".join(rng.choice("ACGTNacgtRY-") for _ in range(rng.randrange(0, 260)))
            rec = "@p%d/%d some text\n%s\n+\n%s\n" % (i, m, s, "I" * len(s))
            dst.append(rec)
            both.append(rec)
    (tmp_path / "a_1.fq").write_text("".join(one))
    (tmp_path / "a_2.fq").write_text("".join(two))
    (tmp_path / "inter.fq").write_text("".join(both))
    (tmp_path / "short_2.fq").write_text("".join(two[:-1]))
    base = dict(os.environ, KPOP_PIPE_FORMAT="reads")
    for block in (None, "300", "5000", "70000"):
        env = dict(base, KPOP_SEQ_BLOCK=block) if block else base
        sh = lambda cmd: subprocess.run(["bash", "-c", cmd], cwd=str(tmp_path), capture_output=True, env=env, timeout=120)
        a = sh("%s -k 7 -L -p a_1.fq a_2.fq | cat" % COUNT)
        b = sh("%s -k 7 -L -s inter.fq | cat" % COUNT)
        assert a.returncode == 0 and b.returncode == 0, (a.stderr, b.stderr)
        ka, ca, ra = _decode_reads_stream(a.stdout)
        kb, cb, rb = _decode_reads_stream(b.stdout)
        assert (ka, ca) == (kb, cb) == (7, 0) and len(ra) == 2 * n and ra == rb, block
        bad = sh("set -o pipefail; %s -k 7 -L -p a_1.fq short_2.fq | cat > /dev/null" % COUNT)
        assert bad.returncode != 0 and b"different numbers of reads" in bad.stderr


def test_countdb_block_parser_equals_line_parser(tmp_path):
    """KPopCountDB -k reads spectra a block at a time; blocks whose lines are all what KPopCount writes go through the threaded
    parser, the others line by line.  Whatever the block size and the thread count -- one block of everything (the old
    behaviour for small files), blocks of a spectrum or two, blocks cut so that plain and odd blocks alternate -- the database
    written is the same file, or the same error is raised (lib/KMerDB.ml:505-575)."""
    import random
    rng = random.Random(21)
    base = "".join("\tg%d\n" % i + "".join("%06x\t%d\n" % (rng.randrange(3000), rng.randrange(1, 500)) for _ in range(rng.randrange(0, 40)))
                   for i in range(30))
    oddities = ["\t", "\n", "\"", "x", "A", "-", "0x1F\n", "\r", "", "\t\t", "1_0", " ", "99999999999\n", "\tg3\n", "+5\n"]
    for it in range(60):
        s = list(base)
        for _ in range(rng.randrange(0, 5) if it else 0):
            s[rng.randrange(len(s))] = rng.choice(oddities)
        data = "".join(s)
        if it % 7 == 3:
            data = data.rstrip("\n")
        if it % 11 == 5:
            data = "abcdef\t1\n" + data
        (tmp_path / "in.KPopSpectra.txt").write_text(data)
        results = []
        for block, threads in (("lines", "1"), ("100000000", "1"), ("64", "4"), ("700", "3"), ("5000", "8")):
            env = dict(os.environ, KPOP_TEXT_BLOCK=block, KPOP_HOST_THREADS=threads, KPOP_HOST_CHUNK="50")
            if block == "lines":  # the reference: every line through the line parser
                env = dict(os.environ, KPOP_COUNTDB_LINES="1")
            out = tmp_path / ("db_%s" % block)
            r = subprocess.run([COUNTDB, "-k", str(tmp_path / "in"), "-o", str(out)], capture_output=True, text=True, errors="replace", env=env)
            db = (tmp_path / ("db_%s.KPopCounter" % block))
            results.append((r.returncode, r.stderr.strip().splitlines()[-1:] if r.returncode else [], db.read_bytes() if r.returncode == 0 else b""))
            if db.exists():
                db.unlink()
        assert all(x == results[0] for x in results[1:]), (it, [(a, b, len(c)) for a, b, c in results])
