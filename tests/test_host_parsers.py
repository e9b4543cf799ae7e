"""CPU test of the host text layer: the threaded spectra parser of the GPU path (read_spectra_hashed, chunks cut at line
boundaries and merged in file order) against the sequential parser (read_spectra_file) on mutated inputs -- same
result or the same error, whatever the chunking.  The harness is compiled from tests/host/parser_diff.cpp."""
import os
import random
import subprocess

import pytest

from conftest import ROOT

# KPOP_TEST_SANITIZE=1 (tools/sanitize_host.sh): the harnesses are built with AddressSanitizer + UBSan like kpop_amd/bin_asan
SAN = ["-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"] if os.environ.get("KPOP_TEST_SANITIZE") == "1" else []


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    out = tmp_path_factory.mktemp("parser_diff") / "parser_diff"
    src = os.path.join(ROOT, "tests", "host", "parser_diff.cpp")
    subprocess.run(["g++", "-O1", "-std=c++17", "-pthread"] + SAN + ["-o", str(out), src, os.path.join(ROOT, "kpop_amd", "host", "kpop_text.cpp")],
                   check=True)
    return str(out)


def test_threaded_parser_agrees_with_sequential_one(harness, tmp_path):
    rng = random.Random(11)
    base = "".join("\tr%d\n" % i + "".join("%03x\t%d\n" % (rng.randrange(4096), rng.randrange(1, 50)) for _ in range(rng.randrange(0, 12)))
                   for i in range(60))
    path = tmp_path / "in.txt"
    for threads, chunk in (("8", "40"), ("3", "1000"), ("1", "1")):
        env = dict(os.environ, KPOP_HOST_THREADS=threads, KPOP_HOST_CHUNK=chunk)
        for it in range(120):
            s = list(base)
            for _ in range(rng.randrange(0, 4)):
                s[rng.randrange(len(s))] = rng.choice(["\t", "\n", "\"", "x", "0", "-", "1e3", "\r", "", "\t\t", "nan", " "])
            data = "".join(s)
            if it % 7 == 0:
                data = data.rstrip("\n")
            if it % 11 == 0:
                data = "aaa\t1\n" + data
            if it % 13 == 0:
                data = ""
            path.write_text(data)
            block = rng.choice(["1", "17", "64", "300", "5000", "100000"])  # the block stream's block size (it grows as needed)
            r = subprocess.run([harness, str(path), "3", block], capture_output=True, text=True, errors="replace", env=env)
            assert r.returncode == 0, (threads, chunk, block, it, r.stdout, r.stderr)


@pytest.fixture(scope="module")
def seq_harness(tmp_path_factory):
    out = tmp_path_factory.mktemp("seq_diff") / "seq_diff"
    host = os.path.join(ROOT, "kpop_amd", "host")
    subprocess.run(["g++", "-O1", "-std=c++17", "-pthread"] + SAN + ["-o", str(out), os.path.join(ROOT, "tests", "host", "seq_diff.cpp"),
                    os.path.join(host, "kpop_text.cpp"), os.path.join(host, "fast_seq.cpp")], check=True)
    return str(out)


def test_block_reader_agrees_with_line_reader(seq_harness, tmp_path):
    """FastSeqReader (blocks cut at record boundaries, records linted by threads) against SeqReader (one line at a time)
    on mutated FASTA and FASTQ, whatever the block size and the thread count; the reads stream round-trips them."""
    rng = random.Random(5)

    def seq(n):
        return "".join(rng.choice("ACGTacgtNn-") for _ in range(n))
    fasta = "".join(">r%d some text\n" % i + "".join(seq(rng.randrange(0, 70)) + "\n" for _ in range(rng.randrange(0, 4))) for i in range(40))
    fastq = "".join("@q%d\n%s\n+\n%s\n" % (i, seq(rng.randrange(0, 80)), "I" * rng.randrange(0, 80)) for i in range(40))
    path = tmp_path / "in.txt"
    for threads, chunk, block in (("8", "50", "64"), ("3", "300", "200"), ("1", "1", "1000"), ("4", "100", "100000")):
        env = dict(os.environ, KPOP_HOST_THREADS=threads, KPOP_HOST_CHUNK=chunk, KPOP_SEQ_BLOCK=block)
        for fmt, base in (("fasta", fasta), ("fastq", fastq)):
            for it in range(80):
                s = list(base)
                for _ in range(rng.randrange(0, 4)):
                    s[rng.randrange(len(s))] = rng.choice([">", "\n", "@", "+", "\r", "\r\n", "", "\n\n", " ", "\t", ">x\n"])
                data = "".join(s)
                if it % 7 == 0:
                    data = data.rstrip("\n")
                if it % 11 == 0:
                    data = "\n\n" + data
                if it % 13 == 0:
                    data = ""
                if it % 17 == 0:
                    data = "ACGT\n" + data
                path.write_text(data)
                r = subprocess.run([seq_harness, str(path), fmt], capture_output=True, text=True, errors="replace", env=env)
                assert r.returncode == 0, (threads, chunk, block, fmt, it, r.stdout, r.stderr)


def test_row_ordering_matches_the_references_string_map(tmp_path):
    """order_rows_by_label (threads sort runs of row numbers, pairwise merges, duplicate detection) against a std::map
    restatement of lib/Twister.ml:78-82,189-204 on random labels: few letters (many duplicates, errors) and many"""
    out = tmp_path / "rows_merge"
    host = os.path.join(ROOT, "kpop_amd", "host")
    subprocess.run(["g++", "-O1", "-std=c++17", "-pthread", "-o", str(out), os.path.join(ROOT, "tests", "host", "rows_merge.cpp"),
                    os.path.join(host, "kpop_text.cpp")], check=True)
    for threads in ("1", "4", "16"):
        env = dict(os.environ, KPOP_HOST_THREADS=threads)
        for seed in range(12):
            for rows, alphabet in ((0, 3), (1, 3), (40, 2), (3000, 6), (70000, 26)):
                r = subprocess.run([str(out), str(seed), str(rows), str(alphabet)], capture_output=True, text=True, env=env)
                assert r.returncode == 0, (threads, seed, rows, alphabet, r.stdout)


def test_number_formatting_equals_printf(tmp_path):
    """append_g (std::to_chars, what the tables and summaries are now printed with) gives printf("%.*g")'s characters:
    random bit patterns, short decimals, halves, the edges of the exponent range, NaNs of either sign; precisions 15
    (the default), 17, 6, 1, and 0 / 20 (which fall back to printf itself)"""
    out = tmp_path / "format_g_check"
    host = os.path.join(ROOT, "kpop_amd", "host")
    subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", "-o", str(out), os.path.join(ROOT, "tests", "host", "format_g_check.cpp"),
                    os.path.join(host, "kpop_text.cpp")], check=True)
    for seed in (1, 2):
        r = subprocess.run([str(out), str(seed), "600000"], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout


def test_centroids_splits_match_the_python_restatement(tmp_path, pyref):
    """SplitsAlgorithm.Centroids (lib/Matrix.ml:361-521,601-612): the drop-in's host implementation against oracle/pyref.py's,
    both drawing from the declared SplitMix64 stream: same splits, same weights to the printed precision, same order"""
    import numpy as np
    from test_cli import write_table
    out = tmp_path / "centroids_run"
    host = os.path.join(ROOT, "kpop_amd", "host")
    subprocess.run(["g++", "-O1", "-std=c++17", "-pthread", "-ffp-contract=off", "-o", str(out), os.path.join(ROOT, "tests", "host", "centroids_run.cpp"),
                    os.path.join(host, "splits.cpp"), os.path.join(host, "kpop_text.cpp")], check=True)
    rng = np.random.RandomState(2)
    for n, d in ((1, 3), (2, 2), (7, 3), (40, 5), (90, 9)):
        centres = rng.normal(size=(4, d)) * 3
        emb = np.array([centres[i % 4] + rng.normal(size=d) * 0.3 for i in range(n)])
        emb = np.array([[float("%.15g" % x) for x in row] for row in emb])
        names = ["leaf %d" % i for i in range(n)]
        write_table(tmp_path / "e.txt", ["D%d" % i for i in range(d)], names, emb)
        r = subprocess.run([str(out), str(tmp_path / "e.txt")], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        want = pyref.splits_text(names, pyref.splits_centroids([list(map(float, row)) for row in emb]))
        assert r.stdout == want, (n, d)


@pytest.fixture(scope="module")
def lookup_harness(tmp_path_factory):
    out = tmp_path_factory.mktemp("lookup_parse") / "lookup_parse"
    subprocess.run(["g++", "-O1", "-std=c++17", "-pthread"] + SAN + ["-o", str(out), os.path.join(ROOT, "tests", "host", "lookup_parse.cpp"),
                    os.path.join(ROOT, "kpop_amd", "host", "kpop_text.cpp")], check=True)
    return str(out)


def _reference_lines(text, number_of):
    """lib/Twister.ml:97-118 and :151-169 line by line: two tab-separated fields (:103-104), a header first (:106-107), names are
    strings, a name that is no column is dropped before its value is read, a column's value must be a float (:153-157).  Of two
    errors in one file the first in file order is the one reported here (in the reference which one surfaces depends on how far
    the reader has run ahead of the workers)."""
    import sys
    sys.path.insert(0, ROOT)
    from oracle import pyref
    out = []
    lines = text.split("\n")
    if lines and lines[-1] == "":
        lines.pop()
    for n, line in enumerate(lines):
        f = line.split("\t")
        if len(f) != 2:
            return "ERROR", "Wrong_number_of_columns(%d, %d, 2)" % (n + 1, len(f))
        if n == 0 and f[0] != "":
            return "ERROR", "Header_expected"
        if f[0] == "":
            try:
                out.append("L " + pyref.strip_quotes(f[1]))
            except ValueError:
                return "ERROR", ""
            continue
        c = number_of(f[0])
        if c is None:
            out.append("-")
            continue
        try:
            v = float(f[1]) if not f[1].lower().startswith(("0x", "-0x")) else float.fromhex(f[1])  # OCaml's float_of_string reads hexadecimal too
        except ValueError:
            return "ERROR", 'Float_expected("%s")' % f[1]
        out.append("%d %.17g" % (c, v))
    return "OK", out


@pytest.mark.parametrize("mode", ["hex", "opaque"])
def test_parser_that_knows_the_columns_says_what_the_reference_says(lookup_harness, tmp_path, mode):
    rng = random.Random(3 if mode == "hex" else 4)
    if mode == "hex":
        names = sorted({"%03x" % rng.randrange(4096) for _ in range(900)})
        table = {n: int(n, 16) for n in names}
    else:
        names = ["%03x" % rng.randrange(4096) for _ in range(300)] + ["AAB", "x|y", "0A", "a b", "k-%d" % 7, "q", "AAB", "0a1f", ""]
        table = {n: i for i, n in enumerate(names)}  # the last of several columns of one name
    (tmp_path / "names.txt").write_text("".join(n + "\n" for n in names if n != ""))
    if mode == "opaque":
        table.pop("", None)
    pool = names[:50] + ["zz", "0A", "ABC", "abc", "fff", "000", "AAB", "x|y", "1234"]
    base = "".join("\tr%d\n" % i + "".join("%s\t%s\n" % (rng.choice(pool), rng.choice(["1", "17", "2.5", "1e3", "0", "123456789012345", "1234567890123456", "three", "0x10", "nan", "-4"]))
                                            for _ in range(rng.randrange(0, 9))) for i in range(40))
    path = tmp_path / "in.txt"
    n_err = n_ok = 0
    for threads, chunk in (("6", "30"), ("1", "1")):
        env = dict(os.environ, KPOP_HOST_THREADS=threads, KPOP_HOST_CHUNK=chunk)
        for it in range(60):
            s = list(base)
            for _ in range(rng.randrange(0, 3)):
                s[rng.randrange(len(s))] = rng.choice(["\n", "x", "0", "A", "a"])
            data = "".join(s)
            if it % 9 == 0:
                data = data.replace("three", "3").replace("0x10", "16").replace("nan", "5")
            path.write_text(data)
            r = subprocess.run([lookup_harness, str(path), mode, str(tmp_path / "names.txt"), rng.choice(["1", "40", "300", "100000"])],
                               capture_output=True, text=True, env=env)
            assert r.returncode == 0, r.stderr
            kind, want = _reference_lines(data, lambda nm: table.get(nm))
            if kind == "ERROR":
                n_err += 1
                assert r.stdout.startswith("ERROR ") and want in r.stdout, (it, want, r.stdout[:200])
            else:
                n_ok += 1
                assert r.stdout.splitlines() == want, (it, mode)
    assert n_ok >= 5 and n_err >= 5
