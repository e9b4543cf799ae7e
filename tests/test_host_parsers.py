"""CPU test of the host text layer: the threaded spectra parser of the GPU path (read_spectra_hashed, chunks cut at line
boundaries and merged in file order) against the sequential parser (read_spectra_file) on mutated inputs -- same
result or the same error, whatever the chunking.  The harness is compiled from tests/host/parser_diff.cpp."""
import os
import random
import subprocess

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    out = tmp_path_factory.mktemp("parser_diff") / "parser_diff"
    src = os.path.join(ROOT, "tests", "host", "parser_diff.cpp")
    subprocess.run(["g++", "-O1", "-std=c++17", "-pthread", "-o", str(out), src, os.path.join(ROOT, "kpop_amd", "host", "kpop_text.cpp")],
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
            r = subprocess.run([harness, str(path), "3"], capture_output=True, text=True, errors="replace", env=env)
            assert r.returncode == 0, (threads, chunk, it, r.stdout, r.stderr)
