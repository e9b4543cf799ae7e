"""The C ABI from a plain C99 host (examples/c_host.c): compiled with gcc against include/kpop_hip.h and the shared library,
run on the GPU, its numbers compared with the oracle -- the boundary carries no Python, C++ or torch types."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, concat

pytestmark = pytest.mark.gpu


def test_c_host_program(tmp_path, oracle, pyref):
    exe = tmp_path / "c_host"
    lib = os.path.join(ROOT, "kpop_amd")
    subprocess.run(["gcc", "-O2", "-std=c99", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "c_host.c"),
                    "-L" + lib, "-lkpop_hip", "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib", "-lm", "-o", str(exe)], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.splitlines()
    reads = ["ACGTACGTTGCA", "TTTTTTTT", "ACGNNACGTAC"]
    k = 3
    for i, s in enumerate(reads):
        want = " ".join("%02x=%d" % (h, c) for h, c in sorted(pyref.count_read(s, k).items()))
        assert lines[i] == ("read %d: %s" % (i, want)).rstrip()
    cols = oracle.enumerate_kmers(k)
    n = len(cols)
    T = np.array([[(c + 1) / 8.0 for c in range(n)], [(n - c) / 4.0 for c in range(n)]])
    bases, offs = concat(reads)
    h, c, o = oracle.count_reads(bases, offs, k)
    tw = oracle.twist(T, cols, h, c.astype(np.float64), o)
    metric = oracle.metric_powers(np.array([0.75, 0.25]))
    ref = np.array([[1.0, 2.0], [3.0, 1.0]])
    dist = oracle.distance_rowwise(ref, tw, metric)
    for i in range(3):
        t = lines[3 + i]
        assert t.startswith("twisted %d: %.15g %.15g |" % (i, tw[i, 0], tw[i, 1])) and t.endswith("spectra_twist identical")
        fused = [float(x) for x in t.split("|")[1].split()[1:3]]
        assert np.allclose(fused, tw[i], rtol=1e-12, atol=0)
        assert lines[6 + i] == "distances %d: %.15g %.15g" % (i, dist[i, 0], dist[i, 1])
    assert len(lines) == 17 and all(l.startswith("summary") for l in lines[9:12])
    # the streaming pipeline from C: two chunks, page-locked buffers, same neighbours, distances of the fused rows
    assert lines[12] == "pipeline: 2 chunks, pinned 1, neighbours identical"
    for i in range(3):
        got = [float(x) for x in lines[13 + i].split(":")[1].split()]
        assert np.allclose(got, dist[i], rtol=1e-12, atol=0)
    # the packed entry points from C: 31 bases in 2 + 1 words, the byte entry point's rows
    assert lines[16] == "packed: 31 bases in 2 + 1 words, rows identical"
