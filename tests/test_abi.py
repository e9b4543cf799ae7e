"""CPU tests of the boundary: the C-ABI library loads, exports every symbol include/kpop_hip.h
declares, and refuses to compute without a GPU (no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "kpop_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(kpop_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported():
    from kpop_amd import _lib
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), "libkpop_hip.so does not export %s" % n
    assert set(_lib.SIGNATURES) == set(names), "python binding and header disagree"


def test_library_is_in_tree():
    from kpop_amd import _lib
    assert os.path.dirname(_lib.LIB_PATH) == os.path.join(ROOT, "kpop_amd")


def test_product_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under kpop_amd/ may import, link or read it."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "kpop_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".c", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "kpop_oracle" not in text and "import oracle" not in text and "from oracle" not in text, (dirpath, f)


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_no_cpu_fallback():
    import kpop_amd
    with pytest.raises(kpop_amd.KPopError):
        kpop_amd.init(0)
    with pytest.raises(kpop_amd.KPopError):
        kpop_amd.count_reads(np.frombuffer(b"ACGTACGT", dtype=np.uint8), [0, 8], 3)
    with pytest.raises(kpop_amd.KPopError):
        kpop_amd.distance_rowwise(np.ones((1, 2)), np.ones((1, 2)), np.ones(2))


def test_metric_matches_oracle(oracle):
    """kpop_metric_compute is O(n_dims) host arithmetic inside the library (lib/Space.ml:88-105)."""
    import kpop_amd
    for n in (1, 9, 64):
        w = oracle.synth_inertia(n)
        assert np.array_equal(kpop_amd.metric_compute(w), oracle.metric_powers(w, 1.0, 1.0, 2.0))
        assert np.array_equal(kpop_amd.metric_compute(w, kpop_amd.METRIC_POWERS, 0.5, 0.7, 1.5),
                              oracle.metric_powers(w, 0.5, 0.7, 1.5))
        assert np.array_equal(kpop_amd.metric_compute(w, kpop_amd.METRIC_FLAT), oracle.metric_flat(n))
    with pytest.raises(kpop_amd.KPopError):
        kpop_amd.metric_compute([1.0], kpop_amd.METRIC_POWERS, -1.0, 1.0, 2.0)  # Negative_power, lib/Space.ml:124


def test_parse_distance():
    import kpop_amd
    assert kpop_amd.parse_distance("euclidean") == (kpop_amd.EUCLIDEAN, 2.0)
    assert kpop_amd.parse_distance("cosine")[0] == kpop_amd.COSINE
    assert kpop_amd.parse_distance("minkowski(1.5)") == (kpop_amd.MINKOWSKI, 1.5)
    for bad in ("manhattan", "minkowski(x)", "minkowski(-1)"):
        with pytest.raises(ValueError):
            kpop_amd.parse_distance(bad)


def test_pack_bases_on_the_host():
    """kpop_pack_bases needs no GPU: 2-bit codes (A0 C1 G2 T3, either case; base i in bits 2 (i % 16).. of word i / 16) and a bit a base
    that is none of ACGTacgt, whatever the number of threads and wherever the batch ends (bin/KPopCount.ml:242-245: Lint.dnaize)"""
    import numpy as np
    from kpop_amd import api
    rng = np.random.RandomState(3)
    lut = {ord(ch): i for i, ch in enumerate("ACGT")}
    lut.update({ord(ch): i for i, ch in enumerate("acgt")})
    for n in (0, 1, 15, 16, 17, 31, 32, 33, 4095, 4096, 4097, 100003):
        b = np.frombuffer(bytes(rng.choice(list(b"ACGTacgtNnRYKM-*.x\x00\xff"), size=n).astype(np.uint8)), dtype=np.uint8) if n else np.zeros(0, np.uint8)
        c, m = api.pack_bases(b, threads=1)
        c3, m3 = api.pack_bases(b, threads=3)
        assert np.array_equal(c, c3) and np.array_equal(m, m3)
        assert len(c) == max((n + 15) // 16, 1) and len(m) == max((n + 31) // 32, 1)
        for i in range(n):
            code, inv = (int(c[i // 16]) >> (2 * (i % 16))) & 3, (int(m[i // 32]) >> (i % 32)) & 1
            if int(b[i]) in lut:
                assert inv == 0 and code == lut[int(b[i])], (n, i, chr(b[i]))
            else:
                assert inv == 1, (n, i, int(b[i]))
        if n % 32:
            assert (int(m[-1]) >> (n % 32)) == (1 << (32 - n % 32)) - 1  # (past the batch's end: no bases)
