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
