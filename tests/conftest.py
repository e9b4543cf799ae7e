import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """Built artefacts are git-ignored: on a fresh checkout build them once (hipcc cross-compiles without a GPU)."""
    import shutil
    need = [os.path.join(ROOT, "kpop_amd", "libkpop_hip.so"), os.path.join(ROOT, "kpop_amd", "bin", "KPopTwistDB"),
            os.path.join(ROOT, "kpop_amd", "bin", "KPopCount"), os.path.join(ROOT, "kpop_amd", "bin", "KPopTwistCA"),
            os.path.join(ROOT, "kpop_amd", "bin", "KPopCountDB"), os.path.join(ROOT, "kpop_amd", "bin", "KPopTwist"),
            os.path.join(ROOT, "kpop_amd", "bin", "KPopTwist_")]
    if all(os.path.exists(p) for p in need):
        return
    if shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc"):
        import __graft_entry__
        __graft_entry__.build()


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def unhex(lst, shape=None):
    a = np.array([float.fromhex(x) for x in lst], dtype=np.float64)
    return a.reshape(shape) if shape is not None else a


def concat(seqs):
    bases = np.frombuffer("".join(seqs).encode(), dtype=np.uint8).copy()
    offs = np.zeros(len(seqs) + 1, dtype=np.uint64)
    if seqs:
        offs[1:] = np.cumsum([len(s) for s in seqs])
    return bases, offs


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure): builds oracle/libkpop_oracle.so on demand."""
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def pyref():
    from oracle import pyref as P
    return P


@pytest.fixture(scope="session")
def kpop():
    """The product, through its C ABI, on GPU 0.  No CPU path: init fails loudly without a GPU."""
    import kpop_amd
    kpop_amd.init(0)
    return kpop_amd
