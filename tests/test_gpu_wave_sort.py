"""The wavefront sort networks (kpop_amd/csrc/wave_sort.h) on their own: DPP moves, row / half-wave swaps and the one-direction
bitonic network are checked against std::sort on the GPU box, apart from the kernels that use them."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_wave_networks_sort_like_std_sort(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    src = os.path.join(ROOT, "tools", "probes", "src", "wave_sort_check.hip")
    exe = str(tmp_path / "wave_sort_check")
    subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-o", exe, src], check=True, timeout=600)
    out = subprocess.run([exe], check=True, timeout=120, capture_output=True, text=True).stdout
    lines = [l for l in out.splitlines() if "bad" in l]
    assert len(lines) == 8 and all(l.endswith("bad 0") for l in lines), out
