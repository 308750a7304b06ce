"""ubench/lazy.cuh (28-bit-limb field: a measured alternative to the shipped 32-bit-limb arithmetic, not part of
the product library -- see DESIGN.md) is kept parity-green by compiling its host build and checking it against the saturated-limb arithmetic."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_lazy_field_and_madd_match_saturated_form(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    exe = str(tmp_path / "lazy_check")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "algebra_amd", "csrc"), "-I", os.path.join(ROOT, "algebra_amd", "csrc", "ubench"),
                           os.path.join(ROOT, "tests", "lazy_host_check.hip"), "-o", exe], timeout=600)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count(": ok") == 3, out.stdout
