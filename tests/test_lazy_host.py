"""csrc/fp28.cuh + ec28.cuh (the carry-free arithmetic of the bucket-accumulation kernels -- 14 x 28-bit limbs for the
Fp384 curves, 9 x 29-bit limbs for BN254 -- and the
curve-isomorphism boundary that lets it consume / produce the reference's canonical limbs) checked on the HOST against
the saturated-limb arithmetic: the templates are __host__ __device__, so the very code of the kernel runs here."""
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
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "algebra_amd", "csrc"),                            os.path.join(ROOT, "tests", "lazy_host_check.hip"), "-o", exe], timeout=600)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count(": ok") == 6, out.stdout   # three base fields + the FFT arithmetic over three scalar fields
