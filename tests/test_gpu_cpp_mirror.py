"""The C++ host mirror (include/ark_hip.hpp: VariableBaseMSM<Curve>, Radix2EvaluationDomain<F>) driven from a compiled
C++ program on the GPU and checked against the oracle -- the compiled-language counterpart of the Python mirror."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_cpp_host_mirror_against_oracle(tmp_path):
    exe = str(tmp_path / "host_mirror_check")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "oracle"),
                           os.path.join(ROOT, "tests", "cpp", "host_mirror_check.cpp"), "-o", exe,
                           "-L", os.path.join(ROOT, "algebra_amd"), "-lark_hip", "-L", os.path.join(ROOT, "oracle"),
                           "-lark_oracle", "-Wl,-rpath," + os.path.join(ROOT, "algebra_amd"),
                           "-Wl,-rpath," + os.path.join(ROOT, "oracle")], timeout=300)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all ok" in out.stdout
