"""pytest configuration: `gpu` marker + oracle build.

`-m "not gpu"` runs here (no GPU): oracle vs golden vectors, host logic, C-ABI symbol checks.
`-m gpu` runs on an MI355X: parity of the HIP path against the oracle, through the C-ABI.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    so = os.path.join(ROOT, "oracle", "libark_oracle.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
