"""Montgomery constants (mirrors test-templates/src/fields.rs:505-560 test_montgomery_config and
fields.rs:385-418 test_fft): R, R2, INV, generator, 2-adic root recomputed with Python ints and
compared with (a) the oracle's table and (b) the HIP kernels' params.hpp."""
import os
import re

import numpy as np
import pytest

import oracle_lib as O
import pyref as P

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("fid,name", list(enumerate(P.FIELD_ORDER)))
def test_oracle_constants(fid, name):
    p, g = P.MODULI[name]
    n = P.nlimbs(p)
    R = (1 << (64 * n)) % p
    assert O.field_limbs(fid) == n
    assert P.from_limbs(O.field_const(fid, 0)) == p
    assert P.from_limbs(O.field_const(fid, 1)) == R
    assert P.from_limbs(O.field_const(fid, 2)) == R * R % p
    assert P.from_limbs(O.field_const(fid, 3)) == g * R % p
    s, t = P.two_adicity(p)
    root = pow(g, t, p)
    assert P.from_limbs(O.field_const(fid, 4)) == root * R % p
    # root^(2^i) == 1 exactly at i = s  (fields.rs:385-418)
    x = root
    for i in range(s):
        assert x != 1
        x = x * x % p
    assert x == 1


def _parse_params():
    txt = open(os.path.join(ROOT, "algebra_amd", "csrc", "params.hpp")).read()
    out = {}
    for m in re.finditer(r"struct (\w+) \{(.*?)\n\};", txt, re.S):
        body = m.group(2)
        d = {}
        for k in ("N", "BITS", "TWO_ADICITY"):
            d[k] = int(re.search(r"int %s = (\d+);" % k, body).group(1))
        d["INV"] = int(re.search(r"INV = (0x[0-9a-f]+)u;", body).group(1), 16)
        for k in ("P", "R", "R2", "GEN", "ROOT"):
            arr = re.search(r"uint32_t %s\[\d+\] = \{(.*?)\};" % k, body).group(1)
            v = 0
            for i, tok in enumerate(arr.split(",")):
                v |= int(tok.strip().rstrip("u"), 16) << (32 * i)
            d[k] = v
        out[m.group(1)] = d
    return out


def test_device_params_header():
    params = _parse_params()
    assert set(params) == set(P.FIELD_ORDER)
    for name, d in params.items():
        p, g = P.MODULI[name]
        n32 = 2 * P.nlimbs(p)
        R = (1 << (32 * n32)) % p
        s, t = P.two_adicity(p)
        assert d["N"] == n32 and d["BITS"] == p.bit_length() and d["TWO_ADICITY"] == s
        assert d["P"] == p and d["R"] == R and d["R2"] == R * R % p
        assert d["GEN"] == g * R % p and d["ROOT"] == pow(g, t, p) * R % p
        assert (d["INV"] * p + 1) % (1 << 32) == 0
        # every modulus leaves a spare top bit and is not all-ones below it: the no-carry
        # Montgomery multiply (montgomery_backend.rs:63-77) is valid, result < 2p before the final subtract
        assert p.bit_length() < 32 * n32
