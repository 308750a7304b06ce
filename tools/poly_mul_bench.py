#!/usr/bin/env python3
"""`&DensePolynomial * &DensePolynomial` (dense.rs:641-656) from HOST coefficient vectors, two ways:
  ark_hip_poly_mul   ONE call: one upload of both factors, both forward transforms + the pointwise product + the inverse
                     transform on the device, one download (what patches/0005 binds);
  three transforms   what patches/0003 alone gives a Rust `&a * &b`: three host-pointer transforms (each both ways over
                     PCIe) with the pointwise product on the host (here: the oracle's field multiplication on all cores).
    python tools/poly_mul_bench.py [LOG_LEN ...]     (two factors of 2^LOG_LEN coefficients each, BLS12-381 Fr)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import algebra_amd as A
import synth as S

FIELD = "BLS12_381_FR"
r = S.R[FIELD]
for lg in [int(x) for x in (sys.argv[1:] or ["16", "20"])]:
    n = 1 << lg
    a = S.gen_scalars(n, 3 + lg, r)
    b = S.gen_scalars(n, 5 + lg, r)
    A.poly_mul_host(FIELD, a, b)                       # warm-up: tables, staging buffers
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        c1 = A.poly_mul_host(FIELD, a, b)
    t_one = (time.perf_counter() - t0) * 1e3 / reps
    dom = A.Radix2EvaluationDomain.new(FIELD, 2 * n - 1)
    import oracle_lib as O                             # (test infrastructure: here the stand-in for the CPU pointwise product)
    fid = O.FID[FIELD]

    def three():
        ea = dom.fft(a)
        eb = dom.fft(b)
        prod = O.field_op(fid, "mul", ea, eb).reshape(-1, 4)
        return dom.ifft(prod)

    three()
    t0 = time.perf_counter()
    for _ in range(2):
        c3 = three()
    t_three = (time.perf_counter() - t0) * 1e3 / 2
    same = bool(np.array_equal(c1, c3[: c1.shape[0]]) and not c3[c1.shape[0]:].any())
    print("2^%d x 2^%d coefficients (domain 2^%d): ark_hip_poly_mul %.2f ms per product (%.0f MiB up, %.0f MiB down); "
          "three host-pointer transforms + host pointwise product %.2f ms; same coefficients: %s"
          % (lg, lg, lg + 1, t_one, 2 * n * 32 / 2**20, (2 * n - 1) * 32 / 2**20, t_three, same), flush=True)
