#!/usr/bin/env python3
"""The transform over GROUP elements (ark_hip_fft_group_in_place_device; fft_in_place<T = Projective>): device time per
transform at a few sizes, and what the reference's CPU loop costs for the same transform ESTIMATED from the oracle's
scalar multiplication timed here on one core: n/2 * log2(n) scalar multiplications (+ n for the inverse's scaling), spread
over the box's usable cores.  Results checked: ifft(fft(P)) == P after into_affine.
usage: python tools/group_fft_bench.py [CURVE] [log_n ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import algebra_amd as A
import synth as S
from algebra_amd import curves as cv
from algebra_amd._lib import check, lib
import bench
import oracle_lib as O   # test infrastructure: here only the timed stand-in for the reference's scalar multiplication

curve = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].isdigit() else "BLS12_381_G1"
sizes = [int(a) for a in sys.argv[1:] if a.isdigit()] or [10, 14, 16]
cid = cv.curve_id(curve)
fname = cv.scalar_field(cid)
r = S.R[fname]
fw = cv.projective_words(cid) // 3
g = O.generator(O.CID[curve])
k4 = S.gen_scalars(64, 3, r)
t0 = time.perf_counter()
for i in range(64):
    O.scalar_mul(O.CID[curve], g, k4[i])
cpu_us = (time.perf_counter() - t0) / 64 * 1e6
cores = bench.usable_cores()
print("# %s: the oracle's scalar multiplication %.0f us on one core; %d usable cores" % (curve, cpu_us, cores))
for k in sizes:
    n = 1 << k
    aff = S.grow_bases(cid, n, S.A0, S.B0, r).cpu().numpy().view(np.uint64).reshape(n, 2 * fw)
    one = O.field_const(O.curve_info(O.CID[curve])[0], 1)
    pts = np.zeros((n, 3 * fw), dtype=np.uint64)
    pts[:, :2 * fw] = aff
    pts[:, 2 * fw:2 * fw + one.size] = one
    dom = A.Radix2EvaluationDomain.new(fname, n)
    d = torch.from_numpy(pts.view(np.int64)).cuda()
    torch.cuda.synchronize()
    dom.fft_group_in_place(curve, d)
    dom.fft_group_in_place(curve, d, inverse=True)
    ok = bool(np.array_equal(A.into_affine(cid, d.cpu().numpy().view(np.uint64).reshape(n, -1)), A.into_affine(cid, pts)))
    reps = 3 if k >= 16 else 6
    t0 = time.perf_counter()
    for _ in range(reps):
        dom.fft_group_in_place(curve, d)
    fwd = (time.perf_counter() - t0) / reps * 1e3
    t0 = time.perf_counter()
    for _ in range(reps):
        dom.fft_group_in_place(curve, d, inverse=True)
    inv = (time.perf_counter() - t0) / reps * 1e3
    est = (n // 2) * k * cpu_us / cores / 1e3
    print("2^%-2d  fft %9.2f ms   ifft %9.2f ms   roundtrip == input: %s   reference CPU loop, estimated: %.0f ms (%d scalar "
          "multiplications / %d cores) -> %.0f x" % (k, fwd, inv, ok, est, (n // 2) * k, cores, est / fwd))
