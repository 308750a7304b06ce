#!/usr/bin/env python3
"""Timing of the callers either side of the MSM that end in a batch normalisation: ScalarMul::batch_mul
(ec/src/scalar_mul/mod.rs:104-251), CurveGroup::normalize_batch (group.rs:302-319) and the one-time table of a
prepared base set.  Device-resident inputs; correctness of the same entry points is covered by tests/.
    python tools/aux_bench.py [CURVE] [LOG_N]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import algebra_amd as A
import synth as S
from algebra_amd import curves as cv
from algebra_amd._lib import check, lib

curve = sys.argv[1] if len(sys.argv) > 1 else "BLS12_381_G1"
logn = int(sys.argv[2]) if len(sys.argv) > 2 else 20
cid = cv.curve_id(curve)
r = S.R[cv.scalar_field(cid)]
n = 1 << logn
L = lib()


def timed(fn, reps=5):
    fn()
    check(L.ark_hip_synchronize(), "sync")
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    check(L.ark_hip_synchronize(), "sync")
    return (time.perf_counter() - t0) * 1e3 / reps


sc = torch.from_numpy(S.gen_scalars(n, 9, r).view(np.int64)).cuda()
gen = S.mul_gen(cid, 0xC0FFEE, r).reshape(1, -1)      # affine k*G
base = A.msm_bigint(cid, gen, S.limbs4(0xBEEF).reshape(1, 4))   # a Projective (x, y, z), z != 1
fw = cv.fe_words(cid)
t = A.BatchMulPreprocessing(cid, base, n)
ms = timed(lambda: t.batch_mul(sc, montgomery=False))
print("%s batch_mul        2^%d scalars: %8.3f ms  (%.3e scalar-muls/s, affine results)" % (curve, logn, ms, n / ms * 1e3))
t.free()
bases = S.grow_bases(cid, n, S.A0, S.B0, r)
# n (x, y, z) triples with a non-trivial z: the cost of normalize_batch does not depend on the values
aw = cv.affine_words(cid)
jac = torch.empty((n, 3 * fw), dtype=torch.int64, device="cuda")
jac[:, :aw] = bases.view(torch.int64).reshape(n, aw)
jac[:, aw:] = torch.from_numpy(np.ascontiguousarray(base).reshape(-1)[aw:].view(np.int64)).cuda()
ms = timed(lambda: A.normalize_batch(cid, jac))
print("%s normalize_batch  2^%d points:  %8.3f ms" % (curve, logn, ms))
del jac
t0 = time.perf_counter()
pb = A.PreparedBases(cid, bases)
check(L.ark_hip_synchronize(), "sync")
print("%s PreparedBases    2^%d bases:   %8.3f s   (%s)" % (curve, logn, time.perf_counter() - t0, pb.info()))
pb.free()
