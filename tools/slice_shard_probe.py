#!/usr/bin/env python3
"""What a bit-slice x base-range hybrid of the sharded MSM would cost per GPU, measured on ONE GPU with the
kernels as they are: rank (b, g) of an 8-GPU job would sum the 64-bit slice g of the scalars over base half b -- which is
VariableBaseMSM::msm_u64 on 2^25 points -- against the base-range shard it would replace (2^23 full-width pairs).  Same for 4 GPUs
(four 64-bit slices over all 2^26 points against 2^24 full-width pairs).  Every result is checked against k*G.
    python tools/slice_shard_probe.py"""
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
import oracle_lib as O
from algebra_amd import curves as cv

cid = cv.curve_id("BLS12_381_G1")
r = S.R["BLS12_381_FR"]


def kg(k):
    return O.to_affine(cid, O.scalar_mul(cid, O.generator(cid), S.limbs4(k % r)))


def timed(fn, reps):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        res = fn()
    return (time.perf_counter() - t0) * 1e3 / reps, res


n_max = 1 << 26
bases = S.grow_bases(cid, n_max, S.A0, S.B0, r)
ab = cv.affine_bytes(cid)
rng = np.random.default_rng(3)
for logn, reps in ((23, 5), (24, 5), (25, 3), (26, 2)):
    n = 1 << logn
    full = S.gen_scalars(n, 77 + logn, r)
    d_full = torch.from_numpy(full.view(np.int64)).cuda()
    ms_f, res = timed(lambda: A.msm_bigint(cid, bases[: n * ab], d_full), reps)
    ok_f = bool(np.array_equal(A.into_affine(cid, res), kg(S.dlog_of_msm(full, S.A0, S.B0, r))))
    u64 = full[:, 1].copy()                          # bits 64..127 of the same scalars: a 64-bit slice
    sl = np.zeros((n, 4), dtype=np.uint64)
    sl[:, 0] = u64
    d_u64 = torch.from_numpy(u64.view(np.int64)).cuda()
    ms_s, res = timed(lambda: A.msm_u64(cid, bases[: n * ab], d_u64), reps)
    ok_s = bool(np.array_equal(A.into_affine(cid, res), kg(S.dlog_of_msm(sl, S.A0, S.B0, r))))
    print("2^%d points:  full-width scalars (a base-range shard) %.2f ms  exact=%s   |   one 64-bit slice (msm_u64) %.2f ms  exact=%s"
          % (logn, ms_f, ok_f, ms_s, ok_s), flush=True)
    del d_full, d_u64
