#!/usr/bin/env python3
"""Window-size sweep for NARROW scalars (the msm_u1/u8/u16/u32/u64 entries): planner's choice against forced c
(ARK_HIP_MSM_C), BLS12_381_G1, device-resident inputs, every result checked against k*G.
    python tools/narrow_sweep.py LOG_N [bits,bits,...]"""
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

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
which = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 8, 16, 32, 64]
cid = cv.curve_id("BLS12_381_G1")
r = S.R[cv.scalar_field(cid)]
n = 1 << logn
bases = S.grow_bases(cid, n, S.A0, S.B0, r)
rng = np.random.default_rng(1)
ENTRY = {1: (A.msm_u1, np.uint8), 8: (A.msm_u8, np.uint8), 16: (A.msm_u16, np.uint16), 32: (A.msm_u32, np.uint32),
         64: (A.msm_u64, np.uint64)}


def timed(fn, reps=3):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        res = fn()
    return (time.perf_counter() - t0) / reps * 1e3, res


print("# BLS12_381_G1 2^%d narrow entries: ms per MSM (planner = no override)" % logn)
for bits in which:
    fn, dt = ENTRY[bits]
    v = rng.integers(0, 1 << bits, size=n, dtype=np.uint64)
    sc = np.zeros((n, 4), dtype=np.uint64)
    sc[:, 0] = v
    kg = S.mul_gen(cid, S.dlog_of_msm(sc, S.A0, S.B0, r), r)
    sg = {1: np.int8, 2: np.int16, 4: np.int32, 8: np.int64}[np.dtype(dt).itemsize]
    d = torch.from_numpy(np.ascontiguousarray(v.astype(dt)).view(sg)).cuda()
    big = torch.from_numpy(sc.view(np.int64)).cuda()
    os.environ.pop("ARK_HIP_MSM_C", None)
    ms, res = timed(lambda: fn(cid, bases, d))
    line = "u%-3d planner %6.2f%s | bigint entry %6.2f |" % (bits, ms, "" if np.array_equal(A.into_affine(cid, res), kg) else "!",
                                                            timed(lambda: A.msm_bigint(cid, bases, big))[0])
    for c in range(max(3, min(bits + 1, 7)), 19):
        if c > bits + 1:
            break
        os.environ["ARK_HIP_MSM_C"] = str(c)
        ms, res = timed(lambda: fn(cid, bases, d))
        line += " c%d %.2f%s" % (c, ms, "" if np.array_equal(A.into_affine(cid, res), kg) else "!")
    os.environ.pop("ARK_HIP_MSM_C", None)
    print(line, flush=True)
