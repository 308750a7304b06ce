#!/usr/bin/env python3
"""Small-n latency tuning: sweeps the level-0 chunk (ARK_HIP_MSM_L0), the bit-stage chunk (ARK_HIP_MSM_CHUNK) and the
window size in one process (the knobs are read per call) for 2^16 ... 2^20, plain and prepared, every result exact."""
import ctypes as C
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
cid = cv.curve_id(curve)
r = S.R[cv.scalar_field(cid)]
L = lib()


def timed(fn, steps=10):
    fn()
    check(L.ark_hip_msm_set_timing(1), "t")
    t0 = time.perf_counter()
    for _ in range(steps):
        res = fn()
    dt = (time.perf_counter() - t0) / steps
    tm = (C.c_double * 8)()
    L.ark_hip_msm_last_timing(tm)
    check(L.ark_hip_msm_set_timing(0), "t")
    return res, dt, list(tm)


for logn in (16, 18, 20):
    n = 1 << logn
    bases = S.grow_bases(cid, n, S.A0, S.B0, r)
    sc = S.gen_scalars(n, 5, r)
    d = torch.from_numpy(sc.view(np.int64)).cuda()
    kg = S.mul_gen(cid, S.dlog_of_msm(sc, S.A0, S.B0, r), r)
    for mode in ("plain", "prepared"):
        cs = [None] + ([logn - 1, logn, logn + 1, logn + 2] if mode == "prepared" else [logn - 2, logn - 1, logn])
        for c in cs:
            key = "ARK_HIP_MSM_C_PREPARED" if mode == "prepared" else "ARK_HIP_MSM_C"
            if c is None:
                os.environ.pop(key, None)
            else:
                os.environ[key] = str(c)
            pb = A.PreparedBases(cid, bases) if mode == "prepared" else None
            fn = (lambda: pb.msm_bigint(d)) if pb else (lambda: A.msm_bigint(cid, bases, d))
            for l0 in (None, 4, 8, 16):
                for ch in (None, 256, 1024):
                    for k, v in (("ARK_HIP_MSM_L0", l0), ("ARK_HIP_MSM_CHUNK", ch)):
                        if v is None:
                            os.environ.pop(k, None)
                        else:
                            os.environ[k] = str(v)
                    res, dt, tm = timed(fn)
                    ok = bool(np.array_equal(A.into_affine(cid, res), kg))
                    print("%s 2^%d %-8s c=%-4s(->%d,W=%d) L0=%-4s chunk=%-5s %.3f ms  [acc %.2f red %.2f]  exact=%s"
                          % (curve, logn, mode, c, int(tm[6]), int(tm[7]), l0, ch, dt * 1e3, tm[3], tm[4], ok), flush=True)
            if pb:
                pb.free()
            os.environ.pop(key, None)
