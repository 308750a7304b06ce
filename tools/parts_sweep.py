#!/usr/bin/env python3
"""Small-n accumulate latency: the longest bucket sets the kernel's time below ~2^18 pairs.  Sweeps the window size
(ARK_HIP_MSM_C) against lanes per run (ARK_HIP_MSM_RUN_PARTS) in one process, plain entry, every result exact.
    python tools/parts_sweep.py [CURVE] [log_n ...] [c-only]"""
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

curve = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].isdigit() else "BLS12_381_G1"
sizes = [int(a) for a in sys.argv[1:] if a.isdigit()] or [16, 17, 18]
C_ONLY = "c-only" in sys.argv   # window size alone, c = the plan's +- 2
cid = cv.curve_id(curve)
r = S.R[cv.scalar_field(cid)]
L = lib()


def timed(fn, steps=12):
    fn()
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


for logn in sizes:
    n = 1 << logn
    bases = S.grow_bases(cid, n, S.A0, S.B0, r)
    sc = S.gen_scalars(n, 5, r)
    d = torch.from_numpy(sc.view(np.int64)).cuda()
    kg = S.mul_gen(cid, S.dlog_of_msm(sc, S.A0, S.B0, r), r)
    os.environ.pop("ARK_HIP_MSM_C", None)
    os.environ.pop("ARK_HIP_MSM_RUN_PARTS", None)
    if C_ONLY:
        wb, ww = C.c_int(), C.c_int()
        check(L.ark_hip_msm_plan(cid, n, 0, C.byref(wb), C.byref(ww)), "plan")
        cands = (None,) + tuple(range(wb.value - 2, wb.value + 3))
    else:
        cands = (None,) + tuple(range(max(6, logn - 6), logn))
    for c in cands:
        for parts in ((None,) if C_ONLY else (None, 2, 4, 8)):
            for k, v in (("ARK_HIP_MSM_C", c), ("ARK_HIP_MSM_RUN_PARTS", parts)):
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = str(v)
            res, dt, tm = timed(lambda: A.msm_bigint(cid, bases, d))
            ok = bool(np.array_equal(A.into_affine(cid, res), kg))
            print("%s 2^%d c=%s parts=%s: %.3f ms  [digits+sort %.2f acc %.2f reduce %.2f  c=%d W=%d]  exact=%s" % (
                curve, logn, c, parts, dt * 1e3, tm[0] + tm[1] + tm[2], tm[3], tm[4], int(tm[6]), int(tm[7]), ok), flush=True)
os.environ.pop("ARK_HIP_MSM_C", None)
os.environ.pop("ARK_HIP_MSM_RUN_PARTS", None)
