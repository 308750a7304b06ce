#!/usr/bin/env python3
"""Window size of the PREPARED path (ARK_HIP_MSM_C_PREPARED) around the library's choice: the table is rebuilt per cell,
12 calls per cell, every result exact.    python tools/prepared_c_sweep.py [CURVE] [log_n ...]"""
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
sizes = [int(a) for a in sys.argv[1:] if a.isdigit()] or [16, 18, 20]
cid = cv.curve_id(curve)
r = S.R[cv.scalar_field(cid)]
L = lib()
for logn in sizes:
    n = 1 << logn
    bases = S.grow_bases(cid, n, S.A0, S.B0, r)
    sc = S.gen_scalars(n, 5, r)
    d = torch.from_numpy(sc.view(np.int64)).cuda()
    kg = S.mul_gen(cid, S.dlog_of_msm(sc, S.A0, S.B0, r), r)
    os.environ.pop("ARK_HIP_MSM_C_PREPARED", None)
    wb, ww = C.c_int(), C.c_int()
    check(L.ark_hip_msm_plan(cid, n, 1, C.byref(wb), C.byref(ww)), "plan")
    for c in (None,) + tuple(range(wb.value - 3, wb.value + 3)):
        if c is None:
            os.environ.pop("ARK_HIP_MSM_C_PREPARED", None)
        else:
            os.environ["ARK_HIP_MSM_C_PREPARED"] = str(c)
        try:
            pb = A.PreparedBases(cid, bases)
        except Exception as e:  # noqa: BLE001
            print("%s 2^%d prepared c=%s: %r" % (curve, logn, c, e), flush=True)
            continue
        pb.msm_bigint(d)
        pb.msm_bigint(d)
        check(L.ark_hip_msm_set_timing(1), "t")
        t0 = time.perf_counter()
        for _ in range(12):
            res = pb.msm_bigint(d)
        dt = (time.perf_counter() - t0) / 12
        tm = (C.c_double * 8)()
        L.ark_hip_msm_last_timing(tm)
        check(L.ark_hip_msm_set_timing(0), "t")
        ok = bool(np.array_equal(A.into_affine(cid, res), kg))
        print("%s 2^%d prepared c=%s: %.3f ms  [digits+sort %.2f acc %.2f reduce %.2f  c=%d W=%d]  exact=%s" % (
            curve, logn, c, dt * 1e3, tm[0] + tm[1] + tm[2], tm[3], tm[4], int(tm[6]), int(tm[7]), ok), flush=True)
        del pb
        torch.cuda.empty_cache()
os.environ.pop("ARK_HIP_MSM_C_PREPARED", None)
