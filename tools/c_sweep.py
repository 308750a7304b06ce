#!/usr/bin/env python3
"""Window-size sweep around the planner's choice (ARK_HIP_MSM_C / ARK_HIP_MSM_C_PREPARED are read per call):
    python tools/c_sweep.py CURVE LOGN[,LOGN...] [plain|prepared|both]
every timed line is checked against k*G."""
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
sizes = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "16,18,20,22").split(",")]
modes = ("plain", "prepared") if (len(sys.argv) <= 3 or sys.argv[3] == "both") else (sys.argv[3],)
cid = cv.curve_id(curve)
r = S.R[cv.scalar_field(cid)]
L = lib()


def timed(fn, steps):
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
    steps = 10 if logn <= 20 else 3
    for mode in modes:
        key = "ARK_HIP_MSM_C_PREPARED" if mode == "prepared" else "ARK_HIP_MSM_C"
        wb, ww = C.c_int(), C.c_int()
        check(L.ark_hip_msm_plan(cid, n, 1 if mode == "prepared" else 0, C.byref(wb), C.byref(ww)), "plan")
        c0 = wb.value
        for c in [None] + list(range(c0 - 3, c0 + 3)):
            if c is not None and (c < 3 or c > 24):
                continue
            if c is None:
                os.environ.pop(key, None)
            else:
                os.environ[key] = str(c)
            try:
                pb = A.PreparedBases(cid, bases) if mode == "prepared" else None
                fn = (lambda: pb.msm_bigint(d)) if pb else (lambda: A.msm_bigint(cid, bases, d))
                res, dt, tm = timed(fn, steps)
                ok = bool(np.array_equal(A.into_affine(cid, res), kg))
                print("%s 2^%d %-8s c=%-7s(->%d,W=%d) %8.3f ms  [sort %.2f acc %.2f red %.2f]  exact=%s"
                      % (curve, logn, mode, "planner" if c is None else c, int(tm[6]), int(tm[7]), dt * 1e3, tm[1] + tm[2], tm[3], tm[4], ok), flush=True)
                if pb:
                    pb.free()
            except Exception as e:  # noqa: BLE001
                print("%s 2^%d %s c=%s: %r" % (curve, logn, mode, c, e), flush=True)
        os.environ.pop(key, None)
    del bases, d
    torch.cuda.empty_cache()
