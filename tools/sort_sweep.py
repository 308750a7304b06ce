#!/usr/bin/env python3
"""Partition-sort geometry sweep: super-bucket bits of pass A (ARK_HIP_MSM_HB) x keys per tile (ARK_HIP_MSM_TILE) against the
library's own choice, device-resident plain MSM, every result checked against k*G; prints total / sort (scan + scatter phases).
    python tools/sort_sweep.py CURVE LOG_N [LOG_N ...]"""
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

curve = sys.argv[1]
cid = cv.curve_id(curve)
r = S.R[cv.scalar_field(cid)]
L = lib()
for logn in [int(x) for x in sys.argv[2:]]:
    n = 1 << logn
    bases = S.grow_bases(cid, n, S.A0, S.B0, r)
    sc = S.gen_scalars(n, 5, r)
    d = torch.from_numpy(sc.view(np.int64)).cuda()
    kg = S.mul_gen(cid, S.dlog_of_msm(sc, S.A0, S.B0, r), r)

    def run(steps=6):
        A.msm_bigint(cid, bases, d)
        check(L.ark_hip_msm_set_timing(1), "t")
        t0 = time.perf_counter()
        for _ in range(steps):
            res = A.msm_bigint(cid, bases, d)
        dt = (time.perf_counter() - t0) / steps
        tm = (C.c_double * 8)()
        L.ark_hip_msm_last_timing(tm)
        check(L.ark_hip_msm_set_timing(0), "t")
        ok = bool(np.array_equal(A.into_affine(cid, res), kg))
        return dt * 1e3, tm[1] + tm[2], ok

    for k in ("ARK_HIP_MSM_HB", "ARK_HIP_MSM_TILE"):
        os.environ.pop(k, None)
    ms, red, ok = run()
    print("%s 2^%d plan %s: library choice %.3f ms (sort %.3f)%s" % (curve, logn, A.msm_plan(cid, n), ms, red, "" if ok else " WRONG"), flush=True)
    for hb in (7, 8, 9, 10, 11):
        line = "  HB=%-2d" % hb
        for tile in (8192, 16384):
            os.environ["ARK_HIP_MSM_HB"] = str(hb)
            os.environ["ARK_HIP_MSM_TILE"] = str(tile)
            ms, red, ok = run()
            line += "  tile %5d: %.3f (%.3f)%s" % (tile, ms, red, "" if ok else "!")
        print(line, flush=True)
    for k in ("ARK_HIP_MSM_HB", "ARK_HIP_MSM_TILE"):
        os.environ.pop(k, None)
    del bases, d
    torch.cuda.empty_cache()
