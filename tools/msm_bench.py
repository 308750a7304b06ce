#!/usr/bin/env python3
"""MSM timing helper for any curve / size (device resident, synthetic P_i = (a + i b)G); every timed configuration is
checked bit-exactly against k*G (tools/synth.py).
    python tools/msm_bench.py CURVE LOG_N [steps] [plain|prepared|both]
env ARK_HIP_MSM_C=<c> / ARK_HIP_MSM_C_PREPARED=<c> override the window size."""
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


def timed(fn, steps):
    fn()
    L = lib()
    check(L.ark_hip_msm_set_timing(1), "t")
    t0 = time.perf_counter()
    for _ in range(steps):
        res = fn()
    dt = (time.perf_counter() - t0) / steps
    tm = (C.c_double * 8)()
    L.ark_hip_msm_last_timing(tm)
    check(L.ark_hip_msm_set_timing(0), "t")
    return res, dt, list(tm)


def main():
    curve, logn = sys.argv[1], int(sys.argv[2])
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    mode = sys.argv[4] if len(sys.argv) > 4 else "plain"
    cid = cv.curve_id(curve)
    r = S.R[cv.scalar_field(cid)]
    n = 1 << logn
    bases = S.grow_bases(cid, n, S.A0, S.B0, r)
    sc = S.gen_scalars(n, 5, r)
    scalars = torch.from_numpy(sc.view(np.int64)).cuda()
    torch.cuda.synchronize()
    kg = S.mul_gen(cid, S.dlog_of_msm(sc, S.A0, S.B0, r), r)
    if mode in ("plain", "both"):
        res, dt, tm = timed(lambda: A.msm_bigint(cid, bases, scalars), steps)
        ok = bool(np.array_equal(A.into_affine(cid, res), kg))
        print("%s 2^%d plain    c=%d W=%d: %.2f ms/MSM  %.3e scalar-muls/s  [digits %.2f sortA %.2f sortB %.2f accumulate %.2f reduce %.2f]  exact=%s"
              % (curve, logn, int(tm[6]), int(tm[7]), dt * 1e3, n / dt, tm[0], tm[1], tm[2], tm[3], tm[4], ok), flush=True)
    if mode in ("prepared", "both"):
        t0 = time.perf_counter()
        pb = A.PreparedBases(cid, bases)
        prep_s = time.perf_counter() - t0
        info = pb.info()
        res, dt, tm = timed(lambda: pb.msm_bigint(scalars), steps)
        ok = bool(np.array_equal(A.into_affine(cid, res), kg))
        print("%s 2^%d prepared c=%d W=%d: %.2f ms/MSM  %.3e scalar-muls/s  [digits %.2f sortA %.2f sortB %.2f accumulate %.2f reduce %.2f]  exact=%s  (table %.2f GiB built in %.2f s)"
              % (curve, logn, int(tm[6]), int(tm[7]), dt * 1e3, n / dt, tm[0], tm[1], tm[2], tm[3], tm[4], ok,
                 info["table_bytes"] / 2**30, prep_s), flush=True)
        pb.free()
    if os.environ.get("MSM_BENCH_HOST"):
        hb = bases.cpu().numpy().view(np.uint64).reshape(n, -1)
        A.msm_bigint(cid, hb, sc)
        t0 = time.perf_counter()
        for _ in range(2):
            rh = A.msm_bigint(cid, hb, sc)
        dth = (time.perf_counter() - t0) / 2
        print("%s 2^%d host-pointer entry (pageable numpy buffers, H2D of %.2f GiB included): %.1f ms/MSM  %.3e scalar-muls/s  exact=%s"
              % (curve, logn, (hb.nbytes + sc.nbytes) / 2**30, dth * 1e3, n / dth,
                 bool(np.array_equal(A.into_affine(cid, rh), kg))), flush=True)


if __name__ == "__main__":
    main()
