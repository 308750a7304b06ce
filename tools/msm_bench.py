#!/usr/bin/env python3
"""MSM timing helper for any curve / size (device resident, synthetic P_i = (a + i b)G), with the exact
k*G check.  usage: python tools/msm_bench.py CURVE LOG_N [steps]     env ARK_HIP_MSM_C=<c> overrides the window"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import algebra_amd as A
from algebra_amd import curves as cv
from algebra_amd._lib import check, lib

R = {"BN254_FR": 21888242871839275222246405745257275088548364400416034343698204186575808495617,
     "BLS12_381_FR": 52435875175126190479447740508185965837690552500527637822603658699938581184513,
     "BLS12_377_FR": 8444461749428370424248824938781546531375899335154063827935233455917409239041}


def limbs4(v):
    return np.array([(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)


def main():
    curve, logn = sys.argv[1], int(sys.argv[2])
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    cid = cv.curve_id(curve)
    r = R[cv.scalar_field(cid)]
    n = 1 << logn
    ab = cv.affine_bytes(cid)
    L = lib()
    gen = np.zeros(cv.affine_words(cid), dtype=np.uint64)
    check(L.ark_hip_curve_generator(cid, gen.ctypes.data_as(C.c_void_p)), "gen")
    mul_gen = lambda k: A.into_affine(cid, A.msm_bigint(cid, gen.reshape(1, -1), limbs4(k % r).reshape(1, 4)))
    a0, b0 = 0xA11CE + (1 << 64), 0xB0B + (3 << 64)
    bases = torch.zeros(n * ab, dtype=torch.uint8, device="cuda")
    bases[:ab] = torch.from_numpy(mul_gen(a0).view(np.uint8)).cuda()
    torch.cuda.synchronize()
    m = 1
    while m < n:
        cnt = min(m, n - m)
        d = np.ascontiguousarray(mul_gen(m * b0))
        check(L.ark_hip_sw_add_affine_device(cid, bases.data_ptr(), bases.data_ptr() + m * ab, cnt, d.ctypes.data_as(C.c_void_p)), "ext")
        m += cnt
    if cv.scalar_field(cid) == "BLS12_381_FR":
        import bench
        sc = bench.gen_scalars(n, 5)          # uniform in [0, r)
    else:
        rng = np.random.default_rng(5)
        sc = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64)
        sc[:, 3] &= np.uint64((1 << 60) - 1)  # < 2^252 < r for the other two scalar fields
    scalars = torch.from_numpy(sc.view(np.int64)).cuda()
    torch.cuda.synchronize()
    res = A.msm_bigint(cid, bases, scalars)
    check(L.ark_hip_msm_set_timing(1), "t")
    t0 = time.perf_counter()
    for _ in range(steps):
        res = A.msm_bigint(cid, bases, scalars)
    dt = (time.perf_counter() - t0) / steps
    tm = (C.c_double * 8)()
    L.ark_hip_msm_last_timing(tm)
    # exact check: k = sum s_i (a + i b)
    vals = sc[:, 0].astype(object) + (sc[:, 1].astype(object) << 64) + (sc[:, 2].astype(object) << 128) + (sc[:, 3].astype(object) << 192) if n <= (1 << 18) else None
    ok = None
    if vals is not None:
        k = (int(np.sum(vals)) * a0 + int(np.dot(vals, np.arange(n, dtype=object))) * b0) % r
        ok = bool(np.array_equal(A.into_affine(cid, res), mul_gen(k)))
    if os.environ.get("MSM_BENCH_HOST"):
        hb = bases.cpu().numpy().view(np.uint64).reshape(n, -1)
        A.msm_bigint(cid, hb, sc)
        t0 = time.perf_counter()
        for _ in range(2):
            rh = A.msm_bigint(cid, hb, sc)
        dth = (time.perf_counter() - t0) / 2
        print("%s 2^%d host-pointer entry (pageable numpy buffers, H2D of %.2f GiB included): %.1f ms/MSM  %.3e scalar-muls/s  same result: %s"
              % (curve, logn, (hb.nbytes + sc.nbytes) / 2**30, dth * 1e3, n / dth, bool(np.array_equal(rh, res)) or
                 bool(np.array_equal(A.into_affine(cid, rh), A.into_affine(cid, res)))))
    print("%s 2^%d c=%d W=%d: %.2f ms/MSM  %.3e scalar-muls/s  [digits %.2f sortA %.2f sortB %.2f accumulate %.2f reduce %.2f]  exact=%s"
          % (curve, logn, int(tm[6]), int(tm[7]), dt * 1e3, n / dt, tm[0], tm[1], tm[2], tm[3], tm[4], ok))


if __name__ == "__main__":
    main()
