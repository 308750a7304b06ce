#!/usr/bin/env python3
"""ScalarMul::batch_mul (ec/src/scalar_mul/mod.rs:104-251) timed the way the reference runs it: table construction
(BatchMulPreprocessing::new) and the batch (batch_mul) separately, per batch size.  Device-resident scalars; every line is
checked against k * g for a few entries through the plain MSM entry (parity proper: tests/test_gpu_msm_prepared.py).
    python tools/batchmul_bench.py CURVE LOG_N [LOG_N ...]      (ARK_HIP_BATCHMUL_WINDOW=w forces the row width)"""
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
L = lib()
check(L.ark_hip_init(0), "init")
r = S.R[cv.scalar_field(cid)]
gen = S.mul_gen(cid, 0xC0FFEE, r).reshape(1, -1)      # affine k*G
base = A.msm_bigint(cid, gen, S.limbs4(0xBEEF).reshape(1, 4))
for logn in [int(a) for a in sys.argv[2:]]:
    n = 1 << logn
    sc = torch.from_numpy(S.gen_scalars(n, 900 + logn, r).view(np.int64)).cuda()
    builds = []
    for _ in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        t = A.BatchMulPreprocessing(cid, base, n)
        builds.append((time.perf_counter() - t0) * 1e3)
        if _ < 3:
            t.free()
    out = t.batch_mul(sc, montgomery=False)
    torch.cuda.synchronize()
    reps = 10 if logn <= 20 else 3
    t0 = time.perf_counter()
    for _ in range(reps):
        out = t.batch_mul(sc, montgomery=False)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / reps
    # spot check: out[i] == sc[i] * base through the MSM entry with the base as the only point
    ok = True
    base_aff = A.into_affine(cid, base).reshape(1, -1)
    for i in (0, n // 2, n - 1):
        e = A.into_affine(cid, A.msm_bigint(cid, base_aff, sc[i:i + 1].cpu().numpy().view(np.uint64)))
        ok &= bool(np.array_equal(out[i].cpu().numpy().view(np.uint64).reshape(-1), e.reshape(-1)))
    t.free()
    print("%s batch_mul 2^%-2d window=%s: table %7.3f ms (best of 4: first %.3f)   batch %9.3f ms  (%.3e scalar-muls/s)   "
          "new + batch_mul %9.3f ms   exact=%s" % (curve, logn, os.environ.get("ARK_HIP_BATCHMUL_WINDOW", "auto"), min(builds),
                                                   builds[0], ms, n / ms * 1e3, min(builds) + ms, ok))
