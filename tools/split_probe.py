#!/usr/bin/env python3
"""Would ONE synchronous MSM finish sooner as two half-size jobs in flight on the two MSM lanes (the second half's sort and
reduction under the first half's accumulate kernel)?  Emulated with two prepared half base sets.
    python tools/split_probe.py [CURVE] [LOG_N ...]"""
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
sizes = [int(a) for a in sys.argv[2:]] or [16, 18, 20, 22]
cid = cv.curve_id(curve)
r = S.R[cv.scalar_field(cid)]
L = lib()
ab = cv.affine_bytes(cid)
for logn in sizes:
    n = 1 << logn
    h = n // 2
    bases = S.grow_bases(cid, n, S.A0, S.B0, r)
    sc = S.gen_scalars(n, 5, r)
    scalars = torch.from_numpy(sc.view(np.int64)).cuda().reshape(n, 4)
    kg = S.mul_gen(cid, S.dlog_of_msm(sc, S.A0, S.B0, r), r)
    pb = A.PreparedBases(cid, bases)
    pa = A.PreparedBases(cid, bases[: h * ab])
    pc = A.PreparedBases(cid, bases[h * ab:])
    s0, s1 = scalars[:h].contiguous(), scalars[h:].contiguous()
    reps = 40 if logn <= 20 else 10

    def whole():
        return pb.msm_bigint(scalars)

    def split():
        j0 = pa.msm_bigint_async(s0)
        j1 = pc.msm_bigint_async(s1)
        return A.sum_projective(cid, np.stack([j0.wait(), j1.wait()]))

    out = {}
    for name, fn in (("whole", whole), ("split", split)):
        res = fn()
        ok = bool(np.array_equal(A.into_affine(cid, res), kg))
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        out[name] = ((time.perf_counter() - t0) * 1e3 / reps, ok)
    print("%s 2^%d: one job %.3f ms (exact=%s) | two half jobs in flight %.3f ms (exact=%s)  plans %s / %s"
          % (curve, logn, out["whole"][0], out["whole"][1], out["split"][0], out["split"][1], pb.info()["window_bits"],
             pa.info()["window_bits"]), flush=True)
    pb.free(); pa.free(); pc.free()
