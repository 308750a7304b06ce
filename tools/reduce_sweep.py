#!/usr/bin/env python3
"""Reduction-geometry sweep of one MSM (base set prepared once): level-0 chunk (ARK_HIP_MSM_L0) x bit-stage chunk
(ARK_HIP_MSM_CHUNK); every configuration checked bit-exactly against k*G.
    python tools/reduce_sweep.py [CURVE] [LOG_N] [plain|prepared]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import algebra_amd as A
import synth as S
from algebra_amd import curves as cv
from msm_bench import timed

curve = sys.argv[1] if len(sys.argv) > 1 else "BLS12_381_G1"
logn = int(sys.argv[2]) if len(sys.argv) > 2 else 24
mode = sys.argv[3] if len(sys.argv) > 3 else "prepared"
cid = cv.curve_id(curve)
r = S.R[cv.scalar_field(cid)]
n = 1 << logn
bases = S.grow_bases(cid, n, S.A0, S.B0, r)
sc = S.gen_scalars(n, 5, r)
scalars = torch.from_numpy(sc.view(np.int64)).cuda()
kg = S.mul_gen(cid, S.dlog_of_msm(sc, S.A0, S.B0, r), r)
pb = A.PreparedBases(cid, bases) if mode == "prepared" else None
run = (lambda: pb.msm_bigint(scalars)) if pb is not None else (lambda: A.msm_bigint(cid, bases, scalars))
for l0 in (None, 8, 16, 32, 64):
    for chunk in (None, 1024, 2048, 4096, 8192, 16384):
        for k, v in (("ARK_HIP_MSM_L0", l0), ("ARK_HIP_MSM_CHUNK", chunk)):
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = str(v)
        res, dt, tm = timed(run, 3)
        ok = bool(np.array_equal(A.into_affine(cid, res), kg))
        print("%s 2^%d %s c=%d W=%d L0=%-4s chunk=%-5s  %.3f ms  [acc %.2f red %.2f]  exact=%s"
              % (curve, logn, mode, int(tm[6]), int(tm[7]), l0, chunk, dt * 1e3, tm[3], tm[4], ok), flush=True)
