#!/usr/bin/env python3
"""One scalar distribution of the reference's MSM bench through msm_bigint, a few calls -- for a kernel trace
(rocprofv3 --kernel-trace -- python tools/dist_probe.py u16 20).  Distributions: random, bool, u8, u16, u32, u64, witness."""
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

name = sys.argv[1] if len(sys.argv) > 1 else "u16"
logn = int(sys.argv[2]) if len(sys.argv) > 2 else 20
cid = cv.curve_id("BLS12_381_G1")
r = S.R[cv.scalar_field(cid)]
n = 1 << logn
bases = S.grow_bases(cid, n, S.A0, S.B0, r)
rng = np.random.default_rng(1)
bits = {"bool": 1, "u8": 8, "u16": 16, "u32": 32, "u64": 64}.get(name)
if name == "witness":   # 60 % zeros, 30 % ones, 5 % minus one, 5 % full width
    sc = S.gen_scalars(n, 77, r)
    u = rng.random(n)
    sc[u < 0.60] = 0
    one = np.zeros(4, dtype=np.uint64)
    one[0] = 1
    sc[(u >= 0.60) & (u < 0.90)] = one
    sc[(u >= 0.90) & (u < 0.95)] = np.array([((r - 1) >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for k in range(4)], dtype=np.uint64)
elif bits is None:
    sc = S.gen_scalars(n, 5, r)
else:
    sc = np.zeros((n, 4), dtype=np.uint64)
    sc[:, 0] = rng.integers(0, 1 << bits, size=n, dtype=np.uint64)
direct = len(sys.argv) > 3 and sys.argv[3] == "direct"   # the msm_u* entry instead of msm_bigint
if direct:
    dt, sg, fn = {1: (np.uint8, np.int8, A.msm_u1), 8: (np.uint8, np.int8, A.msm_u8), 16: (np.uint16, np.int16, A.msm_u16),
                  32: (np.uint32, np.int32, A.msm_u32), 64: (np.uint64, np.int64, A.msm_u64)}[bits]
    d = torch.from_numpy(np.ascontiguousarray(sc[:, 0].astype(dt)).view(sg)).cuda()
    run = lambda: fn(cid, bases, d)
else:
    d = torch.from_numpy(sc.view(np.int64)).cuda()
    run = lambda: A.msm_bigint(cid, bases, d)
kg = S.mul_gen(cid, S.dlog_of_msm(sc, S.A0, S.B0, r), r)
run()
t0 = time.perf_counter()
for _ in range(5):
    res = run()
print("%s 2^%d %s %.2f ms exact=%s" % (name, logn, "direct" if direct else "msm_bigint", (time.perf_counter() - t0) / 5 * 1e3,
                                        bool(np.array_equal(A.into_affine(cid, res), kg))))
