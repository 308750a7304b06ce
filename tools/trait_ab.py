#!/usr/bin/env python3
"""A/B of the piece rules of the trait-surface repeat call (ark_hip_msm_sw, resident-base cache hit) inside ONE process,
alternating: growing pieces (default) / 8 equal pieces / 4 equal pieces.    python tools/trait_ab.py [LOG_N] [ROUNDS]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import synth as S  # noqa: E402
import torch  # noqa: E402
import algebra_amd as A  # noqa: E402
from algebra_amd import curves as cv  # noqa: E402

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cid = cv.curve_id("BLS12_381_G1")
r = S.R[cv.scalar_field(cid)]
n = 1 << logn
bases = S.grow_bases(cid, n, S.A0, S.B0, r)
hb = bases.cpu().numpy().view(np.uint64).reshape(n, -1)
sc = S.gen_scalars(n, 0x7A17, r)
want = S.mul_gen(cid, S.dlog_of_msm(sc, S.A0, S.B0, r), r)
dsc = torch.from_numpy(sc.view(np.int64)).cuda()
A.msm_bigint(cid, hb, sc)
A.msm_bigint(cid, hb, sc)


def run(env):
    for k in ("ARK_HIP_STREAM_PIECES", "ARK_HIP_STREAM_GROWING"):
        os.environ.pop(k, None)
    os.environ.update(env)
    A.msm_bigint(cid, hb, sc)
    t0 = time.perf_counter()
    for _ in range(4):
        res = A.msm_bigint(cid, hb, sc)
    ms = (time.perf_counter() - t0) * 1e3 / 4
    return ms, bool(np.array_equal(A.into_affine(cid, res), want))


t0 = time.perf_counter()
for _ in range(3):
    A.msm_bigint(cid, bases, dsc)
print("2^%d resident: %.2f ms" % (logn, (time.perf_counter() - t0) * 1e3 / 3))
for rd in range(rounds):
    for name, env in (("growing (default)", {}), ("8 equal pieces", {"ARK_HIP_STREAM_PIECES": "8"}),
                      ("4 equal pieces", {"ARK_HIP_STREAM_PIECES": "4"})):
        ms, ok = run(env)
        print("round %d  %-18s %8.2f ms  exact=%s" % (rd, name, ms, ok), flush=True)
