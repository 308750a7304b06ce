#!/usr/bin/env python3
"""The growing scalar pieces of a host-pointer MSM against RESIDENT (pinned) bases: piece weights swept through
ARK_HIP_STREAM_GROW_SCHEDULE on one box, interleaved; every result checked against k*G.
    python tools/grow_sweep.py [LOG_N=24] [reps=4]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import algebra_amd as A
import synth as S
import oracle_lib as O
from algebra_amd import curves as cv

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 24
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
cid = cv.curve_id("BLS12_381_G1")
r = S.R["BLS12_381_FR"]
n = 1 << logn
bases = S.grow_bases(cid, n, S.A0, S.B0, r)
hb = bases.cpu().numpy().view(np.uint64).reshape(n, -1)
sc = S.gen_scalars(n, 0x7A17, r)
want = O.to_affine(cid, O.scalar_mul(cid, O.generator(cid), S.limbs4(S.dlog_of_msm(sc, S.A0, S.B0, r))))
d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
torch.cuda.synchronize()


def timed(fn):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        res = fn()
    return (time.perf_counter() - t0) * 1e3 / reps, res


ms, res = timed(lambda: A.msm_bigint(cid, bases, d_sc))
print("2^%d resident inputs (ark_hip_msm_sw_device): %.2f ms  exact=%s" % (logn, ms, np.array_equal(A.into_affine(cid, res), want)), flush=True)
SCHEDULES = (os.environ.get("GROW_SWEEP_SCHEDULES") or ",1-2-4-8-16-32,1-3-9-27,1-3-8-20,1-4-12-32,1-2-5-12-28,1-3-9-27-60,2-5-12-28,1-4-16,1-3-7-15-30,1-2-4-8-16").replace("-", ",").split(",,") if False else [x.replace("-", ",") for x in (os.environ.get("GROW_SWEEP_SCHEDULES") or ";1-2-4-8-16-32;1-3-9-27;1-3-8-20;1-4-12-32;1-2-5-12-28;1-3-9-27-60;2-5-12-28;1-4-16;1-3-7-15-30;1-2-4-8-16").split(";")]
with A.pin_bases(cid, hb):
    A.msm_bigint(cid, hb, sc)
    for rnd in range(2):
        for s in SCHEDULES:
            if s:
                os.environ["ARK_HIP_STREAM_GROW_SCHEDULE"] = s
            else:
                os.environ.pop("ARK_HIP_STREAM_GROW_SCHEDULE", None)
            ms, res = timed(lambda: A.msm_bigint(cid, hb, sc))
            print("2^%d pinned bases, host scalars, pieces %-16s %.2f ms  exact=%s" % (logn, s or "(default)", ms, np.array_equal(A.into_affine(cid, res), want)), flush=True)
os.environ.pop("ARK_HIP_STREAM_GROW_SCHEDULE", None)
