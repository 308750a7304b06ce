#!/usr/bin/env python3
"""phase breakdown of the u32-scalar MSM at 2^20, plain vs prepared (diagnostic)"""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch, algebra_amd as A, synth as S
from algebra_amd import curves as cv
from algebra_amd._lib import check, lib
L = lib(); cid = 1; r = S.R["BLS12_381_FR"]; n = 1 << 20
bases = S.grow_bases(cid, n, S.A0, S.B0, r)
pb = A.PreparedBases(cid, bases)
rng = np.random.default_rng(1)
for bits in (16, 24, 32, 40, 63):
    sc = np.zeros((n, 4), dtype=np.uint64); sc[:, 0] = rng.integers(0, 1 << bits, size=n, dtype=np.uint64)
    d = torch.from_numpy(sc.view(np.int64)).cuda()
    for label, fn in (("plain", lambda: A.msm_bigint(cid, bases, d)), ("prepared", lambda: pb.msm_bigint(d))):
        fn(); check(L.ark_hip_msm_set_timing(1), "t")
        t0 = time.perf_counter(); fn(); dt = time.perf_counter() - t0
        tm = (C.c_double * 8)(); L.ark_hip_msm_last_timing(tm); check(L.ark_hip_msm_set_timing(0), "t")
        print("u%d %-8s c=%d W=%d  %.2f ms [digits %.2f sortA %.2f sortB %.2f acc %.2f red %.2f dev %.2f]"
              % (bits, label, tm[6], tm[7], dt * 1e3, tm[0], tm[1], tm[2], tm[3], tm[4], tm[5]), flush=True)
