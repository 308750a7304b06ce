#!/usr/bin/env python3
"""Timing of the device-resident transform, interleaved rounds (clock drift hits every round alike): 20 transforms back to
back per round, min / median / max over the rounds.  A/B switches are environment variables the library reads at start-up
(ARK_HIP_FFT_COMPACT=0, ARK_HIP_FFT_LAZY=1): run once per setting on the same box.
usage: python tools/fft_ab.py [log_n ...]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import algebra_amd as A
from algebra_amd._lib import check, lib
import bench

L = lib()
REPS, ROUNDS = 20, 7
print("# " + " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("ARK_HIP_FFT")))
for kf in [int(a) for a in sys.argv[1:]] or [22]:
    nf = 1 << kf
    dom = A.Radix2EvaluationDomain.new("BLS12_381_FR", nf)
    x = torch.from_numpy(bench.gen_scalars(nf, 7).view(np.int64)).cuda()
    y = x.clone()
    torch.cuda.synchronize()
    sref = C.byref(dom._s)
    check(L.ark_hip_fft_in_place_device(dom.field, sref, y.data_ptr()), "fft")
    check(L.ark_hip_ifft_in_place_device(dom.field, sref, y.data_ptr()), "ifft")
    check(L.ark_hip_synchronize(), "sync")
    ok = bool(torch.equal(x, y))
    ms = []
    for r in range(ROUNDS):
        t0 = time.perf_counter()
        for _ in range(REPS):
            check(L.ark_hip_fft_in_place_device(dom.field, sref, y.data_ptr()), "fft")
        check(L.ark_hip_synchronize(), "sync")
        ms.append((time.perf_counter() - t0) * 1e3 / REPS)
        time.sleep(0.02)
    v = sorted(ms)
    print("log_n=%d roundtrip_ok=%s  ms per transform: min %.4f  median %.4f  max %.4f   -> %.2f Gelem/s at the median"
          % (kf, ok, v[0], v[len(v) // 2], v[-1], nf / v[len(v) // 2] / 1e6))
