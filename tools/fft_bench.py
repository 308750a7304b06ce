#!/usr/bin/env python3
"""FFT-only timing helper (device resident): per-pass HIP-event times for a few sizes / pass plans.
usage: python tools/fft_bench.py [log_n ...]      (env ARK_HIP_FFT_KP=5..8 changes stages per pass)"""
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
for kf in [int(a) for a in sys.argv[1:]] or [22]:
    nf = 1 << kf
    dom = A.Radix2EvaluationDomain.new("BLS12_381_FR", nf)
    x = torch.from_numpy(bench.gen_scalars(nf, 7).view(np.int64)).cuda()
    y = x.clone()
    torch.cuda.synchronize()
    sref = C.byref(dom._s)
    for _ in range(3):
        check(L.ark_hip_fft_in_place_device(dom.field, sref, y.data_ptr()), "fft")
        check(L.ark_hip_ifft_in_place_device(dom.field, sref, y.data_ptr()), "ifft")
    check(L.ark_hip_synchronize(), "sync")
    ok = bool(torch.equal(x, y))
    t0 = time.perf_counter()
    for _ in range(20):
        check(L.ark_hip_fft_in_place_device(dom.field, sref, y.data_ptr()), "fft")
    check(L.ark_hip_synchronize(), "sync")
    ms = (time.perf_counter() - t0) * 1e3 / 20
    check(L.ark_hip_fft_set_timing(1), "t")
    check(L.ark_hip_fft_in_place_device(dom.field, sref, y.data_ptr()), "fft")
    ft = (C.c_double * 10)()
    L.ark_hip_fft_last_timing(ft)
    check(L.ark_hip_fft_set_timing(0), "t")
    print("log_n=%d KP=%s roundtrip_ok=%s wall %.3f ms  device %.3f ms  passes %s  -> %.2f Gelem/s, %.1f GB/s algorithmic"
          % (kf, os.environ.get("ARK_HIP_FFT_KP", "8"), ok, ms, ft[0], ["%.3f" % ft[2 + i] for i in range(int(ft[1]))],
             nf / ms / 1e6, 64.0 * nf / ft[0] / 1e6))
