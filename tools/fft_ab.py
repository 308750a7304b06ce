#!/usr/bin/env python3
"""Timing of the device-resident transform on ONE box, one process: 20 transforms back to back per round, rounds either back
to back ("warm": the chip stays at its sustained clocks) or separated by 20 ms of idle time ("cold": every round starts from
the idle power state) -- the second is what a bench leg that follows host-side work measures; beside them eight transforms
per call with three in flight (ark_hip_fft_batch_in_place_device).
A/B switches are environment variables the library reads at start-up (ARK_HIP_FFT_COMPACT=0, ARK_HIP_FFT_LAZY=1).
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
    res = {}
    for label, pause in (("cold (20 ms idle before every round)", 0.02), ("warm (rounds back to back)", 0.0)):
        ms = []
        for r in range(ROUNDS + 2):
            if pause:
                time.sleep(pause)
            t0 = time.perf_counter()
            for _ in range(REPS):
                check(L.ark_hip_fft_in_place_device(dom.field, sref, y.data_ptr()), "fft")
            check(L.ark_hip_synchronize(), "sync")
            if r >= 2:
                ms.append((time.perf_counter() - t0) * 1e3 / REPS)
        res[label] = sorted(ms)
    ys = [x.clone() for _ in range(8)]
    torch.cuda.synchronize()
    ptrs = (C.c_void_p * 8)(*[t.data_ptr() for t in ys])
    check(L.ark_hip_fft_batch_in_place_device(dom.field, sref, ptrs, 8, 0), "batch")
    check(L.ark_hip_synchronize(), "sync")
    mb = []
    for r in range(ROUNDS):
        t0 = time.perf_counter()
        for _ in range(3):
            check(L.ark_hip_fft_batch_in_place_device(dom.field, sref, ptrs, 8, 0), "batch")
        check(L.ark_hip_synchronize(), "sync")
        mb.append((time.perf_counter() - t0) * 1e3 / 24)
    del ys
    print("log_n=%d roundtrip_ok=%s" % (kf, ok))
    for label, v in res.items():
        print("   %-40s ms per transform: min %.4f  median %.4f  max %.4f   -> %.2f Gelem/s at the median"
              % (label, v[0], v[len(v) // 2], v[-1], nf / v[len(v) // 2] / 1e6))
    vb = sorted(mb)
    print("   batch of 8, three in flight (warm)       ms per transform: min %.4f  median %.4f  max %.4f" % (vb[0], vb[len(vb) // 2], vb[-1]))
