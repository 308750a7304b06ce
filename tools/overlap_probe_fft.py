#!/usr/bin/env python3
"""Do two FFTs in flight fill the vector-ALU slots one FFT leaves idle (78 % busy)?  Probe with two logical devices
oversubscribing the one GPU (separate contexts / streams / ping buffers), one host thread each.
    ARK_HIP_OVERSUBSCRIBE=1 python tools/overlap_probe_fft.py [LOG_N] [FFTS_PER_THREAD]"""
import ctypes as C
import os
import sys
import threading
import time

os.environ.setdefault("ARK_HIP_OVERSUBSCRIBE", "1")
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import algebra_amd as A
import synth as S
from algebra_amd._lib import check, lib

FIELD = "BLS12_381_FR"
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 22
K = int(sys.argv[2]) if len(sys.argv) > 2 else 50
n = 1 << logn
L = lib()
ndev = torch.cuda.device_count()
r = S.R[FIELD]
x = torch.from_numpy(S.gen_scalars(n, 3, r).view(np.int64)).cuda()
torch.cuda.synchronize()


def worker(dev, out):
    check(L.ark_hip_set_device(dev), "set_device")
    dom = A.Radix2EvaluationDomain.new(FIELD, n)
    y = x.clone()
    torch.cuda.synchronize()
    sref = C.byref(dom._s)
    for _ in range(3):
        check(L.ark_hip_fft_in_place_device(dom.field, sref, y.data_ptr()), "fft")
    out["ready"].wait()
    t0 = time.perf_counter()
    for _ in range(K):
        check(L.ark_hip_fft_in_place_device(dom.field, sref, y.data_ptr()), "fft")
    check(L.ark_hip_synchronize(), "sync")
    out[dev] = time.perf_counter() - t0


for devs in ([0], [0, ndev], [0, ndev, 2 * ndev]):
    out = {"ready": threading.Barrier(len(devs))}
    th = [threading.Thread(target=worker, args=(d, out)) for d in devs]
    for t in th:
        t.start()
    for t in th:
        t.join()
    wall = max(out[d] for d in devs)
    print("%d context(s) x %d FFTs of 2^%d: %.2f ms wall -> %.4f ms per FFT, %.3e elements/s"
          % (len(devs), K, logn, wall * 1e3, wall * 1e3 / (K * len(devs)), n * K * len(devs) / wall), flush=True)
