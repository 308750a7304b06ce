#!/usr/bin/env python3
"""Does the GPU overlap one MSM's memory-bound phases (digits, sort, reduce) with another MSM's accumulate kernel?
Probe with the library as it is: two logical devices oversubscribing the one GPU (ARK_HIP_OVERSUBSCRIBE=1: separate
contexts, streams and workspaces), one host thread each, against the same total work on a single context.
    ARK_HIP_OVERSUBSCRIBE=1 python tools/overlap_probe.py [LOG_N] [MSMS_PER_THREAD]"""
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
from algebra_amd import curves as cv
from algebra_amd._lib import check, lib

curve = "BLS12_381_G1"
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 24
K = int(sys.argv[2]) if len(sys.argv) > 2 else 6
cid = cv.curve_id(curve)
r = S.R[cv.scalar_field(cid)]
n = 1 << logn
L = lib()
ndev = torch.cuda.device_count()
bases = S.grow_bases(cid, n, S.A0, S.B0, r)
sc = S.gen_scalars(n, 5, r)
scalars = torch.from_numpy(sc.view(np.int64)).cuda()
kg = S.mul_gen(cid, S.dlog_of_msm(sc, S.A0, S.B0, r), r)
torch.cuda.synchronize()


def worker(dev, out):
    check(L.ark_hip_set_device(dev), "set_device")
    pb = A.PreparedBases(cid, bases)
    pb.msm_bigint(scalars)
    out["ready"].wait()        # every worker has built its table and warmed up
    t0 = time.perf_counter()
    for _ in range(K):
        res = pb.msm_bigint(scalars)
    out[dev] = (time.perf_counter() - t0, bool(np.array_equal(A.into_affine(cid, res), kg)))
    pb.free()


for devs in ([0], [0, ndev]):   # logical device ndev wraps onto physical 0
    out = {"ready": threading.Barrier(len(devs))}
    th = [threading.Thread(target=worker, args=(d, out)) for d in devs]
    for t in th:
        t.start()
    for t in th:
        t.join()
    wall = max(out[d][0] for d in devs)
    print("%d context(s) x %d MSMs of 2^%d: %.1f ms wall -> %.2f ms per MSM, %.3e scalar-muls/s, exact=%s"
          % (len(devs), K, logn, wall * 1e3, wall * 1e3 / (K * len(devs)), n * K * len(devs) / wall,
             all(out[d][1] for d in devs)), flush=True)
