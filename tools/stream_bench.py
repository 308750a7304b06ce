#!/usr/bin/env python3
"""Streaming MSM: a prepared, resident base set and scalar vectors that arrive from HOST memory.
Measures the PCIe-inclusive time per MSM (scalars only; the SRS is resident) for
  serial     upload, then compute, one MSM at a time (ark_hip_msm_prepared on pinned / pageable scalars)
  pipelined  ark_hip_msm_prepared_async with two jobs in flight: the upload of MSM k+1 overlaps MSM k's kernels
  resident   scalars already in HBM (the bench.py headline), for reference
and msm_chunks (bases AND scalars streamed from the host in 2^20-pair steps, double-buffered).
    python tools/stream_bench.py [LOG_N] [ROUNDS]"""
import ctypes as C
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
from algebra_amd._lib import check, lib


def main():
    logn = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    curve = "BLS12_381_G1"
    cid = cv.curve_id(curve)
    r = S.R[cv.scalar_field(cid)]
    n = 1 << logn
    L = lib()
    bases = S.grow_bases(cid, n, S.A0, S.B0, r)
    pb = A.PreparedBases(cid, bases)
    nvec = 3
    ptr = C.c_void_p()
    check(L.ark_hip_host_alloc(nvec * n * 32, C.byref(ptr)), "host_alloc")
    pinned = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint64)), shape=(nvec, n, 4))
    want = []
    pageable = []
    for i in range(nvec):
        sc = S.gen_scalars(n, 40 + i, r)
        pinned[i] = sc
        pageable.append(sc)
        want.append(S.mul_gen(cid, S.dlog_of_msm(sc, S.A0, S.B0, r), r))
    dev = [torch.from_numpy(pageable[i].view(np.int64)).cuda() for i in range(nvec)]
    torch.cuda.synchronize()
    ok = True

    def run(label, fn):
        nonlocal ok
        fn(0)  # warm
        t0 = time.perf_counter()
        res = [fn(k) for k in range(rounds)]
        dt = (time.perf_counter() - t0) / rounds
        good = all(np.array_equal(A.into_affine(cid, res[k]), want[k % nvec]) for k in range(rounds))
        ok = ok and good
        print("%-44s %7.2f ms/MSM  %.3e scalar-muls/s  exact=%s" % (label, dt * 1e3, n / dt, good), flush=True)

    run("resident scalars (HBM)", lambda k: pb.msm_bigint(dev[k % nvec]))
    run("serial, pinned host scalars", lambda k: pb.msm_bigint(pinned[k % nvec]))
    run("serial, pageable host scalars", lambda k: pb.msm_bigint(pageable[k % nvec]))
    # pipelined: two jobs in flight
    for label, src in (("pipelined x2, resident scalars (HBM)", dev), ("pipelined x2, pinned host scalars", pinned),
                       ("pipelined x2, pageable host scalars", pageable)):
        pb.msm_bigint_async(src[0]).wait()
        t0 = time.perf_counter()
        pend, res = [], []
        for k in range(rounds):
            pend.append(pb.msm_bigint_async(src[k % nvec]))
            if len(pend) == 2:
                res.append(pend.pop(0).wait())
        while pend:
            res.append(pend.pop(0).wait())
        dt = (time.perf_counter() - t0) / rounds
        good = all(np.array_equal(A.into_affine(cid, res[k]), want[k % nvec]) for k in range(rounds))
        ok = ok and good
        print("%-44s %7.2f ms/MSM  %.3e scalar-muls/s  exact=%s" % (label, dt * 1e3, n / dt, good), flush=True)
    # H2D rate of the scalar upload alone
    d = torch.empty(n * 4, dtype=torch.int64, device="cuda")
    for label, src in (("pinned", pinned[0]), ("pageable", pageable[0])):
        check(L.ark_hip_memcpy_h2d(d.data_ptr(), src.ctypes.data_as(C.c_void_p), n * 32), "h2d")
        t0 = time.perf_counter()
        for _ in range(3):
            check(L.ark_hip_memcpy_h2d(d.data_ptr(), src.ctypes.data_as(C.c_void_p), n * 32), "h2d")
        dt = (time.perf_counter() - t0) / 3
        print("H2D of the %d MiB scalar vector, %s: %.2f ms (%.1f GB/s)" % (n * 32 >> 20, label, dt * 1e3, n * 32 / dt / 1e9))
    pb.free()
    # msm_chunks: everything streamed from the host
    hb = bases.cpu().numpy().view(np.uint64).reshape(n, -1)
    fr = pageable[0]  # canonical < r is also a valid Montgomery residue: msm_chunks takes Fr
    A.msm_chunks(cid, hb[: 1 << 20], fr[: 1 << 20])
    t0 = time.perf_counter()
    out = A.msm_chunks(cid, hb, fr)
    dt = time.perf_counter() - t0
    whole = A.msm_unchecked(cid, bases, dev[0])
    same = bool(np.array_equal(A.into_affine(cid, out), A.into_affine(cid, whole)))
    ok = ok and same
    print("msm_chunks, 2^%d pairs from pageable host memory (%.2f GiB), 2^20-pair steps: %.1f ms  %.3e scalar-muls/s  == msm_unchecked: %s"
          % (logn, (hb.nbytes + fr.nbytes) / 2**30, dt * 1e3, n / dt, same))
    check(L.ark_hip_host_free(ptr), "host_free")
    print("ALL EXACT" if ok else "MISMATCH")


if __name__ == "__main__":
    main()
