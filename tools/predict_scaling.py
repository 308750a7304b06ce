#!/usr/bin/env python3
"""Predicted strong scaling of BASELINE config 4 (ONE BLS12-381 G1 MSM of 2^26 pairs split by base range over N GPUs)
from what ONE GPU can measure: the shard times at 2^26 / N pairs (plain and prepared) and the cost of the exchange with
the ranks EMULATED in one process (ark_hip_test_msm_sharded_emulated: every rank's part sums land where the all-gather
would put them, one kernel adds them, one host tail folds them -- everything but the ncclAllGather call itself).
PREDICTED, UNMEASURED: no multi-GPU node was available to the builder; the RCCL all-gather of N x ~140 KB over xGMI is
an assumed 0.1 ms.    usage: python tools/predict_scaling.py [log_total=26]"""
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
from algebra_amd._lib import check, lib, test_lib

CURVE = "BLS12_381_G1"
ALLGATHER_MS = 0.1   # assumption, labelled in the output


def timed(fn, reps):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    return (time.perf_counter() - t0) * 1e3 / reps, r


def main():
    log_total = int(sys.argv[1]) if len(sys.argv) > 1 else 26
    cid = cv.curve_id(CURVE)
    r = S.R[cv.scalar_field(cid)]
    L = lib()
    n_tot = 1 << log_total
    bases = S.grow_bases(cid, n_tot, S.A0, S.B0, r)
    ab = cv.affine_bytes(cid)
    sc_h = S.gen_scalars(n_tot, 0x5CA1E, r)
    sc = torch.from_numpy(sc_h.view(np.int64)).cuda()
    torch.cuda.synchronize()
    kg = S.mul_gen(cid, S.dlog_of_msm(sc_h, S.A0, S.B0, r), r)
    print("# %s, ONE job of 2^%d pairs; every timed result below is checked against k*G or against the one-GPU result" % (CURVE, log_total))
    plain, prep = {}, {}
    for N in (1, 2, 4, 8):
        n = n_tot // N
        b, s = bases[: n * ab], sc[:n]
        reps = 2 if N == 1 else 4
        ms, res = timed(lambda: A.msm_bigint(cid, b, s), reps)
        plain[N] = ms
        ok = bool(np.array_equal(A.into_affine(cid, res), kg)) if N == 1 else None
        pb = A.PreparedBases(cid, b)
        msp, resp = timed(lambda: pb.msm_bigint(s), reps)
        prep[N] = msp
        okp = bool(np.array_equal(A.into_affine(cid, resp), A.into_affine(cid, res)))
        pb.free()
        del pb
        torch.cuda.empty_cache()
        print("shard of 2^%d / %d = 2^%d pairs on one GPU:  plain %.2f ms   prepared %.2f ms   (whole job == k*G: %s; prepared == plain: %s)"
              % (log_total, N, int(np.log2(n)), ms, msp, ok, okp))
    # the exchange, emulated: N shards of the 2^26 / 8 size run one after the other, then the sum kernel + ONE host tail
    n8 = n_tot // 8
    xch = {}
    for N in (2, 4, 8):
        pbp = (C.c_void_p * N)(*[bases.data_ptr() + i * n8 * ab for i in range(N)])
        psp = (C.c_void_p * N)(*[sc.data_ptr() + i * n8 * 32 for i in range(N)])
        pn = (C.c_size_t * N)(*[n8] * N)
        out = np.zeros(cv.projective_words(cid), dtype=np.uint64)
        path = C.c_int(0)

        def run():
            check(test_lib().ark_hip_test_msm_sharded_emulated(cid, N, pbp, psp, pn, 0, out.ctypes.data_as(C.c_void_p), C.byref(path)), "emulated")
            return out.copy()
        ms, res = timed(run, 3)
        sub = S.mul_gen(cid, S.dlog_of_msm(sc_h[: N * n8], S.A0, S.B0, r), r)
        xch[N] = ms - N * plain[8]
        print("emulated exchange, %d ranks x 2^%d pairs: %.2f ms in all = %d x %.2f (local MSMs incl. their own host tails) %+.2f ms "
              "(sum kernel over %d blocks + header check + one host tail);  path %d (1 = part sums added on the device);  == k*G: %s"
              % (N, int(np.log2(n8)), ms, N, plain[8], xch[N], N, path.value, bool(np.array_equal(A.into_affine(cid, res), sub))))
    print("# PREDICTED (unmeasured) time of the 2^%d job on N GPUs = shard time + max(0, emulated exchange cost of N ranks) + %.1f ms assumed for"
          " the RCCL all-gather of N x ~140 KB" % (log_total, ALLGATHER_MS))
    print("# (the emulated exchange cost is measured with 2^%d-pair shards for every N: the part sums' size depends on the plan, not on N)" % int(np.log2(n8)))
    for label, t in (("plain", plain), ("prepared", prep)):
        for N in (2, 4, 8):
            tn = t[N] + max(0.0, xch[N]) + ALLGATHER_MS
            print("PREDICTED %-8s N=%d: %.2f ms  -> %.2f x the one-GPU %.2f ms  (%.0f %% of linear); %.3e scalar-muls/s"
                  % (label, N, tn, t[1] / tn, t[1], 100.0 * t[1] / tn / N, n_tot / tn * 1e3))


if __name__ == "__main__":
    main()
