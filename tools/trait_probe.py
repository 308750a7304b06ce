#!/usr/bin/env python3
"""Times the trait-surface entry (ark_hip_msm_sw from pageable host memory) at one size: first call (bases + scalars
over PCIe), repeat calls (resident-base cache: scalars only) for several piece counts, cache off, and auto-prepared.
Run once per ARK_HIP_COPY_THREADS setting (the staging pool is created on first use).  Every result is checked
against k*G."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import synth as S  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-n", type=int, default=24)
    ap.add_argument("--curve", default="BLS12_381_G1")
    ap.add_argument("--pieces", default="1,2,4,8")
    ap.add_argument("--auto-prepare", action="store_true")
    args = ap.parse_args()
    import torch
    import algebra_amd as A
    from algebra_amd import curves as cv
    cid = cv.curve_id(args.curve)
    r = S.R[cv.scalar_field(cid)]
    n = 1 << args.log_n
    bases = S.grow_bases(cid, n, S.A0, S.B0, r)
    hb = bases.cpu().numpy().view(np.uint64).reshape(n, -1)
    sc = S.gen_scalars(n, 0x7A17, r)
    want = S.mul_gen(cid, S.dlog_of_msm(sc, S.A0, S.B0, r), r)
    dsc = torch.from_numpy(sc.view(np.int64)).cuda()
    torch.cuda.synchronize()
    tag = "threads=%s n=2^%d %s" % (os.environ.get("ARK_HIP_COPY_THREADS", "default(4)"), args.log_n, args.curve)

    def timed(fn, reps):
        fn()
        t0 = time.perf_counter()
        for _ in range(reps):
            res = fn()
        return (time.perf_counter() - t0) * 1e3 / reps, res

    def ok(res):
        return bool(np.array_equal(A.into_affine(cid, res), want))

    ms, res = timed(lambda: A.msm_bigint(cid, bases, dsc), 3)
    print("%s  resident (ark_hip_msm_sw_device)            %8.2f ms  exact=%s" % (tag, ms, ok(res)), flush=True)
    A.base_cache_clear()
    t0 = time.perf_counter()
    res = A.msm_bigint(cid, hb, sc)
    first = (time.perf_counter() - t0) * 1e3
    gb = (hb.nbytes + sc.nbytes) / 1e9
    print("%s  first call  (miss: %.2f GB over PCIe)        %8.2f ms  exact=%s" % (tag, gb, first, ok(res)), flush=True)
    for p in [int(x) for x in args.pieces.split(",")]:
        os.environ["ARK_HIP_STREAM_PIECES"] = str(p)
        ms, res = timed(lambda: A.msm_bigint(cid, hb, sc), 3)
        print("%s  repeat call (hit), %2d piece(s)               %8.2f ms  exact=%s  %.3e scalar-muls/s"
              % (tag, p, ms, ok(res), n / (ms * 1e-3)), flush=True)
    os.environ.pop("ARK_HIP_STREAM_PIECES", None)
    ms, res = timed(lambda: A.msm_bigint(cid, hb, sc), 3)
    print("%s  repeat call (hit), default pieces             %8.2f ms  exact=%s  cache=%s"
          % (tag, ms, ok(res), A.base_cache_stats()), flush=True)
    if args.auto_prepare:
        A.base_cache_config(-1, 1)
        A.msm_bigint(cid, hb, sc)
        A.msm_bigint(cid, hb, sc)
        ms, res = timed(lambda: A.msm_bigint(cid, hb, sc), 3)
        print("%s  repeat call, auto-prepared table              %8.2f ms  exact=%s" % (tag, ms, ok(res)), flush=True)
        A.base_cache_config(-1, 0)
    A.base_cache_config(0, -1)
    ms, res = timed(lambda: A.msm_bigint(cid, hb, sc), 2)
    print("%s  cache off (bases + scalars streamed)          %8.2f ms  exact=%s" % (tag, ms, ok(res)), flush=True)


if __name__ == "__main__":
    main()
