#!/usr/bin/env python3
"""Times the trait-surface entry (ark_hip_msm_sw from pageable host memory) at one size in each of its modes:
  cache off    bases + scalars streamed over PCIe in pieces, nothing retained;
  pinned       ark_hip_msm_bases_pin: scalars only, growing pieces (first call after the pin and repeat calls);
  cached       the default: device copy validated by a full-content hash on host threads (miss, hit);
  prepared     pinned + auto-prepared per-window table.
Every result is checked against k*G.  ARK_HIP_COPY_THREADS / ARK_HIP_HASH_THREADS are read by the library."""
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
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--auto-prepare", action="store_true")
    ap.add_argument("--schedules", default="", help="';'-separated ARK_HIP_STREAM_SCHEDULE weight lists to time (cache off)")
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
    tag = "copy_threads=%s hash_threads=%s n=2^%d %s" % (os.environ.get("ARK_HIP_COPY_THREADS", "0"),
                                                        os.environ.get("ARK_HIP_HASH_THREADS", "8"), args.log_n, args.curve)

    def timed(fn, reps, warm=True):
        if warm:
            fn()
        t0 = time.perf_counter()
        for _ in range(reps):
            res = fn()
        return (time.perf_counter() - t0) * 1e3 / reps, res

    def ok(res):
        return bool(np.array_equal(A.into_affine(cid, res), want))

    def line(what, ms, res, extra=""):
        print("%s  %-58s %8.2f ms  %.3e scalar-muls/s  exact=%s %s" % (tag, what, ms, n / (ms * 1e-3), ok(res), extra), flush=True)

    gb = (hb.nbytes + sc.nbytes) / 1e9
    ms, res = timed(lambda: A.msm_bigint(cid, bases, dsc), args.reps)
    line("resident (ark_hip_msm_sw_device)", ms, res)
    A.base_cache_config(0, 0)
    ms, res = timed(lambda: A.msm_bigint(cid, hb, sc), args.reps)
    line("cache off: bases + scalars streamed (%.2f GB), default schedule" % gb, ms, res)
    for sched in [x for x in args.schedules.split(";") if x]:
        os.environ["ARK_HIP_STREAM_SCHEDULE"] = sched
        ms, res = timed(lambda: A.msm_bigint(cid, hb, sc), args.reps)
        line("cache off, schedule %s" % sched, ms, res)
    os.environ.pop("ARK_HIP_STREAM_SCHEDULE", None)
    t0 = time.perf_counter()
    pin = A.pin_bases(cid, hb)
    pin_ms = (time.perf_counter() - t0) * 1e3
    ms1, res = timed(lambda: A.msm_bigint(cid, hb, sc), 1, warm=False)
    line("pinned: first call after the pin (pin itself %.1f ms)" % pin_ms, ms1, res)
    ms, res = timed(lambda: A.msm_bigint(cid, hb, sc), args.reps)
    line("pinned: repeat call, growing pieces", ms, res, str(A.base_cache_stats()))
    if args.auto_prepare:
        A.base_cache_config(-1, 1)
        A.msm_bigint(cid, hb, sc)
        ms, res = timed(lambda: A.msm_bigint(cid, hb, sc), args.reps)
        line("pinned + auto-prepared table", ms, res)
        A.base_cache_config(-1, 0)
    pin.unpin()
    A.base_cache_config(64 << 30, 0)
    ms1, res = timed(lambda: A.msm_bigint(cid, hb, sc), 1, warm=False)
    line("verified cache (default): miss (fill + hash)", ms1, res)
    ms, res = timed(lambda: A.msm_bigint(cid, hb, sc), args.reps)
    line("verified cache (default): hit (speculative MSM + full hash)", ms, res, str(A.base_cache_stats()))
    A.base_cache_clear()
    A.base_cache_config(-2, 0)


if __name__ == "__main__":
    main()
