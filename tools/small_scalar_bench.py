#!/usr/bin/env python3
"""The scalar distributions of the reference's MSM bench (bench-templates/src/macros/ec.rs:222-372) at its size
(2^20 points): random, bool, u8, i8, u16, i16, u32, i32, u64, i64 through msm_bigint, the `-direct` forms through
msm_u1/u8/u16/u32/u64, and the 11-way mixed vector -- each result checked against k*G (tools/synth.py).
    python tools/small_scalar_bench.py [LOG_N]"""
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


def to_limbs(vals):
    """list/array of python ints (mod r already) -> [n, 4] uint64"""
    out = np.zeros((len(vals), 4), dtype=np.uint64)
    for k in range(4):
        out[:, k] = [(v >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for v in vals]
    return out


def main():
    logn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    curve = os.environ.get("SKEW_CURVE", "BLS12_381_G1")
    cid = cv.curve_id(curve)
    r = S.R[cv.scalar_field(cid)]
    n = 1 << logn
    bases = S.grow_bases(cid, n, S.A0, S.B0, r)
    pb = A.PreparedBases(cid, bases)
    rng = np.random.default_rng(1)

    def small(bits, signed):
        v = rng.integers(0, 1 << bits, size=n, dtype=np.uint64)
        sc = np.zeros((n, 4), dtype=np.uint64)
        sc[:, 0] = v
        if not signed:
            return sc
        # iN::rand: two's complement value, Scalar::from(negative) = r - |x|
        neg = (v >> np.uint64(bits - 1)) == 1
        mag = np.where(neg, (np.uint64(1) << np.uint64(bits)) - v if bits < 64 else (~v + np.uint64(1)), v)
        vals = [int(m) if not ng else (r - int(m)) % r for m, ng in zip(mag, neg)]
        return to_limbs(vals)

    def mixed():
        s = n // 11
        parts = []
        for bits in (1, 8, 16, 32, 64):
            v = rng.integers(0, 1 << bits, size=s, dtype=np.uint64)
            pos = np.zeros((s, 4), dtype=np.uint64)
            pos[:, 0] = v
            parts.append(pos)
            parts.append(to_limbs([(r - int(x)) % r for x in v]))
        parts.append(S.gen_scalars(n - 10 * s, 99, r))
        allv = np.concatenate(parts)
        return allv[rng.permutation(allv.shape[0])]

    def witness():
        """an R1CS-witness-like vector: 60 % zeros, 30 % ones, 5 % minus one, 5 % full width"""
        sc = S.gen_scalars(n, 77, r)
        u = rng.random(n)
        sc[u < 0.60] = 0
        one = np.zeros(4, dtype=np.uint64)
        one[0] = 1
        sc[(u >= 0.60) & (u < 0.90)] = one
        sc[(u >= 0.90) & (u < 0.95)] = to_limbs([r - 1])[0]
        return sc

    cases = [("random", S.gen_scalars(n, 5, r), None), ("bool", small(1, False), A.msm_u1), ("u8", small(8, False), A.msm_u8),
             ("i8", small(8, True), None), ("u16", small(16, False), A.msm_u16), ("i16", small(16, True), None),
             ("u32", small(32, False), A.msm_u32), ("i32", small(32, True), None), ("u64", small(64, False), A.msm_u64),
             ("i64", small(64, True), None), ("mixed", mixed(), None), ("witness", witness(), None)]
    host = len(sys.argv) > 2 and sys.argv[2] == "host"   # + the host-pointer entry (ark_hip_msm_sw, default settings) for three rows
    hb = bases.cpu().numpy().view(np.uint64).reshape(n, -1) if host else None
    print("# %s 2^%d, device-resident inputs; ms per MSM, every result == k*G" % (curve, logn))
    for name, sc, direct in cases:
        d = torch.from_numpy(np.ascontiguousarray(sc).view(np.int64)).cuda()
        kg = S.mul_gen(cid, S.dlog_of_msm(sc, S.A0, S.B0, r), r)
        line = "%-8s" % name
        for label, fn in (("msm_bigint", lambda: A.msm_bigint(cid, bases, d)), ("prepared", lambda: pb.msm_bigint(d))):
            fn()
            t0 = time.perf_counter()
            for _ in range(3):
                res = fn()
            dt = (time.perf_counter() - t0) / 3
            line += "  %s %6.2f ms (exact=%s)" % (label, dt * 1e3, bool(np.array_equal(A.into_affine(cid, res), kg)))
        if host and name in ("random", "mixed", "witness"):
            hs = np.ascontiguousarray(sc)
            A.msm_bigint(cid, hb, hs)
            t0 = time.perf_counter()
            for _ in range(3):
                res = A.msm_bigint(cid, hb, hs)
            dt = (time.perf_counter() - t0) / 3
            line += "  host-pointer %6.2f ms (exact=%s)" % (dt * 1e3, bool(np.array_equal(A.into_affine(cid, res), kg)))
        if direct is not None:
            # the narrow entries (ark_hip_msm_sw_small_device): scalars in the reference's own integer type
            dt_np = {"bool": np.uint8, "u8": np.uint8, "u16": np.uint16, "u32": np.uint32, "u64": np.uint64}[name]
            sg = {1: np.int8, 2: np.int16, 4: np.int32, 8: np.int64}[np.dtype(dt_np).itemsize]
            dn = torch.from_numpy(np.ascontiguousarray(sc[:, 0].astype(dt_np)).view(sg)).cuda()
            direct(cid, bases, dn)
            t0 = time.perf_counter()
            for _ in range(3):
                res = direct(cid, bases, dn)
            dt = (time.perf_counter() - t0) / 3
            line += "  msm_%s-direct %6.2f ms (exact=%s)" % (name if name != "bool" else "u1", dt * 1e3,
                                                             bool(np.array_equal(A.into_affine(cid, res), kg)))
            mb = 1 if name == "bool" else 0
            pb.msm_small(dn, mb)
            t0 = time.perf_counter()
            for _ in range(3):
                res = pb.msm_small(dn, mb)
            dt = (time.perf_counter() - t0) / 3
            line += "  prepared-direct %6.2f ms (exact=%s)" % (dt * 1e3, bool(np.array_equal(A.into_affine(cid, res), kg)))
        print(line, flush=True)
    pb.free()


if __name__ == "__main__":
    main()
