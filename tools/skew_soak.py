#!/usr/bin/env python3
"""Randomised soak of the non-uniform-scalar paths (width probe and classes, sliced pass B, heavy runs on the side stream,
runs walked by several lanes, narrow-scalar planner): random curve, size 2^17 .. 2^21, random mixture of the width classes
(zeros, +-1, +-u8, +-u16, +-u32, +-u64, 100-bit, full width, a few equal scalars), device-resident msm_bigint /
msm_unchecked (Montgomery) / host-pointer entry, now and then with forced window size or forced run parts -- every result
against k*G in closed form (tools/synth.py).   python tools/skew_soak.py [iterations] [seed] [log_lo log_hi]
(log_lo / log_hi: another size range, e.g. 8 16 for the small-n plan with split runs)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import algebra_amd as A
import synth as S
from algebra_amd import curves as cv

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 2024
rng = np.random.default_rng(seed)
LOG_LO = int(sys.argv[3]) if len(sys.argv) > 4 else 17
LOG_HI = int(sys.argv[4]) if len(sys.argv) > 4 else None
CURVES = ("BLS12_381_G1", "BN254_G1", "BLS12_377_G1", "BLS12_377_G2", "BLS12_381_G2")
LOGMAX = {"BLS12_381_G1": 21, "BN254_G1": 21, "BLS12_377_G1": 21, "BLS12_377_G2": 20, "BLS12_381_G2": 20}
bases = {}


def limbs(vals):
    out = np.zeros((len(vals), 4), dtype=np.uint64)
    for k in range(4):
        out[:, k] = [(v >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for v in vals]
    return out


def class_values(kind, m, r):
    """m scalars of one class as [m, 4] uint64 (canonical)"""
    a = np.zeros((m, 4), dtype=np.uint64)
    if kind == "zero" or m == 0:
        return a
    bits = {"u1": 1, "u8": 8, "u16": 16, "u32": 32, "u64": 64}.get(kind.lstrip("-"))
    if bits is not None:
        a[:, 0] = rng.integers(0, 1 << bits, size=m, dtype=np.uint64) if bits < 64 else rng.integers(0, 1 << 64, size=m, dtype=np.uint64)
        if kind.startswith("-"):   # r - x, limb arithmetic in numpy: r - x for x < 2^64 only borrows through limb 0
            rl = np.array(S.limbs4(r), dtype=np.uint64)
            x = a[:, 0].copy()
            a[:] = rl
            borrow = x > rl[0]
            a[:, 0] = rl[0] - x          # wraps where it borrows
            a[borrow, 1] -= np.uint64(1)  # r's limb 1 is far from zero for all three scalar fields
            a[x == 0] = 0
        return a
    if kind == "b100":
        a[:, 0] = rng.integers(0, 1 << 64, size=m, dtype=np.uint64)
        a[:, 1] = rng.integers(0, 1 << 36, size=m, dtype=np.uint64)
        return a
    if kind == "full":
        return S.gen_scalars(m, int(rng.integers(1, 1 << 30)), r)
    raise ValueError(kind)


KINDS = ("zero", "u1", "-u1", "u8", "-u8", "u16", "-u16", "u32", "-u32", "u64", "-u64", "b100", "full")
bad = 0
for it in range(iters):
    cname = CURVES[int(rng.integers(0, len(CURVES)))]
    cid = cv.curve_id(cname)
    r = S.R[cv.scalar_field(cid)]
    logn = int(rng.integers(LOG_LO, (LOG_HI if LOG_HI is not None else LOGMAX[cname]) + 1))
    n = (1 << logn) - int(rng.integers(0, min(1000, 1 << (logn - 1))))
    ab = cv.affine_words(cid) * 8                     # bytes per affine point (grow_bases returns a flat uint8 tensor)
    if cname not in bases:
        bases[cname] = S.grow_bases(cid, 1 << (LOG_HI if LOG_HI is not None else LOGMAX[cname]), S.A0, S.B0, r)
    b = bases[cname][: n * ab]
    # a random mixture: 1-4 classes with random weights
    k = int(rng.integers(1, 5))
    kinds = [KINDS[int(i)] for i in rng.choice(len(KINDS), size=k, replace=False)]
    w = rng.random(k) ** 2 + 0.01
    cnt = np.floor(w / w.sum() * n).astype(np.int64)
    cnt[0] += n - cnt.sum()
    sc = np.concatenate([class_values(kd, int(m), r) for kd, m in zip(kinds, cnt)])
    if rng.random() < 0.3:
        sc[:: int(rng.integers(2, 50))] = sc[int(rng.integers(0, n))]        # many equal scalars: one heavy bucket per window
    sc = sc[rng.permutation(n)] if rng.random() < 0.7 else sc                 # (unshuffled: classes in blocks -- the sampler's worst case)
    kg = S.mul_gen(cid, S.dlog_of_msm(sc, S.A0, S.B0, r), r)
    env = {}
    u = rng.random()
    if u < 0.2:
        env["ARK_HIP_MSM_RUN_PARTS"] = str(int(rng.choice([2, 4, 8])))
    elif u < 0.35:
        env["ARK_HIP_MSM_C"] = str(int(rng.integers(6, 19)))
    for kk, vv in env.items():
        os.environ[kk] = vv
    mode = int(rng.integers(0, 3))
    try:
        if mode == 0:
            d = torch.from_numpy(np.ascontiguousarray(sc).view(np.int64)).cuda()
            got = A.msm_bigint(cid, b, d)
        elif mode == 1:   # Fr elements in Montgomery form (what the trait's msm passes) on a 4096-pair prefix, then the whole vector
            R = (1 << 256) % r
            ms = limbs([(S.scalar_int(row) * R) % r for row in sc[:4096]])   # (the whole vector when n < 4096)
            d1 = torch.from_numpy(np.ascontiguousarray(ms).view(np.int64)).cuda()
            kg1 = S.mul_gen(cid, S.dlog_of_msm(sc[:4096], S.A0, S.B0, r), r)
            if not np.array_equal(A.into_affine(cid, A.msm_unchecked(cid, b[: min(n, 4096) * ab], d1)), kg1):
                bad += 1
                print("MONTGOMERY MISMATCH", cname, n, kinds, env, flush=True)
            d = torch.from_numpy(np.ascontiguousarray(sc).view(np.int64)).cuda()
            got = A.msm_bigint(cid, b, d)
        else:
            hb = b.cpu().numpy().view(np.uint64).reshape(n, -1)
            got = A.msm_bigint(cid, hb, np.ascontiguousarray(sc))
        ok = bool(np.array_equal(A.into_affine(cid, got), kg))
    finally:
        for kk in env:
            os.environ.pop(kk, None)
    if not ok:
        bad += 1
        print("MISMATCH", cname, n, kinds, cnt.tolist(), env, "mode", mode, flush=True)
    elif it % 10 == 0:
        print("it %d ok: %s n=%d %s %s mode %d" % (it, cname, n, kinds, env, mode), flush=True)
print("skew soak: %d iterations, %d mismatches (seed %d)" % (iters, bad, seed))
sys.exit(1 if bad else 0)
