#!/usr/bin/env python3
"""One-off soak: many MSMs / FFTs of random sizes and seeds against the oracle (looks for rare, timing-dependent
failures in the LDS-staged kernels).  usage: python tools/soak.py [iterations]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import algebra_amd as A
import oracle_lib as O

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(12345)
A4 = np.array([0xA11CE, 1, 2, 0], dtype=np.uint64)
B4 = np.array([0xB0B, 3, 0, 0], dtype=np.uint64)
CURVES = ("BLS12_381_G1", "BN254_G1", "BLS12_377_G2", "BLS12_377_G1")
bases = {c: O.gen_bases(O.CID[c], A4, B4, (1 << 12) if c.endswith("G2") else (1 << 15)) for c in CURVES}
prepared = {}  # (curve, off, n) windows re-prepared every few iterations
bad = 0
for it in range(iters):
    cname = CURVES[it % len(CURVES)]
    cid = O.CID[cname]
    cap = bases[cname].shape[0]
    n = int(rng.integers(1, cap))
    off = int(rng.integers(0, cap - n + 1))
    sc = O.gen_scalars(O.curve_info(cid)[1], 5000 + it, n)
    kind = it % 5
    if kind == 3:  # skewed: many small scalars
        sc[:, 1:] = 0
        sc[:, 0] &= np.uint64(0xFF)
    if kind == 4:  # many equal
        sc[::2] = sc[0]
    bs = bases[cname][off:off + n]
    if it % 7 == 6 and n >= 8:  # duplicate bases with equal scalars (doubling in one bucket), and P / -P pairs (infinity)
        bs = bs.copy()
        bs[1::3] = bs[0]
        sc[1::3] = sc[0]
        fw = bs.shape[1] // 2
        if n >= 16:
            bs[5, :fw] = bs[2, :fw]
            bs[5, fw:] = O.basefield_op(cid, "neg", bs[2, fw:].reshape(1, -1)).reshape(-1)
            sc[5] = sc[2]
    got = A.into_affine(cid, A.msm_bigint(cid, bs, sc))
    exp = O.to_affine(cid, O.msm(cid, bs, sc, O.SIGNED, 8))
    if not np.array_equal(got, exp):
        bad += 1
        print("MSM MISMATCH", cname, n, it)
    # the same sum over a prepared base set (random forced window size now and then), synchronous and as a job
    if it % 4 == 0:
        os.environ["ARK_HIP_MSM_C_PREPARED"] = str(int(rng.integers(3, 18)))
    else:
        os.environ.pop("ARK_HIP_MSM_C_PREPARED", None)
    pb = A.PreparedBases(cid, bs)
    os.environ.pop("ARK_HIP_MSM_C_PREPARED", None)
    job = pb.msm_bigint_async(sc)
    got_p = A.into_affine(cid, pb.msm_bigint(sc))
    got_j = A.into_affine(cid, job.wait())
    pb.free()
    if not (np.array_equal(got_p, exp) and np.array_equal(got_j, exp)):
        bad += 1
        print("PREPARED MSM MISMATCH", cname, n, it, pb)
    log_n = int(rng.integers(1, 17))
    fname = ("BLS12_381_FR", "BN254_FR", "BLS12_377_FR")[it % 3]
    fid = O.FID[fname]
    x = O.gen_scalars(fid, 9000 + it, 1 << log_n, montgomery=True)
    d = A.Radix2EvaluationDomain.new(fname, 1 << log_n)
    if it % 2:
        d = d.get_coset(O.field_const(fid, 3))
    inv = bool(it % 4 >= 2)
    got = (d.ifft(x) if inv else d.fft(x)).reshape(-1)
    exp = O.fft(fid, x, log_n, O.field_const(fid, 3) if it % 2 else None, inv, 8)
    if not np.array_equal(got, exp):
        bad += 1
        print("FFT MISMATCH", fname, log_n, it)
    if it % 3 == 0:  # the same transform as a batch of three device buffers (three streams)
        import torch
        devs = [torch.from_numpy(x.view(np.int64)).cuda() for _ in range(3)]
        d.fft_batch_in_place(devs, inverse=inv)
        if not all(np.array_equal(t.cpu().numpy().view(np.uint64).reshape(-1), exp) for t in devs):
            bad += 1
            print("FFT BATCH MISMATCH", fname, log_n, it)
    if not inv and log_n >= 3:  # ragged short input: degree-aware path
        ln = int(rng.integers(1, (1 << log_n) // 4 + 1))
        full = np.zeros((1 << log_n, 4), dtype=np.uint64)
        full[:ln] = x[:ln]
        if not np.array_equal(d.fft(x[:ln]).reshape(-1), O.fft(fid, full, log_n, O.field_const(fid, 3) if it % 2 else None, False, 8)):
            bad += 1
            print("DEGREE-AWARE FFT MISMATCH", fname, log_n, ln, it)
print("soak: %d iterations, %d mismatches" % (iters, bad))
sys.exit(1 if bad else 0)
