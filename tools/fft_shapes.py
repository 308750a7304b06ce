#!/usr/bin/env python3
"""The reference's FFT bench shapes (poly/benches/fft.rs:55-163, BLS12-381 Fr, Radix2EvaluationDomain, degree 2^4 .. 2^22):
Subgroup FFT, Subgroup FFT on a 4x larger domain (degree-aware path, fft.rs:29-71), Subgroup IFFT, Coset FFT, Coset IFFT
(offset = Fr::GENERATOR = 7), device-resident data, through the C ABI.  One line per degree: ms per transform.
    python tools/fft_shapes.py [min_log] [max_log]"""
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
from algebra_amd._lib import check, lib
import synth as S

L = lib()
FIELD = "BLS12_381_FR"
lo = int(sys.argv[1]) if len(sys.argv) > 1 else 4
hi = int(sys.argv[2]) if len(sys.argv) > 2 else 22
r = S.R[FIELD]


def timed(fn, reps):
    for _ in range(max(3, reps // 4)):  # warm-up: tables cached, clocks up
        fn()
    check(L.ark_hip_synchronize(), "sync")
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    check(L.ark_hip_synchronize(), "sync")
    return (time.perf_counter() - t0) * 1e3 / reps


seven = np.array(S.limbs4(7 * (1 << 256) % r), dtype=np.uint64)  # Fr::GENERATOR = 7, Montgomery form
print("# degree  fft  fft_on_4x_domain  ifft  coset_fft  coset_ifft  fft_batch_of_8   (ms per transform, BLS12-381 Fr, device resident)")
for k in range(lo, hi + 1):
    n = 1 << k
    reps = 200 if k <= 16 else (50 if k <= 20 else 20)
    x = torch.from_numpy(S.gen_scalars(n, 11 + k, r).view(np.int64)).cuda()  # canonical ints < r: valid residues
    dom = A.Radix2EvaluationDomain.new(FIELD, n)
    big = A.Radix2EvaluationDomain.new(FIELD, 4 * n)
    cos = dom.get_coset(seven)
    sref, cref, bref = C.byref(dom._s), C.byref(cos._s), C.byref(big._s)
    y = x.clone()
    ybig = torch.zeros((4 * n, 4), dtype=torch.int64, device="cuda")
    ybig[:n] = x.reshape(n, 4)
    t_fft = timed(lambda: check(L.ark_hip_fft_in_place_device(dom.field, sref, y.data_ptr()), "fft"), reps)
    t_big = timed(lambda: check(L.ark_hip_fft_in_place_degree_aware_device(big.field, bref, ybig.data_ptr(), n), "fft4x"), reps)
    t_ifft = timed(lambda: check(L.ark_hip_ifft_in_place_device(dom.field, sref, y.data_ptr()), "ifft"), reps)
    t_cfft = timed(lambda: check(L.ark_hip_fft_in_place_device(cos.field, cref, y.data_ptr()), "cfft"), reps)
    t_cifft = timed(lambda: check(L.ark_hip_ifft_in_place_device(cos.field, cref, y.data_ptr()), "cifft"), reps)
    ys = [x.clone() for _ in range(8)]
    ptrs = (C.c_void_p * 8)(*[t.data_ptr() for t in ys])
    t_b8 = timed(lambda: check(L.ark_hip_fft_batch_in_place_device(dom.field, sref, ptrs, 8, 0), "batch"), max(3, reps // 8)) / 8
    print("2^%-2d  %8.4f  %8.4f  %8.4f  %8.4f  %8.4f  %8.4f" % (k, t_fft, t_big, t_ifft, t_cfft, t_cifft, t_b8), flush=True)
