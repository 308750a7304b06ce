"""Test-side helpers around the product's C ABI (algebra_amd._lib) -- numpy in, numpy out.  The ark_hip_test_* hooks live in
libark_hip_test.so (test_lib()): the shipped libark_hip.so does not export them."""
import ctypes as C

import numpy as np

import algebra_amd as A
from algebra_amd import curves as cv
from algebra_amd._lib import check, lib, test_lib

OPS = dict(add=0, sub=1, mul=2, sqr=3, neg=4, dbl=5, into_bigint=7, from_bigint=8,
           lazy_mul=20, lazy_sqr=21, lazy_sop2=22)   # the 28-bit-limb device forms (csrc/devops.cuh: field_op_kernel)
PKIND = dict(bkt_add_aff=2, bkt_sub_aff=3, bkt_add_bkt=4, bkt_double=5, bkt_to_jac=6, aff_double_to_bkt=7,
             lazy_add_aff=12, lazy_sub_aff=13, lazy_add_bkt=14, lazy_add_acc=15)   # carry-free forms (LazyK, testops.cuh)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def field_op(field, op, a, b=None):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    words = 6 if field in (2, 4) else 4
    n = a.size // words
    bb = None if b is None else np.ascontiguousarray(b, dtype=np.uint64)
    r = np.zeros_like(a)
    check(test_lib().ark_hip_test_field_op(field, OPS[op], _p(a), _p(bb), _p(r), n), "test_field_op")
    return r


def basefield_op(curve, op, a, b=None):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    n = a.size // cv.fe_words(curve)
    bb = None if b is None else np.ascontiguousarray(b, dtype=np.uint64)
    r = np.zeros_like(a)
    check(test_lib().ark_hip_test_basefield_op(curve, OPS[op], _p(a), _p(bb), _p(r), n), "test_basefield_op")
    return r


def point_op(curve, kind, acc, other=None):
    fw = cv.fe_words(curve)
    acc = np.ascontiguousarray(acc, dtype=np.uint64)
    k = PKIND[kind]
    n = acc.size // (fw * (2 if k == 7 else 4))
    o = None if other is None else np.ascontiguousarray(other, dtype=np.uint64)
    out = np.zeros((n, fw * (3 if k == 6 else 4)), dtype=np.uint64)
    check(test_lib().ark_hip_test_point_op(curve, k, _p(acc), _p(o), _p(out), n), "test_point_op")
    return out


def gpu_extend_bases(curve, seed_bases, n, delta_fn):
    """Grow `seed_bases` (m points, P_i = (a + i b)G) to n points on the GPU: P[i + m] = P[i] + (m b)G.
    delta_fn(m) returns the affine point (m*b)G as numpy limbs.  Returns a CUDA uint8 tensor."""
    import torch
    ab = cv.affine_bytes(curve)
    m = seed_bases.shape[0]
    buf = torch.zeros(n * ab, dtype=torch.uint8, device="cuda")
    buf[: min(m, n) * ab] = torch.from_numpy(np.ascontiguousarray(seed_bases[: min(m, n)]).view(np.uint8).reshape(-1)).cuda()
    torch.cuda.synchronize()
    while m < n:
        cnt = min(m, n - m)
        d = np.ascontiguousarray(delta_fn(m), dtype=np.uint64)
        check(lib().ark_hip_sw_add_affine_device(curve, buf.data_ptr(), buf.data_ptr() + m * ab, cnt, _p(d)),
              "sw_add_affine_device")
        m += cnt
    return buf
