"""ctypes binding of the CPU oracle (oracle/libark_oracle.so). TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_lib = C.CDLL(os.path.join(ROOT, "oracle", "libark_oracle.so"))

FIELDS = ["BN254_FQ", "BN254_FR", "BLS12_381_FQ", "BLS12_381_FR", "BLS12_377_FQ", "BLS12_377_FR"]
CURVES = ["BN254_G1", "BLS12_381_G1", "BLS12_377_G1", "BLS12_377_G2", "BLS12_381_G2"]
FID = {n: i for i, n in enumerate(FIELDS)}
CID = {n: i for i, n in enumerate(CURVES)}

u64p = C.POINTER(C.c_uint64)


def _p(a):
    if a is None:
        return None
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(u64p)


def field_limbs(field):
    return _lib.ark_oracle_field_limbs(field)


def fe_words(curve):
    return _lib.ark_oracle_curve_fe_words(curve)


def curve_info(curve):
    b, s, e = C.c_int(), C.c_int(), C.c_int()
    assert _lib.ark_oracle_curve_info(curve, C.byref(b), C.byref(s), C.byref(e)) == 0
    return b.value, s.value, e.value


def field_const(field, which):
    out = np.zeros(6, dtype=np.uint64)
    n = _lib.ark_oracle_field_const(field, which, _p(out))
    return out[:n].copy()


def generator(curve):
    fw = fe_words(curve)
    out = np.zeros(2 * fw, dtype=np.uint64)
    assert _lib.ark_oracle_curve_generator(curve, _p(out)) == 0
    return out


OPS = dict(add=0, sub=1, mul=2, sqr=3, neg=4, dbl=5, inv=6, into_bigint=7, from_bigint=8)


def field_op(field, op, a, b=None):
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1)
    n = a.size // field_limbs(field)
    r = np.zeros_like(a)
    bb = None if b is None else np.ascontiguousarray(b, dtype=np.uint64)
    assert _lib.ark_oracle_field_op(field, OPS[op], _p(a), _p(bb), _p(r), C.c_size_t(n)) == 0
    return r


def basefield_op(curve, op, a, b=None):
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1)
    n = a.size // fe_words(curve)
    r = np.zeros_like(a)
    bb = None if b is None else np.ascontiguousarray(b, dtype=np.uint64)
    assert _lib.ark_oracle_basefield_op(curve, OPS[op], _p(a), _p(bb), _p(r), C.c_size_t(n)) == 0
    return r


PKIND = dict(jac_add=0, jac_double=1, bkt_add_aff=2, bkt_sub_aff=3, bkt_add_bkt=4, bkt_double=5, bkt_to_jac=6,
             aff_double_to_bkt=7)


def point_op(curve, kind, acc, other=None):
    acc = np.ascontiguousarray(acc, dtype=np.uint64).copy()
    o = None if other is None else np.ascontiguousarray(other, dtype=np.uint64)
    assert _lib.ark_oracle_point_op(curve, PKIND[kind], _p(acc), _p(o)) == 0
    return acc


def to_affine(curve, jac):
    jac = np.ascontiguousarray(jac, dtype=np.uint64)
    fw = fe_words(curve)
    n = jac.size // (3 * fw)
    out = np.zeros(n * 2 * fw, dtype=np.uint64)
    assert _lib.ark_oracle_to_affine(curve, _p(jac), _p(out), C.c_size_t(n)) == 0
    return out.reshape(n, 2 * fw) if n != 1 else out


def scalar_mul(curve, base_xy, scalar4):
    fw = fe_words(curve)
    out = np.zeros(3 * fw, dtype=np.uint64)
    b = np.ascontiguousarray(base_xy, dtype=np.uint64)
    s = np.ascontiguousarray(scalar4, dtype=np.uint64)
    assert _lib.ark_oracle_scalar_mul(curve, _p(b), _p(s), _p(out)) == 0
    return out


def batch_mul(curve, base_jac, scalars):
    """ScalarMul::batch_mul restatement: scalars canonical [n, 4]; returns affine [n, 2*fw]."""
    fw = fe_words(curve)
    b = np.ascontiguousarray(base_jac, dtype=np.uint64)
    s = np.ascontiguousarray(scalars, dtype=np.uint64)
    n = s.size // 4
    out = np.zeros((n, 2 * fw), dtype=np.uint64)
    assert _lib.ark_oracle_batch_mul(curve, _p(b), _p(s), C.c_size_t(n), _p(out)) == 0
    return out


def is_on_curve(curve, xy):
    xy = np.ascontiguousarray(xy, dtype=np.uint64)
    return _lib.ark_oracle_is_on_curve(curve, _p(xy)) == 1


NAIVE, WNAF, SIGNED = 0, 1, 2


def msm(curve, bases, scalars, variant=SIGNED, threads=1, montgomery_scalars=False):
    fw = fe_words(curve)
    bases = np.ascontiguousarray(bases, dtype=np.uint64)
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
    n = min(bases.size // (2 * fw), scalars.size // 4)
    out = np.zeros(3 * fw, dtype=np.uint64)
    fn = _lib.ark_oracle_msm_fr if montgomery_scalars else _lib.ark_oracle_msm
    assert fn(curve, _p(bases), _p(scalars), C.c_size_t(n), variant, threads, _p(out)) == 0
    return out


def make_digits(scalar4, c, num_bits):
    s = np.ascontiguousarray(scalar4, dtype=np.uint64)
    out = np.zeros(300, dtype=np.int64)
    k = _lib.ark_oracle_make_digits(_p(s), c, num_bits, out.ctypes.data_as(C.POINTER(C.c_int64)))
    return out[:k].copy()


def window_size(n):
    return _lib.ark_oracle_window_size(C.c_size_t(n))


def gen_bases(curve, a4, b4, n):
    fw = fe_words(curve)
    out = np.zeros((n, 2 * fw), dtype=np.uint64)
    a = np.ascontiguousarray(a4, dtype=np.uint64)
    b = np.ascontiguousarray(b4, dtype=np.uint64)
    assert _lib.ark_oracle_gen_bases(curve, _p(a), _p(b), C.c_size_t(n), _p(out)) == 0
    return out


def gen_scalars(field, seed, n, montgomery=False):
    out = np.zeros((n, 4), dtype=np.uint64)
    assert _lib.ark_oracle_gen_scalars(field, C.c_uint64(seed), C.c_size_t(n), int(montgomery), _p(out)) == 0
    return out


def msm_dlog(curve, scalars, a4, b4):
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
    out = np.zeros(4, dtype=np.uint64)
    a = np.ascontiguousarray(a4, dtype=np.uint64)
    b = np.ascontiguousarray(b4, dtype=np.uint64)
    assert _lib.ark_oracle_msm_dlog(curve, _p(scalars), C.c_size_t(scalars.size // 4), _p(a), _p(b), _p(out)) == 0
    return out


def fft(field, data, log_n, offset=None, inverse=False, threads=1):
    d = np.ascontiguousarray(data, dtype=np.uint64).reshape(-1).copy()
    assert d.size == 4 << log_n
    o = None if offset is None else np.ascontiguousarray(offset, dtype=np.uint64)
    rc = _lib.ark_oracle_fft(field, _p(d), C.c_uint(log_n), _p(o), int(inverse), threads)
    assert rc == 0, rc
    return d


def domain(field, log_n):
    g, gi, si = (np.zeros(4, dtype=np.uint64) for _ in range(3))
    rc = _lib.ark_oracle_domain(field, C.c_uint(log_n), _p(g), _p(gi), _p(si))
    if rc != 0:
        return None
    return g, gi, si


def dft_naive(field, coeffs, log_n, offset=None):
    c = np.ascontiguousarray(coeffs, dtype=np.uint64)
    out = np.zeros(4 << log_n, dtype=np.uint64)
    o = None if offset is None else np.ascontiguousarray(offset, dtype=np.uint64)
    assert _lib.ark_oracle_dft_naive(field, _p(c), C.c_size_t(c.size // 4), C.c_uint(log_n), _p(o), _p(out)) == 0
    return out
