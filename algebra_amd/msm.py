"""Host mirror of ark_ec::VariableBaseMSM for short-Weierstrass curves.

Reference interface (ec/src/scalar_mul/variable_base/mod.rs:37-151, short_weierstrass/mod.rs:112-119):
    msm(bases, scalars)            -> Result<Projective, usize>   Err(min_len) on length mismatch
    msm_unchecked(bases, scalars)  -> Projective                  truncates to the shorter input
    msm_bigint(bases, bigints)     -> Projective                  scalars already canonical BigInt<4>

Points and scalars are numpy uint64 arrays (host) or torch tensors on the GPU (device), in the
reference's in-memory layout: bases[n, 2*fe_words] (x|y Montgomery limbs, identity all-zero),
scalars[n, 4]; the result is a numpy uint64 array of 3*fe_words limbs (Jacobian x|y|z).
"""
import ctypes as C

import numpy as np

from . import curves as cv
from ._lib import check, lib


class MsmLengthMismatch(ValueError):
    """The reference's Err(min_len) (variable_base/mod.rs:73-77)."""

    def __init__(self, min_len):
        super().__init__("bases and scalars differ in length; shorter is %d" % min_len)
        self.min_len = min_len


def _is_torch(x):
    return type(x).__module__.startswith("torch")


def _rows(x, words):
    if _is_torch(x):
        return x.numel() * x.element_size() // (8 * words)
    return x.size // words


def _host(x):
    a = np.ascontiguousarray(x, dtype=np.uint64)
    return a, a.ctypes.data_as(C.c_void_p)


def _run(curve, bases, scalars, n, montgomery):
    cid = cv.curve_id(curve)
    out = np.zeros(cv.projective_words(cid), dtype=np.uint64)
    L = lib()
    if _is_torch(bases) or _is_torch(scalars):
        if not (_is_torch(bases) and _is_torch(scalars) and bases.is_cuda and scalars.is_cuda):
            raise TypeError("bases and scalars must both be CUDA tensors or both be numpy arrays")
        assert bases.is_contiguous() and scalars.is_contiguous()
        import torch
        torch.cuda.current_stream().synchronize()  # inputs may have been produced on torch's stream
        check(L.ark_hip_msm_sw_device(cid, bases.data_ptr(), scalars.data_ptr(), n, int(montgomery),
                                      out.ctypes.data_as(C.c_void_p)), "ark_hip_msm_sw_device")
    else:
        b, bp = _host(bases)
        s, sp = _host(scalars)
        check(L.ark_hip_msm_sw(cid, bp, sp, n, int(montgomery), out.ctypes.data_as(C.c_void_p)), "ark_hip_msm_sw")
    return out


def msm_unchecked(curve, bases, scalars):
    """VariableBaseMSM::msm_unchecked: scalars are Fr elements (Montgomery); inputs truncated to the shorter."""
    n = min(_rows(bases, cv.affine_words(curve)), _rows(scalars, cv.SCALAR_WORDS))
    return _run(curve, bases, scalars, n, True)


def msm(curve, bases, scalars):
    """VariableBaseMSM::msm / SWCurveConfig::msm: length check, then msm_unchecked."""
    nb, ns = _rows(bases, cv.affine_words(curve)), _rows(scalars, cv.SCALAR_WORDS)
    if nb != ns:
        raise MsmLengthMismatch(min(nb, ns))
    return _run(curve, bases, scalars, nb, True)


def msm_bigint(curve, bases, bigints):
    """VariableBaseMSM::msm_bigint: scalars are canonical BigInt<4> (< r)."""
    n = min(_rows(bases, cv.affine_words(curve)), _rows(bigints, cv.SCALAR_WORDS))
    return _run(curve, bases, bigints, n, False)


def sum_projective(curve, points):
    """Sum of Projective points on the host (Projective: Sum, group.rs:659-663): the multi-GPU combine."""
    cid = cv.curve_id(curve)
    p, pp = _host(points)
    n = p.size // cv.projective_words(cid)
    out = np.zeros(cv.projective_words(cid), dtype=np.uint64)
    check(lib().ark_hip_sw_sum(cid, pp, n, out.ctypes.data_as(C.c_void_p)), "ark_hip_sw_sum")
    return out


def into_affine(curve, points):
    """CurveGroup::into_affine for one or more Projective points (affine.rs:374-396)."""
    cid = cv.curve_id(curve)
    p, pp = _host(points)
    n = p.size // cv.projective_words(cid)
    out = np.zeros((n, cv.affine_words(cid)), dtype=np.uint64)
    check(lib().ark_hip_sw_into_affine(cid, pp, n, out.ctypes.data_as(C.c_void_p)), "ark_hip_sw_into_affine")
    return out[0] if p.ndim == 1 else out


class ChunkedPippenger:
    """Host mirror of ark_ec's streaming accumulator (ec/src/scalar_mul/variable_base/stream_pippenger.rs:10-66):
    buffers (base, bigint) pairs and runs an MSM whenever `max_msm_buffer` pairs are pending; `finalize`
    flushes and returns the total.  Each flush is one device MSM; partial results are summed on the host."""

    def __init__(self, curve, max_msm_buffer):
        self.curve = cv.curve_id(curve)
        self.buf_size = int(max_msm_buffer)
        self._bases = []
        self._scalars = []
        self._result = None

    @classmethod
    def with_size(cls, curve, max_msm_buffer):
        return cls(curve, max_msm_buffer)

    def add(self, base, scalar):
        self._bases.append(np.asarray(base, dtype=np.uint64).reshape(-1))
        self._scalars.append(np.asarray(scalar, dtype=np.uint64).reshape(-1))
        if len(self._bases) == self.buf_size:
            self._flush()

    def _flush(self):
        if not self._bases:
            return
        part = msm_bigint(self.curve, np.stack(self._bases), np.stack(self._scalars))
        self._result = part if self._result is None else sum_projective(self.curve, np.stack([self._result, part]))
        self._bases, self._scalars = [], []

    def finalize(self):
        self._flush()
        if self._result is None:
            return msm_bigint(self.curve, np.zeros((0, cv.affine_words(self.curve)), dtype=np.uint64),
                              np.zeros((0, 4), dtype=np.uint64))
        return self._result


def _small_to_bigint(values, signed_ok=False):
    v = np.asarray(values)
    if v.dtype == np.bool_:
        v = v.astype(np.uint64)
    if not np.issubdtype(v.dtype, np.unsignedinteger):
        raise TypeError("msm_u*: unsigned integer (or bool) scalars expected")
    out = np.zeros((v.size, 4), dtype=np.uint64)
    out[:, 0] = v.astype(np.uint64).reshape(-1)
    return out


def msm_u1(curve, bases, scalars):
    """VariableBaseMSM::msm_u1 (variable_base/mod.rs:89-93): boolean scalars.  On the device these are ordinary
    1-bit scalars: every point with a set bit lands in bucket 0 of window 0 (heavy-bucket path), the other
    windows are empty."""
    return msm_bigint(curve, bases, _small_to_bigint(scalars))


def msm_u8(curve, bases, scalars):
    """VariableBaseMSM::msm_u8 (mod.rs:95-99)."""
    return msm_bigint(curve, bases, _small_to_bigint(np.asarray(scalars, dtype=np.uint8)))


def msm_u16(curve, bases, scalars):
    """VariableBaseMSM::msm_u16 (mod.rs:101-105)."""
    return msm_bigint(curve, bases, _small_to_bigint(np.asarray(scalars, dtype=np.uint16)))


def msm_u32(curve, bases, scalars):
    """VariableBaseMSM::msm_u32 (mod.rs:107-111)."""
    return msm_bigint(curve, bases, _small_to_bigint(np.asarray(scalars, dtype=np.uint32)))


def msm_u64(curve, bases, scalars):
    """VariableBaseMSM::msm_u64 (mod.rs:113-117)."""
    return msm_bigint(curve, bases, _small_to_bigint(np.asarray(scalars, dtype=np.uint64)))


def normalize_batch(curve, points):
    """CurveGroup::normalize_batch (group.rs:302-319) on the device: Projective points (CUDA tensor, 3*fe_words u64
    per point) -> Affine points (CUDA tensor of the same dtype)."""
    import torch
    cid = cv.curve_id(curve)
    assert points.is_cuda and points.is_contiguous()
    n = points.numel() * points.element_size() // (8 * cv.projective_words(cid))
    out = torch.empty(n * cv.affine_words(cid) * 8 // points.element_size(), dtype=points.dtype, device=points.device)
    torch.cuda.current_stream().synchronize()
    check(lib().ark_hip_sw_normalize_batch_device(cid, points.data_ptr(), out.data_ptr(), n), "normalize_batch")
    return out
