"""Dense polynomial multiplication kept on the device: FFT(a), FFT(b), pointwise product, IFFT -- the caller of the
FFT in the reference (`&DensePolynomial * &DensePolynomial`, poly/src/polynomial/univariate/dense.rs:641-656, through
evaluate_over_domain_by_ref / Evaluations::interpolate, univariate/mod.rs:305-360, evaluations/univariate/mod.rs:40-50).
No host round trip between the three transforms."""
import ctypes as C

import numpy as np

from . import curves as cv
from ._lib import check, lib
from .domain import Radix2EvaluationDomain


def poly_mul_host(field, a, b):
    """`&a * &b` through the ONE host-pointer C entry the Rust hook binds (ark_hip_poly_mul: one upload, the three
    transforms and the pointwise product on the device, one download).  numpy in, numpy out (leading zeros dropped)."""
    fid = cv.field_id(field)
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)
    b = np.ascontiguousarray(b, dtype=np.uint64).reshape(-1, 4)
    cap = max(a.shape[0] + b.shape[0] - 1, 1)
    out = np.zeros((cap, 4), dtype=np.uint64)
    out_len = C.c_size_t(0)
    check(lib().ark_hip_poly_mul(fid, a.ctypes.data_as(C.c_void_p), a.shape[0], b.ctypes.data_as(C.c_void_p), b.shape[0],
                                 out.ctypes.data_as(C.c_void_p), C.byref(out_len)), "ark_hip_poly_mul")
    return out[: out_len.value]


def poly_mul(field, a, b):
    """Coefficients of a*b (numpy uint64 [len, 4], Montgomery Fr).  Zero polynomial (empty input) -> empty."""
    import torch
    fid = cv.field_id(field)
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)
    b = np.ascontiguousarray(b, dtype=np.uint64).reshape(-1, 4)
    if a.shape[0] == 0 or b.shape[0] == 0:
        return np.zeros((0, 4), dtype=np.uint64)
    out_len = a.shape[0] + b.shape[0] - 1
    dom = Radix2EvaluationDomain.new(fid, out_len)
    if dom is None:
        raise ValueError("field is not smooth enough to construct domain")  # the reference's expect()
    n = dom.size()
    L = lib()

    def up(x):
        t = torch.zeros((n, 4), dtype=torch.int64, device="cuda")
        t[: x.shape[0]] = torch.from_numpy(x.view(np.int64)).cuda()
        return t

    da, db = up(a), up(b)
    torch.cuda.synchronize()
    sref = C.byref(dom._s)
    ptrs = (C.c_void_p * 2)(da.data_ptr(), db.data_ptr())
    check(L.ark_hip_fft_batch_in_place_device(fid, sref, ptrs, 2, 0), "fft a, b")  # the two transforms in flight together
    check(L.ark_hip_fr_mul_device(fid, da.data_ptr(), db.data_ptr(), da.data_ptr(), n), "pointwise mul")
    check(L.ark_hip_ifft_in_place_device(fid, sref, da.data_ptr()), "ifft")
    check(L.ark_hip_synchronize(), "sync")
    out = da[:out_len].cpu().numpy().view(np.uint64)
    # DensePolynomial::from_coefficients_vec truncates leading zero coefficients (dense.rs truncate_leading_zeros),
    # which is how the reference's interpolate() result looks when a product vanishes at the top
    nz = np.nonzero(out.any(axis=1))[0]
    return out[: (nz[-1] + 1) if nz.size else 0]


class DeviceVec:
    """Fr vector owned by the library's allocator on the current GPU -- the Python mirror of `ark_hip::DeviceVec` (Rust:
    rust/ark-hip/src/device.rs, C++: include/ark_hip.hpp).  numpy in (`from_host`), numpy out (`to_host`), everything in
    between stays in HBM: `evaluate_over_domain` -> `+=`, `-=`, `*=`, `scale`, `negate` -> `interpolate`
    (polynomial/univariate/mod.rs:305-360, evaluations/univariate/mod.rs:40-50, :104-180) with ONE upload per input and
    ONE download.  Asynchronous on the library's stream; `to_host` waits."""

    def __init__(self, field, length, _zero=True):
        self.field = cv.field_id(field)
        self.len = int(length)
        self.cap = self.len
        self.ptr = C.c_void_p(0)
        if self.len:
            check(lib().ark_hip_malloc(self.len * 32, C.byref(self.ptr)), "ark_hip_malloc")
            if _zero:
                check(lib().ark_hip_memset_device(self.ptr, 0, self.len * 32), "ark_hip_memset_device")

    @classmethod
    def from_host(cls, field, x):
        a = np.ascontiguousarray(x, dtype=np.uint64).reshape(-1, 4)
        v = cls(field, a.shape[0], _zero=False)
        check(lib().ark_hip_memcpy_h2d(v.ptr, a.ctypes.data_as(C.c_void_p), a.nbytes), "ark_hip_memcpy_h2d")
        return v

    def to_host(self):
        out = np.empty((self.len, 4), dtype=np.uint64)
        check(lib().ark_hip_memcpy_d2h(out.ctypes.data_as(C.c_void_p), self.ptr, out.nbytes), "ark_hip_memcpy_d2h")
        return out

    def clone(self):
        v = DeviceVec(self.field, self.len, _zero=False)
        check(lib().ark_hip_memcpy_d2d(v.ptr, self.ptr, self.len * 32), "ark_hip_memcpy_d2d")
        return v

    def free(self):
        if self.ptr and self.ptr.value:
            check(lib().ark_hip_free(self.ptr), "ark_hip_free")
            self.ptr = C.c_void_p(0)
            self.len = self.cap = 0

    def __del__(self):
        try:
            self.free()
        except Exception:  # noqa: BLE001 -- interpreter shutdown
            pass

    def __len__(self):
        return self.len

    def resize_zeroed(self, new_len):
        """Vec::resize(new_len, F::zero()) (also truncates)"""
        L = lib()
        if new_len > self.cap:
            v = DeviceVec(self.field, new_len, _zero=False)
            check(L.ark_hip_memcpy_d2d(v.ptr, self.ptr, self.len * 32), "ark_hip_memcpy_d2d")
            old_ptr = self.ptr
            self.ptr, self.cap = v.ptr, new_len
            v.ptr = old_ptr   # freed with v
            v.free()
        old = self.len
        self.len = new_len
        if new_len > old:
            check(L.ark_hip_memset_device(C.c_void_p(self.ptr.value + old * 32), 0, (new_len - old) * 32), "ark_hip_memset_device")

    def _same(self, other):
        if other.len != self.len or other.field != self.field:
            raise ValueError("domains are unequal")   # the reference's assertion

    def __iadd__(self, other):
        self._same(other)
        check(lib().ark_hip_fr_add_device(self.field, self.ptr, other.ptr, self.ptr, self.len), "ark_hip_fr_add_device")
        return self

    def __isub__(self, other):
        self._same(other)
        check(lib().ark_hip_fr_sub_device(self.field, self.ptr, other.ptr, self.ptr, self.len), "ark_hip_fr_sub_device")
        return self

    def __imul__(self, other):
        self._same(other)
        check(lib().ark_hip_fr_mul_device(self.field, self.ptr, other.ptr, self.ptr, self.len), "ark_hip_fr_mul_device")
        return self

    def __itruediv__(self, other):
        """Evaluations /= Evaluations (a zero divisor gives zero, as ark_ff::batch_inversion leaves zeros in place)"""
        self._same(other)
        check(lib().ark_hip_fr_div_device(self.field, self.ptr, other.ptr, self.ptr, self.len), "ark_hip_fr_div_device")
        return self

    def batch_inverse(self):
        check(lib().ark_hip_fr_inverse_device(self.field, self.ptr, self.ptr, self.len), "ark_hip_fr_inverse_device")
        return self

    def scale(self, k):
        k = np.ascontiguousarray(k, dtype=np.uint64).reshape(4)
        check(lib().ark_hip_fr_scale_device(self.field, self.ptr, k.ctypes.data_as(C.c_void_p), self.ptr, self.len),
              "ark_hip_fr_scale_device")
        return self

    def negate(self):
        check(lib().ark_hip_fr_neg_device(self.field, self.ptr, self.ptr, self.len), "ark_hip_fr_neg_device")
        return self

    def evaluate_over_domain(self, domain):
        """coefficients -> evaluations over `domain`, in place (zero-extended; at most size/4 coefficients take the
        degree-aware path).  Returns self (now the evaluations)."""
        have = self.len
        if have > domain.size():
            raise ValueError("more coefficients than the domain size")
        self.resize_zeroed(domain.size())
        check(lib().ark_hip_fft_in_place_degree_aware_device(self.field, C.byref(domain._s), self.ptr, have),
              "ark_hip_fft_in_place_degree_aware_device")
        return self

    def interpolate(self, domain):
        """evaluations over `domain` -> coefficients, in place (Evaluations::interpolate).  Returns self."""
        if self.len != domain.size():
            raise ValueError("domains are unequal")
        check(lib().ark_hip_ifft_in_place_device(self.field, C.byref(domain._s), self.ptr), "ark_hip_ifft_in_place_device")
        return self
