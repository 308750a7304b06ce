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
