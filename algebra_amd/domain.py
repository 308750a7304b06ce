"""Host mirror of ark_poly::Radix2EvaluationDomain (poly/src/domain/radix2/mod.rs:22-164) over the
EvaluationDomain trait surface (poly/src/domain/mod.rs:31-329): new, get_coset,
compute_size_of_domain, size, log_size_of_group, size_inv, group_gen, group_gen_inv, coset_offset,
coset_offset_inv, coset_offset_pow_size, fft / fft_in_place / ifft / ifft_in_place.

Coefficients are numpy uint64 arrays [len, 4] (host) or CUDA torch tensors (device) of Fr elements in
Montgomery form.  As in the reference, inputs shorter than the domain are zero-extended
(radix2/mod.rs:144,151), and a forward transform of at most size/4 coefficients takes the degree-aware path
(radix2/mod.rs:141, fft.rs:29-71): the device skips the butterfly stages that would only copy.
"""
import ctypes as C

import numpy as np

from . import curves as cv
from ._lib import Radix2DomainStruct, check, lib


def _is_torch(x):
    return type(x).__module__.startswith("torch")


class Radix2EvaluationDomain:
    def __init__(self, field, struct):
        self.field = cv.field_id(field)
        self._s = struct

    # ---- constructors -------------------------------------------------------------------------
    @classmethod
    def new(cls, field, num_coeffs):
        """Radix2EvaluationDomain::new (radix2/mod.rs:55-83); returns None where the reference does."""
        s = Radix2DomainStruct()
        rc = lib().ark_hip_radix2_domain_new(cv.field_id(field), num_coeffs, C.byref(s))
        if rc == -2:
            return None
        check(rc, "ark_hip_radix2_domain_new")
        return cls(field, s)

    def get_coset(self, offset):
        """get_coset (radix2/mod.rs:85-92); None for a zero offset."""
        o = np.ascontiguousarray(offset, dtype=np.uint64)
        s = Radix2DomainStruct()
        rc = lib().ark_hip_radix2_domain_get_coset(self.field, C.byref(self._s), o.ctypes.data_as(C.c_void_p),
                                                   C.byref(s))
        if rc == -1:
            return None
        check(rc, "ark_hip_radix2_domain_get_coset")
        return Radix2EvaluationDomain(self.field, s)

    @staticmethod
    def compute_size_of_domain(field, num_coeffs):
        d = Radix2EvaluationDomain.new(field, num_coeffs)
        return None if d is None else d.size()

    # ---- accessors ------------------------------------------------------------------------------
    def size(self):
        return int(self._s.size)

    def log_size_of_group(self):
        return int(self._s.log_size_of_group)

    def _fe(self, name):
        return np.array(list(getattr(self._s, name)), dtype=np.uint64)

    def size_as_field_element(self):
        return self._fe("size_as_field_element")

    def size_inv(self):
        return self._fe("size_inv")

    def group_gen(self):
        return self._fe("group_gen")

    def group_gen_inv(self):
        return self._fe("group_gen_inv")

    def coset_offset(self):
        return self._fe("offset")

    def coset_offset_inv(self):
        return self._fe("offset_inv")

    def coset_offset_pow_size(self):
        return self._fe("offset_pow_size")

    # ---- transforms ---------------------------------------------------------------------------
    def _resize(self, x):
        n = self.size()
        if _is_torch(x):
            import torch
            assert x.is_cuda and x.is_contiguous()
            rows = x.numel() * x.element_size() // 32
            if rows > n:
                raise ValueError("more coefficients than the domain size")
            if rows == n:
                return x
            y = torch.zeros((n, x.shape[-1]) if x.dim() == 2 else (n * 32 // x.element_size(),), dtype=x.dtype,
                            device=x.device)
            y.view(-1)[: x.numel()] = x.view(-1)
            return y
        a = np.ascontiguousarray(x, dtype=np.uint64).reshape(-1, 4)
        if a.shape[0] > n:
            raise ValueError("more coefficients than the domain size")
        if a.shape[0] < n:
            a = np.concatenate([a, np.zeros((n - a.shape[0], 4), dtype=np.uint64)])
        return a

    def _run(self, x, inverse, copy):
        L = lib()
        rows = (x.numel() * x.element_size() // 32) if _is_torch(x) else (np.asarray(x).size // 4)
        if not inverse and rows * 4 <= self.size() and rows > 0:
            return self._run_degree_aware(x, rows)
        x = self._resize(x)
        if _is_torch(x):
            import torch
            if copy:
                x = x.clone()
            torch.cuda.current_stream().synchronize()
            fn = L.ark_hip_ifft_in_place_device if inverse else L.ark_hip_fft_in_place_device
            check(fn(self.field, C.byref(self._s), x.data_ptr()), "ark_hip_(i)fft_in_place_device")
            check(L.ark_hip_synchronize(), "ark_hip_synchronize")
            return x
        if copy or not x.flags["WRITEABLE"]:
            x = x.copy()
        fn = L.ark_hip_ifft_in_place if inverse else L.ark_hip_fft_in_place
        check(fn(self.field, C.byref(self._s), x.ctypes.data_as(C.c_void_p)), "ark_hip_(i)fft_in_place")
        return x

    def _run_degree_aware(self, x, rows):
        """fft_in_place for coeffs.len() * 4 <= size (radix2/mod.rs:141): the buffer is grown to the domain size
        without copying zeros over PCIe and the library skips the leading stages."""
        L = lib()
        n = self.size()
        if _is_torch(x):
            import torch
            assert x.is_cuda and x.is_contiguous()
            y = torch.empty((n, 4), dtype=torch.int64, device=x.device)
            y.view(-1)[: rows * 4] = x.view(torch.int64).view(-1)
            torch.cuda.current_stream().synchronize()
            check(L.ark_hip_fft_in_place_degree_aware_device(self.field, C.byref(self._s), y.data_ptr(), rows),
                  "ark_hip_fft_in_place_degree_aware_device")
            check(L.ark_hip_synchronize(), "ark_hip_synchronize")
            return y if x.dtype == torch.int64 else y.view(x.dtype)
        a = np.ascontiguousarray(x, dtype=np.uint64).reshape(-1, 4)
        y = np.empty((n, 4), dtype=np.uint64)
        y[:rows] = a
        check(L.ark_hip_fft_in_place_degree_aware(self.field, C.byref(self._s), y.ctypes.data_as(C.c_void_p), rows),
              "ark_hip_fft_in_place_degree_aware")
        return y

    def fft(self, coeffs):
        """EvaluationDomain::fft (domain/mod.rs:92-96): evaluations over the (coset of the) subgroup."""
        return self._run(coeffs, False, True)

    def fft_in_place(self, coeffs):
        """fft_in_place (radix2/mod.rs:140-147).  Returns the (possibly re-allocated, zero-extended) buffer."""
        return self._run(coeffs, False, False)

    def ifft(self, evals):
        return self._run(evals, True, True)

    def ifft_in_place(self, evals):
        """ifft_in_place (radix2/mod.rs:150-153)."""
        return self._run(evals, True, False)

    def fft_batch_in_place(self, polys, inverse=False):
        """Several transforms over this domain at once (the polynomials a prover transforms together): CUDA tensors of
        exactly size() elements each, transformed in place, up to three in flight on the GPU
        (ark_hip_fft_batch_in_place_device; the reference's callers loop over fft_in_place).  Returns `polys`."""
        import torch
        L = lib()
        n = self.size()
        ptrs = (C.c_void_p * len(polys))()
        for i, x in enumerate(polys):
            assert _is_torch(x) and x.is_cuda and x.is_contiguous() and x.numel() * x.element_size() == n * 32
            ptrs[i] = x.data_ptr()
        torch.cuda.current_stream().synchronize()
        check(L.ark_hip_fft_batch_in_place_device(self.field, C.byref(self._s), ptrs, len(polys), int(bool(inverse))),
              "ark_hip_fft_batch_in_place_device")
        check(L.ark_hip_synchronize(), "ark_hip_synchronize")
        return polys

    def ifft_batch_in_place(self, evals):
        return self.fft_batch_in_place(evals, inverse=True)

    def __repr__(self):
        return "Radix-2 multiplicative subgroup of size %d" % self.size()

    def fft_group_in_place(self, curve, points, inverse=False):
        """EvaluationDomain::fft_in_place / ifft_in_place for T = Projective<P> (poly/src/domain/mod.rs:332-362; the
        reference's own use: poly/src/test.rs:57): `points` = size() Jacobian points (x | y | z limbs, the reference's
        Projective) of `curve`, whose scalar field is this domain's -- numpy uint64 [n, 3 * fe_words] in host memory or a
        torch CUDA tensor of the same bytes; transformed in place and returned.  Pad with identities (z = 0) yourself."""
        from . import curves as cv
        L = lib()
        cid = cv.curve_id(curve)
        if cv.field_id(cv.scalar_field(cid)) != self.field:   # (the C entry checks it too: the domain's generator in the curve's Fr)
            raise ValueError("a domain over %s cannot transform points of %s (scalar field %s)"
                             % (cv.FIELDS[self.field], cv.curve_name(cid), cv.scalar_field(cid)))
        n = self.size()
        if _is_torch(points):
            import torch
            assert points.is_cuda and points.is_contiguous() and points.numel() * points.element_size() == n * cv.projective_words(cid) * 8
            torch.cuda.current_stream().synchronize()
            check(L.ark_hip_fft_group_in_place_device(cid, C.byref(self._s), points.data_ptr(), 1 if inverse else 0),
                  "ark_hip_fft_group_in_place_device")
            check(L.ark_hip_synchronize(), "ark_hip_synchronize")
            return points
        a = np.ascontiguousarray(points, dtype=np.uint64)
        assert a.size == n * cv.projective_words(cid)
        check(L.ark_hip_fft_group_in_place(cid, C.byref(self._s), a.ctypes.data_as(C.c_void_p), 1 if inverse else 0),
              "ark_hip_fft_group_in_place")
        return a

