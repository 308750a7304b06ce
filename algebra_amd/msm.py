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


def msm_plan(curve, n, prepared=False):
    """(window bits, windows) the library would use for n pairs (ark_hip_msm_plan: host arithmetic only)."""
    c, w = C.c_int(), C.c_int()
    check(lib().ark_hip_msm_plan(cv.curve_id(curve), int(n), int(bool(prepared)), C.byref(c), C.byref(w)), "ark_hip_msm_plan")
    return c.value, w.value


MSM_WIDTH_TOP = (0, 1, 8, 16, 32, 64, 128, 192, 256)


def msm_plan_widths(curve, n, max_bits, counts):
    """(window bits, windows) of a plain msm_bigint whose scalars fall into the width classes `counts` (nine entries:
    scalars with MSM_WIDTH_TOP[k-1] < bits(min(s, r - s)) <= MSM_WIDTH_TOP[k]) -- ark_hip_msm_plan_widths: the plan the
    device entry uses after its width probe (msm_signed's partition, variable_base/mod.rs:251-336)."""
    arr = (C.c_uint32 * 9)(*[int(x) for x in counts])
    c, w = C.c_int(), C.c_int()
    check(lib().ark_hip_msm_plan_widths(cv.curve_id(curve), int(n), int(max_bits), arr, C.byref(c), C.byref(w)),
          "ark_hip_msm_plan_widths")
    return c.value, w.value


class ResidentBases:
    """A base array pinned on the GPU (ark_hip_msm_bases_pin): while the pin lives the caller does NOT modify the
    array, and every host-array msm / msm_bigint / msm_u* whose bases are this array -- or a row range of it -- runs
    against the resident copy and uploads only its scalars.  The Rust twin (ark_hip::msm::ResidentBases) holds the
    shared borrow of the slice, which makes the promise a compile-time fact; here the array is made read-only for the
    pin's lifetime when numpy allows it.  Use as a context manager or call unpin()."""

    def __init__(self, curve, bases):
        self.curve = cv.curve_id(curve)
        a = np.asarray(bases)
        if a.dtype != np.uint64 or not a.flags.c_contiguous:
            raise TypeError("pin needs a C-contiguous uint64 array (a copy would be pinned at another address)")
        self.bases = a
        self.n = a.size // cv.affine_words(self.curve)
        self._was_writeable = bool(a.flags.writeable)
        check(lib().ark_hip_msm_bases_pin(self.curve, a.ctypes.data_as(C.c_void_p), self.n), "ark_hip_msm_bases_pin")
        self._pinned = True
        try:
            a.flags.writeable = False
        except ValueError:
            pass

    def unpin(self):
        if self._pinned:
            self._pinned = False
            check(lib().ark_hip_msm_bases_unpin(self.curve, self.bases.ctypes.data_as(C.c_void_p), self.n),
                  "ark_hip_msm_bases_unpin")
            if self._was_writeable:
                try:
                    self.bases.flags.writeable = True
                except ValueError:
                    pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.unpin()

    def __del__(self):
        try:
            self.unpin()
        except Exception:
            pass


def pin_bases(curve, bases):
    """ResidentBases(curve, bases): `with A.pin_bases(cid, srs): ...` keeps the SRS on the GPU for the block."""
    return ResidentBases(curve, bases)


def base_cache_config(budget_bytes=-1, auto_prepare_after=-1):
    """Verified resident-base cache of the host-array entry (ark_hip_msm_cache_config; on by default with a quarter of
    the device memory): numpy base sets passed to msm / msm_bigint are kept on the GPU per (array address, length) and
    validated on EVERY call by a hash of their full content -- an edited array is re-uploaded, never used stale.
    0 bytes disables it (every call then streams its bases), -2 restores the default budget."""
    check(lib().ark_hip_msm_cache_config(int(budget_bytes), int(auto_prepare_after)), "ark_hip_msm_cache_config")


def base_cache_clear():
    check(lib().ark_hip_msm_cache_clear(), "ark_hip_msm_cache_clear")


def base_cache_stats():
    out = (C.c_uint64 * 8)()
    check(lib().ark_hip_msm_cache_stats(out), "ark_hip_msm_cache_stats")
    return dict(zip(("entries", "bytes", "hits", "misses", "refreshed", "evicted", "pinned", "pinned_hits"),
                    [int(v) for v in out]))


def base_cache_hash_stats():
    """the verified cache's validation pass: calls streamed because the host was too busy to hash in time, the latest pass
    (us), its smoothed rate (MB/s), host threads per pass"""
    out = (C.c_uint64 * 4)()
    check(lib().ark_hip_msm_cache_hash_stats(out), "ark_hip_msm_cache_hash_stats")
    return dict(zip(("busy_streamed", "last_hash_us", "hash_mb_per_s", "threads"), [int(v) for v in out]))


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


class HashMapPippenger:
    """Host mirror of ark_ec's HashMapPippenger (stream_pippenger.rs:68-128): scalars of equal bases are summed in
    Fr (Montgomery residues add like integers mod r) before they reach the MSM; a flush runs one device MSM over the
    distinct bases (the Fr -> BigInt conversion of the reference's flush happens on the device)."""

    def __init__(self, curve, max_msm_buffer):
        self.curve = cv.curve_id(curve)
        self.buf_size = int(max_msm_buffer)
        self._r = cv.SCALAR_MODULUS[cv.scalar_field(self.curve)]
        self._map = {}
        self._result = None

    @staticmethod
    def _int(limbs):
        return sum(int(x) << (64 * i) for i, x in enumerate(limbs))

    def add(self, base, scalar):
        key = np.asarray(base, dtype=np.uint64).reshape(-1).tobytes()
        s = self._int(np.asarray(scalar, dtype=np.uint64).reshape(-1))
        self._map[key] = (self._map.get(key, 0) + s) % self._r
        if len(self._map) == self.buf_size:
            self._flush()

    def _flush(self):
        if not self._map:
            return
        bases = np.frombuffer(b"".join(self._map.keys()), dtype=np.uint64).reshape(len(self._map), -1)
        scalars = np.array([[(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)] for v in self._map.values()],
                           dtype=np.uint64)
        part = msm_unchecked(self.curve, bases, scalars)
        self._result = part if self._result is None else sum_projective(self.curve, np.stack([self._result, part]))
        self._map = {}

    def finalize(self):
        self._flush()
        if self._result is None:
            return msm_bigint(self.curve, np.zeros((0, cv.affine_words(self.curve)), dtype=np.uint64),
                              np.zeros((0, 4), dtype=np.uint64))
        return self._result


class MsmJob:
    """An MSM in flight on the device (ark_hip_msm_job): wait() blocks, finishes it and returns the Projective."""

    def __init__(self, curve, handle, keep):
        self.curve = curve
        self._h = handle
        self._keep = keep  # inputs must outlive the job

    def wait(self):
        if self._h is None:
            raise RuntimeError("job already waited for")
        out = np.zeros(cv.projective_words(self.curve), dtype=np.uint64)
        h, self._h = self._h, None
        check(lib().ark_hip_msm_wait(h, out.ctypes.data_as(C.c_void_p)), "ark_hip_msm_wait")
        self._keep = None
        return out

    def __del__(self):  # a job dropped without wait() would hold its device slot for ever
        if getattr(self, "_h", None) is not None:
            try:
                h, self._h = self._h, None
                lib().ark_hip_msm_wait(h, None)
            except Exception:  # noqa: BLE001 -- interpreter shutdown
                pass


def msm_bigint_async(curve, bases, bigints, montgomery=False):
    """Enqueue msm_bigint on device-resident inputs (CUDA tensors) and return an MsmJob."""
    cid = cv.curve_id(curve)
    if not (_is_torch(bases) and _is_torch(bigints) and bases.is_cuda and bigints.is_cuda):
        raise TypeError("msm_bigint_async takes CUDA tensors")
    import torch
    torch.cuda.current_stream().synchronize()
    n = min(_rows(bases, cv.affine_words(cid)), _rows(bigints, cv.SCALAR_WORDS))
    h = C.c_void_p()
    check(lib().ark_hip_msm_sw_device_async(cid, bases.data_ptr(), bigints.data_ptr(), n, int(montgomery), C.byref(h)),
          "ark_hip_msm_sw_device_async")
    return MsmJob(cid, h, (bases, bigints))


class PreparedBases:
    """A fixed base set (an SRS) resident on the GPU together with its table of per-window multiples
    (ark_hip_msm_bases): the `bases` argument of VariableBaseMSM::msm held across calls.  msm / msm_unchecked /
    msm_bigint have the reference's meaning over the first len(scalars) bases."""

    def __init__(self, curve, bases):
        self.curve = cv.curve_id(curve)
        self._h = C.c_void_p()
        L = lib()
        if _is_torch(bases):
            import torch
            assert bases.is_cuda and bases.is_contiguous()
            torch.cuda.current_stream().synchronize()
            self.n = _rows(bases, cv.affine_words(self.curve))
            check(L.ark_hip_msm_bases_prepare_device(self.curve, bases.data_ptr(), self.n, C.byref(self._h)),
                  "ark_hip_msm_bases_prepare_device")
        else:
            b, bp = _host(bases)
            self.n = b.size // cv.affine_words(self.curve)
            check(L.ark_hip_msm_bases_prepare(self.curve, bp, self.n, C.byref(self._h)), "ark_hip_msm_bases_prepare")

    def info(self):
        n, c, w, tb = C.c_size_t(), C.c_int(), C.c_int(), C.c_size_t()
        check(lib().ark_hip_msm_bases_info(self._h, C.byref(n), C.byref(c), C.byref(w), C.byref(tb)), "bases_info")
        return {"n": n.value, "window_bits": c.value, "windows": w.value, "table_bytes": tb.value}

    def _run(self, scalars, n, montgomery):
        out = np.zeros(cv.projective_words(self.curve), dtype=np.uint64)
        L = lib()
        if _is_torch(scalars):
            import torch
            assert scalars.is_cuda and scalars.is_contiguous()
            torch.cuda.current_stream().synchronize()
            check(L.ark_hip_msm_prepared_device(self._h, scalars.data_ptr(), n, int(montgomery),
                                                out.ctypes.data_as(C.c_void_p)), "ark_hip_msm_prepared_device")
        else:
            s, sp = _host(scalars)
            check(L.ark_hip_msm_prepared(self._h, sp, n, int(montgomery), out.ctypes.data_as(C.c_void_p)),
                  "ark_hip_msm_prepared")
        return out

    def msm_bigint(self, bigints):
        return self._run(bigints, min(self.n, _rows(bigints, cv.SCALAR_WORDS)), False)

    def msm_unchecked(self, scalars):
        return self._run(scalars, min(self.n, _rows(scalars, cv.SCALAR_WORDS)), True)

    def msm(self, scalars):
        ns = _rows(scalars, cv.SCALAR_WORDS)
        if ns != self.n:
            raise MsmLengthMismatch(min(ns, self.n))
        return self._run(scalars, ns, True)

    def msm_bigint_async(self, bigints, montgomery=False):
        """Enqueue and return an MsmJob.  Host scalars (numpy, ideally in pinned memory) upload on the copy stream
        while the previous MSM computes; CUDA tensors are used in place."""
        n = min(self.n, _rows(bigints, cv.SCALAR_WORDS))
        h = C.c_void_p()
        L = lib()
        if _is_torch(bigints):
            import torch
            torch.cuda.current_stream().synchronize()
            check(L.ark_hip_msm_prepared_device_async(self._h, bigints.data_ptr(), n, int(montgomery), C.byref(h)),
                  "ark_hip_msm_prepared_device_async")
            return MsmJob(self.curve, h, (self, bigints))
        s, sp = _host(bigints)
        check(L.ark_hip_msm_prepared_async(self._h, sp, n, int(montgomery), C.byref(h)), "ark_hip_msm_prepared_async")
        return MsmJob(self.curve, h, (self, s))

    def msm_small(self, scalars, max_bits=0):
        """msm_u1 / u8 / u16 / u32 / u64 against this base set: `scalars` is a CUDA tensor of 1-, 2-, 4- or 8-byte
        integers (ark_hip_msm_prepared_small_device)."""
        import torch
        assert _is_torch(scalars) and scalars.is_cuda and scalars.is_contiguous()
        torch.cuda.current_stream().synchronize()
        n = min(self.n, scalars.numel())
        out = np.zeros(cv.projective_words(self.curve), dtype=np.uint64)
        check(lib().ark_hip_msm_prepared_small_device(self._h, scalars.data_ptr(), n, scalars.element_size(), max_bits,
                                                      out.ctypes.data_as(C.c_void_p)), "ark_hip_msm_prepared_small_device")
        return out

    def free(self):
        if self._h is not None and self._h.value:
            check(lib().ark_hip_msm_bases_free(self._h), "ark_hip_msm_bases_free")
        self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def msm_chunks(curve, bases, scalars, step=0):
    """VariableBaseMSM::msm_chunks (variable_base/mod.rs:119-150) over host arrays: Fr (Montgomery) scalars, streams
    aligned at their end, steps of `step` pairs (0 = the reference's 2^20); uploads overlap the previous step."""
    cid = cv.curve_id(curve)
    b, bp = _host(bases)
    s, sp = _host(scalars)
    nb, ns = b.size // cv.affine_words(cid), s.size // cv.SCALAR_WORDS
    if ns > nb:
        raise AssertionError("scalars_stream.len() <= bases_stream.len()")  # the reference's assert!
    out = np.zeros(cv.projective_words(cid), dtype=np.uint64)
    check(lib().ark_hip_msm_sw_chunks(cid, bp, nb, sp, ns, step, out.ctypes.data_as(C.c_void_p)), "ark_hip_msm_sw_chunks")
    return out


def msm_bigint_multi(curve, n_gpus, bases, bigints, montgomery=False):
    """One MSM over n_gpus GPUs from this process (ark_hip_msm_sw_multi): host arrays, even base-range split."""
    cid = cv.curve_id(curve)
    b, bp = _host(bases)
    s, sp = _host(bigints)
    n = min(b.size // cv.affine_words(cid), s.size // cv.SCALAR_WORDS)
    out = np.zeros(cv.projective_words(cid), dtype=np.uint64)
    check(lib().ark_hip_msm_sw_multi(cid, n_gpus, bp, sp, n, int(montgomery), out.ctypes.data_as(C.c_void_p)),
          "ark_hip_msm_sw_multi")
    return out


class BatchMulPreprocessing:
    """Host mirror of ark_ec::scalar_mul::BatchMulPreprocessing (ec/src/scalar_mul/mod.rs:156-251): the table of
    multiples of one group element lives on the GPU; batch_mul(v) returns v[i] * base as Affine points."""

    def __init__(self, curve, base, num_scalars=0):
        """BatchMulPreprocessing::new(base, num_scalars); base: Projective limbs (x|y|z)."""
        self.curve = cv.curve_id(curve)
        b, bp = _host(base)
        self._h = C.c_void_p()
        check(lib().ark_hip_batch_mul_table_new(self.curve, bp, int(num_scalars), C.byref(self._h)),
              "ark_hip_batch_mul_table_new")

    def batch_mul(self, scalars, montgomery=True):
        """v: Fr elements (Montgomery) as in the reference; montgomery=False for canonical BigInt<4>."""
        L = lib()
        if _is_torch(scalars):
            import torch
            assert scalars.is_cuda and scalars.is_contiguous()
            n = _rows(scalars, cv.SCALAR_WORDS)
            out = torch.empty((n, cv.affine_words(self.curve)), dtype=torch.int64, device=scalars.device)
            torch.cuda.current_stream().synchronize()
            check(L.ark_hip_batch_mul_device(self._h, scalars.data_ptr(), n, int(montgomery), out.data_ptr()),
                  "ark_hip_batch_mul_device")
            return out
        s, sp = _host(scalars)
        n = s.size // cv.SCALAR_WORDS
        out = np.zeros((n, cv.affine_words(self.curve)), dtype=np.uint64)
        check(L.ark_hip_batch_mul(self._h, sp, n, int(montgomery), out.ctypes.data_as(C.c_void_p)), "ark_hip_batch_mul")
        return out

    def free(self):
        if self._h is not None and self._h.value:
            check(lib().ark_hip_batch_mul_table_free(self._h), "ark_hip_batch_mul_table_free")
        self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def batch_mul(curve, base, scalars, montgomery=True):
    """ScalarMul::batch_mul (ec/src/scalar_mul/mod.rs:104-107): table + batch_mul_with_preprocessing."""
    t = BatchMulPreprocessing(curve, base, _rows(scalars, cv.SCALAR_WORDS))
    try:
        return t.batch_mul(scalars, montgomery)
    finally:
        t.free()


def _msm_small(curve, bases, scalars, dtype, max_bits=0):
    """ark_hip_msm_sw_small(_device): the scalars cross the boundary as the reference's own narrow integers."""
    cid = cv.curve_id(curve)
    out = np.zeros(cv.projective_words(cid), dtype=np.uint64)
    L = lib()
    nbytes = np.dtype(dtype).itemsize
    if _is_torch(scalars):
        if not (_is_torch(bases) and bases.is_cuda and scalars.is_cuda):
            raise TypeError("bases and scalars must both be CUDA tensors or both be host arrays")
        assert bases.is_contiguous() and scalars.is_contiguous() and scalars.element_size() == nbytes
        import torch
        torch.cuda.current_stream().synchronize()
        n = min(_rows(bases, cv.affine_words(cid)), scalars.numel())
        check(L.ark_hip_msm_sw_small_device(cid, bases.data_ptr(), scalars.data_ptr(), n, nbytes, max_bits,
                                            out.ctypes.data_as(C.c_void_p)), "ark_hip_msm_sw_small_device")
        return out
    v = np.asarray(scalars)
    if v.dtype == np.bool_:
        v = v.astype(np.uint8)          # Rust's bool is one byte, 0 or 1
    if not np.issubdtype(v.dtype, np.unsignedinteger):
        raise TypeError("msm_u*: unsigned integer (or bool) scalars expected")
    v = np.ascontiguousarray(v.reshape(-1), dtype=dtype)
    b, bp = _host(bases)
    n = min(_rows(b, cv.affine_words(cid)), v.size)
    check(L.ark_hip_msm_sw_small(cid, bp, v.ctypes.data_as(C.c_void_p), n, nbytes, max_bits,
                                 out.ctypes.data_as(C.c_void_p)), "ark_hip_msm_sw_small")
    return out


def msm_u1(curve, bases, scalars):
    """VariableBaseMSM::msm_u1 (variable_base/mod.rs:89-93; msm_binary :373-390): boolean scalars, one byte each as in
    the reference's &[bool].  One window of one bucket: the sum of the selected bases (heavy-run kernels)."""
    return _msm_small(curve, bases, scalars, np.uint8, 1)


def msm_u8(curve, bases, scalars):
    """VariableBaseMSM::msm_u8 (mod.rs:95-99)."""
    return _msm_small(curve, bases, scalars, np.uint8)


def msm_u16(curve, bases, scalars):
    """VariableBaseMSM::msm_u16 (mod.rs:101-105)."""
    return _msm_small(curve, bases, scalars, np.uint16)


def msm_u32(curve, bases, scalars):
    """VariableBaseMSM::msm_u32 (mod.rs:107-111)."""
    return _msm_small(curve, bases, scalars, np.uint32)


def msm_u64(curve, bases, scalars):
    """VariableBaseMSM::msm_u64 (mod.rs:113-117)."""
    return _msm_small(curve, bases, scalars, np.uint64)


def normalize_batch(curve, points):
    """CurveGroup::normalize_batch (group.rs:302-319) on the device: Projective points (CUDA tensor or numpy array,
    3*fe_words u64 per point) -> Affine points (same kind)."""
    cid = cv.curve_id(curve)
    if not _is_torch(points):   # host array: the entry the Rust hook behind CurveGroup::normalize_batch binds
        p = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, cv.projective_words(cid))
        out = np.zeros((p.shape[0], cv.affine_words(cid)), dtype=np.uint64)
        check(lib().ark_hip_sw_normalize_batch(cid, p.ctypes.data_as(C.c_void_p), p.shape[0], out.ctypes.data_as(C.c_void_p)),
              "ark_hip_sw_normalize_batch")
        return out
    import torch
    assert points.is_cuda and points.is_contiguous()
    n = points.numel() * points.element_size() // (8 * cv.projective_words(cid))
    out = torch.empty(n * cv.affine_words(cid) * 8 // points.element_size(), dtype=points.dtype, device=points.device)
    torch.cuda.current_stream().synchronize()
    check(lib().ark_hip_sw_normalize_batch_device(cid, points.data_ptr(), out.data_ptr(), n), "normalize_batch")
    return out
