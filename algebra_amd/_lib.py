"""ctypes binding of libark_hip.so (the C ABI declared in include/ark_hip.h).

There is no CPU fallback: if the shared library is missing or no GPU is visible, every compute
entry point raises.  Build it with ``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C algebra_amd/csrc -j8``.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ARK_HIP_LIB") or os.path.join(_HERE, "libark_hip.so")  # ARK_HIP_LIB: A/B builds (tools/)

u64p = C.POINTER(C.c_uint64)


class ArkHipError(RuntimeError):
    def __init__(self, code, what):
        super().__init__("%s failed with code %d" % (what, code))
        self.code = code


class Radix2DomainStruct(C.Structure):
    """``ark_hip_radix2_domain``: mirror of Radix2EvaluationDomain<F>'s fields (poly/src/domain/radix2/mod.rs:22-42)."""
    _fields_ = [
        ("size", C.c_uint64),
        ("log_size_of_group", C.c_uint32),
        ("_pad", C.c_uint32),
        ("size_as_field_element", C.c_uint64 * 4),
        ("size_inv", C.c_uint64 * 4),
        ("group_gen", C.c_uint64 * 4),
        ("group_gen_inv", C.c_uint64 * 4),
        ("offset", C.c_uint64 * 4),
        ("offset_inv", C.c_uint64 * 4),
        ("offset_pow_size", C.c_uint64 * 4),
    ]


# symbol -> (restype, argtypes); must list every function include/ark_hip.h declares
SYMBOLS = {
    "ark_hip_device_count": (C.c_int, []),
    "ark_hip_init": (C.c_int, [C.c_int]),
    "ark_hip_set_device": (C.c_int, [C.c_int]),
    "ark_hip_get_device": (C.c_int, []),
    "ark_hip_host_alloc": (C.c_int, [C.c_size_t, C.POINTER(C.c_void_p)]),
    "ark_hip_host_free": (C.c_int, [C.c_void_p]),
    "ark_hip_msm_sw_device_async": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_void_p)]),
    "ark_hip_msm_wait": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ark_hip_msm_bases_prepare": (C.c_int, [C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "ark_hip_msm_bases_prepare_device": (C.c_int, [C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "ark_hip_msm_bases_free": (C.c_int, [C.c_void_p]),
    "ark_hip_msm_bases_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                         C.POINTER(C.c_size_t)]),
    "ark_hip_msm_prepared": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]),
    "ark_hip_msm_prepared_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]),
    "ark_hip_msm_prepared_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_void_p)]),
    "ark_hip_msm_prepared_device_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_void_p)]),
    "ark_hip_msm_sw_chunks": (C.c_int, [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]),
    "ark_hip_msm_sw_multi": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]),
    "ark_hip_msm_sw_multi_device": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                              C.POINTER(C.c_size_t), C.c_int, C.c_void_p]),
    "ark_hip_batch_mul_table_new": (C.c_int, [C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "ark_hip_batch_mul_table_free": (C.c_int, [C.c_void_p]),
    "ark_hip_batch_mul": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]),
    "ark_hip_batch_mul_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]),
    "ark_hip_fft_in_place_degree_aware": (C.c_int, [C.c_int, C.POINTER(Radix2DomainStruct), C.c_void_p, C.c_size_t]),
    "ark_hip_fft_in_place_degree_aware_device": (C.c_int, [C.c_int, C.POINTER(Radix2DomainStruct), C.c_void_p, C.c_size_t]),
    "ark_hip_fft_batch_in_place_device": (C.c_int, [C.c_int, C.POINTER(Radix2DomainStruct), C.POINTER(C.c_void_p), C.c_size_t,
                                                    C.c_int]),
    "ark_hip_shutdown": (None, []),
    "ark_hip_synchronize": (C.c_int, []),
    "ark_hip_version": (C.c_char_p, []),
    "ark_hip_host_threads": (C.c_int, [C.POINTER(C.c_int)]),
    "ark_hip_curve_info": (C.c_int, [C.c_int] + [C.POINTER(C.c_int)] * 4),
    "ark_hip_malloc": (C.c_int, [C.c_size_t, C.POINTER(C.c_void_p)]),
    "ark_hip_free": (C.c_int, [C.c_void_p]),
    "ark_hip_memcpy_h2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "ark_hip_memcpy_d2h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "ark_hip_curve_generator": (C.c_int, [C.c_int, C.c_void_p]),
    "ark_hip_msm_sw": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]),
    "ark_hip_msm_sw_small": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p]),
    "ark_hip_msm_sw_small_device": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p]),
    "ark_hip_msm_prepared_small_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p]),
    "ark_hip_msm_bases_pin": (C.c_int, [C.c_int, C.c_void_p, C.c_size_t]),
    "ark_hip_msm_bases_unpin": (C.c_int, [C.c_int, C.c_void_p, C.c_size_t]),
    "ark_hip_msm_cache_config": (C.c_int, [C.c_longlong, C.c_int]),
    "ark_hip_msm_cache_clear": (C.c_int, []),
    "ark_hip_msm_cache_stats": (C.c_int, [C.POINTER(C.c_uint64)]),
    "ark_hip_msm_cache_hash_stats": (C.c_int, [C.POINTER(C.c_uint64)]),
    "ark_hip_msm_sw_device": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]),
    "ark_hip_msm_plan": (C.c_int, [C.c_int, C.c_size_t, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "ark_hip_msm_plan_widths": (C.c_int, [C.c_int, C.c_size_t, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_int),
                                          C.POINTER(C.c_int)]),
    "ark_hip_msm_set_timing": (C.c_int, [C.c_int]),
    "ark_hip_msm_last_timing": (C.c_int, [C.POINTER(C.c_double)]),
    "ark_hip_sw_sum": (C.c_int, [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "ark_hip_sw_into_affine": (C.c_int, [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "ark_hip_sw_add_affine_device": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "ark_hip_sw_normalize_batch_device": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]),
    "ark_hip_radix2_domain_new": (C.c_int, [C.c_int, C.c_size_t, C.POINTER(Radix2DomainStruct)]),
    "ark_hip_radix2_domain_get_coset": (C.c_int, [C.c_int, C.POINTER(Radix2DomainStruct), C.c_void_p,
                                                  C.POINTER(Radix2DomainStruct)]),
    "ark_hip_fft_in_place": (C.c_int, [C.c_int, C.POINTER(Radix2DomainStruct), C.c_void_p]),
    "ark_hip_ifft_in_place": (C.c_int, [C.c_int, C.POINTER(Radix2DomainStruct), C.c_void_p]),
    "ark_hip_fft_in_place_device": (C.c_int, [C.c_int, C.POINTER(Radix2DomainStruct), C.c_void_p]),
    "ark_hip_ifft_in_place_device": (C.c_int, [C.c_int, C.POINTER(Radix2DomainStruct), C.c_void_p]),
    "ark_hip_fr_mul_device": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "ark_hip_poly_mul": (C.c_int, [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_size_t)]),
    "ark_hip_sw_normalize_batch": (C.c_int, [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "ark_hip_fr_pow": (C.c_int, [C.c_int, C.c_void_p, C.c_uint64, C.c_void_p]),
    "ark_hip_fft_axis_device": (C.c_int, [C.c_int, C.c_void_p, C.c_uint, C.c_size_t, C.c_void_p]),
    "ark_hip_comm_unique_id": (C.c_int, [C.c_void_p]),
    "ark_hip_comm_init": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "ark_hip_comm_info": (C.c_int, [C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "ark_hip_comm_destroy": (C.c_int, []),
    "ark_hip_msm_sw_device_sharded": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]),
    "ark_hip_msm_prepared_device_sharded": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]),
    "ark_hip_msm_prepared_multi": (C.c_int, [C.c_int, C.POINTER(C.c_void_p), C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]),
    "ark_hip_fft_sharded_device": (C.c_int, [C.c_int, C.POINTER(Radix2DomainStruct), C.c_void_p, C.c_int]),
    "ark_hip_fft_shard_local_device": (C.c_int, [C.c_int, C.POINTER(Radix2DomainStruct), C.c_int, C.c_int, C.c_void_p, C.c_int]),
    "ark_hip_fft_shard_cross_device": (C.c_int, [C.c_int, C.POINTER(Radix2DomainStruct), C.c_int, C.c_void_p, C.c_void_p, C.c_int]),
    "ark_hip_fr_add_device": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "ark_hip_fr_sub_device": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "ark_hip_fr_neg_device": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]),
    "ark_hip_fft_group_in_place": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_int]),
    "ark_hip_fft_group_in_place_device": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_int]),
    "ark_hip_fr_div_device": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "ark_hip_fr_inverse_device": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]),
    "ark_hip_fr_scale_device": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "ark_hip_memcpy_d2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "ark_hip_memset_device": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t]),
    "ark_hip_fft_set_kernel": (C.c_int, [C.c_int]),
    "ark_hip_fft_set_timing": (C.c_int, [C.c_int]),
    "ark_hip_fft_last_timing": (C.c_int, [C.POINTER(C.c_double)]),
}

# the test hooks: exported by libark_hip_test.so only (the same objects as libark_hip.so + csrc/capi_test.hip); declared in
# include/ark_hip.h under ARK_HIP_TEST_HOOKS
TEST_SYMBOLS = {
    "ark_hip_test_field_op": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "ark_hip_test_basefield_op": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "ark_hip_test_host_basefield_op": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "ark_hip_test_point_op": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "ark_hip_test_msm_sharded_emulated": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                                    C.POINTER(C.c_size_t), C.c_int, C.c_void_p, C.POINTER(C.c_int)]),
    "ark_hip_test_base_hash": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64)]),
    "ark_hip_test_msm_host_fold": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
}
TEST_LIB_PATH = os.path.join(_HERE, "libark_hip_test.so")

_lib = None
_test_lib = None


def lib():
    """The loaded library; raises (loudly) when the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "algebra_amd: %s not found -- the HIP extension is not built and there is no CPU fallback "
                "(run `make -C algebra_amd/csrc -j8` or __graft_entry__.build())" % LIB_PATH)
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7 and cannot
        # initialise if a second copy (the system ROCm one this library would otherwise pull in)
        # is already loaded.  Importing torch first makes the dynamic linker resolve our
        # DT_NEEDED libamdhip64.so.7 to the copy torch loaded.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        l = C.CDLL(LIB_PATH)
        _bind(l, SYMBOLS, not os.environ.get("ARK_HIP_LIB"))   # an A/B build of an older revision (tools/): its entries only
        _lib = l
    return _lib


def _bind(l, table, strict):
    for name, (res, args) in table.items():
        try:
            fn = getattr(l, name)  # AttributeError if the library does not export it
        except AttributeError:
            if not strict:
                continue
            raise
        fn.restype = res
        fn.argtypes = args


def test_lib():
    """libark_hip_test.so: the product library's objects PLUS the ark_hip_test_* hooks (device field / point arithmetic, the
    host tail, the hashing pass) -- for tests/ and tools/ only.  A separate shared object, hence a separate library instance
    (its own contexts and helper pool) next to lib()."""
    global _test_lib
    if _test_lib is None:
        if not os.path.exists(TEST_LIB_PATH):
            raise ImportError("algebra_amd: %s not found (run `make -C algebra_amd/csrc -j8`)" % TEST_LIB_PATH)
        try:
            import torch  # noqa: F401  (one HIP runtime per process: see lib())
        except ImportError:
            pass
        l = C.CDLL(TEST_LIB_PATH)
        _bind(l, SYMBOLS, True)
        _bind(l, TEST_SYMBOLS, True)
        _test_lib = l
    return _test_lib


def check(code, what):
    if code != 0:
        raise ArkHipError(code, what)
