"""algebra_amd -- MI355X-native MSM and radix-2 FFT behind the arkworks trait surface.

Host-side mirror (Python, for tests/bench/multi-GPU plumbing) of the two reference interfaces the
HIP library replaces:

    ark_ec::VariableBaseMSM            -> algebra_amd.msm   (msm, msm_unchecked, msm_bigint)
    ark_poly::Radix2EvaluationDomain   -> algebra_amd.domain.Radix2EvaluationDomain

All arithmetic runs in libark_hip.so (hand-written HIP for gfx950) through the C ABI of
include/ark_hip.h; this package holds no arithmetic and no CPU fallback.
"""
from . import curves  # noqa: F401
from ._lib import ArkHipError, LIB_PATH, lib  # noqa: F401
from .msm import (BatchMulPreprocessing, batch_mul, ChunkedPippenger, HashMapPippenger, MsmJob, MsmLengthMismatch, PreparedBases, into_affine,  # noqa: F401
                  msm, msm_bigint, msm_bigint_async, msm_bigint_multi, msm_chunks, msm_u1, msm_u8, msm_u16, msm_u32,
                  msm_u64, msm_unchecked, normalize_batch, sum_projective, base_cache_config, base_cache_clear,
                  base_cache_stats, base_cache_hash_stats, ResidentBases, pin_bases, msm_plan, msm_plan_widths, MSM_WIDTH_TOP)
from .domain import Radix2EvaluationDomain  # noqa: F401
from .poly import DeviceVec, poly_mul, poly_mul_host  # noqa: F401
