//! ark-hip: plugs libark_hip.so (MI355X MSM + radix-2 FFT) into ark-ec / ark-poly 0.6.
//!
//! Two ways in:
//! * **patched arkworks** (patches/0001-0005): the upstream curve configs override `SWCurveConfig::msm` /
//!   `msm_bigint` behind their `hip` feature and call [`msm::sw_msm`] / [`msm::sw_msm_bigint`]; `G1Projective` stays
//!   the same type, `VariableBaseMSM::{msm, msm_unchecked, msm_bigint, msm_chunks}`, `ChunkedPippenger` and
//!   `HashMapPippenger` all reach the GPU, and so do `CurveGroup::normalize_batch` and `ScalarMul::batch_mul`
//!   (patches/0004).  ark-poly's `hip` feature does the same for `Radix2EvaluationDomain` (0003) and for
//!   `&DensePolynomial * &DensePolynomial` (0005: one upload, one download).
//! * **unmodified arkworks**: [`hip_sw_config!`] declares a wrapper `SWCurveConfig` (every item of the trait
//!   delegated, `msm` on the GPU) and [`domain::HipRadix2EvaluationDomain`] wraps the evaluation domain.
//!
//! * **chains of transforms**: [`device::DeviceVec`] / [`device::DeviceEvaluations`] keep coefficient and evaluation
//!   vectors in HBM -- `evaluate_over_domain` -> pointwise `+=`, `-=`, `*=` -> `interpolate` with one upload per input
//!   and one download, instead of two PCIe crossings per transform through the `EvaluationDomain` hook.
//!
//! Beyond the reference's surface: [`msm::PreparedBases`] (a fixed SRS resident on the GPU with its per-window
//! multiples), [`msm::MsmJob`] (asynchronous MSMs), [`msm::msm_multi`] (one MSM over all GPUs of the node).
//!
//! SOURCE ONLY: the image this repository is built in has no Rust toolchain; the identical C ABI is exercised by
//! tests/ through ctypes and through the compiled C++ mirror (include/ark_hip.hpp).
pub mod device;
pub mod domain;
pub mod msm;
pub use ark_hip_sys as sys;
pub use ark_hip_sys::{BLS12_377_G1, BLS12_377_G2, BLS12_381_G1, BLS12_381_G2, BN254_G1};
pub use device::{DeviceError, DeviceEvaluations, DeviceVec};
pub use msm::{serve_group_coefficients, sw_msm, sw_msm_bigint, HipServed};
#[cfg(feature = "ec-hook")]
pub use msm::{sw_batch_mul, sw_msm_small, sw_normalize_batch};
