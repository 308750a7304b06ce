//! ark-hip: plugs libark_hip.so (MI355X MSM + radix-2 FFT) into ark-ec / ark-poly.
//!
//! * `msm::Hip*Config`  -- `SWCurveConfig`s whose `msm` runs on the GPU: `Projective<HipBls12_381G1Config>::msm(..)`
//! * `domain::HipRadix2EvaluationDomain<F>` -- an `EvaluationDomain<F>` whose (i)fft runs on the GPU
//!
//! SOURCE ONLY: the image this repository is built in has no Rust toolchain; the identical C ABI is exercised by
//! tests/ through ctypes.  See INTEGRATION.md.
pub mod domain;
pub mod msm;
pub mod sys;
