//! MSM hook.  `SWCurveConfig::msm` is the reference's designed override point
//! (ec/src/models/short_weierstrass/mod.rs:111-119); `<Projective<P> as VariableBaseMSM>::msm` forwards to it
//! (group.rs:650-657).  A wrapper config delegates every constant to the upstream config and overrides `msm`.
use crate::sys;
use ark_ec::short_weierstrass::{Affine, Projective, SWCurveConfig};
use ark_ec::{scalar_mul::variable_base::VariableBaseMSM, CurveConfig};
use ark_ff::{BigInt, PrimeField};
use core::ffi::c_int;

/// Implemented for each upstream config served by libark_hip.so.
pub trait HipCurve: SWCurveConfig {
    const CURVE_ID: c_int;
    /// u64 words of one base-field element (4, 6 or 12)
    const FE_WORDS: usize;
}

/// Layout guard.  arkworks types are not `#[repr(C)]`; in practice `Fp(BigInt([u64; N]), PhantomData)`,
/// `Affine { x, y, infinity: () }` and `Projective { x, y, z }` are contiguous in declaration order.
/// The shim refuses to run (and falls back to the CPU path) if the sizes disagree.
fn layout_ok<P: HipCurve>() -> bool {
    core::mem::size_of::<Affine<P>>() == 2 * P::FE_WORDS * 8
        && core::mem::size_of::<Projective<P>>() == 3 * P::FE_WORDS * 8
        && core::mem::size_of::<P::ScalarField>() == 32
}

/// `VariableBaseMSM::msm_unchecked` on the GPU (scalars are Fr in Montgomery form: the `into_bigint`
/// pass of variable_base/mod.rs:60-62 runs on the device).  `None` => caller uses the CPU default.
pub fn msm_fr<P: HipCurve>(bases: &[Affine<P>], scalars: &[P::ScalarField]) -> Option<Projective<P>> {
    if !layout_ok::<P>() {
        return None;
    }
    let n = bases.len().min(scalars.len());
    let mut out = core::mem::MaybeUninit::<Projective<P>>::uninit();
    let rc = unsafe {
        sys::ark_hip_msm_sw(P::CURVE_ID, bases.as_ptr() as *const u64, scalars.as_ptr() as *const u64, n, 1,
                            out.as_mut_ptr() as *mut u64)
    };
    (rc == 0).then(|| unsafe { out.assume_init() })
}

/// `VariableBaseMSM::msm_bigint` on the GPU (canonical BigInt<4> scalars) -- what the reference benches
/// (bench-templates/src/macros/ec.rs:240) and ChunkedPippenger (stream_pippenger.rs:48) call.
pub fn msm_bigint<P: HipCurve>(bases: &[Affine<P>], bigints: &[<P::ScalarField as PrimeField>::BigInt])
                               -> Option<Projective<P>>
where
    P::ScalarField: PrimeField<BigInt = BigInt<4>>,
{
    if !layout_ok::<P>() {
        return None;
    }
    let n = bases.len().min(bigints.len());
    let mut out = core::mem::MaybeUninit::<Projective<P>>::uninit();
    let rc = unsafe {
        sys::ark_hip_msm_sw(P::CURVE_ID, bases.as_ptr() as *const u64, bigints.as_ptr() as *const u64, n, 0,
                            out.as_mut_ptr() as *mut u64)
    };
    (rc == 0).then(|| unsafe { out.assume_init() })
}

/// Declares `$name`, a drop-in `SWCurveConfig` equal to `$up` except that `msm` runs on the MI355X.
#[macro_export]
macro_rules! hip_sw_config {
    ($name:ident, $up:ty, $id:expr, $words:expr) => {
        #[derive(Clone, Default, PartialEq, Eq)]
        pub struct $name;
        impl ark_ec::CurveConfig for $name {
            type BaseField = <$up as ark_ec::CurveConfig>::BaseField;
            type ScalarField = <$up as ark_ec::CurveConfig>::ScalarField;
            const COFACTOR: &'static [u64] = <$up as ark_ec::CurveConfig>::COFACTOR;
            const COFACTOR_INV: Self::ScalarField = <$up as ark_ec::CurveConfig>::COFACTOR_INV;
        }
        impl ark_ec::short_weierstrass::SWCurveConfig for $name {
            const COEFF_A: Self::BaseField = <$up as ark_ec::short_weierstrass::SWCurveConfig>::COEFF_A;
            const COEFF_B: Self::BaseField = <$up as ark_ec::short_weierstrass::SWCurveConfig>::COEFF_B;
            const GENERATOR: ark_ec::short_weierstrass::Affine<Self> = ark_ec::short_weierstrass::Affine::new_unchecked(
                <$up as ark_ec::short_weierstrass::SWCurveConfig>::GENERATOR.x,
                <$up as ark_ec::short_weierstrass::SWCurveConfig>::GENERATOR.y,
            );
            #[inline(always)]
            fn mul_by_a(e: Self::BaseField) -> Self::BaseField {
                <$up as ark_ec::short_weierstrass::SWCurveConfig>::mul_by_a(e)
            }
            /// short_weierstrass/mod.rs:112-119: length check stays here, the sum runs on the GPU,
            /// any device error falls back to the reference's CPU path.
            fn msm(bases: &[ark_ec::short_weierstrass::Affine<Self>], scalars: &[Self::ScalarField])
                   -> Result<ark_ec::short_weierstrass::Projective<Self>, usize> {
                if bases.len() != scalars.len() {
                    return Err(bases.len().min(scalars.len()));
                }
                Ok($crate::msm::msm_fr::<Self>(bases, scalars)
                    .unwrap_or_else(|| ark_ec::scalar_mul::variable_base::VariableBaseMSM::msm_unchecked(bases, scalars)))
            }
        }
        impl $crate::msm::HipCurve for $name {
            const CURVE_ID: core::ffi::c_int = $id;
            const FE_WORDS: usize = $words;
        }
    };
}

#[cfg(feature = "bls12-381")]
hip_sw_config!(HipBls12_381G1Config, ark_bls12_381::g1::Config, crate::sys::BLS12_381_G1, 6);
#[cfg(feature = "bls12-381")]
hip_sw_config!(HipBls12_381G2Config, ark_bls12_381::g2::Config, crate::sys::BLS12_381_G2, 12);
#[cfg(feature = "bn254")]
hip_sw_config!(HipBn254G1Config, ark_bn254::g1::Config, crate::sys::BN254_G1, 4);
#[cfg(feature = "bls12-377")]
hip_sw_config!(HipBls12_377G1Config, ark_bls12_377::g1::Config, crate::sys::BLS12_377_G1, 6);
#[cfg(feature = "bls12-377")]
hip_sw_config!(HipBls12_377G2Config, ark_bls12_377::g2::Config, crate::sys::BLS12_377_G2, 12);

// So that `ark_ec::...` paths used above resolve without the user importing them.
#[allow(unused_imports)]
use {CurveConfig as _, VariableBaseMSM as _};

/// A base set (an SRS) kept in GPU memory across MSMs: uploaded once, then every `msm_bigint` call moves only the
/// scalars (what `ChunkedPippenger` / a prover's commit loop wants).
pub struct ResidentBases<P: HipCurve> {
    d_bases: *mut core::ffi::c_void,
    d_scalars: *mut core::ffi::c_void,
    n: usize,
    _p: core::marker::PhantomData<P>,
}
impl<P: HipCurve> ResidentBases<P> {
    pub fn upload(bases: &[Affine<P>]) -> Option<Self> {
        if !layout_ok::<P>() {
            return None;
        }
        let (mut db, mut ds) = (core::ptr::null_mut(), core::ptr::null_mut());
        let bytes = core::mem::size_of_val(bases);
        unsafe {
            if sys::ark_hip_malloc(bytes, &mut db) != 0 || sys::ark_hip_malloc(bases.len() * 32, &mut ds) != 0 {
                return None;
            }
            if sys::ark_hip_memcpy_h2d(db, bases.as_ptr() as *const _, bytes) != 0 {
                return None;
            }
        }
        Some(Self { d_bases: db, d_scalars: ds, n: bases.len(), _p: core::marker::PhantomData })
    }
    pub fn msm_bigint(&self, bigints: &[BigInt<4>]) -> Option<Projective<P>> {
        let n = self.n.min(bigints.len());
        let mut out = core::mem::MaybeUninit::<Projective<P>>::uninit();
        let rc = unsafe {
            if sys::ark_hip_memcpy_h2d(self.d_scalars, bigints.as_ptr() as *const _, n * 32) != 0 {
                return None;
            }
            sys::ark_hip_msm_sw_device(P::CURVE_ID, self.d_bases, self.d_scalars, n, 0, out.as_mut_ptr() as *mut u64)
        };
        (rc == 0).then(|| unsafe { out.assume_init() })
    }
}
impl<P: HipCurve> Drop for ResidentBases<P> {
    fn drop(&mut self) {
        unsafe {
            sys::ark_hip_free(self.d_bases);
            sys::ark_hip_free(self.d_scalars);
        }
    }
}
