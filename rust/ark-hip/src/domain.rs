//! FFT hook: a newtype over `Radix2EvaluationDomain<F>` implementing the open `EvaluationDomain<F>` trait
//! (poly/src/domain/mod.rs:31-329).  Every accessor delegates; `fft_in_place` / `ifft_in_place` go to the GPU
//! when the coefficient type is the field itself and fall back to the inner CPU implementation otherwise
//! (`T = G1Projective` must keep working, poly/src/test.rs:57).
use crate::sys::{self, ark_hip_radix2_domain};
use ark_ff::{FftField, PrimeField};
use ark_poly::domain::{DomainCoeff, EvaluationDomain, Radix2EvaluationDomain};
use ark_serialize::{CanonicalDeserialize, CanonicalSerialize};
use core::any::TypeId;
use core::ffi::c_int;

/// Scalar fields served by libark_hip.so.
pub trait HipField: FftField + PrimeField + 'static {
    const FIELD_ID: c_int;
}
#[cfg(feature = "bls12-381")]
impl HipField for ark_bls12_381::Fr { const FIELD_ID: c_int = sys::BLS12_381_FR; }
#[cfg(feature = "bn254")]
impl HipField for ark_bn254::Fr { const FIELD_ID: c_int = sys::BN254_FR; }
#[cfg(feature = "bls12-377")]
impl HipField for ark_bls12_377::Fr { const FIELD_ID: c_int = sys::BLS12_377_FR; }

#[derive(Copy, Clone, Hash, Eq, PartialEq, Debug, CanonicalSerialize, CanonicalDeserialize)]
pub struct HipRadix2EvaluationDomain<F: HipField>(pub Radix2EvaluationDomain<F>);

fn limbs<F: PrimeField>(x: &F) -> [u64; 4] {
    // Fp is (BigInt<4>, PhantomData): the Montgomery residue is the in-memory value
    debug_assert_eq!(core::mem::size_of::<F>(), 32);
    unsafe { *(x as *const F as *const [u64; 4]) }
}

impl<F: HipField> HipRadix2EvaluationDomain<F> {
    fn c_struct(&self) -> ark_hip_radix2_domain {
        let d = &self.0;
        ark_hip_radix2_domain {
            size: d.size,
            log_size_of_group: d.log_size_of_group,
            _pad: 0,
            size_as_field_element: limbs(&d.size_as_field_element),
            size_inv: limbs(&d.size_inv),
            group_gen: limbs(&d.group_gen),
            group_gen_inv: limbs(&d.group_gen_inv),
            offset: limbs(&d.offset),
            offset_inv: limbs(&d.offset_inv),
            offset_pow_size: limbs(&d.offset_pow_size),
        }
    }
    fn on_device<T: DomainCoeff<F>>(&self, v: &mut Vec<T>, inverse: bool) -> bool {
        if TypeId::of::<T>() != TypeId::of::<F>() || core::mem::size_of::<T>() != 32 {
            return false;
        }
        v.resize(self.0.size as usize, T::zero()); // radix2/mod.rs:144,151
        let dom = self.c_struct();
        let rc = unsafe {
            if inverse {
                sys::ark_hip_ifft_in_place(F::FIELD_ID, &dom, v.as_mut_ptr() as *mut u64)
            } else {
                sys::ark_hip_fft_in_place(F::FIELD_ID, &dom, v.as_mut_ptr() as *mut u64)
            }
        };
        rc == 0 // on failure the buffer is untouched (the device works on its own copy)
    }
}

impl<F: HipField> EvaluationDomain<F> for HipRadix2EvaluationDomain<F> {
    type Elements = <Radix2EvaluationDomain<F> as EvaluationDomain<F>>::Elements;
    fn new(num_coeffs: usize) -> Option<Self> { Radix2EvaluationDomain::new(num_coeffs).map(Self) }
    fn get_coset(&self, offset: F) -> Option<Self> { self.0.get_coset(offset).map(Self) }
    fn compute_size_of_domain(num_coeffs: usize) -> Option<usize> {
        Radix2EvaluationDomain::<F>::compute_size_of_domain(num_coeffs)
    }
    fn size(&self) -> usize { self.0.size() }
    fn log_size_of_group(&self) -> u64 { self.0.log_size_of_group() }
    fn size_inv(&self) -> F { self.0.size_inv() }
    fn group_gen(&self) -> F { self.0.group_gen() }
    fn group_gen_inv(&self) -> F { self.0.group_gen_inv() }
    fn coset_offset(&self) -> F { self.0.coset_offset() }
    fn coset_offset_inv(&self) -> F { self.0.coset_offset_inv() }
    fn coset_offset_pow_size(&self) -> F { self.0.coset_offset_pow_size() }
    fn fft_in_place<T: DomainCoeff<F>>(&self, coeffs: &mut Vec<T>) {
        if !self.on_device(coeffs, false) {
            self.0.fft_in_place(coeffs)
        }
    }
    fn ifft_in_place<T: DomainCoeff<F>>(&self, evals: &mut Vec<T>) {
        if !self.on_device(evals, true) {
            self.0.ifft_in_place(evals)
        }
    }
    fn elements(&self) -> Self::Elements { self.0.elements() }
}
