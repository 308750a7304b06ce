//! FFT hook for unmodified arkworks: a newtype over `Radix2EvaluationDomain<F>` implementing the open
//! `EvaluationDomain<F>` trait (poly/src/domain/mod.rs:31-329).  Every required item delegates; `fft_in_place` /
//! `ifft_in_place` go to the GPU when the coefficient type is the field itself and fall back to the inner CPU
//! implementation otherwise (`T = G1Projective` must keep working, poly/src/test.rs:57).  With patches/0003 the
//! upstream `Radix2EvaluationDomain` does this itself behind ark-poly's `hip` feature.
use ark_ff::FftField;
use ark_poly::domain::{DomainCoeff, EvaluationDomain, Radix2EvaluationDomain};
use ark_serialize::{CanonicalDeserialize, CanonicalSerialize};

#[derive(Copy, Clone, Hash, Eq, PartialEq, Debug, CanonicalSerialize, CanonicalDeserialize)]
pub struct HipRadix2EvaluationDomain<F: FftField>(pub Radix2EvaluationDomain<F>);

impl<F: FftField> HipRadix2EvaluationDomain<F> {
    fn on_device<T: DomainCoeff<F>>(&self, v: &mut ark_std::vec::Vec<T>, inverse: bool) -> bool {
        let d = &self.0;
        ark_hip_sys::radix2_fft_in_place(
            d.size,
            d.log_size_of_group,
            &[d.size_inv, d.group_gen, d.group_gen_inv, d.offset, d.offset_inv],
            v,
            T::zero(),
            inverse,
        )
    }
}

impl<F: FftField> EvaluationDomain<F> for HipRadix2EvaluationDomain<F> {
    type Elements = <Radix2EvaluationDomain<F> as EvaluationDomain<F>>::Elements;
    fn new(num_coeffs: usize) -> Option<Self> {
        Radix2EvaluationDomain::new(num_coeffs).map(Self)
    }
    fn get_coset(&self, offset: F) -> Option<Self> {
        self.0.get_coset(offset).map(Self)
    }
    fn compute_size_of_domain(num_coeffs: usize) -> Option<usize> {
        Radix2EvaluationDomain::<F>::compute_size_of_domain(num_coeffs)
    }
    fn size(&self) -> usize {
        self.0.size()
    }
    fn log_size_of_group(&self) -> u64 {
        self.0.log_size_of_group()
    }
    fn size_inv(&self) -> F {
        self.0.size_inv()
    }
    fn group_gen(&self) -> F {
        self.0.group_gen()
    }
    fn group_gen_inv(&self) -> F {
        self.0.group_gen_inv()
    }
    fn coset_offset(&self) -> F {
        self.0.coset_offset()
    }
    fn coset_offset_inv(&self) -> F {
        self.0.coset_offset_inv()
    }
    fn coset_offset_pow_size(&self) -> F {
        self.0.coset_offset_pow_size()
    }
    fn fft_in_place<T: DomainCoeff<F>>(&self, coeffs: &mut ark_std::vec::Vec<T>) {
        if !self.on_device(coeffs, false) {
            self.0.fft_in_place(coeffs)
        }
    }
    fn ifft_in_place<T: DomainCoeff<F>>(&self, evals: &mut ark_std::vec::Vec<T>) {
        if !self.on_device(evals, true) {
            self.0.ifft_in_place(evals)
        }
    }
    fn elements(&self) -> Self::Elements {
        self.0.elements()
    }
}
