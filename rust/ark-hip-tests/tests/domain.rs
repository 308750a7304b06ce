//! The radix-2 domain tests of the reference (poly/src/domain/radix2/mod.rs:351-600), restated over
//! `HipRadix2EvaluationDomain` -- the newtype route for unmodified arkworks.  (The in-crate originals run through the hook of
//! patches/0003 with `cargo test -p ark-poly --features hip`: rust/ci.sh does both.)
use ark_bls12_381::Fr;
use ark_ff::{FftField, Field, One, Zero};
use ark_hip::domain::HipRadix2EvaluationDomain as Dom;
use ark_poly::{
    univariate::DensePolynomial, DenseUVPolynomial, EvaluationDomain, Polynomial, Radix2EvaluationDomain,
};
use ark_std::{test_rng, UniformRand};

/// radix2/mod.rs:351-392 `test_fft_correctness`: every evaluation against `Polynomial::evaluate`, fft/ifft and the coset
/// pair as inverses.
#[test]
fn fft_correctness() {
    let log_degree = 5;
    let degree = 1usize << log_degree;
    let rand_poly = DensePolynomial::<Fr>::rand(degree - 1, &mut test_rng());
    for log_domain_size in log_degree..(log_degree + 2) {
        let domain_size = 1usize << log_domain_size;
        let domain = Dom::<Fr>::new(domain_size).unwrap();
        let coset_domain = domain.get_coset(Fr::GENERATOR).unwrap();
        let poly_evals = domain.fft(&rand_poly.coeffs);
        let poly_coset_evals = coset_domain.fft(&rand_poly.coeffs);
        for (i, (x, coset_x)) in domain.elements().zip(coset_domain.elements()).enumerate() {
            assert_eq!(poly_evals[i], rand_poly.evaluate(&x));
            assert_eq!(poly_coset_evals[i], rand_poly.evaluate(&coset_x));
        }
        let from_subgroup = DensePolynomial::from_coefficients_vec(domain.ifft(&poly_evals));
        let from_coset = DensePolynomial::from_coefficients_vec(coset_domain.ifft(&poly_coset_evals));
        assert_eq!(rand_poly, from_subgroup, "degree = {}, domain size = {}", degree, domain_size);
        assert_eq!(rand_poly, from_coset, "degree = {}, domain size = {}", degree, domain_size);
    }
}

/// radix2/mod.rs:394-410 `degree_aware_fft_correctness`: a short input on a domain four times its size (the device's
/// degree-aware entry: only the coefficients cross PCIe).
#[test]
fn degree_aware_fft_correctness() {
    let num_coeffs = 1usize << 5;
    let rand_poly = DensePolynomial::<Fr>::rand(num_coeffs - 1, &mut test_rng());
    let domain = Dom::<Fr>::new(num_coeffs * 4).unwrap();
    let coset_domain = domain.get_coset(Fr::GENERATOR).unwrap();
    let evals = domain.fft(&rand_poly.coeffs);
    let coset_evals = coset_domain.fft(&rand_poly.coeffs);
    for (i, (x, coset_x)) in domain.elements().zip(coset_domain.elements()).enumerate() {
        assert_eq!(evals[i], rand_poly.evaluate(&x));
        assert_eq!(coset_evals[i], rand_poly.evaluate(&coset_x));
    }
}

/// radix2/mod.rs:432-537 `parallel_fft_consistency`: the device against the reference's CPU domain on every size up to
/// 2^15 and every input length, all four transforms.
#[test]
fn device_matches_the_cpu_domain() {
    let rng = &mut test_rng();
    for log_d in 0..=15u32 {
        let d = 1usize << log_d;
        for len in [1usize, d / 4 + 1, d / 2 + 1, d] {
            let len = len.min(d);
            let v: Vec<Fr> = (0..len).map(|_| Fr::rand(rng)).collect();
            let cpu = Radix2EvaluationDomain::<Fr>::new(d).unwrap();
            let gpu = Dom::<Fr>(cpu);
            assert_eq!(gpu.fft(&v), cpu.fft(&v), "fft 2^{log_d} len {len}");
            assert_eq!(gpu.ifft(&v), cpu.ifft(&v), "ifft 2^{log_d} len {len}");
            let (cc, gc) = (cpu.get_coset(Fr::GENERATOR).unwrap(), gpu.get_coset(Fr::GENERATOR).unwrap());
            assert_eq!(gc.fft(&v), cc.fft(&v), "coset fft 2^{log_d} len {len}");
            assert_eq!(gc.ifft(&v), cc.ifft(&v), "coset ifft 2^{log_d} len {len}");
        }
    }
}

/// radix2/mod.rs:581-600 `test_fft_ifft_identity`, at a size where the device runs several passes.
#[test]
fn fft_ifft_identity() {
    let rng = &mut test_rng();
    let domain = Dom::<Fr>::new(1 << 18).unwrap();
    let v: Vec<Fr> = (0..domain.size()).map(|_| Fr::rand(rng)).collect();
    let mut w = v.clone();
    domain.fft_in_place(&mut w);
    assert_ne!(w, v);
    domain.ifft_in_place(&mut w);
    assert_eq!(w, v);
    // and the constants the device is handed are the domain's own (radix2/mod.rs:551-579)
    assert_eq!(domain.group_gen().pow([domain.size() as u64]), Fr::one());
    assert_eq!(domain.group_gen() * domain.group_gen_inv(), Fr::one());
    assert_eq!(domain.size_inv() * Fr::from(domain.size() as u64), Fr::one());
    assert!(!domain.coset_offset().is_zero());
}
