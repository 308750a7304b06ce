//! `test_group!(name; Group; sw)` (test-templates/src/groups.rs:438-470) is what every arkworks curve crate runs over its
//! groups (curves/bls12_381/src/curves/tests/mod.rs:9-10): group laws, affine conversion incl. `normalize_batch`,
//! serialisation, cofactor operations and -- through the `msm` arm (:189-214) -- the whole MSM suite of
//! test-templates/src/msm.rs: `test_var_base_msm` (:17-34), `test_var_base_msm_mixed_scalars` (:36-72),
//! `test_var_base_msm_specialized` (:74-110: msm_u1 .. msm_u64), `test_chunked_pippenger` (:112-131),
//! `test_hashmap_pippenger` (:133-157).  Here it runs with every MSM, normalize_batch and batch_mul on the GPU.
use ark_algebra_test_templates::*;

/// Route 1 (patches/0001-0005): the upstream types, their configs' hooks behind the `hip` feature.
mod patched {
    use super::*;
    use ark_bls12_377::{G1Projective as Bls377G1, G2Projective as Bls377G2};
    use ark_bls12_381::{G1Projective as Bls381G1, G2Projective as Bls381G2};
    use ark_bn254::G1Projective as Bn254G1;

    test_group!(bls12_381_g1; Bls381G1; sw);
    test_group!(bls12_381_g2; Bls381G2; sw);
    test_group!(bls12_377_g1; Bls377G1; sw);
    test_group!(bls12_377_g2; Bls377G2; sw);
    test_group!(bn254_g1; Bn254G1; sw);

    /// the marker the typed layer is bounded by is there for all five configs, with the library's ids
    #[test]
    fn configs_declare_themselves_served() {
        use ark_hip::HipServed;
        assert_eq!(<ark_bls12_381::g1::Config as HipServed>::CURVE, ark_hip::BLS12_381_G1);
        assert_eq!(<ark_bls12_381::g2::Config as HipServed>::CURVE, ark_hip::BLS12_381_G2);
        assert_eq!(<ark_bls12_377::g1::Config as HipServed>::CURVE, ark_hip::BLS12_377_G1);
        assert_eq!(<ark_bls12_377::g2::Config as HipServed>::CURVE, ark_hip::BLS12_377_G2);
        assert_eq!(<ark_bn254::g1::Config as HipServed>::CURVE, ark_hip::BN254_G1);
        assert_eq!(<ark_bls12_381::g1::Config as HipServed>::SCALAR_FIELD, ark_hip::sys::BLS12_381_FR);
        assert_eq!(<ark_bn254::g1::Config as HipServed>::SCALAR_FIELD, ark_hip::sys::BN254_FR);
    }
}

/// Route 2 (unmodified arkworks): the wrapper configs `hip_sw_config!` declares -- different types, same curves.
mod wrapped {
    use super::*;
    use ark_ec::short_weierstrass::Projective;
    use ark_hip_curves::*;

    type HipBls381G1 = Projective<HipBls12_381G1Config>;
    type HipBls381G2 = Projective<HipBls12_381G2Config>;
    type HipBls377G1 = Projective<HipBls12_377G1Config>;
    type HipBls377G2 = Projective<HipBls12_377G2Config>;
    type HipBn254G1 = Projective<HipBn254G1Config>;

    test_group!(hip_bls12_381_g1; HipBls381G1; sw);
    test_group!(hip_bls12_381_g2; HipBls381G2; sw);
    test_group!(hip_bls12_377_g1; HipBls377G1; sw);
    test_group!(hip_bls12_377_g2; HipBls377G2; sw);
    test_group!(hip_bn254_g1; HipBn254G1; sw);
}

/// Sizes the templates do not reach (their MSMs stop at 2^10 .. 11 x 2^10 pairs): 2^16 pairs against the reference's CPU
/// Pippenger (`msm_bigint_default`, patches/0001), and a base slice edited between two calls (the verified cache).
#[test]
fn msm_2_16_matches_the_cpu_pippenger_and_sees_an_edited_base() {
    use ark_bls12_381::{Fr, G1Affine, G1Projective};
    use ark_ec::{scalar_mul::variable_base::VariableBaseMSM, CurveGroup};
    use ark_ff::PrimeField;
    use ark_std::UniformRand;
    let rng = &mut ark_std::test_rng();
    let n = 1usize << 16;
    let g = G1Projective::rand(rng);
    let mut bases: Vec<G1Affine> = G1Projective::normalize_batch(&(0..n).map(|i| g * Fr::from(i as u64 + 1)).collect::<Vec<_>>());
    let scalars: Vec<Fr> = (0..n).map(|_| Fr::rand(rng)).collect();
    let bigints: Vec<_> = scalars.iter().map(|s| s.into_bigint()).collect();
    let cpu = ark_ec::scalar_mul::variable_base::msm_bigint_default::<G1Projective>(&bases, &bigints);
    assert_eq!(G1Projective::msm(&bases, &scalars).unwrap(), cpu);
    assert_eq!(G1Projective::msm_bigint(&bases, &bigints), cpu);
    bases[n / 3] = (g * Fr::from(0xdead_beefu64)).into_affine(); // same address, same length, new content
    let cpu2 = ark_ec::scalar_mul::variable_base::msm_bigint_default::<G1Projective>(&bases, &bigints);
    assert_ne!(cpu, cpu2);
    assert_eq!(G1Projective::msm(&bases, &scalars).unwrap(), cpu2);
}

/// `fft_in_place::<G1Projective>` (poly/src/test.rs:57 uses it): the transform over group elements reaches the device once
/// the curve's Projective is registered -- any typed entry point does it, or `serve_group_coefficients` directly.
#[test]
fn group_coefficients_round_trip_through_the_domain() {
    use ark_bls12_381::{Fr, G1Projective};
    use ark_poly::{EvaluationDomain, Radix2EvaluationDomain};
    use ark_std::UniformRand;
    ark_hip::serve_group_coefficients::<ark_bls12_381::g1::Config>();
    let rng = &mut ark_std::test_rng();
    let n = 1usize << 10;
    let d = Radix2EvaluationDomain::<Fr>::new(n).unwrap();
    let pts: Vec<G1Projective> = (0..n).map(|_| G1Projective::rand(rng)).collect();
    let mut v = pts.clone();
    d.fft_in_place(&mut v);
    assert_ne!(v, pts);
    d.ifft_in_place(&mut v);
    assert_eq!(v, pts);
}
