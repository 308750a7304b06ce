//! `SWCurveConfig`s equal to the upstream ones except that `msm` runs on the MI355X:
//! `Projective::<HipBls12_381G1Config>::msm(&bases, &scalars)`.
use ark_hip::hip_sw_config;

hip_sw_config!(HipBls12_381G1Config, ark_bls12_381::g1::Config, ark_hip::BLS12_381_G1);
hip_sw_config!(HipBls12_381G2Config, ark_bls12_381::g2::Config, ark_hip::BLS12_381_G2);
hip_sw_config!(HipBls12_377G1Config, ark_bls12_377::g1::Config, ark_hip::BLS12_377_G1);
hip_sw_config!(HipBls12_377G2Config, ark_bls12_377::g2::Config, ark_hip::BLS12_377_G2);
hip_sw_config!(HipBn254G1Config, ark_bn254::g1::Config, ark_hip::BN254_G1);
