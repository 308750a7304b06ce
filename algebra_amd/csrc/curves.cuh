// Curve configurations served by the device MSM (ids shared with include/ark_hip.h).
// Constants come from the reference's curve crates via tools/gen_constants.py (params.hpp):
//   curves/bn254/src/curves/g1.rs:16-63, curves/bls12_381/src/curves/g1.rs:28-59, g2.rs:40-75,
//   curves/bls12_377/src/curves/g1.rs:30-60, g2.rs:22-72 (+ fields/fq2.rs:12-13 NONRESIDUE = -5).
#pragma once
#include "ec.cuh"

#ifndef ARK_ACC_MIN_WAVES_G1
#define ARK_ACC_MIN_WAVES_G1 1
#endif
namespace arkhip {

struct BN254_G1 {
  static constexpr int ACC_MIN_WAVES = ARK_ACC_MIN_WAVES_G1;   // min waves per SIMD requested for the accumulate kernel
  static constexpr bool RELAXED = true;   // bucket accumulation on residues in [0, 2p) (fp.cuh, ec.cuh)
  static constexpr int ID = 0;
  static constexpr bool LAZY_A = true;    // accumulate kernels on carry-free limbs: 9 x 29 bits (fp28.cuh / ec28.cuh)
  typedef Fp<BN254_FQ> F;
  typedef F FA;                           // field type of the bucket-accumulation kernels
  static constexpr bool RELAXED_A = RELAXED;
  typedef BN254_FR S;
};
struct BLS12_381_G1 {
  static constexpr int ACC_MIN_WAVES = ARK_ACC_MIN_WAVES_G1;   // min waves per SIMD requested for the accumulate kernel
  static constexpr bool RELAXED = true;   // bucket accumulation on residues in [0, 2p) (fp.cuh, ec.cuh)
  static constexpr int ID = 1;
  static constexpr bool LAZY_A = true;    // accumulate kernels on carry-free 28-bit limbs (fp28.cuh / ec28.cuh)
  typedef Fp<BLS12_381_FQ> F;
  typedef F FA;                           // field type of the bucket-accumulation kernels
  static constexpr bool RELAXED_A = RELAXED;
  typedef BLS12_381_FR S;
};
struct BLS12_377_G1 {
  static constexpr int ACC_MIN_WAVES = ARK_ACC_MIN_WAVES_G1;   // min waves per SIMD requested for the accumulate kernel
  static constexpr bool RELAXED = true;   // bucket accumulation on residues in [0, 2p) (fp.cuh, ec.cuh)
  static constexpr int ID = 2;
  static constexpr bool LAZY_A = true;
  typedef Fp<BLS12_377_FQ> F;
  typedef F FA;                           // field type of the bucket-accumulation kernels
  static constexpr bool RELAXED_A = RELAXED;
  typedef BLS12_377_FR S;
};
struct BLS12_377_G2 {
  static constexpr int ACC_MIN_WAVES = 1;
  static constexpr bool RELAXED = false;
  static constexpr int ID = 3;
  static constexpr bool LAZY_A = true;    // accumulate / reduction kernels on carry-free 28-bit limbs, lane pairs (fp28x2.cuh)
  typedef Fp2<BLS12_377_FQ, 5> F;
  typedef Fp2Half<BLS12_377_FQ, 5> FA;    // bucket accumulation: one Fp2 element per lane PAIR (fp.cuh)
  static constexpr bool RELAXED_A = true;
  typedef BLS12_377_FR S;
};
struct BLS12_381_G2 {
  static constexpr int ACC_MIN_WAVES = 1;
  static constexpr bool RELAXED = false;
  static constexpr int ID = 4;
  static constexpr bool LAZY_A = true;
  typedef Fp2<BLS12_381_FQ, 1> F;
  typedef Fp2Half<BLS12_381_FQ, 1> FA;    // bucket accumulation: one Fp2 element per lane PAIR (fp.cuh)
  static constexpr bool RELAXED_A = true;
  typedef BLS12_381_FR S;
};

}  // namespace arkhip
