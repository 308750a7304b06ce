/*
 * TEST INFRASTRUCTURE (oracle) -- CPU restatement of the reference's MSM / radix-2 FFT hot path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call this library;
 * the product (libark_hip.so) never links or loads it.
 *
 * Parity pinned: tests/test_oracle_golden.py checks this oracle against the reference's own
 * known-answer vectors (k*G tables for BLS12-381 G1/G2, RFC 9380 points for BLS12-377 G2).
 *
 * All buffers are little-endian u64 limbs, Montgomery form, exactly the reference's in-memory
 * layout: Fp = [u64; N] (ff/src/biginteger/mod.rs:34), Affine = x|y with identity = all-zero
 * (short_weierstrass/affine.rs:91-104), Projective = x|y|z (group.rs:34-41), Fp2 = c0|c1
 * (quadratic_extension.rs:100-106). Scalars are 4-limb canonical BigInts unless stated.
 */
#ifndef ARK_ORACLE_H
#define ARK_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* field ids: 0 BN254_FQ, 1 BN254_FR, 2 BLS12_381_FQ, 3 BLS12_381_FR, 4 BLS12_377_FQ, 5 BLS12_377_FR
 * curve ids: 0 BN254_G1, 1 BLS12_381_G1, 2 BLS12_377_G1, 3 BLS12_377_G2, 4 BLS12_381_G2 */

int ark_oracle_field_limbs(int field);          /* u64 limbs per element */
int ark_oracle_curve_fe_words(int curve);       /* u64 words per base-field element (4, 6 or 12) */
int ark_oracle_curve_info(int curve, int* base_field, int* scalar_field, int* ext_degree);
int ark_oracle_field_const(int field, int which, uint64_t* out); /* 0 p, 1 R, 2 R2, 3 GEN, 4 ROOT ; returns n */
int ark_oracle_curve_generator(int curve, uint64_t* out_xy);

/* elementwise field ops on n elements. op: 0 add, 1 sub, 2 mul, 3 sqr, 4 neg, 5 dbl, 6 inv,
 * 7 into_bigint (Montgomery -> canonical), 8 from_bigint (canonical -> Montgomery) */
int ark_oracle_field_op(int field, int op, const uint64_t* a, const uint64_t* b, uint64_t* r, size_t n);
/* same for the curve's base field (Fp or Fp2): op 0 add, 1 sub, 2 mul, 3 sqr, 4 neg, 5 dbl, 6 inv */
int ark_oracle_basefield_op(int curve, int op, const uint64_t* a, const uint64_t* b, uint64_t* r, size_t n);

/* point ops. kinds: 0 jac += jac, 1 jac double, 2 bucket += affine, 3 bucket -= affine,
 * 4 bucket += bucket, 5 bucket double, 6 bucket -> jac, 7 affine double_to_bucket */
int ark_oracle_point_op(int curve, int kind, uint64_t* acc, const uint64_t* other);
int ark_oracle_to_affine(int curve, const uint64_t* jac, uint64_t* out_xy, size_t n);
int ark_oracle_scalar_mul(int curve, const uint64_t* base_xy, const uint64_t* scalar4, uint64_t* out_jac);
int ark_oracle_is_on_curve(int curve, const uint64_t* xy);
/* ScalarMul::batch_mul through BatchMulPreprocessing (ec/src/scalar_mul/mod.rs:104-251): out[i] = scalars[i] * base
 * as affine points; scalars = n x 4 canonical limbs; base = Projective x|y|z */
int ark_oracle_batch_mul(int curve, const uint64_t* base_jac, const uint64_t* scalars, size_t n, uint64_t* out_xy);

/* MSM. variant: 0 naive sum of double-and-add, 1 msm_bigint_wnaf (threads/2 chunks x 2 threads),
 * 2 msm_signed (full reference dispatch). scalars = n x 4 canonical limbs. out = Jacobian x|y|z. */
int ark_oracle_msm(int curve, const uint64_t* bases, const uint64_t* scalars, size_t n, int variant, int threads,
                   uint64_t* out_jac);
/* scalars given as Montgomery-form Fr elements (VariableBaseMSM::msm / msm_unchecked entry) */
int ark_oracle_msm_fr(int curve, const uint64_t* bases, const uint64_t* scalars_mont, size_t n, int variant,
                      int threads, uint64_t* out_jac);
/* signed base-2^c digits of one scalar (make_digits); returns digit count */
int ark_oracle_make_digits(const uint64_t* scalar4, int c, int num_bits, int64_t* out);
int ark_oracle_window_size(size_t n); /* c chosen by msm_bigint_wnaf_parallel for n points */

/* synthetic inputs: bases P_i = (a + i*b)*G ; uniform scalars in [0, r) by top-limb-masked rejection */
int ark_oracle_gen_bases(int curve, const uint64_t* a4, const uint64_t* b4, size_t n, uint64_t* out_xy);
int ark_oracle_gen_scalars(int field, uint64_t seed, size_t n, int montgomery, uint64_t* out);
/* k = sum_i s_i * (a + i*b) mod r  (discrete log of the MSM result w.r.t. G for gen_bases inputs) */
int ark_oracle_msm_dlog(int curve, const uint64_t* scalars, size_t n, const uint64_t* a4, const uint64_t* b4,
                        uint64_t* out_k4);

/* Radix-2 FFT over a scalar field, in place, natural order in and out.
 * data: 2^log_n Montgomery elements. offset: coset offset (Montgomery) or NULL for the subgroup.
 * inverse != 0: ifft (includes the size_inv / offset_inv^i scaling). */
int ark_oracle_fft(int field, uint64_t* data, unsigned log_n, const uint64_t* offset, int inverse, int threads);
/* domain constants as Radix2EvaluationDomain::new computes them: group_gen, group_gen_inv, size_inv */
int ark_oracle_domain(int field, unsigned log_n, uint64_t* group_gen, uint64_t* group_gen_inv, uint64_t* size_inv);
/* direct evaluation out[k] = sum_j coeffs[j] * (offset * g^k)^j (Horner), for small n */
int ark_oracle_dft_naive(int field, const uint64_t* coeffs, size_t ncoeffs, unsigned log_n, const uint64_t* offset,
                         uint64_t* out);

#ifdef __cplusplus
}
#endif
#endif
