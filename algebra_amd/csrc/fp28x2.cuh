// Fp2 = Fp[u] / (u^2 - beta) on carry-free 28-bit limbs, ONE ELEMENT PER LANE PAIR (even lane: c0, odd lane: c1): the G2
// bucket arithmetic of the MSM (BASELINE config 5: BLS12-377 G2; BLS12-381 G2).
//
// Role in the reference: ff/src/fields/models/quadratic_extension.rs:626-670 (mul via Fp::sum_of_products, square by
// complex squaring), with the curve configs' small non-residues (curves/bls12_377/src/fields/fq2.rs:12-51: beta = -5;
// bls12_381 fq2.rs: beta = -1).  Values never leave the accumulate / reduction kernels: memory holds the reference's
// canonical limbs (Fp2Half, fp.cuh).
//
// Why: the saturated lane-pair form (Fp2Half) spends 432 v_mad_u64_u32 + 432 v_addc_co_u32 on the two-product sum behind
// one Fp2 product.  On 14 x 28-bit limbs the same sum is 588 multiply-adds and nothing else (FpL::sop2), a square is ONE
// product per lane (complex squaring: 392) and Y3 = R (Q - X3) - Y1 PPP is one four-product sum per lane (FpL::sop4: 980).
//
// Lane roles for A B, with a = this lane's component of A, pa = the partner's (DPP quad_perm [1,0,3,2]):
//     even (c0):  a0 b0 + (beta a1) b1          odd (c1):  a0 b1 + a1 b0
// beta a1 = NB (K p - a1) with NB = -beta, formed limb-wise from a spread K p (beta_neg) -- limbs below (NB + 1) 2^28.
//
// OPERAND CLASSES (limb sizes in units of 2^28; a column of 14 terms per product + 14 reduction terms must stay below
// 2^64 = 18.28 x 14 x 2^56):
//   mul<KA>(A, B):   even column  cA0 cB0 + (NB + 1) cB1 + 1,  odd column  cA0 cB1 + cA1 cB0 + 1
//                    -> A1 and B1 normalised (class 1); A0, B0 may be semi-normalised (class 3): 9 + 6 + 1 = 16
//   sqr<KW>(A):      A normalised;  even  (a0 + a1)(a0 + beta a1): 2 x 7 + 1 = 15;  result: c0 semi-normalised (s + 2 c1 for
//                    beta = -5), c1 normalised -- a legal A or B of mul
//   mul_sub<KA, KY>: all four operands normalised: 1 + 6 + 2 + 5 + 1 = 15
// Value bounds (units of p) are written at the call sites (ec28x2.cuh); R' / p >= 2520.
#pragma once
#include "fp28.cuh"

namespace arkhip {

template <class P_, int NEG_BETA>
struct Fp2L {
  typedef P_ P;
  typedef FpL<P> B;
  typedef Fp2Half<P, NEG_BETA> M;   // the same element in memory: canonical 32-bit limbs, one component per lane
  static constexpr int L = B::L;
  static constexpr int NB = NEG_BETA;
  static_assert(B::W == 28 && B::SEMI2, "lane-pair Fp2 is laid out for the 14 x 28-bit geometry");
  static_assert(NB == 1 || NB == 5, "non-residues served: -1 (BLS12-381), -5 (BLS12-377)");
  static_assert(14 * (9 + (NB + 1) + 1) < 256 && 14 * (2 * (NB + 2) + 1) < 256 && 14 * (1 + (NB + 1) + 2 + NB + 1) < 256,
                "column bounds of mul / sqr / mul_sub");
  B v;  // this lane's component

  ARK_DEV static bool odd() { return (threadIdx.x & 1u) != 0; }
  ARK_DEV static u32 swap1(u32 x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (u32)__builtin_amdgcn_mov_dpp((int)x, 0xB1, 0xF, 0xF, true);  // quad_perm [1,0,3,2]: the pair partner's value
#else
    return x;
#endif
  }
  ARK_DEV static B partner(const B& x) {
    B r;
#pragma unroll
    for (int i = 0; i < L; i++) r.l[i] = swap1(x.l[i]);
    return r;
  }
  ARK_DEV static bool both(bool mine) { return mine && (swap1(mine ? 1u : 0u) != 0); }
  ARK_DEV static Fp2L zero() { return Fp2L{B::zero()}; }
  ARK_DEV static Fp2L one() { return Fp2L{odd() ? B::zero() : B::one()}; }

  // beta x = NB (K p - x) for a normalised x below (K - 1/2) p: limb i = (NB K p spread, each limb lending NB 2^28) - NB x_i,
  // never negative, below (NB + 1) 2^28; value NB (K p - x) <= NB K p
  template <int K>
  ARK_DEV static B beta_neg(const B& x) {
    B r;
#pragma unroll
    for (int i = 0; i < L; i++) r.l[i] = B::template kp_spread_any<NB * K, NB>(i) - (u32)NB * x.l[i];
    return r;
  }
  // A B; A1 normalised and below (KA - 1/2) p, B1 normalised, A0 / B0 up to semi-normalised.  Output normalised, below
  // (|A0| |B0| + NB KA |B1|) / R' + 1 (even), (|A0| |B1| + |A1| |B0|) / R' + 1 (odd).
  template <int KA>
  ARK_DEV static Fp2L mul(const Fp2L& a, const Fp2L& b) {
    const B pa = partner(a.v), pb = partner(b.v);
    const B bz = beta_neg<KA>(pa);
    const bool o = odd();
    B X, Z;
#pragma unroll
    for (int i = 0; i < L; i++) {
      X.l[i] = o ? pa.l[i] : a.v.l[i];   // a0
      Z.l[i] = o ? a.v.l[i] : bz.l[i];   // a1 (odd lane) / beta a1 (even lane)
    }
    return Fp2L{B::sop2(X, b.v, Z, pb)};
  }
  // A^2 by complex squaring (quadratic_extension.rs:268-320): with t = a0 a1,
  //   a0^2 + beta a1^2 = (a0 + a1)(a0 + beta a1) - (1 + beta) t = s + (NB - 1) t,        2 a0 a1 = (2 a0) a1
  // even lane: s; odd lane: c1 = (2 a0) a1 directly -- ONE product per lane; then c0 = s + ((NB - 1) / 2) c1.
  // A normalised, below (KW - 1/2) p.  `zero` (optional): A = 0 mod p, decided on the two products (both below 2 p):
  // s = c1 = 0 mod p exactly when A^2 = 0, i.e. A = 0 (Fp2 is a field).
  template <int KW>
  ARK_DEV static Fp2L sqr(const Fp2L& a, bool* zero = nullptr) {
    static_assert((NB - 1) % 2 == 0, "c0 = s + ((NB - 1) / 2) c1");
    const B pa = partner(a.v);
    const B bz = beta_neg<KW>(pa);
    const bool o = odd();
    B X, Y;
#pragma unroll
    for (int i = 0; i < L; i++) {
      X.l[i] = o ? (pa.l[i] << 1) : a.v.l[i] + pa.l[i];   // 2 a0        | a0 + a1       (limbs < 2^29)
      Y.l[i] = o ? a.v.l[i] : a.v.l[i] + bz.l[i];         // a1          | a0 + beta a1  (limbs < (NB + 2) 2^28)
    }
    const B prod = B::mul(X, Y);
    if (zero) *zero = both(prod.is_zero_or_p());
    if constexpr (NB == 1) {
      return Fp2L{prod};
    } else {
      const B pp = partner(prod);   // even lane receives c1
      Fp2L r;
#pragma unroll
      for (int i = 0; i < L; i++) r.v.l[i] = o ? prod.l[i] : prod.l[i] + (u32)((NB - 1) / 2) * pp.l[i];
      return r;
    }
  }
  // A B - Y D under ONE reduction per lane (Y3 = R (Q - X3) - Y1 PPP): all four operands normalised; A1 below
  // (KA - 1/2) p, Y below (KY - 1/2) p.
  //   even:  a0 b0 + beta a1 b1 + (KY p - y0) d0 + NB y1 d1          (- beta y1 d1 = + NB y1 d1: no negation needed)
  //   odd:   a0 b1 + a1 b0 + (KY p - y0) d1 + (KY p - y1) d0
  template <int KA, int KY>
  ARK_DEV static Fp2L mul_sub(const Fp2L& a, const Fp2L& b, const Fp2L& y, const Fp2L& d) {
    const B pa = partner(a.v), pb = partner(b.v), pd = partner(d.v), py = partner(y.v);
    const B ny = B::template neg_semi<KY>(y.v);   // this lane's component of -Y, limbs < 2^29
    const B pny = partner(ny);
    const B bz = beta_neg<KA>(pa);
    const bool o = odd();
    B X0, X1, X2, X3;
#pragma unroll
    for (int i = 0; i < L; i++) {
      X0.l[i] = o ? pa.l[i] : a.v.l[i];
      X1.l[i] = o ? a.v.l[i] : bz.l[i];
      X2.l[i] = o ? pny.l[i] : ny.l[i];
      X3.l[i] = o ? ny.l[i] : (u32)NB * py.l[i];
    }
    return Fp2L{B::sop4(X0, b.v, X1, pb, X2, d.v, X3, pd)};
  }
  // component-wise differences (the lanes are independent)
  template <int K>
  ARK_DEV static Fp2L sub_sweep(const Fp2L& a, const Fp2L& b) { return Fp2L{B::template sub_sweep<K>(a.v, b.v)}; }
  template <int K>
  ARK_DEV static Fp2L sub_b_2c_norm(const Fp2L& a, const Fp2L& b, const Fp2L& c) {
    return Fp2L{B::template sub_b_2c_norm<K>(a.v, b.v, c.v)};
  }
  // component-wise product with the residue 1: brings a repacked canonical value (below 256 p) below 1.11 p
  ARK_DEV static Fp2L reduce_small(const Fp2L& a) { return Fp2L{B::mul(a.v, B::one())}; }

  // ---- the boundary with memory (Fp2Half: canonical limbs of this lane's component) ----
  ARK_DEV static Fp2L from_canonical(const M& m) { return Fp2L{B::unpack32_shl(m.v.l)}; }   // the residue itself, below 256 p
  ARK_DEV M to_canonical() const {   // normalised value below 256 p -> canonical
    return M{v.template shr_mod<B::SH>().to_canonical_bits()};
  }
};

}  // namespace arkhip
