// The mixed addition of the bucket-accumulation kernel on carry-free 28-bit limbs (fp28.cuh).
//
// Bases and buckets live in HBM in the reference's form: canonical residues with radix R = 2^(32 N) = 2^384.  fp28.cuh
// computes with radix R' = 2^(28 L) = 2^392.  A gathered coordinate is converted by REPACKING ONLY: its canonical limbs
// shifted left by 8 bits are the integer x R 2^8 = x R', i.e. the residue x itself in the new radix -- unreduced (below
// 256 p), which is all a multiplication operand has to be: x2 and y2 enter the mixed addition only through the products
// U2 = x2 ZZ1 and S2 = y2 ZZZ1, whose outputs are small again (R' / p > 2^11).  The first point of a bucket (accumulator
// still at infinity) is brought below 1.13 p by one product with the residue 1.  Buckets leave through a division by 2^8
// mod p (v R' -> v R, FpL::shr_mod) and one conditional subtraction, in the reference's canonical form.
//
// Differences are formed WITHOUT carry sweeps (FpL::sub_semi: K p with limbs that cannot be outrun by a normalised
// subtrahend); only X3, which is subtracted from twice, is swept once.  "Is this difference zero mod p" is asked of the
// product that follows it (PP = P^2 is 0 mod p exactly when P is), where it is a comparison with 0 and p.
//
// Formulas and branches: ec/src/models/short_weierstrass/bucket.rs:168-238 (madd-2008-s), affine.rs:169-201
// (mdbl-2008-s-1, a = 0).  Bounds in the comments are in units of p; "n" = normalised limbs (< 2^28), "s" =
// semi-normalised (< 3 2^28).
//
// Two limb geometries share these formulas (FpL::W / L from params.hpp).  The bounds in the comments are for 14 x 28
// bits (Fp384: R' / p > 2048, bases enter below 2^8 p).  For 9 x 29 bits (BN254 Fq: R' / p >= 169, bases enter below
// 2^5 p) the same chain gives
//   x2, y2 < 32;  first point / U2 / S2 < 32 * 1.06 / 169 + 1 < 1.21;  P in (0.9, 7.21), R in (0.8, 3.21) -- both SWEPT to
//   normalised limbs (FpL::sub_op: a 29-bit column has no room for two semi-normalised operands, 90 2^58 > 2^64);
//   PP < 7.21^2 / 169 + 1 < 1.31 (the zero test needs < 2);  PPP < 7.21 * 1.31 / 169 + 1 < 1.06;  Q < 5.07 * 1.31 / 169 + 1
//   < 1.04;  R^2 < 1.07;  X3 = R^2 - PPP - 2 Q + 4p in (0.8, 5.07);  t = Q - X3 + 6p < 7.04 (s);
//   Y3 < (3.21 * 7.04 + 2 * 1.06) / 169 + 1 < 1.15;  ZZ3 < 1.06 * 1.31 / 169 + 1 < 1.01;  ZZZ3 < 1.01
//   columns: (n x s) 9 x 3 2^58 + (limbs < 2^30) x n 9 x 2 2^58 + reduction 9 x 2^58 = 54 2^58 < 2^64
//   full addition: U1 = X1 ZZ2 < 5.07 * 32 / 169 + 1 < 1.97 (hence P = U2 - U1 + 3p), S1 < 1.21 * 32 / 169 + 1 < 1.23,
//   P in (1.03, 4.21), PP < 1.11; leaving: shr_mod<5> output < 5.07 / 32 + 1 < 2.
#pragma once
#include "ec.cuh"
#include "fp28.cuh"

namespace arkhip {

template <class P>
struct XYZZL {
  FpL<P> x, y, zz, zzz;  // x < 5.01, y < 1.13, zz, zzz < 1.01 (all n)
  bool inf;
};

// a gathered base (canonical 32-bit limbs of the reference layout; the digit's sign already applied to y) as
// multiplication operands of the 28-bit form: the residues x, y themselves, below 256 p (n)
template <class P>
ARK_HD void lazy_from_affine(const Fp<P>& x, const Fp<P>& y, FpL<P>& xl, FpL<P>& yl) {
  xl = FpL<P>::unpack32_shl(x.l);
  yl = FpL<P>::unpack32_shl(y.l);
}

// stored bucket (canonical XYZZ, radix R) -> accumulator: repack as above, then one product with the residue 1 per
// coordinate brings it below 256 / 2048 + 1 < 1.13 (a later piece of a streamed MSM, MsmPiece: once per bucket)
template <class P>
ARK_HD XYZZL<P> lazy_from_bucket(const XYZZ<Fp<P>>& b) {
  typedef FpL<P> F;
  XYZZL<P> r;
  r.inf = b.is_zero();
  const F one = F::one();
  r.x = F::mul(F::unpack32_shl(b.x.l), one);
  r.y = F::mul(F::unpack32_shl(b.y.l), one);
  r.zz = F::mul(F::unpack32_shl(b.zz.l), one);
  r.zzz = F::mul(F::unpack32_shl(b.zzz.l), one);
  return r;
}
// accumulator -> bucket in the reference's canonical form: bits(v R) = bits(v R') / 2^8 mod p
template <class P>
ARK_HD XYZZ<Fp<P>> lazy_to_bucket(const XYZZL<P>& a) {
  if (a.inf) return XYZZ<Fp<P>>::zero();
  // shr_mod output < v / 2^8 + p: below 2 p for every coordinate (x < 5.01 -> 5.01 / 256 + 1)
  constexpr int SH = FpL<P>::SH;
  XYZZ<Fp<P>> r;
  r.x = a.x.template shr_mod<SH>().to_canonical_bits();
  r.y = a.y.template shr_mod<SH>().to_canonical_bits();
  r.zz = a.zz.template shr_mod<SH>().to_canonical_bits();
  r.zzz = a.zzz.template shr_mod<SH>().to_canonical_bits();
  return r;
}

// affine doubling (mdbl-2008-s-1, a = 0), all in 28-bit limbs; x2, y2 the operands of lazy_from_affine (< 256 p).  Rare
// branch (equal points in one bucket): out of line so that its registers do not burden the hot loops.
template <class P>
ARK_COLD_HD void xyzz_mdbl_lazy_xy(XYZZL<P>& acc, const FpL<P>& x2, const FpL<P>& y2) {
  typedef FpL<P> F;
  const F one = F::one();
  const F x1 = F::mul(x2, one);                                   // < 1.13 (n)
  const F y = F::mul(y2, one);                                    // < 1.13 (n)
  const F u = F::template sub<0>(F::add_lazy(y, y), F::zero());   // 2 y, < 2.3 (n)
  const F v = F::sqr(u);                                          // < 1.01
  const F w = F::mul(u, v);                                       // < 1.01
  const F s = F::mul(x1, v);                                      // < 1.01
  const F xx = F::sqr(x1);                                        // < 1.01
  const F m = F::template sub<0>(F::add_lazy(F::add_lazy(xx, xx), xx), F::zero());  // 3 xx, < 3.03 (n)
  acc.x = F::template sub_b_2c_norm<4>(F::sqr(m), F::zero(), s); // m^2 - 2 s + 4p: (1.9, 5.01), n
  const F t = F::template sub_semi<6>(s, acc.x);                  // (0.9, 7.02), s
  acc.y = F::sop2(m, t, F::template neg_semi<2>(w), y);           // m t - w y: < (3.03 * 7.02 + 2 * 1.13) / 2048 + 1 < 1.02
  acc.zz = v;
  acc.zzz = w;
  acc.inf = false;
}
// the same for the base at `src` (reference layout, canonical limbs; neg: the digit's sign): the accumulate kernels'
// form -- it re-reads the base, so that nothing of their hot loop has to stay alive for a doubling that almost never comes
template <class P>
ARK_COLD_HD void xyzz_mdbl_lazy(XYZZL<P>& acc, const char* src, bool neg) {
  const Affine<Fp<P>> b = Affine<Fp<P>>::load(src);
  FpL<P> x2, y2;
  lazy_from_affine<P>(b.x, Fp<P>::cond_neg(b.y, neg), x2, y2);
  xyzz_mdbl_lazy_xy<P>(acc, x2, y2);
}

// acc += (x2, y2): a non-identity base from lazy_from_affine (x2, y2 < 256, n; the digit's sign is in y2 already).
// Returns true when the base EQUALS the accumulated point: the caller then replaces the accumulator by the doubling of
// the base (xyzz_mdbl_lazy) -- left to the caller so that x2 / y2 need not outlive the two products that consume them
// (kept for a doubling that almost never comes they cost 28 registers, or 28 scratch stores per addition).
template <class P>
ARK_HD bool xyzz_madd_lazy(XYZZL<P>& acc, const FpL<P>& x2, const FpL<P>& y2) {
  typedef FpL<P> F;
  if (acc.inf) {
    const F one = F::one();
    acc.x = F::mul(x2, one);                                      // < 256 / 2048 + 1 < 1.13
    acc.y = F::mul(y2, one);                                      // < 1.13
    acc.zz = one;
    acc.zzz = one;
    acc.inf = false;
    return false;
  }
  const F u2 = F::mul(x2, acc.zz);                                // < 256 * 1.01 / 2048 + 1 < 1.13 (n)
  const F s2 = F::mul(y2, acc.zzz);                               // < 1.13 (n)
  const F pd = F::template sub_op<6>(u2, acc.x);                  // U2 - X1 + 6p: (0.99, 7.13), s (n on 29-bit limbs)
  const F rd = F::template sub_op<2>(s2, acc.y);                  // S2 - Y1 + 2p: (0.87, 3.13), s (n on 29-bit limbs)
  const F pp = F::sqr(pd);                                        // < 7.13^2 / 2048 + 1 < 1.03 (n)
  if (pp.is_zero_or_p()) {                                        // P = 0 mod p: same x -- doubling or infinity (bucket.rs:176-200)
    if (F::sqr(rd).is_zero_or_p()) return true;                   // R = 0 mod p as well: the same point
    acc.inf = true;
    return false;
  }
  const F ppp = F::mul(pd, pp);                                   // < 7.13 * 1.03 / 2048 + 1 < 1.01 (n)
  const F q = F::mul(acc.x, pp);                                  // < 5.01 * 1.03 / 2048 + 1 < 1.01 (n)
  const F x3 = F::template sub_b_2c_norm<4>(F::sqr(rd), ppp, q);  // R^2 - PPP - 2 Q + 4p: (0.97, 5.01), n
  const F t = F::template sub_semi<6>(q, x3);                     // Q - X3 + 6p: (0.99, 7.01), s
  // Y3 = R (Q - X3) - Y1 PPP as ONE sum of two products: R t + (2p - Y1) PPP < (3.13 * 7.01 + 2 * 1.01) / 2048 + 1 < 1.02
  // (column bound: 14 x (3 2^28)^2 + 14 x 2^29 2^28 + 14 x 2^56 < 2^63.4)
  acc.y = F::sop2(rd, t, F::template neg_semi<2>(acc.y), ppp);
  acc.zz = F::mul(acc.zz, pp);                                    // < 1.01
  acc.zzz = F::mul(acc.zzz, ppp);                                 // < 1.01
  acc.x = x3;
  return false;
}

// ---- full addition (the reduction and heavy-run kernels) ------------------------------------------------------------
// acc += b, add-2008-s (bucket.rs:256-337) in 28-bit limbs.  b's coordinates may be a stored bucket's canonical limbs
// repacked (below 256 p: lazy_operands_of) or another accumulator's (small): they enter through products with acc's
// small coordinates only.  Bounds for the large case:
//   U1 = X1 ZZ2 < 5.01 * 256 / 2048 + 1 < 1.63, U2 = X2 ZZ1 < 1.13, S1 = Y1 ZZZ2 < 1.15, S2 = Y2 ZZZ1 < 1.13  (n)
//   P = U2 - U1 + 2p in (0.37, 3.13), R = S2 - S1 + 2p in (0.85, 3.13)  (s);  PP < 1.01, PPP, Q < 1.01  (n)
//   X3 = R^2 - PPP - 2 Q + 4p in (0.99, 5.01) (n);  Y3 < (3.13 * 7.01 + 2 * 1.01) / 2048 + 1 < 1.02;  ZZ3, ZZZ3 < 1.01
// equal points (P = R = 0 mod p): xyzz_dbl_lazy, in place (rare).
template <class P>
struct XYZZOperands { FpL<P> x, y, zz, zzz; bool inf; };
template <class P>
ARK_HD XYZZOperands<P> lazy_operands_of(const XYZZ<Fp<P>>& b) {   // a stored bucket as multiplication operands
  XYZZOperands<P> r;
  r.inf = b.is_zero();
  r.x = FpL<P>::unpack32_shl(b.x.l);
  r.y = FpL<P>::unpack32_shl(b.y.l);
  r.zz = FpL<P>::unpack32_shl(b.zz.l);
  r.zzz = FpL<P>::unpack32_shl(b.zzz.l);
  return r;
}
// acc = 2 acc (dbl-2008-s-1, a = 0; bucket.rs:112-146) for an accumulator that is not at infinity, in 28-bit limbs: X3 and
// Y3 of the XYZZ doubling are functions of (X1, Y1) alone -- exactly the affine doubling's -- and ZZ3 = V ZZ1, ZZZ3 = W ZZZ1
// with the V, W it leaves in zz / zzz.  acc.x < 5.07, acc.y < 1.23 (n) are legal operands of xyzz_mdbl_lazy_xy (< 256 p), whose
// first two products bring them below 1.13 again.  (Rounds 3-5 went through the canonical form and the saturated formulas, out
// of line: the accumulator handed to that call lived in scratch memory.)
template <class P>
ARK_COLD_HD void xyzz_dbl_lazy(XYZZL<P>& acc) {
  typedef FpL<P> F;
  XYZZL<P> d;
  xyzz_mdbl_lazy_xy<P>(d, acc.x, acc.y);
  acc.zz = F::mul(d.zz, acc.zz);                                  // < 1.01
  acc.zzz = F::mul(d.zzz, acc.zzz);                               // < 1.01
  acc.x = d.x;
  acc.y = d.y;
}
template <class P>
ARK_HD void xyzz_add_lazy(XYZZL<P>& acc, const FpL<P>& bx, const FpL<P>& by, const FpL<P>& bzz, const FpL<P>& bzzz, bool binf) {
  typedef FpL<P> F;
  if (binf) return;
  if (acc.inf) {   // acc = b, brought below 1.13 p
    const F one = F::one();
    acc.x = F::mul(bx, one);
    acc.y = F::mul(by, one);
    acc.zz = F::mul(bzz, one);
    acc.zzz = F::mul(bzzz, one);
    acc.inf = false;
    return;
  }
  const F u1 = F::mul(acc.x, bzz);
  const F u2 = F::mul(bx, acc.zz);
  const F s1 = F::mul(acc.y, bzzz);
  const F s2 = F::mul(by, acc.zzz);
  const F pd = F::template sub_op<3>(u2, u1);                     // U1 < 1.97 on 29-bit limbs (stored bucket: 32 p)
  const F rd = F::template sub_op<2>(s2, s1);
  const F pp = F::sqr(pd);
  if (pp.is_zero_or_p()) {
    if (F::sqr(rd).is_zero_or_p()) {
      xyzz_dbl_lazy<P>(acc);
    } else {
      acc.inf = true;
    }
    return;
  }
  const F ppp = F::mul(pd, pp);
  const F q = F::mul(u1, pp);
  const F x3 = F::template sub_b_2c_norm<4>(F::sqr(rd), ppp, q);
  const F t = F::template sub_semi<6>(q, x3);
  acc.y = F::sop2(rd, t, F::template neg_semi<2>(s1), ppp);
  acc.zz = F::mul(F::mul(acc.zz, bzz), pp);
  acc.zzz = F::mul(F::mul(acc.zzz, bzzz), ppp);
  acc.x = x3;
}

}  // namespace arkhip
