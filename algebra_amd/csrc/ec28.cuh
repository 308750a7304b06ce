// The mixed addition of the bucket-accumulation kernel on carry-free 28-bit limbs (fp28.cuh), with the change of
// Montgomery radix absorbed by a curve isomorphism.
//
// Bases and buckets live in HBM in the reference's form: canonical residues with radix R = 2^(32 N) = 2^384.  fp28.cuh
// computes with radix R' = 2^(28 L) = 2^392.  Converting every gathered coordinate (a multiplication by R'/R = 2^8 mod p)
// would cost two products per mixed addition.  Instead the limbs are only REPACKED: the integer x R read with radix R'
// is the residue u^2 x, u = 2^-4 mod p.  For a curve y^2 = x^3 + b (a = 0, every curve served here: ec.cuh) the map
//       (x, y)  ->  (u^2 x, u^3 y)
// is an isomorphism onto y^2 = x^3 + u^6 b, and neither the addition nor the a = 0 doubling formulas mention b: the
// kernel simply adds points of that isomorphic curve.  The repacked x IS u^2 x; the repacked y is u^2 y and needs one
// cheap division by 16 mod p (FpL::shr_mod<4>, ~100 simple instructions) to become u^3 y.
// A bucket (X', Y', ZZ', ZZZ') of the isomorphic curve maps back to E as (X', Y', u^2 ZZ', u^3 ZZZ') -- still an XYZZ
// representative (ZZ^3 = ZZZ^2 is preserved) -- and storing a value V with radix R means storing the radix-R' bits of
// u^2 V: four divisions by powers of two mod p per BUCKET (not per addition), then a conditional subtraction to the
// canonical range.  Loading a stored bucket (a later piece of a streamed MSM, MsmPiece) is the same in reverse with the
// projective freedom (l^2 X, l^3 Y, l^2 ZZ, l^3 ZZZ), l = u, which again needs right shifts only.
//
// Formulas and branches: ec/src/models/short_weierstrass/bucket.rs:168-238 (madd-2008-s), affine.rs:169-201
// (mdbl-2008-s-1, a = 0).  Bounds in the comments are in units of p; "n" = normalised limbs.
#pragma once
#include "ec.cuh"
#include "fp28.cuh"

namespace arkhip {

template <class P>
struct XYZZL {
  FpL<P> x, y, zz, zzz;  // x < 5.1, y < 1.1, zz, zzz < 1.04 (all n)
  bool inf;
};

// a gathered base (canonical 32-bit limbs of the reference layout) as a point of the isomorphic curve
template <class P>
ARK_HD void lazy_from_affine(const Fp<P>& x, const Fp<P>& y, FpL<P>& xl, FpL<P>& yl) {
  xl = FpL<P>::unpack32(x.l);                              // u^2 x, < 1
  yl = FpL<P>::unpack32(y.l).template shr_mod<4>();        // u^3 y, < 1.07
}

// stored bucket (canonical XYZZ of E, radix R) -> accumulator on the isomorphic curve, representative scaled by l = u:
//   X'' = u^4 X, Y'' = u^6 Y, ZZ'' = u^2 ZZ, ZZZ'' = u^3 ZZZ;  the stored bits of V are the radix-R' bits of u^2 V
template <class P>
ARK_HD XYZZL<P> lazy_from_bucket(const XYZZ<Fp<P>>& b) {
  XYZZL<P> r;
  r.inf = b.is_zero();
  r.x = FpL<P>::unpack32(b.x.l).template shr_mod<8>();     // u^2 X -> u^4 X, < 1.01
  r.y = FpL<P>::unpack32(b.y.l).template shr_mod<16>();    // u^2 Y -> u^6 Y, < 1.01
  r.zz = FpL<P>::unpack32(b.zz.l);                         // u^2 ZZ
  r.zzz = FpL<P>::unpack32(b.zzz.l).template shr_mod<4>(); // u^2 ZZZ -> u^3 ZZZ, < 1.07
  return r;
}
// accumulator -> bucket of E in the reference's canonical form:
//   (X', Y', u^2 ZZ', u^3 ZZZ') then bits(V) = radix-R' residue u^2 V
template <class P>
ARK_HD XYZZ<Fp<P>> lazy_to_bucket(const XYZZL<P>& a) {
  if (a.inf) return XYZZ<Fp<P>>::zero();
  // shr_mod output < v / 2^K + p: below 2 p for every input bound here (x < 5.1 -> 5.1 / 256 + 1)
  XYZZ<Fp<P>> r;
  r.x = a.x.template shr_mod<8>().to_canonical_bits();
  r.y = a.y.template shr_mod<8>().to_canonical_bits();
  r.zz = a.zz.template shr_mod<16>().to_canonical_bits();
  r.zzz = a.zzz.template shr_mod<20>().to_canonical_bits();
  return r;
}

// affine doubling on the isomorphic curve (mdbl-2008-s-1, a = 0), all in 28-bit limbs.  Rare branch (equal points in
// one bucket): kept out of line so that its registers do not burden the hot loop.
template <class P>
__host__ __device__ __attribute__((noinline)) void xyzz_mdbl_lazy(XYZZL<P>& acc, const FpL<P>& x1, const FpL<P>& y1, bool neg) {
  typedef FpL<P> F;
  const F y = neg ? F::template neg<2>(y1) : y1;                  // < 2 (n)
  const F u = F::template sub<0>(F::add_lazy(y, y), F::zero());   // 2 y, < 4 (n)
  const F v = F::sqr(u);                                          // < 1.01
  const F w = F::mul(u, v);                                       // < 1.01
  const F s = F::mul(x1, v);                                      // < 1.01
  const F xx = F::sqr(x1);                                        // < 1.01
  const F m = F::template sub<0>(F::add_lazy(F::add_lazy(xx, xx), xx), F::zero());  // 3 xx, < 3.03 (n)
  acc.x = F::template sub_b_2c<4>(F::sqr(m), F::zero(), s);       // m^2 - 2 s + 4p: (1.9, 5.01)
  const F t = F::template sub<6>(s, acc.x);                       // (0.9, 7.02)
  acc.y = F::sop2(m, t, F::template neg<2>(w), y);                // m t - w y: < (3.03 * 7.02 + 2 * 2) / 2048 + 1 < 1.02
  acc.zz = v;
  acc.zzz = w;
  acc.inf = false;
}

// acc += (x2, +-y2), (x2, y2) a non-identity point of the isomorphic curve from lazy_from_affine (x2 < 1, y2 < 1.07, n)
template <class P>
ARK_HD void xyzz_madd_lazy(XYZZL<P>& acc, const FpL<P>& x2, const FpL<P>& y2, bool neg) {
  typedef FpL<P> F;
  if (acc.inf) {
    acc.x = x2;
    acc.y = neg ? F::template neg<2>(y2) : y2;                    // < 2
    acc.zz = F::one();
    acc.zzz = F::one();
    acc.inf = false;
    return;
  }
  const F u2 = F::mul(x2, acc.zz);                                // < 1.01
  const F s2 = F::mul(y2, acc.zzz);                               // < 1.01
  const F pd = F::template sub<6>(u2, acc.x);                     // U2 - X1 + 6p: (0.9, 7.01), n
  const F rd = neg ? F::template negsub<4>(s2, acc.y)             // -S2 - Y1 + 4p: (0.9, 4]
                   : F::template sub<2>(s2, acc.y);               //  S2 - Y1 + 2p: (0, 3.01)
  if (pd.is_zero_mod_p()) {                                       // same x: doubling or infinity (bucket.rs:176-200)
    if (rd.is_zero_mod_p()) xyzz_mdbl_lazy<P>(acc, x2, y2, neg);
    else acc.inf = true;
    return;
  }
  const F pp = F::sqr(pd);                                        // < 7.01^2 / 2048 + 1 < 1.03
  const F ppp = F::mul(pd, pp);                                   // < 1.01
  const F q = F::mul(acc.x, pp);                                  // < 5.1 * 1.03 / 2048 + 1 < 1.01
  const F x3 = F::template sub_b_2c<4>(F::sqr(rd), ppp, q);       // R^2 - PPP - 2 Q + 4p: (0.97, 5.01), n
  const F t = F::template sub<6>(q, x3);                          // Q - X3 + 6p: (0.99, 7.01)
  // Y3 = R (Q - X3) - Y1 PPP as ONE sum of two products: R t + (2p - Y1) PPP < (4 * 7.01 + 2 * 1.01) / 2048 + 1 < 1.02
  acc.y = F::sop2(rd, t, F::template neg<2>(acc.y), ppp);
  acc.zz = F::mul(acc.zz, pp);                                    // < 1.01
  acc.zzz = F::mul(acc.zzz, ppp);                                 // < 1.01
  acc.x = x3;
}

}  // namespace arkhip
