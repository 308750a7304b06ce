// Bucket additions over Fp2 (G2) on carry-free 28-bit limbs, one element per lane pair (fp28x2.cuh).
//
// Formulas and branches: ec/src/models/short_weierstrass/bucket.rs:168-238 (madd-2008-s), :256-337 (add-2008-s); equal
// points are doubled in the same limbs (lazy2_mdbl / lazy2_dbl below; rare).  Bases and buckets
// live in HBM in the reference's canonical radix-2^384 form; a gathered coordinate enters by the shifted repack (the
// residue itself in radix 2^392, below 256 p: Fp2L::from_canonical) as the B operand of a product whose A operand is one of
// the accumulator's small coordinates; buckets leave through a division by 2^8 mod p (Fp2L::to_canonical).
//
// Bounds in the comments: units of p, worst case over the two curves served (BLS12-381: R' / p >= 2520, NB = 1;
// BLS12-377: R' / p >= 38996, NB = 5); "n" = normalised limbs, "s" = semi-normalised (< 3 2^28).  Every control-flow
// condition is pair-uniform.
#pragma once
#include "ec.cuh"
#include "fp28x2.cuh"

namespace arkhip {

template <class FL2>
struct XYZZL2 {
  FL2 x, y, zz, zzz;  // x < 7.2 (n), y < 1.11 (n), zz, zzz < 1.02 (n)
  bool inf;
};

template <class FL2>
ARK_DEV XYZZL2<FL2> lazy2_inf() {
  XYZZL2<FL2> r;
  r.inf = true;
  r.x = r.y = r.zz = r.zzz = FL2::zero();
  return r;
}
// stored bucket (canonical XYZZ over Fp2Half) -> accumulator: repack, then one product with the residue 1 per component
template <class FL2>
ARK_DEV XYZZL2<FL2> lazy2_from_bucket(const XYZZ<typename FL2::M>& b) {
  XYZZL2<FL2> r;
  r.inf = b.is_zero();   // pair-wide
  r.x = FL2::reduce_small(FL2::from_canonical(b.x));       // < 256 / 2520 + 1 < 1.11
  r.y = FL2::reduce_small(FL2::from_canonical(b.y));
  r.zz = FL2::reduce_small(FL2::from_canonical(b.zz));
  r.zzz = FL2::reduce_small(FL2::from_canonical(b.zzz));
  return r;
}
template <class FL2>
ARK_DEV XYZZ<typename FL2::M> lazy2_to_bucket(const XYZZL2<FL2>& a) {
  typedef typename FL2::M M;
  if (a.inf) return XYZZ<M>::zero();
  return XYZZ<M>{a.x.to_canonical(), a.y.to_canonical(), a.zz.to_canonical(), a.zzz.to_canonical()};  // x: 7.2 / 256 + 1 < 2
}

// ---- doubling (equal points in one bucket: duplicate bases only) -- mdbl-2008-s-1 / dbl-2008-s-1 with a = 0
// (affine.rs:169-201, bucket.rs:112-146) on the lane pair's carry-free limbs, expanded in place.  (Rounds 4-5 went through the
// canonical form and the saturated lane-pair formulas out of line; the accumulator handed to that call lived in scratch.)
// acc = 2 (x1, y1) for an affine point with both coordinates normalised and below 1.5 p:
//   U = 2 Y1 < 2.3 (n);  V = U^2: c0 < 3.1 (s), c1 < 1.01;  W = U V < (2.3 * 3.1 + NB 4 * 1.01) / R' + 1 < 1.02;  S = X1 V < 1.01;
//   XX = X1^2 brought below 1.01 by a product with 1;  M = 3 XX < 3.03 (n);  M^2: c0 < 3.1 (s), c1 < 1.01;
//   X3 = M^2 - 2 S + 4p in (1.9, 7.1) (n);  t = S - X3 + 8p in (0.9, 7.12) (n);  Y3 = M t - Y1 W < 1.04 (as the mixed addition's)
template <class FL2>
ARK_COLD_DEV void lazy2_mdbl_small(XYZZL2<FL2>& acc, const FL2& x1, const FL2& y1) {
  typedef FL2 F;
  typedef typename F::B B;
  const F u = F{B::template sub<0>(B::add_lazy(y1.v, y1.v), B::zero())};
  const F v = F::template sqr<4>(u);
  const F w = F::template mul<4>(u, v);
  const F s = F::template mul<2>(x1, v);
  const F xx = F::reduce_small(F::template sqr<2>(x1));
  const F m = F{B::template sub<0>(B::add_lazy(B::add_lazy(xx.v, xx.v), xx.v), B::zero())};
  const F x3 = F::template sub_b_2c_norm<4>(F::template sqr<4>(m), F::zero(), s);
  const F t = F::template sub_sweep<8>(s, x3);
  acc.y = F::template mul_sub<4, 2>(m, t, y1, w);
  acc.x = x3;
  acc.zz = F::reduce_small(v);                                    // < 1.01, normalised (V's c0 is semi-normalised)
  acc.zzz = w;
  acc.inf = false;
}
// acc = 2 (base at `src`, canonical layout; neg: the digit's sign): the accumulate kernels' form -- it re-reads the base, so
// that nothing of their hot loop has to stay alive for a doubling that almost never comes
template <class FL2>
ARK_COLD_DEV void lazy2_mdbl(XYZZL2<FL2>& acc, const char* src, bool neg) {
  typedef typename FL2::M M;
  const Affine<M> b = Affine<M>::load(src);
  lazy2_mdbl_small<FL2>(acc, FL2::reduce_small(FL2::from_canonical(b.x)),
                        FL2::reduce_small(FL2::from_canonical(M::cond_neg(b.y, neg))));
}
// acc = 2 acc for an accumulator that is not at infinity: X3, Y3 are the affine doubling's of (X1, Y1); ZZ3 = V ZZ1, ZZZ3 = W ZZZ1
template <class FL2>
ARK_COLD_DEV void lazy2_dbl(XYZZL2<FL2>& acc) {
  typedef FL2 F;
  XYZZL2<FL2> d;
  lazy2_mdbl_small<FL2>(d, F::reduce_small(acc.x), acc.y);        // x < 7.2 -> 1.01; y < 1.11
  acc.zz = F::template mul<2>(d.zz, acc.zz);                      // < 1.01
  acc.zzz = F::template mul<2>(d.zzz, acc.zzz);                   // < 1.01
  acc.x = d.x;
  acc.y = d.y;
}

// acc += (x2, y2): a non-identity base (Fp2L::from_canonical: n, < 256; the digit's sign already in y2).  Returns true when
// the base EQUALS the accumulated point: the caller then replaces the accumulator by the doubling of the base.
template <class FL2>
ARK_DEV bool xyzz_madd_lazy2(XYZZL2<FL2>& acc, const FL2& x2, const FL2& y2) {
  typedef FL2 F;
  if (acc.inf) {
    acc.x = F::reduce_small(x2);                                  // < 1.11
    acc.y = F::reduce_small(y2);
    acc.zz = F::one();
    acc.zzz = F::one();
    acc.inf = false;
    return false;
  }
  const F u2 = F::template mul<2>(acc.zz, x2);                    // (1.02 * 256 + NB 2 * 256) / R' + 1 < 1.31 (n)
  const F s2 = F::template mul<2>(acc.zzz, y2);                   // < 1.31 (n)
  const F pd = F::template sub_sweep<8>(u2, acc.x);               // U2 - X1 + 8p in (0.8, 9.31), n
  const F rd = F::template sub_sweep<2>(s2, acc.y);               // S2 - Y1 + 2p in (0.89, 3.31), n
  bool pz;
  const F pp = F::template sqr<10>(pd, &pz);                      // s < 18.7 * (9.31 + NB 10) / R' + 1 < 1.15, c1 < 1.07; c0 < 3.1 (s)
  if (pz) {                                                       // P = 0: same x -- doubling or infinity (bucket.rs:176-200)
    bool rz;
    (void)F::template sqr<4>(rd, &rz);
    if (rz) return true;
    acc.inf = true;
    return false;
  }
  const F ppp = F::template mul<10>(pd, pp);                      // (9.31 * 3.1 + NB 10 * 1.07) / R' + 1 < 1.02 (n)
  const F q = F::template mul<8>(acc.x, pp);                      // (7.2 * 3.1 + NB 8 * 1.07) / R' + 1 < 1.02 (n)
  const F rr = F::template sqr<4>(rd);                            // c0 < 3.1 (s), c1 < 1.01
  const F x3 = F::template sub_b_2c_norm<4>(rr, ppp, q);          // R^2 - PPP - 2 Q + 4p in (0.9, 7.1), n
  const F t = F::template sub_sweep<8>(q, x3);                    // Q - X3 + 8p in (0.9, 9.02), n
  acc.y = F::template mul_sub<4, 2>(rd, t, acc.y, ppp);           // (3.31 * 9.02 + NB 4 * 9.02 + 2 * 1.02 + NB 1.11 * 1.02) / R' + 1 < 1.04
  acc.zz = F::template mul<2>(acc.zz, pp);                        // < 1.01
  acc.zzz = F::template mul<2>(acc.zzz, ppp);                     // < 1.01
  acc.x = x3;
  return false;
}

// ---- full addition (the reduction and heavy-run kernels) ----
// acc += b; b's coordinates are a stored bucket's canonical limbs repacked (below 256 p) or another accumulator's.
//   U1 = X1 ZZ2 < (7.2 * 256 + NB 8 * 256) / R' + 1 < 2.55;  U2, S2 < 1.31;  S1 = Y1 ZZZ2 < (1.11 * 256 + NB 2 * 256) / R' + 1 < 1.32
//   P = U2 - U1 + 4p in (1.4, 5.31), R = S2 - S1 + 2p in (0.6, 3.31)  (n)
template <class FL2>
struct XYZZOperands2 { FL2 x, y, zz, zzz; bool inf; };
template <class FL2>
ARK_DEV XYZZOperands2<FL2> lazy2_operands_of(const XYZZ<typename FL2::M>& b) {
  XYZZOperands2<FL2> r;
  r.inf = b.is_zero();
  r.x = FL2::from_canonical(b.x);
  r.y = FL2::from_canonical(b.y);
  r.zz = FL2::from_canonical(b.zz);
  r.zzz = FL2::from_canonical(b.zzz);
  return r;
}
template <class FL2>
ARK_DEV void xyzz_add_lazy2(XYZZL2<FL2>& acc, const FL2& bx, const FL2& by, const FL2& bzz, const FL2& bzzz, bool binf) {
  typedef FL2 F;
  if (binf) return;
  if (acc.inf) {   // acc = b, brought below 1.11 p
    acc.x = F::reduce_small(bx);
    acc.y = F::reduce_small(by);
    acc.zz = F::reduce_small(bzz);
    acc.zzz = F::reduce_small(bzzz);
    acc.inf = false;
    return;
  }
  const F u1 = F::template mul<8>(acc.x, bzz);
  const F u2 = F::template mul<2>(acc.zz, bx);
  const F s1 = F::template mul<2>(acc.y, bzzz);
  const F s2 = F::template mul<2>(acc.zzz, by);
  const F pd = F::template sub_sweep<4>(u2, u1);
  const F rd = F::template sub_sweep<2>(s2, s1);
  bool pz;
  const F pp = F::template sqr<6>(pd, &pz);
  if (pz) {
    bool rz;
    (void)F::template sqr<4>(rd, &rz);
    if (rz) {
      lazy2_dbl<FL2>(acc);
    } else {
      acc.inf = true;
    }
    return;
  }
  const F ppp = F::template mul<6>(pd, pp);
  const F q = F::template mul<4>(u1, pp);
  const F rr = F::template sqr<4>(rd);
  const F x3 = F::template sub_b_2c_norm<4>(rr, ppp, q);
  const F t = F::template sub_sweep<8>(q, x3);
  acc.y = F::template mul_sub<4, 2>(rd, t, s1, ppp);
  acc.zz = F::template mul<2>(F::template mul<2>(acc.zz, bzz), pp);      // inner: (1.02 * 256 + NB 2 * 256) / R' + 1 < 1.31
  acc.zzz = F::template mul<2>(F::template mul<2>(acc.zzz, bzzz), ppp);
  acc.x = x3;
}

}  // namespace arkhip
