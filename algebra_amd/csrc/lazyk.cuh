// What the carry-free accumulate / reduction kernels need from a curve, in one place: LazyK<C> binds the memory-side field
// (canonical limbs: Fp, or the lane-pair Fp2Half), the accumulator in carry-free limbs and the additions on it --
// ec28.cuh for the prime-field curves (one lane per bucket), ec28x2.cuh for G2 (one lane PAIR per bucket).
#pragma once
#include "curves.cuh"
#include "ec28.cuh"
#include "ec28x2.cuh"

namespace arkhip {

template <class C, int LANES_ = C::FA::LANES>
struct LazyK;

template <class C>
struct LazyK<C, 1> {
  typedef typename C::F FM;            // the field as it lives in memory
  typedef typename FM::P P;
  typedef FpL<P> FL;
  typedef XYZZL<P> Acc;
  static constexpr u32 LANES = 1;
  static constexpr int WORDS = 4 * FL::L + 1;   // one parked accumulator: four coordinates + the infinity flag
  ARK_HD static Acc inf() {
    Acc a;
    a.inf = true;
    a.x = a.y = a.zz = a.zzz = FL::zero();
    return a;
  }
  ARK_HD static Acc from_bucket(const XYZZ<FM>& b) { return lazy_from_bucket<P>(b); }
  ARK_HD static XYZZ<FM> to_bucket(const Acc& a) { return lazy_to_bucket<P>(a); }
  // acc += (+-) p, a non-identity base as gathered; true: p EQUALS the accumulated point (the caller doubles the base)
  ARK_HD static bool madd(Acc& acc, const Affine<FM>& p, bool neg) {
    FL lx, ly;
    lazy_from_affine<P>(p.x, FM::cond_neg(p.y, neg), lx, ly);
    return xyzz_madd_lazy<P>(acc, lx, ly);
  }
  ARK_HD static void mdbl(Acc& out, const char* src, bool neg) { xyzz_mdbl_lazy<P>(out, src, neg); }
  ARK_HD static void add(Acc& acc, const XYZZ<FM>& b) {
    const XYZZOperands<P> o = lazy_operands_of<P>(b);
    xyzz_add_lazy<P>(acc, o.x, o.y, o.zz, o.zzz, o.inf);
  }
  ARK_HD static void add_acc(Acc& acc, const Acc& b) { xyzz_add_lazy<P>(acc, b.x, b.y, b.zz, b.zzz, b.inf); }
  ARK_DEV static void park(const Acc& a, char* slot) {
    u32* w = (u32*)slot;
#pragma unroll
    for (int i = 0; i < FL::L; i++) {
      w[i] = a.x.l[i];
      w[FL::L + i] = a.y.l[i];
      w[2 * FL::L + i] = a.zz.l[i];
      w[3 * FL::L + i] = a.zzz.l[i];
    }
    w[4 * FL::L] = a.inf ? 1u : 0u;
  }
  ARK_DEV static Acc unpark(const char* slot) {
    const u32* w = (const u32*)slot;
    Acc a;
#pragma unroll
    for (int i = 0; i < FL::L; i++) {
      a.x.l[i] = w[i];
      a.y.l[i] = w[FL::L + i];
      a.zz.l[i] = w[2 * FL::L + i];
      a.zzz.l[i] = w[3 * FL::L + i];
    }
    a.inf = w[4 * FL::L] != 0u;
    return a;
  }
};

// G2: a lane pair owns the bucket; every lane holds ONE component of each coordinate
template <class C>
struct LazyK<C, 2> {
  typedef typename C::FA FM;           // Fp2Half
  typedef typename FM::P P;
  typedef Fp2L<P, FM::NEG_BETA_> FL;
  typedef XYZZL2<FL> Acc;
  static constexpr u32 LANES = 2;
  static constexpr int L = FL::L;
  static constexpr int WORDS = 8 * L + 1;       // per PAIR: four coordinates x two components + the infinity flag
  ARK_DEV static Acc inf() { return lazy2_inf<FL>(); }
  ARK_DEV static Acc from_bucket(const XYZZ<FM>& b) { return lazy2_from_bucket<FL>(b); }
  ARK_DEV static XYZZ<FM> to_bucket(const Acc& a) { return lazy2_to_bucket<FL>(a); }
  ARK_DEV static bool madd(Acc& acc, const Affine<FM>& p, bool neg) {
    return xyzz_madd_lazy2<FL>(acc, FL::from_canonical(p.x), FL::from_canonical(FM::cond_neg(p.y, neg)));
  }
  ARK_DEV static void mdbl(Acc& out, const char* src, bool neg) { lazy2_mdbl<FL>(out, src, neg); }
  ARK_DEV static void add(Acc& acc, const XYZZ<FM>& b) {
    const XYZZOperands2<FL> o = lazy2_operands_of<FL>(b);
    xyzz_add_lazy2<FL>(acc, o.x, o.y, o.zz, o.zzz, o.inf);
  }
  ARK_DEV static void add_acc(Acc& acc, const Acc& b) { xyzz_add_lazy2<FL>(acc, b.x, b.y, b.zz, b.zzz, b.inf); }
  // the pair's slot: coordinate k, component c at words [(2 k + c) L, (2 k + c + 1) L); both lanes write the flag
  ARK_DEV static void park(const Acc& a, char* slot) {
    u32* w = (u32*)slot + (FL::odd() ? L : 0);
#pragma unroll
    for (int i = 0; i < L; i++) {
      w[i] = a.x.v.l[i];
      w[2 * L + i] = a.y.v.l[i];
      w[4 * L + i] = a.zz.v.l[i];
      w[6 * L + i] = a.zzz.v.l[i];
    }
    ((u32*)slot)[8 * L] = a.inf ? 1u : 0u;
  }
  ARK_DEV static Acc unpark(const char* slot) {
    const u32* w = (const u32*)slot + (FL::odd() ? L : 0);
    Acc a;
#pragma unroll
    for (int i = 0; i < L; i++) {
      a.x.v.l[i] = w[i];
      a.y.v.l[i] = w[2 * L + i];
      a.zz.v.l[i] = w[4 * L + i];
      a.zzz.v.l[i] = w[6 * L + i];
    }
    a.inf = ((const u32*)slot)[8 * L] != 0u;
    return a;
  }
};

}  // namespace arkhip
