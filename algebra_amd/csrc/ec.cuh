// Short-Weierstrass point arithmetic (a = 0 curves) on the device.
//
// Replaces the reference's bucket / projective formulas on the MSM hot path:
//   ec/src/models/short_weierstrass/bucket.rs:21-30 (Bucket = extended Jacobian X,Y,ZZ,ZZZ),
//   :78-83 (ZERO = (1,1,0,0)), :112-146 (double_in_place), :168-244 (+= / -= Affine, madd-2008-s),
//   :256-343 (+= / -= Bucket, add-2008-s), :389-397 (Bucket -> Projective: (X*ZZ, Y*ZZZ, ZZ)),
//   ec/src/models/short_weierstrass/affine.rs:91-104 (identity = (0,0) when ZeroFlag = ()),
//   :169-201 (double_to_bucket, mdbl-2008-s-1),
//   ec/src/models/short_weierstrass/group.rs:171-221 (Jacobian doubling, a = 0), :450-538 (Jacobian add).
// All five curves served here have COEFF_A = 0 (bn254 g1.rs:21, bls12_381 g1.rs:43, g2.rs:58,
// bls12_377 g1.rs:42, g2.rs:47), so the a*ZZ^2 term of the doubling formulas is dropped.
// Only the group element matters for parity (the reference compares after into_affine()), so
// the formulas need to be correct, not operation-for-operation identical.
#pragma once
#include "fp.cuh"

namespace arkhip {

template <class F>
struct Affine {
  F x, y;
  static constexpr int BYTES = 2 * F::FULL_BYTES;   // in memory (an Fp2Half lane holds half of each coordinate)
  ARK_HD bool is_zero() const {  // affine.rs:91-104
    if constexpr (F::LANES == 2) {
      const bool zx = x.is_zero();  // pair-wide tests: evaluated by both lanes, never short-circuited
      const bool zy = y.is_zero();
      return zx && zy;
    } else {
      return x.is_zero() && y.is_zero();
    }
  }
  ARK_HD static Affine load(const void* p) {
    Affine a;
    a.x = F::load(p);
    a.y = F::load((const char*)p + F::FULL_BYTES);
    return a;
  }
};

template <class F>
struct XYZZ {
  F x, y, zz, zzz;
  static constexpr int BYTES = 4 * F::FULL_BYTES;
  ARK_HD static XYZZ zero() { return XYZZ{F::one(), F::one(), F::zero(), F::zero()}; }  // bucket.rs:78-83
  ARK_HD bool is_zero() const { return zz.is_zero(); }
  ARK_HD static XYZZ load(const void* p) {
    XYZZ r;
    const char* q = (const char*)p;
    r.x = F::load(q);
    r.y = F::load(q + F::FULL_BYTES);
    r.zz = F::load(q + 2 * F::FULL_BYTES);
    r.zzz = F::load(q + 3 * F::FULL_BYTES);
    return r;
  }
  ARK_HD void store(void* p) const {
    char* q = (char*)p;
    x.store(q);
    y.store(q + F::FULL_BYTES);
    zz.store(q + 2 * F::FULL_BYTES);
    zzz.store(q + 3 * F::FULL_BYTES);
  }
  ARK_HD static XYZZ from_affine(const Affine<F>& a) {
    if (a.is_zero()) return zero();
    return XYZZ{a.x, a.y, F::one(), F::one()};
  }
  ARK_HD static XYZZ neg(const XYZZ& a) { return XYZZ{a.x, F::neg(a.y), a.zz, a.zzz}; }  // bucket.rs:158-166
};

template <class F>
struct Jac {
  F x, y, z;
  static constexpr int BYTES = 3 * F::FULL_BYTES;
  ARK_HD static Jac zero() { return Jac{F::one(), F::one(), F::zero()}; }  // group.rs:145-151
  ARK_HD bool is_zero() const { return z.is_zero(); }
  ARK_HD void store(void* p) const {
    char* q = (char*)p;
    x.store(q);
    y.store(q + F::FULL_BYTES);
    z.store(q + 2 * F::FULL_BYTES);
  }
};

// ---- Montgomery's trick inside one lane (ff/src/fields/mod.rs:358-385 batch_inversion) -----------------------------
// A lane owns the B values z_j = get_z(t + j * stride), j < B (index < n): one pass multiplies them up (zeros skipped,
// as the reference does), ONE Fermat inversion (~1.5 N log2(p) products) inverts the product, a second pass in reverse
// order peels off the individual inverses: 3 products per value + 1/B of an inversion instead of one inversion each.
// Consecutive lanes own consecutive indices (stride = number of lanes), so every pass is a coalesced sweep.
// emit(index, z_inverse, nonzero) is called once per owned index, in descending j.
template <class F, int B, class GetZ, class Emit>
ARK_DEV void lane_batch_inverse(size_t t, size_t stride, size_t n, GetZ get_z, Emit emit) {
  F pre[B];
  u32 nzmask = 0;
  F run = F::one();
#pragma unroll
  for (int j = 0; j < B; j++) {
    const size_t i = t + (size_t)j * stride;
    if (i < n) {
      const F z = get_z(i);
      if (!z.is_zero()) {
        run = F::mul(run, z);
        nzmask |= 1u << j;
      }
    }
    pre[j] = run;
  }
  F inv = F::inverse(run);
#pragma unroll
  for (int j = B - 1; j >= 0; j--) {
    const size_t i = t + (size_t)j * stride;
    if (i < n) {
      if ((nzmask >> j) & 1u) {
        const F z = get_z(i);
        const F zi = j > 0 ? F::mul(inv, pre[j > 0 ? j - 1 : 0]) : inv;
        inv = F::mul(inv, z);
        emit(i, zi, true);
      } else {
        emit(i, F::zero(), false);
      }
    }
  }
}
// values per lane: bounded by the registers the B running products take (B * limbs)
template <class F> struct LaneBatch { static constexpr int B = F::BYTES > 64 ? 4 : 8; };

// XYZZ (x, y, zz, zzz) -> affine (x / zz, y / zzz), identity -> (0, 0), with the lane-batched inversion above
// (ZZ^-1 = (ZZ * ZZZ^-1)^2 since ZZ^3 = ZZZ^2).  in: n * XYZZ::BYTES, out: n * Affine::BYTES.
template <class F>
__global__ void __launch_bounds__(128) xyzz_to_affine_batched_kernel(const char* __restrict__ in, char* __restrict__ out,
                                                                     size_t n, size_t lanes) {
  constexpr int B = LaneBatch<F>::B;
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= lanes) return;
  lane_batch_inverse<F, B>(
      t, lanes, n, [&](size_t i) { return F::load(in + i * XYZZ<F>::BYTES + 3 * F::BYTES); },
      [&](size_t i, const F& zzzi, bool nonzero) {
        F x = F::zero(), y = F::zero();
        if (nonzero) {
          const char* p = in + i * XYZZ<F>::BYTES;
          const F zzi = F::sqr(F::mul(F::load(p + 2 * F::BYTES), zzzi));
          x = F::mul(F::load(p), zzi);
          y = F::mul(F::load(p + F::BYTES), zzzi);
        }
        x.store(out + i * Affine<F>::BYTES);
        y.store(out + i * Affine<F>::BYTES + F::BYTES);
      });
}
template <class F>
static inline void xyzz_to_affine_batched_launch(const void* d_xyzz, void* d_affine, size_t n, hipStream_t stream) {
  constexpr int B = LaneBatch<F>::B;
  const size_t lanes = (n + B - 1) / B;
  hipLaunchKernelGGL((xyzz_to_affine_batched_kernel<F>), dim3((unsigned)((lanes + 127) / 128)), dim3(128), 0, stream,
                     (const char*)d_xyzz, (char*)d_affine, n, lanes);
}

// affine doubling into XYZZ (mdbl-2008-s-1, a = 0)            affine.rs:169-201
template <class F>
ARK_HD XYZZ<F> xyzz_mdbl(const F& x1, const F& y1) {
  F u = F::dbl(y1);
  F v = F::sqr(u);
  F w = F::mul(u, v);
  F s = F::mul(x1, v);
  F xx = F::sqr(x1);
  F m = F::add(F::dbl(xx), xx);
  XYZZ<F> r;
  r.x = F::sub(F::sqr(m), F::dbl(s));
  r.y = F::sub(F::mul(m, F::sub(s, r.x)), F::mul(w, y1));
  r.zz = v;
  r.zzz = w;
  return r;
}

// XYZZ doubling (dbl-2008-s-1, a = 0)                         bucket.rs:112-146
template <class F>
ARK_HD XYZZ<F> xyzz_dbl(const XYZZ<F>& p) {
  if (p.is_zero()) return p;
  F u = F::dbl(p.y);
  F v = F::sqr(u);
  F w = F::mul(u, v);
  F s = F::mul(p.x, v);
  F xx = F::sqr(p.x);
  F m = F::add(F::dbl(xx), xx);
  XYZZ<F> r;
  r.x = F::sub(F::sqr(m), F::dbl(s));
  r.y = F::sub(F::mul(m, F::sub(s, r.x)), F::mul(w, p.y));
  r.zz = F::mul(v, p.zz);
  r.zzz = F::mul(w, p.zzz);
  return r;
}

// acc += (x2, y2)  with (x2,y2) a non-identity affine point   bucket.rs:168-238 (madd-2008-s)
// Branches: acc = inf -> copy; same x: same y -> mdbl, opposite y -> inf.
template <class F>
ARK_HD void xyzz_madd(XYZZ<F>& acc, const F& x2, const F& y2) {
  if (acc.is_zero()) {
    acc.x = x2; acc.y = y2; acc.zz = F::one(); acc.zzz = F::one();
    return;
  }
  F p = F::sub(F::mul(x2, acc.zz), acc.x);
  F r = F::sub(F::mul(y2, acc.zzz), acc.y);
  if (p.is_zero()) {
    if (r.is_zero()) acc = xyzz_mdbl<F>(x2, y2);
    else acc = XYZZ<F>::zero();
    return;
  }
  F pp = F::sqr(p);
  F ppp = F::mul(p, pp);
  F q = F::mul(acc.x, pp);
  F x3 = F::sub(F::sub(F::sqr(r), ppp), F::dbl(q));
  F y3 = F::sub(F::mul(r, F::sub(q, x3)), F::mul(acc.y, ppp));
  acc.x = x3;
  acc.y = y3;
  acc.zz = F::mul(acc.zz, pp);
  acc.zzz = F::mul(acc.zzz, ppp);
}

// The same mixed addition on relaxed residues (Fp::mul_r & co., values in [0, 2p)): the bucket-accumulation loop
// of the MSM.  x2 canonical, y2 canonical or the relaxed negation 2p - y (F::neg_r); acc's coordinates relaxed, except
// that infinity is the exact (1, 1, 0, 0).
template <class F>
ARK_HD void xyzz_madd_relaxed(XYZZ<F>& acc, const F& x2, const F& y2) {
  if (acc.is_zero()) {
    acc.x = x2; acc.y = y2; acc.zz = F::one(); acc.zzz = F::one();
    return;
  }
  F p = F::sub_r(F::mul_r(x2, acc.zz), acc.x);
  F r = F::sub_r(F::mul_r(y2, acc.zzz), acc.y);
  const bool pz = p.is_zero_mod_p();  // evaluated unconditionally: over a lane pair (Fp2Half) a test is an exchange
  const bool rz = r.is_zero_mod_p();
  if (pz) {
    if (rz) acc = xyzz_mdbl<F>(x2, y2.canonical());   // canonical arithmetic (y2 may be the relaxed 2p - y); rare
    else acc = XYZZ<F>::zero();
    return;
  }
  F pp = F::sqr_r(p);
  F ppp = F::mul_r(p, pp);
  F q = F::mul_r(acc.x, pp);
  F x3 = F::sub_r(F::sub_r(F::sqr_r(r), ppp), F::dbl_r(q));
  F y3;
  if constexpr (F::FUSED_Y3) y3 = F::sop2_r(r, F::sub_r(q, x3), F::neg_r(acc.y), ppp);  // R (Q - X3) - Y1 PPP, one reduction
  else y3 = F::sub_r(F::mul_r(r, F::sub_r(q, x3)), F::mul_r(acc.y, ppp));
  acc.x = x3;
  acc.y = y3;
  acc.zz = F::mul_r(acc.zz, pp);
  acc.zzz = F::mul_r(acc.zzz, ppp);
}
template <class F>
ARK_HD XYZZ<F> xyzz_canonical(const XYZZ<F>& a) {
  if (a.is_zero()) return XYZZ<F>::zero();
  return XYZZ<F>{a.x.canonical(), a.y.canonical(), a.zz.canonical(), a.zzz.canonical()};
}

// acc += b (both XYZZ)                                        bucket.rs:256-337 (add-2008-s)
template <class F>
ARK_HD void xyzz_add(XYZZ<F>& acc, const XYZZ<F>& b) {
  if (b.is_zero()) return;
  if (acc.is_zero()) { acc = b; return; }
  F u1 = F::mul(acc.x, b.zz);
  F u2 = F::mul(b.x, acc.zz);
  F s1 = F::mul(acc.y, b.zzz);
  F s2 = F::mul(b.y, acc.zzz);
  F p = F::sub(u2, u1);
  F r = F::sub(s2, s1);
  if (p.is_zero()) {
    if (r.is_zero()) acc = xyzz_dbl<F>(acc);
    else acc = XYZZ<F>::zero();
    return;
  }
  F pp = F::sqr(p);
  F ppp = F::mul(p, pp);
  F q = F::mul(u1, pp);
  F x3 = F::sub(F::sub(F::sqr(r), ppp), F::dbl(q));
  F y3 = F::sub(F::mul(r, F::sub(q, x3)), F::mul(s1, ppp));
  acc.x = x3;
  acc.y = y3;
  acc.zz = F::mul(F::mul(acc.zz, b.zz), pp);
  acc.zzz = F::mul(F::mul(acc.zzz, b.zzz), ppp);
}

// The same full addition on relaxed residues (values in [0, 2p), infinity the exact (1, 1, 0, 0)): the bucket
// reduction.  Both operands may be relaxed; the rare doubling branch goes through canonical arithmetic.
template <class F>
ARK_HD void xyzz_add_relaxed(XYZZ<F>& acc, const XYZZ<F>& b) {
  const bool bz = b.is_zero();   // evaluated unconditionally (pair-wide exchanges over Fp2Half)
  const bool az = acc.is_zero();
  if (bz) return;
  if (az) { acc = b; return; }
  F u1 = F::mul_r(acc.x, b.zz);
  F u2 = F::mul_r(b.x, acc.zz);
  F s1 = F::mul_r(acc.y, b.zzz);
  F s2 = F::mul_r(b.y, acc.zzz);
  F p = F::sub_r(u2, u1);
  F r = F::sub_r(s2, s1);
  const bool pz = p.is_zero_mod_p();
  const bool rz = r.is_zero_mod_p();
  if (pz) {
    if (rz) acc = xyzz_dbl<F>(xyzz_canonical<F>(acc));
    else acc = XYZZ<F>::zero();
    return;
  }
  F pp = F::sqr_r(p);
  F ppp = F::mul_r(p, pp);
  F q = F::mul_r(u1, pp);
  F x3 = F::sub_r(F::sub_r(F::sqr_r(r), ppp), F::dbl_r(q));
  F y3;
  if constexpr (F::FUSED_Y3) y3 = F::sop2_r(r, F::sub_r(q, x3), F::neg_r(s1), ppp);
  else y3 = F::sub_r(F::mul_r(r, F::sub_r(q, x3)), F::mul_r(s1, ppp));
  acc.x = x3;
  acc.y = y3;
  acc.zz = F::mul_r(F::mul_r(acc.zz, b.zz), pp);
  acc.zzz = F::mul_r(F::mul_r(acc.zzz, b.zzz), ppp);
}

// XYZZ -> Jacobian (X*ZZ, Y*ZZZ, ZZ): valid because ZZ^3 = ZZZ^2   bucket.rs:389-397
template <class F>
ARK_HD Jac<F> xyzz_to_jac(const XYZZ<F>& p) {
  if (p.is_zero()) return Jac<F>::zero();
  return Jac<F>{F::mul(p.x, p.zz), F::mul(p.y, p.zzz), p.zz};
}

}  // namespace arkhip
