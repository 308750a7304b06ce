// Unsaturated ("lazy") prime-field arithmetic for the MSM bucket-accumulation kernel.
//
// Why: on gfx950 the widest integer multiply-add is v_mad_u64_u32 (32x32+64 -> 64).  With saturated
// 32-bit limbs every partial product needs a second instruction to catch the carry out of the 64-bit
// accumulator (v_addc_co_u32 through VCC), and that carry instruction costs as much issue time as the
// multiply (profiles/r1_ubench_instruction_rates.txt: 5.4 cycles each).  With 28-bit limbs a column of a
// product-scanning multiplication -- up to 2L products of (< 2^30)^2 plus the Montgomery terms -- fits
// the 64-bit accumulator, so the inner loop is v_mad_u64_u32 only.  Additions need no carry chain at all
// (limb-wise), subtractions are limb-wise signed followed by one carry sweep, and no conditional
// subtraction of p is ever needed: the Montgomery radix 2^(28 L) exceeds p by >= 2^8, so values bounded by
// 8p going into a product come out below 1.04 p.
//
// Scope: values live in this form only inside msm_accumulate (ec formulas of
// ec/src/models/short_weierstrass/bucket.rs:168-238, madd-2008-s).  Bases are converted once per MSM
// (x * 2^(28L) mod p, 28-bit limbs), buckets are converted back to the reference's canonical Montgomery
// form (R = 2^(64 N)) when stored, so everything downstream -- and every result -- is bit-identical.
#pragma once
#include "../ec.cuh"

namespace arkhip {

template <class P_>
struct FpLazy {
  typedef P_ P;
  static constexpr int L = P::LZ_L;
  static constexpr int N = P::N;  // 32-bit limbs of the canonical form
  static constexpr u32 MASK = (1u << 28) - 1u;
  static constexpr int BYTES = 4 * L;
  u32 l[L];  // value = sum l[i] 2^(28 i); "normalised": every l[i] < 2^28

  ARK_HD static FpLazy zero() {
    FpLazy r;
#pragma unroll
    for (int i = 0; i < L; i++) r.l[i] = 0;
    return r;
  }
  ARK_HD bool limbs_all_zero() const {
    u32 o = 0;
#pragma unroll
    for (int i = 0; i < L; i++) o |= l[i];
    return o == 0;
  }

  // Montgomery product a*b*2^(-28L) mod p, output normalised and < a*b/2^(28L) + p.
  // Inputs: limbs < 2^30 (a column then stays below 2^64).
  ARK_HD static FpLazy mul(const FpLazy& a, const FpLazy& b) {
    u32 m[L];
    FpLazy r;
    u64 acc = 0;
#pragma unroll
    for (int k = 0; k < L; k++) {
#pragma unroll
      for (int i = 0; i <= k; i++) acc += (u64)a.l[i] * b.l[k - i];
#pragma unroll
      for (int i = 0; i < k; i++) acc += (u64)m[i] * P::LZ_KP[1][k - i];
      m[k] = ((u32)acc * P::LZ_INV) & MASK;
      acc += (u64)m[k] * P::LZ_KP[1][0];
      acc >>= 28;
    }
#pragma unroll
    for (int k = L; k < 2 * L - 1; k++) {
#pragma unroll
      for (int i = k - L + 1; i < L; i++) acc += (u64)a.l[i] * b.l[k - i];
#pragma unroll
      for (int i = k - L + 1; i < L; i++) acc += (u64)m[i] * P::LZ_KP[1][k - i];
      r.l[k - L] = (u32)acc & MASK;
      acc >>= 28;
    }
    r.l[L - 1] = (u32)acc;
    return r;
  }
  ARK_HD static FpLazy sqr(const FpLazy& a) { return mul(a, a); }

  // limb-wise sum, no carries (limbs grow by one bit)
  ARK_HD static FpLazy add_lazy(const FpLazy& a, const FpLazy& b) {
    FpLazy r;
#pragma unroll
    for (int i = 0; i < L; i++) r.l[i] = a.l[i] + b.l[i];
    return r;
  }
  // carry sweep over signed limbs; the value must be in [0, 2^(28L))
  ARK_HD static FpLazy normalise(const int* d) {
    FpLazy r;
    int carry = 0;
#pragma unroll
    for (int i = 0; i < L - 1; i++) {
      int v = d[i] + carry;
      r.l[i] = (u32)v & MASK;
      carry = v >> 28;  // arithmetic shift: floor division
    }
    r.l[L - 1] = (u32)(d[L - 1] + carry);
    return r;
  }
  // a - b + K p, normalised   (caller guarantees a - b + K p >= 0)
  template <int K>
  ARK_HD static FpLazy sub(const FpLazy& a, const FpLazy& b) {
    int d[L];
#pragma unroll
    for (int i = 0; i < L; i++) d[i] = (int)a.l[i] - (int)b.l[i] + (int)P::LZ_KP[K][i];
    return normalise(d);
  }
  // a - b - c + K p
  template <int K>
  ARK_HD static FpLazy sub2(const FpLazy& a, const FpLazy& b, const FpLazy& c) {
    int d[L];
#pragma unroll
    for (int i = 0; i < L; i++) d[i] = (int)a.l[i] - (int)b.l[i] - (int)c.l[i] + (int)P::LZ_KP[K][i];
    return normalise(d);
  }
  // K p - a - b
  template <int K>
  ARK_HD static FpLazy negsub(const FpLazy& a, const FpLazy& b) {
    int d[L];
#pragma unroll
    for (int i = 0; i < L; i++) d[i] = (int)P::LZ_KP[K][i] - (int)a.l[i] - (int)b.l[i];
    return normalise(d);
  }
  // is the (normalised, < 9p) value a multiple of p?  Cheap low-limb filter, exact compare behind it.
  ARK_HD bool is_zero_mod_p() const {
    bool hit = false;
#pragma unroll
    for (int k = 0; k < 9; k++) hit |= (l[0] == P::LZ_KP[k][0]);
    if (!hit) return false;
    bool any = false;
#pragma unroll
    for (int k = 0; k < 9; k++) {
      u32 o = 0;
#pragma unroll
      for (int i = 0; i < L; i++) o |= l[i] ^ P::LZ_KP[k][i];
      any |= (o == 0);
    }
    return any;
  }

  // ---- conversions with the canonical form (Fp<P>: 32-bit limbs, R = 2^(32N), value < p) ----
  // pack normalised 28-bit limbs of a value < 2^(32N) into 32-bit limbs (no reduction)
  ARK_HD void pack32(u32* out) const {
#pragma unroll
    for (int j = 0; j < N; j++) {
      const int bit = 32 * j;
      const int i = bit / 28, sh = bit % 28;
      // 32 bits starting at 28-bit-limb i, offset sh: spans limbs i, i+1 (and i+2 when sh > 24)
      u64 v = (u64)l[i] >> sh;
      if (i + 1 < L) v |= (u64)l[i + 1] << (28 - sh);
      if (i + 2 < L) v |= (u64)l[i + 2] << (56 - sh);
      out[j] = (u32)v;
    }
  }
  ARK_HD static FpLazy unpack32(const u32* in) {
    FpLazy r;
#pragma unroll
    for (int i = 0; i < L; i++) {
      const int bit = 28 * i;
      const int j = bit / 32, sh = bit % 32;
      u64 v = 0;
      if (j < N) v = (u64)in[j] >> sh;
      if (j + 1 < N) v |= (u64)in[j + 1] << (32 - sh);
      r.l[i] = (u32)v & MASK;
    }
    return r;
  }
  // canonical Montgomery (R = 2^(32N)) -> lazy form: x*2^(28L), normalised, < p
  ARK_HD static FpLazy from_canonical(const Fp<P>& a) {
    Fp<P> c;
#pragma unroll
    for (int i = 0; i < N; i++) c.l[i] = P::LZ_CIN[i];
    Fp<P> t = Fp<P>::mul(a, c);
    return unpack32(t.l);
  }
  // lazy (normalised, < 2^(28L)) -> canonical Montgomery (R = 2^(32N)), fully reduced:
  // one lazy product with the raw integer 2^(32N) mod p brings x*2^(28L) to x*2^(32N) below 1.04 p.
  ARK_HD Fp<P> to_canonical() const {
    FpLazy k = unpack32(P::R);
    FpLazy t = mul(*this, k);
    u32 w[N];
    t.pack32(w);
    return Fp<P>::reduce_once(w);
  }
  ARK_HD static FpLazy one() { return unpack32(P::LZ_CIN); }  // 2^(28L) mod p
  ARK_HD static FpLazy load(const void* p) {  // L u32 words, 16-byte aligned rows of 4*ceil(L/4) words
    FpLazy r;
#if defined(__HIP_DEVICE_COMPILE__)
    const u32* q = (const u32*)p;
    if constexpr (L % 4 == 0) {
#pragma unroll
      for (int i = 0; i < L / 4; i++) {
        uint4 v = ((const uint4*)q)[i];
        r.l[4 * i] = v.x; r.l[4 * i + 1] = v.y; r.l[4 * i + 2] = v.z; r.l[4 * i + 3] = v.w;
      }
    } else {
#pragma unroll
      for (int i = 0; i < L / 2; i++) {
        uint2 v = ((const uint2*)q)[i];
        r.l[2 * i] = v.x; r.l[2 * i + 1] = v.y;
      }
    }
#else
    __builtin_memcpy(r.l, p, BYTES);
#endif
    return r;
  }
  ARK_HD void store(void* p) const {
#if defined(__HIP_DEVICE_COMPILE__)
    u32* q = (u32*)p;
#pragma unroll
    for (int i = 0; i < L / 2; i++) ((uint2*)q)[i] = make_uint2(l[2 * i], l[2 * i + 1]);
#else
    __builtin_memcpy(p, l, BYTES);
#endif
  }
};

// ---- bucket (XYZZ) in lazy coordinates, with an explicit infinity flag ---------------------------
template <class P>
struct XYZZLazy {
  FpLazy<P> x, y, zz, zzz;  // x < 6p, y < 4p, zz/zzz < 1.04p; all normalised
  bool inf;
};

// rare branch of the mixed addition (same x): doubling through the canonical form, or infinity.
// Kept out of line so that its registers do not burden the hot loop.
template <class P>
__host__ __device__ __attribute__((noinline)) void xyzz_madd_lazy_special(XYZZLazy<P>& acc, const FpLazy<P>& x2,
                                                                          const FpLazy<P>& y2, bool neg, bool same_y) {
  typedef FpLazy<P> L;
  if (!same_y) {
    acc.inf = true;
    return;
  }
  Fp<P> cx = x2.to_canonical();
  Fp<P> cy = y2.to_canonical();
  if (neg) cy = Fp<P>::neg(cy);
  XYZZ<Fp<P>> d = xyzz_mdbl<Fp<P>>(cx, cy);
  acc.x = L::from_canonical(d.x);
  acc.y = L::from_canonical(d.y);
  acc.zz = L::from_canonical(d.zz);
  acc.zzz = L::from_canonical(d.zzz);
}

// acc += (x2, +-y2), (x2, y2) a non-identity affine point in lazy form (canonical values < p).
// Same formulas and branches as xyzz_madd (bucket.rs:168-238); bounds in the comments are in units of p.
template <class P>
ARK_HD void xyzz_madd_lazy(XYZZLazy<P>& acc, const FpLazy<P>& x2, const FpLazy<P>& y2, bool neg) {
  typedef FpLazy<P> L;
  if (acc.inf) {
    acc.x = x2;
    acc.y = neg ? L::template negsub<1>(y2, L::zero()) : y2;
    acc.zz = L::one();
    acc.zzz = L::one();
    acc.inf = false;
    return;
  }
  L u2 = L::mul(x2, acc.zz);                                   // < 1.04
  L s2 = L::mul(y2, acc.zzz);                                  // < 1.04
  L pd = L::template sub<6>(u2, acc.x);                        // (0, 7.04)
  L rd = neg ? L::template negsub<6>(s2, acc.y)                // (0.96, 6]
             : L::template sub<4>(s2, acc.y);                  // (0, 5.04)
  if (pd.is_zero_mod_p()) {
    xyzz_madd_lazy_special<P>(acc, x2, y2, neg, rd.is_zero_mod_p());
    return;
  }
  L pp = L::sqr(pd);
  L ppp = L::mul(pd, pp);
  L q = L::mul(acc.x, pp);
  L x3 = L::template sub2<4>(L::sqr(rd), ppp, L::add_lazy(q, q));  // (0.88, 5.04)
  L t = L::template sub<6>(q, x3);                                 // (0.96, 7.04)
  L y3 = L::template sub<2>(L::mul(rd, t), L::mul(acc.y, ppp));    // (0.96, 3.04)
  acc.zz = L::mul(acc.zz, pp);
  acc.zzz = L::mul(acc.zzz, ppp);
  acc.x = x3;
  acc.y = y3;
}

template <class P>
ARK_HD XYZZ<Fp<P>> xyzz_from_lazy(const XYZZLazy<P>& a) {
  if (a.inf) return XYZZ<Fp<P>>::zero();
  return XYZZ<Fp<P>>{a.x.to_canonical(), a.y.to_canonical(), a.zz.to_canonical(), a.zzz.to_canonical()};
}

}  // namespace arkhip
