// Prime-field arithmetic for gfx950 (MI355X), Montgomery form, 32-bit limbs.
//
// Replaces, on the device, the reference's Fp<MontBackend<C,N>,N>:
//   ff/src/fields/models/fp/montgomery_backend.rs:129-171 (add/sub/double/neg),
//   :181-246 (mul), :250-317 (square), :396-412 (into_bigint)
// The in-memory layout is the reference's: N64 little-endian u64 limbs == 2*N64 little-endian
// u32 limbs (ff/src/biginteger/mod.rs:34), Montgomery residue with R = 2^(64*N64), always fully
// reduced to [0, p) -- so every value has exactly one bit pattern and results are bit-identical
// to the reference whatever multiplication schedule is used.
//
// gfx950 has no 64x64 multiplier: the widest integer MAC is v_mad_u64_u32 (32x32+64 -> 64, with
// carry-out in VCC, quarter rate).  The multiply below is a product-scanning (Comba) Montgomery
// multiplication: column k accumulates  sum a[i]*b[k-i] + sum m[i]*p[k-i]  in a 96-bit
// accumulator {lo64, hi32} with one v_mad_u64_u32 + v_addc_co_u32 per 32x32 product; modulus
// limbs live in SGPRs (they are wave-uniform constants).  Measured: 58.7 G Fp384-mul/s,
// 124 G Fp256-mul/s per MI355X (profiles/r1_ubench_instruction_rates.txt).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "params.hpp"

namespace arkhip {
typedef uint32_t u32;
typedef uint64_t u64;

#define ARK_DEV __device__ __forceinline__
#define ARK_HD __host__ __device__ __forceinline__
// The equal-points branch of a bucket addition (a doubling: duplicate bases only).  ARK_COLD_INLINE=1 (default since round 6)
// expands it in place so that no kernel carries a call frame in scratch memory; 0 restores the out-of-line calls of rounds
// 3-5 (an accumulator handed to a call by reference lives in scratch: 240-368 B per lane) for A/B builds.
#ifndef ARK_COLD_INLINE
#define ARK_COLD_INLINE 1
#endif
#if ARK_COLD_INLINE
#define ARK_COLD_HD __host__ __device__ __forceinline__
#define ARK_COLD_DEV __device__ __forceinline__
#else
#define ARK_COLD_HD __host__ __device__ __attribute__((noinline))
#define ARK_COLD_DEV __device__ __attribute__((noinline))
#endif

// ---- 96-bit column accumulator --------------------------------------------------------------
// One product step: {lo64, hi32} += A * B.  FIRST opens a column: the carry initialises the top word (0 + 0 + carry),
// so the shift between columns needs no zeroing move.
#define ARK_S(A, B) "v_mad_u64_u32 %0, vcc, %" #A ", %" #B ", %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
#define ARK_F(A, B) "v_mad_u64_u32 %0, vcc, %" #A ", %" #B ", %0\n\tv_addc_co_u32 %1, vcc, 0, 0, vcc\n\t"
// N steps in ONE asm statement (hipcc pads every asm statement boundary with wait states; a column of the product is
// two statements: its a*b terms, its m*p terms).  Operands: %0 lo64, %1 hi32, then the N (x, y) pairs.
#define ARK_T1 ARK_S(2, 3)
#define ARK_T2 ARK_T1 ARK_S(4, 5)
#define ARK_T3 ARK_T2 ARK_S(6, 7)
#define ARK_T4 ARK_T3 ARK_S(8, 9)
#define ARK_T5 ARK_T4 ARK_S(10, 11)
#define ARK_T6 ARK_T5 ARK_S(12, 13)
#define ARK_T7 ARK_T6 ARK_S(14, 15)
#define ARK_T8 ARK_T7 ARK_S(16, 17)
#define ARK_T9 ARK_T8 ARK_S(18, 19)
#define ARK_T10 ARK_T9 ARK_S(20, 21)
#define ARK_T11 ARK_T10 ARK_S(22, 23)
#define ARK_T12 ARK_T11 ARK_S(24, 25)
#define ARK_T13 ARK_T12 ARK_S(26, 27)
#define ARK_I1 ARK_F(2, 3)
#define ARK_I2 ARK_I1 ARK_S(4, 5)
#define ARK_I3 ARK_I2 ARK_S(6, 7)
#define ARK_I4 ARK_I3 ARK_S(8, 9)
#define ARK_I5 ARK_I4 ARK_S(10, 11)
#define ARK_I6 ARK_I5 ARK_S(12, 13)
#define ARK_I7 ARK_I6 ARK_S(14, 15)
#define ARK_I8 ARK_I7 ARK_S(16, 17)
#define ARK_I9 ARK_I8 ARK_S(18, 19)
#define ARK_I10 ARK_I9 ARK_S(20, 21)
#define ARK_I11 ARK_I10 ARK_S(22, 23)
#define ARK_I12 ARK_I11 ARK_S(24, 25)
#define ARK_I13 ARK_I12 ARK_S(26, 27)
// operand lists: x[LO + i] * y[K - LO - i] (both in VGPRs) / m[LO + i] * p[K - LO - i] (modulus limbs: SGPRs)
#define ARK_OV(i) "v"(x[LO + i]), "v"(y[K - LO - i])
#define ARK_OV1 ARK_OV(0)
#define ARK_OV2 ARK_OV1, ARK_OV(1)
#define ARK_OV3 ARK_OV2, ARK_OV(2)
#define ARK_OV4 ARK_OV3, ARK_OV(3)
#define ARK_OV5 ARK_OV4, ARK_OV(4)
#define ARK_OV6 ARK_OV5, ARK_OV(5)
#define ARK_OV7 ARK_OV6, ARK_OV(6)
#define ARK_OV8 ARK_OV7, ARK_OV(7)
#define ARK_OV9 ARK_OV8, ARK_OV(8)
#define ARK_OV10 ARK_OV9, ARK_OV(9)
#define ARK_OV11 ARK_OV10, ARK_OV(10)
#define ARK_OV12 ARK_OV11, ARK_OV(11)
#define ARK_OV13 ARK_OV12, ARK_OV(12)
#define ARK_OP(i) "v"(x[LO + i]), "s"(PL<P, K - LO - i>::v)
#define ARK_OP1 ARK_OP(0)
#define ARK_OP2 ARK_OP1, ARK_OP(1)
#define ARK_OP3 ARK_OP2, ARK_OP(2)
#define ARK_OP4 ARK_OP3, ARK_OP(3)
#define ARK_OP5 ARK_OP4, ARK_OP(4)
#define ARK_OP6 ARK_OP5, ARK_OP(5)
#define ARK_OP7 ARK_OP6, ARK_OP(6)
#define ARK_OP8 ARK_OP7, ARK_OP(7)
#define ARK_OP9 ARK_OP8, ARK_OP(8)
#define ARK_OP10 ARK_OP9, ARK_OP(9)
#define ARK_OP11 ARK_OP10, ARK_OP(10)
#define ARK_OP12 ARK_OP11, ARK_OP(11)
#define ARK_OP13 ARK_OP12, ARK_OP(12)
#define ARK_STMT(n, STR, HI, OPS) \
  if constexpr (HIX - LO + 1 == n) asm(STR##n : "+v"(c.lo), HI(c.hi) : OPS##n : "vcc")
#define ARK_HI_RW "+v"
#define ARK_HI_W "=&v"
#define ARK_ALL(STR, HI, OPS)                                                                                         \
  ARK_STMT(1, STR, HI, OPS); ARK_STMT(2, STR, HI, OPS); ARK_STMT(3, STR, HI, OPS); ARK_STMT(4, STR, HI, OPS);         \
  ARK_STMT(5, STR, HI, OPS); ARK_STMT(6, STR, HI, OPS); ARK_STMT(7, STR, HI, OPS); ARK_STMT(8, STR, HI, OPS);         \
  ARK_STMT(9, STR, HI, OPS); ARK_STMT(10, STR, HI, OPS); ARK_STMT(11, STR, HI, OPS); ARK_STMT(12, STR, HI, OPS);      \
  ARK_STMT(13, STR, HI, OPS)

struct Acc96 { u64 lo; u32 hi; };
// -p0^-1 mod 2^64 by Newton iteration (host Montgomery product: a compile-time constant per modulus)
constexpr u64 host_inv64(u64 p0) {
  u64 z = 1;
  for (int i = 0; i < 6; i++) z *= 2 - p0 * z;
  return 0 - z;
}
ARK_DEV void acc_shift(Acc96& c) { c.lo = (c.lo >> 32) | ((u64)c.hi << 32); c.hi = 0; }

// compile-time access to modulus limbs (keeps them immediates -> s_mov)
template <class P, int I> struct PL { static constexpr u32 v = P::P[I]; };

// c += sum_{i=LO..HIX} x[i] * y[K-i]   (both operands in VGPRs); FIRST: these are the first terms of the column
template <int LO, int HIX, int K, bool FIRST = false>
ARK_DEV void col_vv(Acc96& c, const u32* x, const u32* y) {
  static_assert(HIX - LO + 1 <= 13, "one asm statement takes at most 13 products (30 operands)");
  if constexpr (FIRST) {
    static_assert(HIX >= LO, "a column opens with at least one product");
    ARK_ALL(ARK_I, ARK_HI_W, ARK_OV);
  } else {
    ARK_ALL(ARK_T, ARK_HI_RW, ARK_OV);
  }
}
// c += sum_{i=LO..HIX} m[i] * p[K-i]   (p = modulus, SGPR operands)
template <class P, int LO, int HIX, int K>
ARK_DEV void col_vp(Acc96& c, const u32* x) {
  static_assert(HIX - LO + 1 <= 13, "one asm statement takes at most 13 products (30 operands)");
  ARK_ALL(ARK_T, ARK_HI_RW, ARK_OP);
}
// squaring column: c += sum_{i+j=K, i<j} 2*a[i]*a[j] + (K even ? a[K/2]^2 : 0), done as doubled operand d[] = 2a
// (not used yet: mul(a,a) is the square in this round)

template <class P, int K>
ARK_DEV void mont_cols_lo(Acc96& c, const u32* a, const u32* b, u32* m) {
  constexpr int N = P::N;
  col_vv<0, K, K, true>(c, a, b);
  col_vp<P, 0, K - 1, K>(c, m);
  m[K] = (u32)c.lo * P::INV;
  col_vp<P, K, K, K>(c, m);
  acc_shift(c);
  if constexpr (K + 1 < N) mont_cols_lo<P, K + 1>(c, a, b, m);
}
template <class P, int K>
ARK_DEV void mont_cols_hi(Acc96& c, const u32* a, const u32* b, const u32* m, u32* t) {
  constexpr int N = P::N;
  col_vv<K - N + 1, N - 1, K, true>(c, a, b);
  col_vp<P, K - N + 1, N - 1, K>(c, m);
  t[K - N] = (u32)c.lo;
  acc_shift(c);
  if constexpr (K + 1 < 2 * N - 1) mont_cols_hi<P, K + 1>(c, a, b, m, t);
}

// the same columns for a sum of two products  a*b + a2*b2  under ONE interleaved reduction
// (montgomery_backend.rs:415-516 sum_of_products, M = 2): per column up to 3N products, < 2^70 -- fits the 96-bit
// accumulator
template <class P, int K>
ARK_DEV void mont2_cols_lo(Acc96& c, const u32* a, const u32* b, const u32* a2, const u32* b2, u32* m) {
  constexpr int N = P::N;
  col_vv<0, K, K, true>(c, a, b);
  col_vv<0, K, K>(c, a2, b2);
  col_vp<P, 0, K - 1, K>(c, m);
  m[K] = (u32)c.lo * P::INV;
  col_vp<P, K, K, K>(c, m);
  acc_shift(c);
  if constexpr (K + 1 < N) mont2_cols_lo<P, K + 1>(c, a, b, a2, b2, m);
}
template <class P, int K>
ARK_DEV void mont2_cols_hi(Acc96& c, const u32* a, const u32* b, const u32* a2, const u32* b2, const u32* m, u32* t) {
  constexpr int N = P::N;
  col_vv<K - N + 1, N - 1, K, true>(c, a, b);
  col_vv<K - N + 1, N - 1, K>(c, a2, b2);
  col_vp<P, K - N + 1, N - 1, K>(c, m);
  t[K - N] = (u32)c.lo;
  acc_shift(c);
  if constexpr (K + 1 < 2 * N - 1) mont2_cols_hi<P, K + 1>(c, a, b, a2, b2, m, t);
}

// ... and of four products (an Fp2 sum of two products, per component): up to 5N products per column, < 2^71
template <class P, int K>
ARK_DEV void mont4_cols_lo(Acc96& c, const u32* const* x, const u32* const* y, u32* m) {
  constexpr int N = P::N;
  col_vv<0, K, K, true>(c, x[0], y[0]);
  col_vv<0, K, K>(c, x[1], y[1]);
  col_vv<0, K, K>(c, x[2], y[2]);
  col_vv<0, K, K>(c, x[3], y[3]);
  col_vp<P, 0, K - 1, K>(c, m);
  m[K] = (u32)c.lo * P::INV;
  col_vp<P, K, K, K>(c, m);
  acc_shift(c);
  if constexpr (K + 1 < N) mont4_cols_lo<P, K + 1>(c, x, y, m);
}
template <class P, int K>
ARK_DEV void mont4_cols_hi(Acc96& c, const u32* const* x, const u32* const* y, const u32* m, u32* t) {
  constexpr int N = P::N;
  col_vv<K - N + 1, N - 1, K, true>(c, x[0], y[0]);
  col_vv<K - N + 1, N - 1, K>(c, x[1], y[1]);
  col_vv<K - N + 1, N - 1, K>(c, x[2], y[2]);
  col_vv<K - N + 1, N - 1, K>(c, x[3], y[3]);
  col_vp<P, K - N + 1, N - 1, K>(c, m);
  t[K - N] = (u32)c.lo;
  acc_shift(c);
  if constexpr (K + 1 < 2 * N - 1) mont4_cols_hi<P, K + 1>(c, x, y, m, t);
}

// ---- the field element ----------------------------------------------------------------------
template <class P_>
struct Fp {
  typedef P_ P;
  static constexpr int N = P::N;        // 32-bit limbs
  static constexpr int WORDS64 = N / 2; // u64 words in memory
  static constexpr int BYTES = 4 * N;
  static constexpr int LANES = 1;       // lanes an element is spread over (Fp2Half: 2)
  static constexpr int FULL_BYTES = BYTES;
  static constexpr bool FUSED_Y3 = true;  // has sop2_r (ec.cuh: Y3 of the mixed addition under one reduction)
  u32 l[N];

  ARK_HD static Fp zero() {
    Fp r;
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = 0;
    return r;
  }
  ARK_HD static Fp one() {  // Montgomery R (montgomery_backend.rs:21)
    Fp r;
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = P::R[i];
    return r;
  }
#if !defined(__HIP_DEVICE_COMPILE__)
  // host-only helpers: the element on 64-bit limbs (N is even for every field here)
  static_assert(N % 2 == 0, "64-bit host limbs");
  static constexpr u64 host_p(int i) { return ((u64)P::P[2 * i + 1] << 32) | P::P[2 * i]; }
  static void host_limbs(const Fp& a, u64* x) {
    for (int i = 0; i < N / 2; i++) x[i] = ((u64)a.l[2 * i + 1] << 32) | a.l[2 * i];
  }
  static Fp host_from_limbs(const u64* x) {
    Fp r;
    for (int i = 0; i < N / 2; i++) {
      r.l[2 * i] = (u32)x[i];
      r.l[2 * i + 1] = (u32)(x[i] >> 32);
    }
    return r;
  }
  static Fp host_reduce_once(const u64* t) {   // t < 2p: t - p if that is not negative
    u64 d[N / 2];
    unsigned long long bw = 0;
    for (int i = 0; i < N / 2; i++) d[i] = __builtin_subcll(t[i], host_p(i), bw, &bw);
    Fp r;
    for (int i = 0; i < N / 2; i++) {
      const u64 v = bw ? t[i] : d[i];
      r.l[2 * i] = (u32)v;
      r.l[2 * i + 1] = (u32)(v >> 32);
    }
    return r;
  }
#endif
  ARK_HD bool is_zero() const {
    u32 o = 0;
#pragma unroll
    for (int i = 0; i < N; i++) o |= l[i];
    return o == 0;
  }
  ARK_HD static bool eq(const Fp& a, const Fp& b) {
    u32 o = 0;
#pragma unroll
    for (int i = 0; i < N; i++) o |= a.l[i] ^ b.l[i];
    return o == 0;
  }
  // r = t - p if t >= p else t       (t < 2p < 2^(32N): every modulus here has a spare top bit)
  // Carry chains are written with __builtin_addc/__builtin_subc: hipcc lowers them to one
  // v_add_co/v_addc_co (v_sub_co/v_subb_co) per limb.
  ARK_HD static Fp reduce_once(const u32* t) {
#if !defined(__HIP_DEVICE_COMPILE__)
    {   // host (the MSM's serial tail): the same on 64-bit limbs -- the 32-bit carry chains below cost the host more than its products
      u64 x[N / 2];
      for (int i = 0; i < N / 2; i++) x[i] = ((u64)t[2 * i + 1] << 32) | t[2 * i];
      return host_reduce_once(x);
    }
#endif
    u32 d[N];
    u32 borrow = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
      u32 bo;
      d[i] = __builtin_subc(t[i], (u32)P::P[i], borrow, &bo);
      borrow = bo;
    }
    Fp r;
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = borrow ? t[i] : d[i];
    return r;
  }
  ARK_HD static Fp add(const Fp& a, const Fp& b) {  // montgomery_backend.rs:129-136
#if !defined(__HIP_DEVICE_COMPILE__)
    {
      u64 x[N / 2], y[N / 2];
      host_limbs(a, x);
      host_limbs(b, y);
      unsigned long long c = 0;
      for (int i = 0; i < N / 2; i++) x[i] = __builtin_addcll(x[i], y[i], c, &c);
      return host_reduce_once(x);   // (a + b < 2p < 2^(32 N): no carry out)
    }
#endif
    u32 t[N];
    u32 c = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
      u32 co;
      t[i] = __builtin_addc(a.l[i], b.l[i], c, &co);
      c = co;
    }
    return reduce_once(t);
  }
  ARK_HD static Fp dbl(const Fp& a) {  // montgomery_backend.rs:151-171
#if !defined(__HIP_DEVICE_COMPILE__)
    {
      u64 x[N / 2];
      host_limbs(a, x);
      for (int i = N / 2 - 1; i > 0; i--) x[i] = (x[i] << 1) | (x[i - 1] >> 63);
      x[0] <<= 1;
      return host_reduce_once(x);
    }
#endif
    u32 t[N];
#pragma unroll
    for (int i = N - 1; i > 0; i--) t[i] = (a.l[i] << 1) | (a.l[i - 1] >> 31);
    t[0] = a.l[0] << 1;
    return reduce_once(t);
  }
  ARK_HD static Fp sub(const Fp& a, const Fp& b) {  // montgomery_backend.rs:138-149
#if !defined(__HIP_DEVICE_COMPILE__)
    {
      u64 x[N / 2], y[N / 2];
      host_limbs(a, x);
      host_limbs(b, y);
      unsigned long long bw = 0;
      for (int i = 0; i < N / 2; i++) x[i] = __builtin_subcll(x[i], y[i], bw, &bw);
      const u64 mask = 0ull - (u64)bw;
      unsigned long long c = 0;
      for (int i = 0; i < N / 2; i++) x[i] = __builtin_addcll(x[i], host_p(i) & mask, c, &c);
      return host_from_limbs(x);
    }
#endif
    u32 d[N];
    u32 borrow = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
      u32 bo;
      d[i] = __builtin_subc(a.l[i], b.l[i], borrow, &bo);
      borrow = bo;
    }
    const u32 mask = 0u - borrow;
    Fp r;
    u32 c = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
      u32 co;
      r.l[i] = __builtin_addc(d[i], (u32)P::P[i] & mask, c, &co);
      c = co;
    }
    return r;
  }
  ARK_HD static Fp neg(const Fp& a) {  // 0 stays 0 (fp/mod.rs Neg)
#if !defined(__HIP_DEVICE_COMPILE__)
    {
      u64 x[N / 2], r[N / 2];
      host_limbs(a, x);
      u64 nz64 = 0;
      for (int i = 0; i < N / 2; i++) nz64 |= x[i];
      const u64 mask = nz64 ? ~0ull : 0ull;
      unsigned long long bw = 0;
      for (int i = 0; i < N / 2; i++) r[i] = __builtin_subcll(host_p(i) & mask, x[i], bw, &bw);
      return host_from_limbs(r);
    }
#endif
    u32 nz = 0;
#pragma unroll
    for (int i = 0; i < N; i++) nz |= a.l[i];
    const u32 mask = nz ? 0xffffffffu : 0u;
    Fp r;
    u32 borrow = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
      u32 bo;
      r.l[i] = __builtin_subc((u32)P::P[i] & mask, a.l[i], borrow, &bo);
      borrow = bo;
    }
    return r;
  }
  // Montgomery product a*b*R^-1 mod p  (montgomery_backend.rs:181-246; product-scanning schedule)
  ARK_HD static Fp mul(const Fp& a, const Fp& b) {
#if defined(__HIP_DEVICE_COMPILE__)
    Acc96 c{0, 0};
    u32 m[N];
    u32 t[N];
    mont_cols_lo<P, 0>(c, a.l, b.l, m);
    mont_cols_hi<P, N>(c, a.l, b.l, m, t);
    t[N - 1] = (u32)c.lo;
    return reduce_once(t);
#else
    // host build of the same function (used only for the serial tail of the MSM -- the
    // 255-doubling window combine -- and for domain constants): 64-bit-limb CIOS.
    constexpr int M = N / 2;
    u64 x[M], y[M], pp[M], t[M + 2];
    for (int i = 0; i < M; i++) {
      x[i] = ((u64)a.l[2 * i + 1] << 32) | a.l[2 * i];
      y[i] = ((u64)b.l[2 * i + 1] << 32) | b.l[2 * i];
      pp[i] = ((u64)P::P[2 * i + 1] << 32) | P::P[2 * i];
    }
    constexpr u64 inv = host_inv64(((u64)P::P[1] << 32) | P::P[0]);  // -p^-1 mod 2^64, folded at compile time
    for (int i = 0; i < M + 2; i++) t[i] = 0;
    for (int i = 0; i < M; i++) {
      unsigned __int128 c = 0;
      for (int j = 0; j < M; j++) {
        c += (unsigned __int128)x[j] * y[i] + t[j];
        t[j] = (u64)c;
        c >>= 64;
      }
      c += t[M];
      t[M] = (u64)c;
      t[M + 1] = (u64)(c >> 64);
      u64 mm = t[0] * inv;
      c = (unsigned __int128)mm * pp[0] + t[0];
      c >>= 64;
      for (int j = 1; j < M; j++) {
        c += (unsigned __int128)mm * pp[j] + t[j];
        t[j - 1] = (u64)c;
        c >>= 64;
      }
      c += t[M];
      t[M - 1] = (u64)c;
      t[M] = t[M + 1] + (u64)(c >> 64);
    }
    return host_reduce_once(t);
#endif
  }
  ARK_HD static Fp sqr(const Fp& a) { return mul(a, a); }  // montgomery_backend.rs:250-317

  // ---- "relaxed" residues in [0, 2p) for the MSM inner loop -------------------------------------------
  // Every modulus served here satisfies 4p <= 2^(32N) (p/R = 0.10, 0.007, 0.19 for the three base fields), so a
  // Montgomery product of operands < 2p is itself < 2p without the final conditional subtraction:
  // (a b + m p)/R < p (4p/R + 1) <= 2p.  Sums and differences are brought back below 2p with one
  // conditional +-2p.  Saves the 2N-instruction tail of every product in the bucket accumulation; values are
  // made canonical again (one conditional subtraction) before they are stored.
  ARK_HD static Fp mul_r(const Fp& a, const Fp& b) {
    static_assert((P::P[N - 1] >> 30) == 0, "relaxed arithmetic needs 4p <= R");
#if defined(__HIP_DEVICE_COMPILE__)
    Acc96 c{0, 0};
    u32 m[N];
    Fp r;
    mont_cols_lo<P, 0>(c, a.l, b.l, m);
    mont_cols_hi<P, N>(c, a.l, b.l, m, r.l);
    r.l[N - 1] = (u32)c.lo;
    return r;
#else
    return mul(a, b);  // canonical is a valid relaxed representative
#endif
  }
  ARK_HD static Fp sqr_r(const Fp& a) { return mul_r(a, a); }
  // t (< 4p) -> t or t - 2p, whichever lies in [0, 2p)
  ARK_HD static Fp reduce_2p(const u32* t) {
    u32 d[N];
    u32 borrow = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
      const u32 p2 = (i == 0) ? ((u32)P::P[0] << 1) : (((u32)P::P[i] << 1) | ((u32)P::P[i - 1] >> 31));
      u32 bo;
      d[i] = __builtin_subc(t[i], p2, borrow, &bo);
      borrow = bo;
    }
    Fp r;
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = borrow ? t[i] : d[i];
    return r;
  }
  ARK_HD static Fp add_r(const Fp& a, const Fp& b) {
    u32 t[N];
    u32 c = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
      u32 co;
      t[i] = __builtin_addc(a.l[i], b.l[i], c, &co);
      c = co;
    }
    return reduce_2p(t);
  }
  ARK_HD static Fp dbl_r(const Fp& a) {
    u32 t[N];
#pragma unroll
    for (int i = N - 1; i > 0; i--) t[i] = (a.l[i] << 1) | (a.l[i - 1] >> 31);
    t[0] = a.l[0] << 1;
    return reduce_2p(t);
  }
  ARK_HD static Fp sub_r(const Fp& a, const Fp& b) {  // a - b (+ 2p if negative)
    u32 d[N];
    u32 borrow = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
      u32 bo;
      d[i] = __builtin_subc(a.l[i], b.l[i], borrow, &bo);
      borrow = bo;
    }
    const u32 mask = 0u - borrow;
    Fp r;
    u32 c = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
      const u32 p2 = (i == 0) ? ((u32)P::P[0] << 1) : (((u32)P::P[i] << 1) | ((u32)P::P[i - 1] >> 31));
      u32 co;
      r.l[i] = __builtin_addc(d[i], p2 & mask, c, &co);
      c = co;
    }
    return r;
  }
  // The same relaxed domain [0, 2p) for fields whose 4p may exceed R (BLS12-381 Fr: p = 0.45 R) -- the FFT butterflies.
  // add_r2: the sum of two relaxed values may carry out of the N limbs; it is then certainly >= 2p.
  ARK_HD static Fp add_r2(const Fp& a, const Fp& b) {
    u32 t[N], d[N];
    u32 c = 0, borrow = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
      u32 co;
      t[i] = __builtin_addc(a.l[i], b.l[i], c, &co);
      c = co;
    }
#pragma unroll
    for (int i = 0; i < N; i++) {
      const u32 p2 = (i == 0) ? ((u32)P::P[0] << 1) : (((u32)P::P[i] << 1) | ((u32)P::P[i - 1] >> 31));
      u32 bo;
      d[i] = __builtin_subc(t[i], p2, borrow, &bo);
      borrow = bo;
    }
    const bool keep = borrow != 0 && c == 0;  // t < 2p and no carry: t stands
    Fp r;
#pragma unroll
    for (int i = 0; i < N; i++) r.l[i] = keep ? t[i] : d[i];
    return r;
  }
  // relaxed a (< 2p) times CANONICAL b (< p): (2p^2 + R p) / R < 2p for every p < R/2 -- no condition on 4p <= R
  ARK_HD static Fp mul_r1(const Fp& a, const Fp& b) {
#if defined(__HIP_DEVICE_COMPILE__)
    Acc96 c{0, 0};
    u32 m[N];
    Fp r;
    mont_cols_lo<P, 0>(c, a.l, b.l, m);
    mont_cols_hi<P, N>(c, a.l, b.l, m, r.l);
    r.l[N - 1] = (u32)c.lo;
    return r;
#else
    return mul(a.canonical(), b);
#endif
  }
  // 2p - a for a relaxed a (<= 2p): a representative of -a in [0, 2p]; only ever used as a multiplication operand
  ARK_HD static Fp neg_r(const Fp& a) {
    Fp r;
    u32 borrow = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
      const u32 p2 = (i == 0) ? ((u32)P::P[0] << 1) : (((u32)P::P[i] << 1) | ((u32)P::P[i - 1] >> 31));
      u32 bo;
      r.l[i] = __builtin_subc(p2, a.l[i], borrow, &bo);
      borrow = bo;
    }
    return r;
  }
  // a*b + c*d with one interleaved reduction on relaxed operands (all <= 2p): (8p^2 + m p)/R < p (8p/R + 1), i.e.
  // already < 2p when 8p <= R (BLS12-381, BLS12-377); otherwise (BN254: < 2.6p) one conditional -2p follows.
  // Saves one of the two reductions of  Y3 = R (Q - X3) - Y1 PPP  in every bucket addition.
  ARK_HD static Fp sop2_r(const Fp& a, const Fp& b, const Fp& c2, const Fp& d) {
#if defined(__HIP_DEVICE_COMPILE__)
    Acc96 c{0, 0};
    u32 m[N];
    u32 t[N];
    mont2_cols_lo<P, 0>(c, a.l, b.l, c2.l, d.l, m);
    mont2_cols_hi<P, N>(c, a.l, b.l, c2.l, d.l, m, t);
    t[N - 1] = (u32)c.lo;
    if constexpr ((P::P[N - 1] >> 29) == 0) {
      Fp r;
#pragma unroll
      for (int i = 0; i < N; i++) r.l[i] = t[i];
      return r;
    } else {
      return reduce_2p(t);
    }
#else
    return add(mul(a.canonical(), b.canonical()), mul(c2.canonical(), d.canonical()));
#endif
  }
  // x0*y0 + x1*y1 + x2*y2 + x3*y3 under one reduction (device only).  The caller guarantees sum < R*p * (1 or 3):
  // FOLD = false: sum <= R p, result < 2p as it stands; FOLD = true: sum <= 3 R p, one conditional -2p follows.
  template <bool FOLD>
  ARK_DEV static Fp sop4_r(const Fp& x0, const Fp& y0, const Fp& x1, const Fp& y1, const Fp& x2, const Fp& y2, const Fp& x3,
                           const Fp& y3) {
    const u32* xs[4] = {x0.l, x1.l, x2.l, x3.l};
    const u32* ys[4] = {y0.l, y1.l, y2.l, y3.l};
    Acc96 c{0, 0};
    u32 m[N];
    u32 t[N];
    mont4_cols_lo<P, 0>(c, xs, ys, m);
    mont4_cols_hi<P, N>(c, xs, ys, m, t);
    t[N - 1] = (u32)c.lo;
    if constexpr (FOLD) {
      return reduce_2p(t);
    } else {
      Fp r;
#pragma unroll
      for (int i = 0; i < N; i++) r.l[i] = t[i];
      return r;
    }
  }
  // canonical form: a*b + c*d for operands < p, except that `c2` may be any N-limb value up to 6p (a small multiple
  // of a negated element: the -beta*a1 term of an Fp2 product); (p^2 + 6p^2 + m p)/R < p (7p/R + 1) < 2p.
  ARK_HD static Fp sop2(const Fp& a, const Fp& b, const Fp& c2, const Fp& d) {
#if defined(__HIP_DEVICE_COMPILE__)
    Acc96 c{0, 0};
    u32 m[N];
    u32 t[N];
    mont2_cols_lo<P, 0>(c, a.l, b.l, c2.l, d.l, m);
    mont2_cols_hi<P, N>(c, a.l, b.l, c2.l, d.l, m, t);
    t[N - 1] = (u32)c.lo;
    return reduce_once(t);
#else
    return add(mul(a, b), mul(reduce_full(c2), d));
#endif
  }
  // host helper: any N-limb value -> [0, p) by repeated subtraction (only small multiples of p occur)
  ARK_HD static Fp reduce_full(const Fp& a) {
    Fp r = a;
    for (int it = 0; it < 8; it++) {
      u32 d[N];
      u32 borrow = 0;
#pragma unroll
      for (int i = 0; i < N; i++) {
        u32 bo;
        d[i] = __builtin_subc(r.l[i], (u32)P::P[i], borrow, &bo);
        borrow = bo;
      }
      if (borrow) break;
#pragma unroll
      for (int i = 0; i < N; i++) r.l[i] = d[i];
    }
    return r;
  }
  ARK_HD bool is_zero_mod_p() const {  // relaxed value: 0 or p
    if (l[0] != 0 && l[0] != (u32)P::P[0]) return false;  // almost always decided by the low limb (2 compares, not 2N)
    u32 o = 0, q = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
      o |= l[i];
      q |= l[i] ^ (u32)P::P[i];
    }
    return o == 0 || q == 0;
  }
  // relaxed value in [0, 2p] -> [0, p).  Two conditional subtractions: 2p itself occurs (neg_r of a zero component of
  // an Fp2 coordinate); only used where residues leave the relaxed domain (end of a kernel, the rare doubling branch)
  ARK_HD Fp canonical() const { return reduce_once(reduce_once(l).l); }
  // Out-of-line copy for the extension-field formulas: an XYZZ addition over Fp2 would otherwise inline ~40 copies
  // of this 700-instruction sequence (minutes of compile time per kernel, scratch spills).  Operands travel by value
  // so the AMDGPU calling convention keeps them in VGPRs.
  __host__ __device__ __attribute__((noinline)) static Fp mul_call(Fp a, Fp b) { return mul(a, b); }
  __host__ __device__ __attribute__((noinline)) static Fp sop2_call(Fp a, Fp b, Fp c, Fp d) { return sop2(a, b, c, d); }
  // Montgomery -> canonical integer (montgomery_backend.rs:396-412): multiply by 1
  ARK_HD static Fp from_mont(const Fp& a) {
    Fp o = zero();
    o.l[0] = 1;
    return mul(a, o);
  }
  ARK_HD static Fp to_mont(const Fp& a) {  // canonical -> Montgomery: a * R2 * R^-1
    Fp r2;
#pragma unroll
    for (int i = 0; i < N; i++) r2.l[i] = P::R2[i];
    return mul(a, r2);
  }
  ARK_HD static Fp cond_neg(const Fp& a, bool n) { return n ? neg(a) : a; }
  // a^(p-2) (Fermat).  The reference uses a binary extended Euclid (montgomery_backend.rs:319-378);
  // the value is the same canonical residue.  Off the hot path: domain constants, into_affine,
  // synthetic-base generation.
  ARK_HD static Fp inverse(const Fp& a) {
    u32 e[N];
    u32 borrow = 2;
    for (int i = 0; i < N; i++) {
      u64 x = (u64)P::P[i] - borrow;
      e[i] = (u32)x;
      borrow = (u32)(x >> 63);
    }
    Fp r = one();
    for (int i = 32 * N - 1; i >= 0; i--) {
      r = sqr(r);
      if ((e[i >> 5] >> (i & 31)) & 1) r = mul(r, a);
    }
    return r;
  }

  // ---- memory: 16-byte vector loads/stores of the reference layout ----
  ARK_HD static Fp load(const void* p) {
    Fp r;
#if defined(__HIP_DEVICE_COMPILE__)
    const uint4* q = (const uint4*)p;
#pragma unroll
    for (int i = 0; i < N / 4; i++) {
      uint4 v = q[i];
      r.l[4 * i] = v.x; r.l[4 * i + 1] = v.y; r.l[4 * i + 2] = v.z; r.l[4 * i + 3] = v.w;
    }
#else
    __builtin_memcpy(r.l, p, BYTES);  // host pointers are only 8-byte aligned
#endif
    return r;
  }
  ARK_HD void store(void* p) const {
#if defined(__HIP_DEVICE_COMPILE__)
    uint4* q = (uint4*)p;
#pragma unroll
    for (int i = 0; i < N / 4; i++) q[i] = make_uint4(l[4 * i], l[4 * i + 1], l[4 * i + 2], l[4 * i + 3]);
#else
    __builtin_memcpy(p, l, BYTES);
#endif
  }
};

// ---- quadratic extension Fp2 = Fp[u]/(u^2 - BETA), BETA = -NEG_BETA (small) -------------------
// Replaces ff/src/fields/models/quadratic_extension.rs:268-320 (add/sub/neg/double), :626-670
// (mul via sum_of_products, square) and the curve configs' nonresidue helpers
// (curves/bls12_377/src/fields/fq2.rs:12-51: NONRESIDUE = -5; bls12_381 fq2.rs: -1).
// Any correct schedule gives the reference's bits (components are canonical Fp residues);
// this one is Karatsuba (3 base multiplications).
template <class P_, int NEG_BETA>
struct Fp2 {
  typedef Fp<P_> B;
  typedef P_ P;
  static constexpr int WORDS64 = 2 * B::WORDS64;
  static constexpr int BYTES = 2 * B::BYTES;
  static constexpr int LANES = 1;
  static constexpr int FULL_BYTES = BYTES;
  static constexpr bool FUSED_Y3 = false;
  B c0, c1;

  ARK_HD static Fp2 zero() { return Fp2{B::zero(), B::zero()}; }
  ARK_HD static Fp2 one() { return Fp2{B::one(), B::zero()}; }
  ARK_HD bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
  ARK_HD static bool eq(const Fp2& a, const Fp2& b) { return B::eq(a.c0, b.c0) && B::eq(a.c1, b.c1); }
  ARK_HD static Fp2 add(const Fp2& a, const Fp2& b) { return Fp2{B::add(a.c0, b.c0), B::add(a.c1, b.c1)}; }
  ARK_HD static Fp2 sub(const Fp2& a, const Fp2& b) { return Fp2{B::sub(a.c0, b.c0), B::sub(a.c1, b.c1)}; }
  ARK_HD static Fp2 dbl(const Fp2& a) { return Fp2{B::dbl(a.c0), B::dbl(a.c1)}; }
  ARK_HD static Fp2 neg(const Fp2& a) { return Fp2{B::neg(a.c0), B::neg(a.c1)}; }
  ARK_HD static Fp2 cond_neg(const Fp2& a, bool n) { return n ? neg(a) : a; }
  // conj(a) / (c0^2 - beta c1^2)      quadratic_extension.rs:322-339
  ARK_HD static Fp2 inverse(const Fp2& a) {
    B norm = B::add(B::sqr(a.c0), mul_neg_beta(B::sqr(a.c1)));
    B ni = B::inverse(norm);
    return Fp2{B::mul(a.c0, ni), B::neg(B::mul(a.c1, ni))};
  }
  // x * NEG_BETA for the small non-residues used here (1 or 5)
  ARK_HD static B mul_neg_beta(const B& x) {
    if constexpr (NEG_BETA == 1) return x;
    else if constexpr (NEG_BETA == 5) return B::add(B::dbl(B::dbl(x)), x);
    else { static_assert(NEG_BETA == 1 || NEG_BETA == 5, "unsupported nonresidue"); return x; }
  }
  // NEG_BETA * (p - x): an N-limb representative (<= NEG_BETA * p, not reduced) of beta * x, used only as an operand of
  // Fp::sop2
  ARK_HD static B neg_beta_times_neg(const B& x) {
    static_assert((u64)(1 + NEG_BETA) * ((u64)P::P[B::N - 1] + 1) < (1ull << 32), "NEG_BETA * p must fit N limbs");
    B n;
    u32 borrow = 0;
#pragma unroll
    for (int i = 0; i < B::N; i++) {
      u32 bo;
      n.l[i] = __builtin_subc((u32)P::P[i], x.l[i], borrow, &bo);
      borrow = bo;
    }
    if constexpr (NEG_BETA == 1) return n;
    B q;  // 4n + n
#pragma unroll
    for (int i = B::N - 1; i > 0; i--) q.l[i] = (n.l[i] << 2) | (n.l[i - 1] >> 30);
    q.l[0] = n.l[0] << 2;
    u32 c = 0;
#pragma unroll
    for (int i = 0; i < B::N; i++) {
      u32 co;
      q.l[i] = __builtin_addc(q.l[i], n.l[i], c, &co);
      c = co;
    }
    return q;
  }
  // (a0 + a1 u)(b0 + b1 u) = (a0 b0 + beta a1 b1) + (a0 b1 + a1 b0) u: two sums of two products, each under ONE
  // Montgomery reduction (quadratic_extension.rs:646-654 via Fp::sum_of_products, montgomery_backend.rs:415-516):
  // 4 limb products + 2 reductions = the multiply work of Karatsuba's 3 full products, without its 5 additions.
  ARK_HD static Fp2 mul(const Fp2& a, const Fp2& b) {
    Fp2 r;
    r.c0 = B::sop2_call(a.c0, b.c0, neg_beta_times_neg(a.c1), b.c1);
    r.c1 = B::sop2_call(a.c0, b.c1, a.c1, b.c0);
    return r;
  }
  ARK_HD static Fp2 mul_karatsuba(const Fp2& a, const Fp2& b) {
    B v0 = B::mul_call(a.c0, b.c0);
    B v1 = B::mul_call(a.c1, b.c1);
    B s = B::mul_call(B::add(a.c0, a.c1), B::add(b.c0, b.c1));
    Fp2 r;
    r.c1 = B::sub(B::sub(s, v0), v1);
    r.c0 = B::sub(v0, mul_neg_beta(v1));  // v0 + beta*v1
    return r;
  }
  ARK_HD static Fp2 sqr(const Fp2& a) {
    // (a0 + a1 u)^2 = (a0^2 + beta a1^2) + 2 a0 a1 u ; with t = a0*a1:
    // a0^2 + beta a1^2 = (a0 + a1)(a0 + beta a1) - (1 + beta) t
    B t = B::mul_call(a.c0, a.c1);
    B s = B::mul_call(B::add(a.c0, a.c1), B::sub(a.c0, mul_neg_beta(a.c1)));
    Fp2 r;
    r.c1 = B::dbl(t);
    // -(1+beta) t = (NEG_BETA - 1) t
    if constexpr (NEG_BETA == 1) r.c0 = s;
    else r.c0 = B::add(s, B::dbl(B::dbl(t)));  // NEG_BETA == 5: + 4t
    return r;
  }
  ARK_HD static Fp2 load(const void* p) {
    Fp2 r;
    r.c0 = B::load(p);
    r.c1 = B::load((const char*)p + B::BYTES);
    return r;
  }
  ARK_HD void store(void* p) const {
    c0.store(p);
    c1.store((char*)p + B::BYTES);
  }
};

// ---- Fp2 element spread over a LANE PAIR: the even lane holds c0, the odd lane c1 --------------------------------
// The bucket-accumulation kernel over Fp2 (G2) needs ~2x the registers of the G1 kernel when one lane owns a whole
// element (XYZZ accumulator = 96 VGPRs alone): one wave per SIMD, and at one wave per SIMD the multiplier pipeline
// runs at ~64% of its rate (profiles/r1_ubench_instruction_rates.txt, "mad+addc @1 waves/SIMD").  Splitting every
// element over two adjacent lanes halves the register need per lane -- the G1 kernel's budget, two waves per SIMD --
// for the same multiply work:   lane c0:  a0 b0 + (beta a1) b1      lane c1:  a0 b1 + a1 b0
// each ONE sum of two products under one Montgomery reduction (Fp::sop2_r); the partner's operands arrive through
// DPP quad_perm [1,0,3,2] (v_mov_dpp, full rate, no LDS).  Additions are component-wise.  Every branch condition is
// made pair-uniform (a zero test is the AND of both lanes' tests), so the two lanes never diverge.
// Values are "relaxed" residues (< 2p) as in the G1 kernel; `canonical()` folds them back below p.
template <class P_, int NEG_BETA>
struct Fp2Half {
  typedef Fp<P_> B;
  typedef P_ P;
  static constexpr int N = B::N;
  static constexpr int BYTES = B::BYTES;           // bytes this lane holds
  static constexpr int FULL_BYTES = 2 * B::BYTES;  // bytes of the whole element in memory (c0 | c1)
  static constexpr int LANES = 2;
  static constexpr int NEG_BETA_ = NEG_BETA;
  static constexpr bool FUSED_Y3 = true;
  B v;

  // (device-only type; the bodies are visible to the host pass as well because kernel templates are parsed there)
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
    for (int i = 0; i < N; i++) r.l[i] = swap1(x.l[i]);
    return r;
  }
  ARK_DEV static bool both(bool mine) { return mine && (swap1(mine ? 1u : 0u) != 0); }
  ARK_DEV static Fp2Half zero() { return Fp2Half{B::zero()}; }
  ARK_DEV static Fp2Half one() { return Fp2Half{odd() ? B::zero() : B::one()}; }
  ARK_DEV bool is_zero() const { return both(v.is_zero()); }
  ARK_DEV bool is_zero_mod_p() const { return both(v.is_zero_mod_p()); }
  ARK_DEV static Fp2Half add_r(const Fp2Half& a, const Fp2Half& b) { return Fp2Half{B::add_r(a.v, b.v)}; }
  ARK_DEV static Fp2Half sub_r(const Fp2Half& a, const Fp2Half& b) { return Fp2Half{B::sub_r(a.v, b.v)}; }
  ARK_DEV static Fp2Half dbl_r(const Fp2Half& a) { return Fp2Half{B::dbl_r(a.v)}; }
  ARK_DEV static Fp2Half neg(const Fp2Half& a) { return Fp2Half{B::neg(a.v)}; }  // canonical input (a base's y)
  ARK_DEV static Fp2Half cond_neg(const Fp2Half& a, bool n) { return n ? neg(a) : a; }
  ARK_DEV Fp2Half canonical() const { return Fp2Half{v.canonical()}; }
  // NEG_BETA * (2p - x) for a relaxed x: an N-limb representative (<= 2 NEG_BETA p) of beta * x, operand of sop2_r only
  ARK_DEV static B beta_times(const B& x) {
    static_assert((u64)(1 + 2 * NEG_BETA) * ((u64)P::P[N - 1] + 1) < (1ull << 32), "2 NEG_BETA p must fit N limbs");
    B n = B::neg_r(x);
    if constexpr (NEG_BETA == 1) return n;
    B q;  // 4n + n
#pragma unroll
    for (int i = N - 1; i > 0; i--) q.l[i] = (n.l[i] << 2) | (n.l[i - 1] >> 30);
    q.l[0] = n.l[0] << 2;
    u32 c = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
      u32 co;
      q.l[i] = __builtin_addc(q.l[i], n.l[i], c, &co);
      c = co;
    }
    return q;
  }
  // relaxed product: operands < 2p; the sum of products is < (4 + 4 NEG_BETA) p^2... (24 p^2 for beta = -5 over
  // BLS12-377, 8 p^2 for beta = -1 over BLS12-381): (sum + m p) / R < 2p for both fields (p/R = 0.0063, 0.102)
  ARK_DEV static Fp2Half mul_r(const Fp2Half& a, const Fp2Half& b) {
    static_assert((P::P[N - 1] >> 29) == 0, "sop2_r without a final subtraction needs 8p <= R");
    const B pa = partner(a.v), pb = partner(b.v);
    const bool o = odd();
    B X, Z;
    const B bz = beta_times(pa);
#pragma unroll
    for (int i = 0; i < N; i++) {
      X.l[i] = o ? pa.l[i] : a.v.l[i];   // a0
      Z.l[i] = o ? a.v.l[i] : bz.l[i];   // a1 (odd lane) / beta a1 (even lane)
    }
    return Fp2Half{B::sop2_r(X, b.v, Z, pb)};  // even: a0 b0 + beta a1 b1 ; odd: a0 b1 + a1 b0
  }
  ARK_DEV static Fp2Half neg_r(const Fp2Half& a) { return Fp2Half{B::neg_r(a.v)}; }
  // a*b + c*d over Fp2 (relaxed operands; c is typically a neg_r): FOUR base products per lane under one reduction,
  //   even: a0 b0 + beta a1 b1 + c0 d0 + beta c1 d1        odd: a0 b1 + a1 b0 + c0 d1 + c1 d0
  // bound (8 + 8 NEG_BETA) p^2: within R p over BLS12-377 (48 p < R); over BLS12-381 (16 p^2 > R p) one fold follows.
  ARK_DEV static Fp2Half sop2_r(const Fp2Half& a, const Fp2Half& b, const Fp2Half& c, const Fp2Half& d) {
    constexpr bool FOLD = ((u64)(8 + 8 * NEG_BETA) * ((u64)P::P[N - 1] + 1)) > (1ull << 32);
    static_assert(((u64)(8 + 8 * NEG_BETA) * ((u64)P::P[N - 1] + 1)) <= 3 * (1ull << 32), "sum of four products exceeds 3 R p");
    const B pa = partner(a.v), pb = partner(b.v), pc = partner(c.v), pd = partner(d.v);
    const B ba = beta_times(pa), bc = beta_times(pc);
    const bool o = odd();
    B X0, X1, X2, X3;
#pragma unroll
    for (int i = 0; i < N; i++) {
      X0.l[i] = o ? pa.l[i] : a.v.l[i];
      X1.l[i] = o ? a.v.l[i] : ba.l[i];
      X2.l[i] = o ? pc.l[i] : c.v.l[i];
      X3.l[i] = o ? c.v.l[i] : bc.l[i];
    }
    return Fp2Half{B::template sop4_r<FOLD>(X0, b.v, X1, pb, X2, d.v, X3, pd)};
  }
  // relaxed square by complex squaring: with t = a0 a1,  a0^2 + beta a1^2 = (a0 + a1)(a0 + beta a1) - (1 + beta) t.
  // even lane: s = (a0 + a1)(a0 + beta a1); odd lane: t = a0 a1 -- ONE plain product per lane (288 multiplies instead
  // of the 432 of mul_r(a, a)); then c0 = s + (NEG_BETA - 1) t, c1 = 2 t.
  ARK_DEV static Fp2Half sqr_r(const Fp2Half& a) {
    static_assert((u64)(4 + 4 * NEG_BETA) * ((u64)P::P[N - 1] + 1) < (1ull << 32), "2p * (2 + 2 NEG_BETA) p must stay below R p");
    const B pa = partner(a.v);
    const bool o = odd();
    const B sum = B::add_r(a.v, pa);   // a0 + a1 (used by the even lane)
    const B bz = beta_times(pa);       // even lane: NEG_BETA (2p - a1) = beta a1 (mod p), <= 2 NEG_BETA p
    B wide;                            // a0 + beta a1, unreduced (<= (2 + 2 NEG_BETA) p < 2^(32N)): a product operand only
    {
      u32 cy = 0;
#pragma unroll
      for (int i = 0; i < N; i++) {
        u32 co;
        wide.l[i] = __builtin_addc(a.v.l[i], bz.l[i], cy, &co);
        cy = co;
      }
    }
    B X, Y;
#pragma unroll
    for (int i = 0; i < N; i++) {
      X.l[i] = o ? pa.l[i] : sum.l[i];
      Y.l[i] = o ? a.v.l[i] : wide.l[i];
    }
    const B prod = B::mul_r(X, Y);     // even: s ; odd: t
    const B pt = partner(prod);        // even lane receives t
    B even;
    if constexpr (NEG_BETA == 1) even = prod;
    else even = B::add_r(prod, B::dbl_r(B::dbl_r(pt)));  // NEG_BETA == 5: s + 4 t
    const B oddv = B::dbl_r(prod);
    Fp2Half r;
#pragma unroll
    for (int i = 0; i < N; i++) r.v.l[i] = o ? oddv.l[i] : even.l[i];
    return r;
  }
  // canonical forms (the rare doubling branch and point conversions): relaxed arithmetic + one fold
  ARK_DEV static Fp2Half mul(const Fp2Half& a, const Fp2Half& b) { return mul_r(a, b).canonical(); }
  ARK_DEV static Fp2Half sqr(const Fp2Half& a) { return sqr_r(a).canonical(); }
  ARK_DEV static Fp2Half add(const Fp2Half& a, const Fp2Half& b) { return Fp2Half{B::add(a.v, b.v)}; }
  ARK_DEV static Fp2Half sub(const Fp2Half& a, const Fp2Half& b) { return Fp2Half{B::sub(a.v, b.v)}; }
  ARK_DEV static Fp2Half dbl(const Fp2Half& a) { return Fp2Half{B::dbl(a.v)}; }
  // this lane's component of the element stored at p (c0 | c1)
  ARK_DEV static Fp2Half load(const void* p) { return Fp2Half{B::load((const char*)p + (odd() ? B::BYTES : 0))}; }
  ARK_DEV void store(void* p) const { v.store((char*)p + (odd() ? B::BYTES : 0)); }
};

}  // namespace arkhip
