// Carry-free prime-field arithmetic on 28-bit limbs for the bucket-accumulation kernel of the MSM (Fp384 base fields).
//
// Why: gfx950's widest integer multiply-add is v_mad_u64_u32 (32x32+64 -> 64, carry-out only).  With saturated 32-bit
// limbs (fp.cuh) every partial product needs a v_addc_co_u32 behind it to catch the carry out of the 64-bit accumulator;
// that pair issues in ~10.8 cycles per wave at the accumulate kernel's occupancy (two waves per SIMD).  With 28-bit limbs
// a whole column of a product-scanning Montgomery multiplication -- up to 2 x 14 products of (< 2^30) x (< 2^28) -- fits
// the 64-bit accumulator, so the inner loop is v_mad_u64_u32 ONLY: 392 multiply-adds instead of 288 + 288 instructions.
// Measured on MI355X (profiles/r3_ubench_product_rate.txt, csrc/ubench/mulbench.hip): 62.1 against 52.8 G products/s at
// two waves per SIMD, 73.4 against 61.8 at eight.
//
// Values are Montgomery residues with radix R' = 2^(28 L) = 2^392 (L = 14), NOT fully reduced: R' exceeds p by a factor
// >= 2^11, so a product of operands below A p and B p comes out below (A B / 2048 + 1) p -- no conditional subtraction
// exists anywhere; sums are limb-wise (no carry chain); differences add a multiple of p first and end in one carry sweep.
// The bound of every intermediate of the mixed addition is written next to the code that produces it (ec28.cuh).
//
// Role in the reference: the same field, ff/src/fields/models/fp/montgomery_backend.rs:129-246 -- there with R = 2^(64 N)
// and canonical results.  This form never leaves the accumulate kernel: bases enter through a repacking of their
// canonical limbs shifted left by 28 L - 32 N = 8 bits (x R 2^8 = x R': the SAME residue in the new radix, merely not
// reduced -- below 256 p, which a multiplication operand may be) and buckets are stored in the reference's canonical
// form again, so everything downstream -- and every result -- is bit-identical.
#pragma once
#include "fp.cuh"

namespace arkhip {

// ---- 64-bit column accumulator, multiply-adds only (device) ----------------------------------------------------
// One column of a product as ONE dependency chain of v_mad_u64_u32 into a 64-bit accumulator that starts from the
// previous column's carry: no instruction joins partial sums (the compiler, left to itself, splits a column into two
// chains and adds them: +27 64-bit additions per product).  Up to 13 products per asm statement (30 operands); hipcc
// pads statement boundaries with wait states, so a column is at most three statements.
#define ARK_LM(A, B) "v_mad_u64_u32 %0, vcc, %" #A ", %" #B ", %0\n\t"
#define ARK_LT1 ARK_LM(1, 2)
#define ARK_LT2 ARK_LT1 ARK_LM(3, 4)
#define ARK_LT3 ARK_LT2 ARK_LM(5, 6)
#define ARK_LT4 ARK_LT3 ARK_LM(7, 8)
#define ARK_LT5 ARK_LT4 ARK_LM(9, 10)
#define ARK_LT6 ARK_LT5 ARK_LM(11, 12)
#define ARK_LT7 ARK_LT6 ARK_LM(13, 14)
#define ARK_LT8 ARK_LT7 ARK_LM(15, 16)
#define ARK_LT9 ARK_LT8 ARK_LM(17, 18)
#define ARK_LT10 ARK_LT9 ARK_LM(19, 20)
#define ARK_LT11 ARK_LT10 ARK_LM(21, 22)
#define ARK_LT12 ARK_LT11 ARK_LM(23, 24)
#define ARK_LT13 ARK_LT12 ARK_LM(25, 26)
#define ARK_LOP(i) "v"(x[LO + i]), "s"(PLZ<P, K - LO - i>::v)
#define ARK_LOP1 ARK_LOP(0)
#define ARK_LOP2 ARK_LOP1, ARK_LOP(1)
#define ARK_LOP3 ARK_LOP2, ARK_LOP(2)
#define ARK_LOP4 ARK_LOP3, ARK_LOP(3)
#define ARK_LOP5 ARK_LOP4, ARK_LOP(4)
#define ARK_LOP6 ARK_LOP5, ARK_LOP(5)
#define ARK_LOP7 ARK_LOP6, ARK_LOP(6)
#define ARK_LOP8 ARK_LOP7, ARK_LOP(7)
#define ARK_LOP9 ARK_LOP8, ARK_LOP(8)
#define ARK_LOP10 ARK_LOP9, ARK_LOP(9)
#define ARK_LOP11 ARK_LOP10, ARK_LOP(10)
#define ARK_LOP12 ARK_LOP11, ARK_LOP(11)
#define ARK_LOP13 ARK_LOP12, ARK_LOP(12)
#define ARK_LSTMT(n, OPS) \
  if constexpr (HIX - LO + 1 == n) asm(ARK_LT##n : "+v"(c) : OPS##n : "vcc")
#define ARK_LALL(OPS)                                                                                       \
  ARK_LSTMT(1, OPS); ARK_LSTMT(2, OPS); ARK_LSTMT(3, OPS); ARK_LSTMT(4, OPS); ARK_LSTMT(5, OPS); ARK_LSTMT(6, OPS); \
  ARK_LSTMT(7, OPS); ARK_LSTMT(8, OPS); ARK_LSTMT(9, OPS); ARK_LSTMT(10, OPS); ARK_LSTMT(11, OPS); ARK_LSTMT(12, OPS); \
  ARK_LSTMT(13, OPS)
// modulus limbs (28-bit form) as immediates -> SGPRs
template <class P, int I> struct PLZ { static constexpr u32 v = P::LZ_KP[1][I]; };
// c += sum_{i = LO..HIX} x[i] * y[K - i]
template <int LO, int HIX, int K>
ARK_DEV void lcol_vv(u64& c, const u32* x, const u32* y) {
  if constexpr (HIX >= LO) {
    if constexpr (HIX - LO + 1 > 13) {
      lcol_vv<LO, LO + 12, K>(c, x, y);
      lcol_vv<LO + 13, HIX, K>(c, x, y);
    } else {
      ARK_LALL(ARK_OV);
    }
  }
}
// c += sum_{i = LO..HIX} m[i] * p[K - i]
template <class P, int LO, int HIX, int K>
ARK_DEV void lcol_vp(u64& c, const u32* x) {
  if constexpr (HIX >= LO) {
    if constexpr (HIX - LO + 1 > 13) {
      lcol_vp<P, LO, LO + 12, K>(c, x);
      lcol_vp<P, LO + 13, HIX, K>(c, x);
    } else {
      ARK_LALL(ARK_LOP);
    }
  }
}

template <class P_>
struct FpL {
  typedef P_ P;
  static constexpr int L = P::LZ_L;   // limbs of W bits
  static constexpr int W = P::LZ_W;   // 28 (Fp384: 14 limbs) or 29 (Fp256: 9 limbs)
  static constexpr int N = P::N;      // 32-bit limbs of the canonical form
  static constexpr u32 MASK = (1u << W) - 1u;
  static_assert(W * L >= 32 * N, "the W-bit form must hold every N x 32-bit value");
  // may BOTH operands of a product be semi-normalised (limbs < 3 2^W)?  L x 9 2^(2W) + L x 2^(2W) < 2^64
  static constexpr bool SEMI2 = 10 * L < (1 << (64 - 2 * W));
  static_assert(4 * L < (1 << (64 - 2 * W)), "a normalised x semi-normalised column must fit 64 bits");
  u32 l[L];  // value = sum l[i] 2^(28 i); "normalised": every l[i] < 2^28

  ARK_HD static FpL zero() {
    FpL r;
#pragma unroll
    for (int i = 0; i < L; i++) r.l[i] = 0;
    return r;
  }
  // 2^(28 L) mod p: the residue of 1 (normalised, < p)
  ARK_HD static FpL one() { return unpack32(P::LZ_CIN); }

  // Montgomery product a b 2^(-28 L) mod p: normalised output below a b / 2^(28 L) + p.
  // Column bound: 14 products a_i b_j + 14 products m_i p_j + carry: with both operands semi-normalised (limbs < 3 2^28,
  // sub_semi below) 14 x 9 2^56 + 14 x 2^56 + 2^36 < 2^63.2; one normalised operand admits limbs up to 2^31 on the other.
  //
  // Two accumulator chains per column.  v_mad_u64_u32 into ONE accumulator is a serial dependency, and with the two
  // waves per SIMD the accumulate kernel's registers allow the multiplier's latency is exposed (product rate at 1 / 2 /
  // 4 / 8 waves per SIMD with a single chain: 53.8 / 62.8 / 70.0 / 74.3 G/s, profiles/r3_ubench_product_rate.txt).  The
  // operand part A_k = sum a_i b_(k-i) of a column depends on nothing but the inputs, the reduction part
  // B_k = carry + sum m_i p_(k-i) on the earlier columns: written as separate sums (one 64-bit addition joins them) the
  // scheduler interleaves column k's reduction chain with column k+1's operand chain.
  ARK_HD static FpL mul_c(const FpL& a, const FpL& b) {
    u32 m[L];
    FpL r;
    u64 carry = 0;
#pragma unroll
    for (int k = 0; k < L; k++) {
      u64 A = 0, B = carry;
#pragma unroll
      for (int i = 0; i <= k; i++) A += (u64)a.l[i] * b.l[k - i];
#pragma unroll
      for (int i = 0; i < k; i++) B += (u64)m[i] * P::LZ_KP[1][k - i];
      u64 t = A + B;
      m[k] = ((u32)t * P::LZ_INV) & MASK;
      t += (u64)m[k] * P::LZ_KP[1][0];
      carry = t >> W;
    }
#pragma unroll
    for (int k = L; k < 2 * L - 1; k++) {
      u64 A = 0, B = carry;
#pragma unroll
      for (int i = k - L + 1; i < L; i++) A += (u64)a.l[i] * b.l[k - i];
#pragma unroll
      for (int i = k - L + 1; i < L; i++) B += (u64)m[i] * P::LZ_KP[1][k - i];
      const u64 t = A + B;
      r.l[k - L] = (u32)t & MASK;
      carry = t >> W;
    }
    r.l[L - 1] = (u32)carry;
    return r;
  }
#if defined(__HIP_DEVICE_COMPILE__)
  // ---- device forms: every column ONE multiply-add chain (lcol_vv / lcol_vp above) -------------------------------
  // Product rate at the accumulate kernel's two waves per SIMD (profiles/r3_ubench_product_rate.txt): 75.6 G/s against
  // 69.3 for the compiler's scheduling of mul_c (which splits each column in two chains and joins them with a 64-bit
  // addition) and 58.7 for the saturated product.
  template <int K, int NP>
  ARK_DEV static void asm_cols_lo(u64& c, const u32* a, const u32* b, const u32* a2, const u32* b2, u32* m,
                                  const u32* a3 = nullptr, const u32* b3 = nullptr, const u32* a4 = nullptr,
                                  const u32* b4 = nullptr) {
    lcol_vv<0, K, K>(c, a, b);
    if constexpr (NP >= 2) lcol_vv<0, K, K>(c, a2, b2);
    if constexpr (NP >= 3) lcol_vv<0, K, K>(c, a3, b3);
    if constexpr (NP >= 4) lcol_vv<0, K, K>(c, a4, b4);
    lcol_vp<P, 0, K - 1, K>(c, m);
    m[K] = ((u32)c * P::LZ_INV) & MASK;
    lcol_vp<P, K, K, K>(c, m);
    c >>= W;
    if constexpr (K + 1 < L) asm_cols_lo<K + 1, NP>(c, a, b, a2, b2, m, a3, b3, a4, b4);
  }
  template <int K, int NP>
  ARK_DEV static void asm_cols_hi(u64& c, const u32* a, const u32* b, const u32* a2, const u32* b2, const u32* m, u32* r,
                                  const u32* a3 = nullptr, const u32* b3 = nullptr, const u32* a4 = nullptr,
                                  const u32* b4 = nullptr) {
    lcol_vv<K - L + 1, L - 1, K>(c, a, b);
    if constexpr (NP >= 2) lcol_vv<K - L + 1, L - 1, K>(c, a2, b2);
    if constexpr (NP >= 3) lcol_vv<K - L + 1, L - 1, K>(c, a3, b3);
    if constexpr (NP >= 4) lcol_vv<K - L + 1, L - 1, K>(c, a4, b4);
    lcol_vp<P, K - L + 1, L - 1, K>(c, m);
    r[K - L] = (u32)c & MASK;
    c >>= W;
    if constexpr (K + 1 < 2 * L - 1) asm_cols_hi<K + 1, NP>(c, a, b, a2, b2, m, r, a3, b3, a4, b4);
  }
  // squaring columns: cross products a_i * (2 a_(K-i)) for i < K - i, then the diagonal
  template <int K>
  ARK_DEV static void asm_sq_lo(u64& c, const u32* a, const u32* d, u32* m) {
    if constexpr (K >= 1) lcol_vv<0, (K - 1) / 2, K>(c, a, d);
    if constexpr (K % 2 == 0) lcol_vv<K / 2, K / 2, K>(c, a, a);
    lcol_vp<P, 0, K - 1, K>(c, m);
    m[K] = ((u32)c * P::LZ_INV) & MASK;
    lcol_vp<P, K, K, K>(c, m);
    c >>= W;
    if constexpr (K + 1 < L) asm_sq_lo<K + 1>(c, a, d, m);
  }
  template <int K>
  ARK_DEV static void asm_sq_hi(u64& c, const u32* a, const u32* d, const u32* m, u32* r) {
    lcol_vv<K - L + 1, (K - 1) / 2, K>(c, a, d);
    if constexpr (K % 2 == 0) lcol_vv<K / 2, K / 2, K>(c, a, a);
    lcol_vp<P, K - L + 1, L - 1, K>(c, m);
    r[K - L] = (u32)c & MASK;
    c >>= W;
    if constexpr (K + 1 < 2 * L - 1) asm_sq_hi<K + 1>(c, a, d, m, r);
  }
  ARK_DEV static FpL mul(const FpL& a, const FpL& b) {
    u32 m[L];
    FpL r;
    u64 c = 0;
    asm_cols_lo<0, 1>(c, a.l, b.l, nullptr, nullptr, m);
    asm_cols_hi<L, 1>(c, a.l, b.l, nullptr, nullptr, m, r.l);
    r.l[L - 1] = (u32)c;
    return r;
  }
  ARK_DEV static FpL sop2(const FpL& a, const FpL& b, const FpL& x, const FpL& y) {
    u32 m[L];
    FpL r;
    u64 c = 0;
    asm_cols_lo<0, 2>(c, a.l, b.l, x.l, y.l, m);
    asm_cols_hi<L, 2>(c, a.l, b.l, x.l, y.l, m, r.l);
    r.l[L - 1] = (u32)c;
    return r;
  }
  // a b + c d + e f + g h under ONE reduction (the Y3 of a bucket addition over Fp2: fp28x2.cuh).  The caller keeps the
  // column sum -- 4 L limb products + L reduction terms -- below 2^64 (operand classes stated at the call site).
  ARK_DEV static FpL sop4(const FpL& a, const FpL& b, const FpL& x, const FpL& y, const FpL& e, const FpL& f, const FpL& g,
                          const FpL& h) {
    u32 m[L];
    FpL r;
    u64 c = 0;
    asm_cols_lo<0, 4>(c, a.l, b.l, x.l, y.l, m, e.l, f.l, g.l, h.l);
    asm_cols_hi<L, 4>(c, a.l, b.l, x.l, y.l, m, r.l, e.l, f.l, g.l, h.l);
    r.l[L - 1] = (u32)c;
    return r;
  }
  ARK_DEV static FpL sqr(const FpL& a) {
    u32 m[L], d[L];
    FpL r;
#pragma unroll
    for (int i = 0; i < L; i++) d[i] = a.l[i] << 1;
    u64 c = 0;
    asm_sq_lo<0>(c, a.l, d, m);
    asm_sq_hi<L>(c, a.l, d, m, r.l);
    r.l[L - 1] = (u32)c;
    return r;
  }
#else
  // host (tests/lazy_host_check.hip; the window combine never uses this form): the portable statements
  static FpL mul(const FpL& a, const FpL& b) { return mul_c(a, b); }
  static FpL sqr(const FpL& a) { return sqr_c(a); }
  static FpL sop2(const FpL& a, const FpL& b, const FpL& x, const FpL& y) { return sop2_c(a, b, x, y); }
  static FpL sop4(const FpL& a, const FpL& b, const FpL& x, const FpL& y, const FpL& e, const FpL& f, const FpL& g,
                  const FpL& h) { return sop4_c(a, b, x, y, e, f, g, h); }
#endif
  ARK_HD static FpL sop4_c(const FpL& a, const FpL& b, const FpL& x, const FpL& y, const FpL& e, const FpL& f, const FpL& g,
                           const FpL& h) {
    u32 m[L];
    FpL r;
    u64 carry = 0;
#pragma unroll
    for (int k = 0; k < 2 * L - 1; k++) {
      u64 t = carry;
      const int lo = k < L ? 0 : k - L + 1, hi = k < L ? k : L - 1;
#pragma unroll
      for (int i = lo; i <= hi; i++)
        t += (u64)a.l[i] * b.l[k - i] + (u64)x.l[i] * y.l[k - i] + (u64)e.l[i] * f.l[k - i] + (u64)g.l[i] * h.l[k - i];
#pragma unroll
      for (int i = lo; i <= (k < L ? k - 1 : L - 1); i++) t += (u64)m[i] * P::LZ_KP[1][k - i];
      if (k < L) {
        m[k] = ((u32)t * P::LZ_INV) & MASK;
        t += (u64)m[k] * P::LZ_KP[1][0];
      } else {
        r.l[k - L] = (u32)t & MASK;
      }
      carry = t >> W;
    }
    r.l[L - 1] = (u32)carry;
    return r;
  }
  // a^2 of a normalised or semi-normalised a (limbs < 3 2^28): every cross product once, against the doubled limb (< 6 2^28;
  // per column at most 7 x 18 2^56 + 9 2^56 + 14 x 2^56 + carry < 2^63.3).  105 + 196 multiply-adds instead of 392.
  // (The saturated form's dedicated square lost to mul(a, a) on its doubling carries -- DESIGN 4; here doubling is a shift.)
  ARK_HD static FpL sqr_c(const FpL& a) {
    u32 m[L], d[L];
    FpL r;
#pragma unroll
    for (int i = 0; i < L; i++) d[i] = a.l[i] << 1;
    u64 carry = 0;
#pragma unroll
    for (int k = 0; k < L; k++) {
      u64 A = 0, B = carry;
#pragma unroll
      for (int i = 0; 2 * i < k; i++) A += (u64)a.l[i] * d[k - i];
      if (k % 2 == 0) A += (u64)a.l[k / 2] * a.l[k / 2];
#pragma unroll
      for (int i = 0; i < k; i++) B += (u64)m[i] * P::LZ_KP[1][k - i];
      u64 t = A + B;
      m[k] = ((u32)t * P::LZ_INV) & MASK;
      t += (u64)m[k] * P::LZ_KP[1][0];
      carry = t >> W;
    }
#pragma unroll
    for (int k = L; k < 2 * L - 1; k++) {
      u64 A = 0, B = carry;
#pragma unroll
      for (int i = k - L + 1; 2 * i < k; i++) A += (u64)a.l[i] * d[k - i];
      if (k % 2 == 0) A += (u64)a.l[k / 2] * a.l[k / 2];
#pragma unroll
      for (int i = k - L + 1; i < L; i++) B += (u64)m[i] * P::LZ_KP[1][k - i];
      const u64 t = A + B;
      r.l[k - L] = (u32)t & MASK;
      carry = t >> W;
    }
    r.l[L - 1] = (u32)carry;
    return r;
  }
  // a b + c d under ONE reduction (montgomery_backend.rs:415-516 sum_of_products, M = 2): the Y3 of every bucket
  // addition.  All four operands normalised (28 products + 14 reduction terms of < 2^56 per column: < 2^62);
  // output normalised, below (a b + c d) / 2^(28 L) + p.
  ARK_HD static FpL sop2_c(const FpL& a, const FpL& b, const FpL& c, const FpL& d) {
    u32 m[L];
    FpL r;
    u64 carry = 0;
#pragma unroll
    for (int k = 0; k < L; k++) {
      u64 A = 0, C = 0, B = carry;   // three independent chains (see mul)
#pragma unroll
      for (int i = 0; i <= k; i++) A += (u64)a.l[i] * b.l[k - i];
#pragma unroll
      for (int i = 0; i <= k; i++) C += (u64)c.l[i] * d.l[k - i];
#pragma unroll
      for (int i = 0; i < k; i++) B += (u64)m[i] * P::LZ_KP[1][k - i];
      u64 t = A + C + B;
      m[k] = ((u32)t * P::LZ_INV) & MASK;
      t += (u64)m[k] * P::LZ_KP[1][0];
      carry = t >> W;
    }
#pragma unroll
    for (int k = L; k < 2 * L - 1; k++) {
      u64 A = 0, C = 0, B = carry;
#pragma unroll
      for (int i = k - L + 1; i < L; i++) A += (u64)a.l[i] * b.l[k - i];
#pragma unroll
      for (int i = k - L + 1; i < L; i++) C += (u64)c.l[i] * d.l[k - i];
#pragma unroll
      for (int i = k - L + 1; i < L; i++) B += (u64)m[i] * P::LZ_KP[1][k - i];
      const u64 t = A + C + B;
      r.l[k - L] = (u32)t & MASK;
      carry = t >> W;
    }
    r.l[L - 1] = (u32)carry;
    return r;
  }

  // limb-wise sum, no carries (limbs grow by one bit): a valid `mul` operand beside a normalised one
  ARK_HD static FpL add_lazy(const FpL& a, const FpL& b) {
    FpL r;
#pragma unroll
    for (int i = 0; i < L; i++) r.l[i] = a.l[i] + b.l[i];
    return r;
  }
  // carry sweep over signed limbs; the value must lie in [0, 2^(28 L))
  ARK_HD static FpL normalise(const int* d) {
    FpL r;
    int carry = 0;
#pragma unroll
    for (int i = 0; i < L - 1; i++) {
      const int v = d[i] + carry;
      r.l[i] = (u32)v & MASK;
      carry = v >> W;  // arithmetic shift: floor division
    }
    r.l[L - 1] = (u32)(d[L - 1] + carry);
    return r;
  }
  // a - b + K p, normalised          (the caller guarantees a - b + K p >= 0; limbs of a, b below 2^30)
  template <int K>
  ARK_HD static FpL sub(const FpL& a, const FpL& b) {
    int d[L];
#pragma unroll
    for (int i = 0; i < L; i++) d[i] = (int)a.l[i] - (int)b.l[i] + (int)P::LZ_KP[K][i];
    return normalise(d);
  }
  // a - b - 2 c + K p
  template <int K>
  ARK_HD static FpL sub_b_2c(const FpL& a, const FpL& b, const FpL& c) {
    int d[L];
#pragma unroll
    for (int i = 0; i < L; i++) d[i] = (int)a.l[i] - (int)b.l[i] - 2 * (int)c.l[i] + (int)P::LZ_KP[K][i];
    return normalise(d);
  }
  // K p - a - b
  template <int K>
  ARK_HD static FpL negsub(const FpL& a, const FpL& b) {
    int d[L];
#pragma unroll
    for (int i = 0; i < L; i++) d[i] = (int)P::LZ_KP[K][i] - (int)a.l[i] - (int)b.l[i];
    return normalise(d);
  }
  // K p - a
  template <int K>
  ARK_HD static FpL neg(const FpL& a) {
    int d[L];
#pragma unroll
    for (int i = 0; i < L; i++) d[i] = (int)P::LZ_KP[K][i] - (int)a.l[i];
    return normalise(d);
  }
  // ---- differences WITHOUT a carry sweep ----------------------------------------------------------------------
  // K p written with limbs d_i in [H 2^28 - H, (H + 1) 2^28) (every limb lends H 2^28 to the one below it): for normalised
  // subtrahends a limb-wise  a_i - b_i + d_i  can then not go negative, so the difference needs no signed carry sweep at
  // all: 2 instructions per limb.  The result is "semi-normalised" -- limbs below (H + 2) 2^28 -- which a product accepts
  // on BOTH sides for H = 1 (14 x (3 2^28)^2 + 14 x 2^56 < 2^63.1).  The top limb lends nothing and must dominate the
  // subtrahend's: floor(K p / 2^364) - H >= floor(b / 2^364), true whenever K exceeds the subtrahend's bound by >= 1/2.
  template <int K, int H>
  static constexpr u32 kp_spread(int i) {
    return P::LZ_KP[K][i] + (i < L - 1 ? ((u32)H << W) : 0u) - (i > 0 ? (u32)H : 0u);
  }
  // a - b + K p, semi-normalised (a, b normalised)
  template <int K>
  ARK_HD static FpL sub_semi(const FpL& a, const FpL& b) {
    FpL r;
#pragma unroll
    for (int i = 0; i < L; i++) r.l[i] = a.l[i] - b.l[i] + kp_spread<K, 1>(i);
    return r;
  }
  // a - b + K p swept to normalised limbs (a's limbs below 3 2^W, b normalised): an unsigned sweep, the spread limbs of K p
  // cannot go negative
  template <int K>
  ARK_HD static FpL sub_sweep(const FpL& a, const FpL& b) {
    FpL r;
    u32 carry = 0;
#pragma unroll
    for (int i = 0; i < L - 1; i++) {
      const u32 v = a.l[i] - b.l[i] + kp_spread_any<K, 1>(i) + carry;
      r.l[i] = v & MASK;
      carry = v >> W;
    }
    r.l[L - 1] = a.l[L - 1] - b.l[L - 1] + kp_spread_any<K, 1>(L - 1) + carry;
    return r;
  }
  // a - b + K p as an operand that may meet ANOTHER semi-normalised operand in a product (a square of it, R t in Y3):
  // semi-normalised where the limb width leaves room for that (SEMI2: 14 x 28 bits), otherwise swept (9 x 29 bits)
  template <int K>
  ARK_HD static FpL sub_op(const FpL& a, const FpL& b) {
    if constexpr (SEMI2) return sub_semi<K>(a, b);
    else return sub_sweep<K>(a, b);
  }
  // limb i of k p for ANY k with k p < 2^(W L) (the LZ_KP table stops at 8): compile-time arithmetic on LZ_KP[1]
  static constexpr u32 kp_limb(int k, int i) {
    u64 carry = 0;
    u32 out = 0;
    for (int j = 0; j <= i; j++) {
      const u64 t = (u64)P::LZ_KP[1][j] * (u64)k + carry;
      out = j < L - 1 ? (u32)(t & MASK) : (u32)t;
      carry = t >> W;
    }
    return out;
  }
  // k p with limbs that each lend H 2^W to the one below (kp_spread for arbitrary k)
  template <int K, int H>
  static constexpr u32 kp_spread_any(int i) {
    return kp_limb(K, i) + (i < L - 1 ? ((u32)H << W) : 0u) - (i > 0 ? (u32)H : 0u);
  }
  // K p - a, limbs below 2^29 (a normalised)
  template <int K>
  ARK_HD static FpL neg_semi(const FpL& a) {
    FpL r;
#pragma unroll
    for (int i = 0; i < L; i++) r.l[i] = kp_spread<K, 1>(i) - a.l[i];
    return r;
  }
  // neg ? K p - a : a    (a normalised; limbs below 2^29)
  template <int K>
  ARK_HD static FpL cond_neg_semi(const FpL& a, bool neg) {
    FpL r;
#pragma unroll
    for (int i = 0; i < L; i++) r.l[i] = neg ? kp_spread<K, 1>(i) - a.l[i] : a.l[i];
    return r;
  }
  // a - b - 2 c + K p, NORMALISED (a, b, c normalised): limbs a_i - b_i - 2 c_i + d_i with d_i >= 3 2^28 - 3 stay
  // non-negative, so the carry sweep is unsigned
  template <int K>
  ARK_HD static FpL sub_b_2c_norm(const FpL& a, const FpL& b, const FpL& c) {
    FpL r;
    u32 carry = 0;
#pragma unroll
    for (int i = 0; i < L - 1; i++) {
      const u32 v = a.l[i] - b.l[i] - 2u * c.l[i] + kp_spread<K, 3>(i) + carry;
      r.l[i] = v & MASK;
      carry = v >> W;
    }
    r.l[L - 1] = a.l[L - 1] - b.l[L - 1] - 2u * c.l[L - 1] + kp_spread<K, 3>(L - 1) + carry;
    return r;
  }
  // is the normalised value, known to lie below 2 p, zero mod p (i.e. 0 or p)?  One-limb filter, exact compare behind it.
  ARK_HD bool is_zero_or_p() const {
    if (l[0] != 0u && l[0] != P::LZ_KP[1][0]) return false;
    u32 o0 = 0, o1 = 0;
#pragma unroll
    for (int i = 0; i < L; i++) {
      o0 |= l[i];
      o1 |= l[i] ^ P::LZ_KP[1][i];
    }
    return o0 == 0u || o1 == 0u;
  }
  // is the (normalised, < 9 p) value a multiple of p?  One-limb filter, exact compare behind it.
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

  // ---- the boundary with the canonical form (N x 32-bit limbs) ----
  // repacking only: the INTEGER is unchanged (what it means as a residue depends on the radix: v = x 2^(32 N) read with
  // radix 2^(28 L) is the residue x 2^-(28 L - 32 N) -- the curve isomorphism of ec28.cuh absorbs that factor)
  ARK_HD static FpL unpack32(const u32* in) {
    FpL r;
#pragma unroll
    for (int i = 0; i < L; i++) {
      const int bit = W * i;
      const int j = bit / 32, sh = bit % 32;
      u64 v = 0;
      if (j < N) v = (u64)in[j] >> sh;
      if (j + 1 < N) v |= (u64)in[j + 1] << (32 - sh);
      r.l[i] = (u32)v & MASK;
    }
    return r;
  }
  // the canonical limbs shifted left by SH = 28 L - 32 N bits: x R 2^SH = x R', i.e. the residue x itself in this form's
  // radix, unreduced (below 2^SH p; normalised limbs) -- good as ONE operand of a product: x y / R' < 2^SH y p / R'
  static constexpr int SH = W * L - 32 * N;
  ARK_HD static FpL unpack32_shl(const u32* in) {
    static_assert(SH >= 0 && SH < W, "shift within the lowest limb");
    FpL r;
#pragma unroll
    for (int i = 0; i < L; i++) {
      const int bit = W * i - SH;   // input bit that lands at this limb's bit 0
      if (bit < 0) {
        r.l[i] = (in[0] << SH) & MASK;
      } else {
        const int j = bit / 32, sh = bit % 32;
        u64 v = 0;
        if (j < N) v = (u64)in[j] >> sh;
        if (j + 1 < N) v |= (u64)in[j + 1] << (32 - sh);
        r.l[i] = (u32)v & MASK;
      }
    }
    return r;
  }
  // normalised limbs of a value below 2^(32 N) -> 32-bit limbs (no reduction)
  ARK_HD void pack32(u32* out) const {
#pragma unroll
    for (int j = 0; j < N; j++) {
      const int bit = 32 * j;
      const int i = bit / W, sh = bit % W;
      u64 v = (u64)l[i] >> sh;
      if (i + 1 < L) v |= (u64)l[i + 1] << (W - sh);
      if (i + 2 < L) v |= (u64)l[i + 2] << (2 * W - sh);
      out[j] = (u32)v;
    }
  }
  // v 2^-K mod p for a normalised v, K <= 24: one Montgomery step with radix 2^K.  Output normalised, below v / 2^K + p.
  template <int K>
  ARK_HD FpL shr_mod() const {
    static_assert(K >= 1 && K <= 24, "single-limb step");
    u32 m = (l[0] * P::LZ_INV) & ((1u << K) - 1u);   // v + m p = 0 mod 2^K
#if defined(__HIP_DEVICE_COMPILE__)
    // hipcc (ROCm 7.2) miscompiles  (u64)(x & 0xffffff) * c  for a constant c < 2^24: both operands qualify for the 24-bit
    // multiplier, whose demanded-bits rule drops the mask, and the product is then emitted as a full v_mad_u64_u32 on
    // the UNMASKED x (seen with K = 24 on the 256-bit fields: csrc/ubench/lazycheck.hip; with K <= 8, the only case the
    // accumulate kernels use, the product fits 32 bits and a genuine 24-bit multiply is emitted).  Hide the known bits.
    asm volatile("" : "+v"(m));
#endif
    u32 t[L + 1];
    u64 acc = 0;
#pragma unroll
    for (int i = 0; i < L; i++) {
      acc += (u64)l[i] + (u64)m * P::LZ_KP[1][i];
      t[i] = (u32)acc & MASK;
      acc >>= W;
    }
    t[L] = (u32)acc;
    FpL r;
#pragma unroll
    for (int i = 0; i < L; i++) r.l[i] = ((t[i] >> K) | (t[i + 1] << (W - K))) & MASK;
    r.l[L - 1] = (t[L - 1] >> K) | (t[L] << (W - K));  // top limb keeps whatever is left (value < 2^(28 L))
    return r;
  }
  // normalised value below 2 p (and below 2^(32 N)) -> canonical limbs in [0, p)
  ARK_HD Fp<P> to_canonical_bits() const {
    u32 w[N];
    pack32(w);
    return Fp<P>::reduce_once(w);
  }
};

}  // namespace arkhip
