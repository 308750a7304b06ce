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

template <class P_>
struct FpL {
  typedef P_ P;
  static constexpr int L = P::LZ_L;   // 28-bit limbs
  static constexpr int N = P::N;      // 32-bit limbs of the canonical form
  static constexpr u32 MASK = (1u << 28) - 1u;
  static_assert(28 * L >= 32 * N, "the 28-bit form must hold every N x 32-bit value");
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
  ARK_HD static FpL mul(const FpL& a, const FpL& b) {
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
      carry = t >> 28;
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
      carry = t >> 28;
    }
    r.l[L - 1] = (u32)carry;
    return r;
  }
  // the single-chain form (kept for the microbenchmark's A/B: csrc/ubench/mulbench.hip)
  ARK_HD static FpL mul_chain1(const FpL& a, const FpL& b) {
    u32 m[L];
    FpL r;
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
  // a^2 of a normalised or semi-normalised a (limbs < 3 2^28): every cross product once, against the doubled limb (< 6 2^28;
  // per column at most 7 x 18 2^56 + 9 2^56 + 14 x 2^56 + carry < 2^63.3).  105 + 196 multiply-adds instead of 392.
  // (The saturated form's dedicated square lost to mul(a, a) on its doubling carries -- DESIGN 4; here doubling is a shift.)
  ARK_HD static FpL sqr(const FpL& a) {
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
      carry = t >> 28;
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
      carry = t >> 28;
    }
    r.l[L - 1] = (u32)carry;
    return r;
  }
  // a b + c d under ONE reduction (montgomery_backend.rs:415-516 sum_of_products, M = 2): the Y3 of every bucket
  // addition.  All four operands normalised (28 products + 14 reduction terms of < 2^56 per column: < 2^62);
  // output normalised, below (a b + c d) / 2^(28 L) + p.
  ARK_HD static FpL sop2(const FpL& a, const FpL& b, const FpL& c, const FpL& d) {
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
      carry = t >> 28;
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
      carry = t >> 28;
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
      carry = v >> 28;  // arithmetic shift: floor division
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
    return P::LZ_KP[K][i] + (i < L - 1 ? ((u32)H << 28) : 0u) - (i > 0 ? (u32)H : 0u);
  }
  // a - b + K p, semi-normalised (a, b normalised)
  template <int K>
  ARK_HD static FpL sub_semi(const FpL& a, const FpL& b) {
    FpL r;
#pragma unroll
    for (int i = 0; i < L; i++) r.l[i] = a.l[i] - b.l[i] + kp_spread<K, 1>(i);
    return r;
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
      carry = v >> 28;
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
      const int bit = 28 * i;
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
  static constexpr int SH = 28 * L - 32 * N;
  ARK_HD static FpL unpack32_shl(const u32* in) {
    static_assert(SH >= 0 && SH < 28, "shift within the lowest limb");
    FpL r;
#pragma unroll
    for (int i = 0; i < L; i++) {
      const int bit = 28 * i - SH;   // input bit that lands at this limb's bit 0
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
      const int i = bit / 28, sh = bit % 28;
      u64 v = (u64)l[i] >> sh;
      if (i + 1 < L) v |= (u64)l[i + 1] << (28 - sh);
      if (i + 2 < L) v |= (u64)l[i + 2] << (56 - sh);
      out[j] = (u32)v;
    }
  }
  // v 2^-K mod p for a normalised v, K <= 24: one Montgomery step with radix 2^K.  Output normalised, below v / 2^K + p.
  template <int K>
  ARK_HD FpL shr_mod() const {
    static_assert(K >= 1 && K <= 24, "single-limb step");
    const u32 m = (l[0] * P::LZ_INV) & ((1u << K) - 1u);   // v + m p = 0 mod 2^K
    u32 t[L + 1];
    u64 acc = 0;
#pragma unroll
    for (int i = 0; i < L; i++) {
      acc += (u64)l[i] + (u64)m * P::LZ_KP[1][i];
      t[i] = (u32)acc & MASK;
      acc >>= 28;
    }
    t[L] = (u32)acc;
    FpL r;
#pragma unroll
    for (int i = 0; i < L; i++) r.l[i] = ((t[i] >> K) | (t[i + 1] << (28 - K))) & MASK;
    r.l[L - 1] = (t[L - 1] >> K) | (t[L] << (28 - K));  // top limb keeps whatever is left (value < 2^(28 L))
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
