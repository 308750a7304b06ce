// Radix-2 FFT / IFFT over a scalar field on one MI355X, in place, natural order in and out.
//
// Replaces the reference's Radix2EvaluationDomain kernels:
//   poly/src/domain/radix2/mod.rs:140-153 (fft_in_place / ifft_in_place dispatch),
//   poly/src/domain/radix2/fft.rs:74-119 (in_order_fft/ifft_in_place, coset scaling),
//   :124-187 (roots_of_unity table), :190-210 (butterflies), :252-349 (io_helper/oi_helper),
//   :373-380 (derange), poly/src/domain/mod.rs:115-148 (distribute_powers).
// Contract: forward X[j] = sum_i x[i] (h g^j)^i ; inverse
// x[i] = n^-1 h^-i sum_j X[j] g^-ij ; all values canonical Montgomery residues, so any correct
// schedule is bit-identical to the reference.
//
// GPU organisation: decimation-in-frequency with the reference's own twiddle convention
// (stage with half-width `gap` uses roots[j * n/(2 gap)]), executed as P = ceil(k/8) passes over
// HBM; each pass runs up to 8 consecutive stages on a tile held in LDS.  A tile is
// 2^kp butterfly rows x 8 adjacent columns (256-byte contiguous segments in HBM), the last pass
// takes its 8 "columns" from the top index bits so that the bit-reversed write-back -- which
// replaces the reference's serial derange() -- is also made of 256-byte segments.  Coset
// pre-scaling (x[i] *= h^i) is fused into the first pass's loads, and the inverse transform's
// n^-1 h^-i scaling into the last pass's stores (two-level power tables).
#pragma once
#include <hip/hip_runtime.h>
#include <map>
#include <string.h>
#include <array>
#include <mutex>
#include "msm.cuh"  // DevBuf, ARK_HIP_TRY

namespace arkhip {

static constexpr int FFT_LANE_BITS = 2;    // 4 adjacent elements = 128 B segments (tile 2^(8+2) x 32 B = 32 KiB LDS: 4 workgroups per CU)
static constexpr int FFT_MAX_KP = 8;       // stages per pass
static constexpr int FFT_SINGLE_MAX = 10;  // whole transform in one workgroup up to 2^10
static constexpr int PW_LO_BITS = 10;      // two-level power tables: h^i = hi[i >> 10] * lo[i & 1023]

__device__ __forceinline__ u32 bitrev32(u32 x, int bits) { return bits == 0 ? 0u : (__brev(x) >> (32 - bits)); }

// out[j] = base^(j * stride), j < count   (binary exponentiation; tables are tiny or built once)
template <class FP>
__global__ void __launch_bounds__(256) fft_pow_table_kernel(const u32* __restrict__ base, u64 stride, u32 count,
                                                            const u32* __restrict__ mulby, u32* __restrict__ out) {
  typedef Fp<FP> F;
  u32 j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= count) return;
  F b = F::load(base);
  F r = mulby ? F::load(mulby) : F::one();
  u64 e = (u64)j * stride;
  while (e) {
    if (e & 1) r = F::mul(r, b);
    b = F::sqr(b);
    e >>= 1;
  }
  r.store(out + (size_t)j * F::N);
}
// the same with base and multiplier passed BY VALUE (kernel arguments: nothing to stage, nothing to wait for)
struct FftElemArg { u32 w[8]; };   // one element of a served scalar field: 4 x 64 bits
template <class FP>
__global__ void __launch_bounds__(256) fft_pow_table_val_kernel(FftElemArg base, FftElemArg mulby, int has_mul, u32 count,
                                                                u32* __restrict__ out) {
  typedef Fp<FP> F;
  static_assert(F::N == 8, "FftElemArg holds 8 words");
  u32 j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= count) return;
  F b = F::load(base.w);
  F r = has_mul ? F::load(mulby.w) : F::one();
  u32 e = j;
  while (e) {
    if (e & 1) r = F::mul(r, b);
    b = F::sqr(b);
    e >>= 1;
  }
  r.store(out + (size_t)j * F::N);
}
// roots[j] = hi[j >> LO] * lo[j & mask]
template <class FP>
__global__ void __launch_bounds__(256) fft_expand_table_kernel(const u32* __restrict__ lo, const u32* __restrict__ hi,
                                                               int lo_bits, size_t count, u32* __restrict__ out) {
  typedef Fp<FP> F;
  size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= count) return;
  F a = F::load(lo + (j & ((1u << lo_bits) - 1)) * F::N);
  F b = F::load(hi + (j >> lo_bits) * F::N);
  F::mul(a, b).store(out + j * F::N);
}

struct FftPassArgs {
  int k;       // log2 n
  int s0;      // first global stage of this pass
  int kp;      // stages in this pass
  int t;       // log2 lanes
  int last;    // last pass: lanes = top bits, bit-reversed write-back
  int zskip;   // first executed pass of a degree-aware transform (fft.rs:29-71): the input holds only the first
               // n >> zskip coefficients, the first zskip stages are not run -- their effect on a zero-padded input,
               // x'[pos] = x[pos mod d] * w^((pos mod d) * bitrev_z(pos / d)), d = n >> zskip, is applied by the loads
  const u32* roots;    // w^j, j < n/2; compact: followed by the table of every later stage (below)
  int compact;         // 1: roots holds T_0 | T_1 | ... with T_s[i] = w^(i << s), i < n >> (s + 1) -- T_s starts at entry
                       // n - (n >> s): a stage-s twiddle is T_s[index], not roots[index << s] (round 5)
  const u32* pre_lo;   // first pass: x[pos] *= pre_hi[pos >> 10] * pre_lo[pos & 1023]   (nullable)
  const u32* pre_hi;
  const u32* post_lo;  // last pass: out[pos] *= post_hi[pos >> 10] * post_lo[pos & 1023]  (nullable)
  const u32* post_hi;
  const u32* post_const;  // last pass: out *= const (nullable; used when post tables are absent)
  const u32* pre_full;    // round 5: the expanded tables h^pos (pre) / c h^-pos (post), one entry per position (nullable): one
  const u32* post_full;   // product per element instead of two, for 32 more bytes read -- the saturated kernel only
};

// element <-> two 16-byte LDS planes
template <class F>
__device__ __forceinline__ F fft_unpack(const uint4& a, const uint4& b) {
  F x;
  x.l[0] = a.x; x.l[1] = a.y; x.l[2] = a.z; x.l[3] = a.w;
  x.l[4] = b.x; x.l[5] = b.y; x.l[6] = b.z; x.l[7] = b.w;
  return x;
}
template <class F>
__device__ __forceinline__ void fft_pack(const F& x, uint4& a, uint4& b) {
  a = make_uint4(x.l[0], x.l[1], x.l[2], x.l[3]);
  b = make_uint4(x.l[4], x.l[5], x.l[6], x.l[7]);
}

static constexpr int FFT_THREADS = 256;
static constexpr int FFT_MAX_EPT = 4;  // elements per lane: tiles of up to 1024 elements (32 KiB of LDS)

// One pass = load tile -> kp butterfly stages in LDS -> store tile.  Loops over the lane's elements /
// butterflies are fully unrolled with predication so that all global loads of a phase (tile rows,
// twiddles of a stage) are in flight together before the first dependent use.
#ifndef ARK_FFT_MIN_WAVES
#define ARK_FFT_MIN_WAVES 1   // A/B builds: 5 caps the kernel at 96 registers (27 spilled) for a fifth workgroup per CU
#endif
// TH: threads per workgroup = a quarter of the largest tile (256: 1024-element tiles, 32 KiB of LDS, four workgroups per CU;
// 512 / 1024: 2048- / 4096-element tiles of the round-6 plans with fewer passes -- 64 / 128 KiB, two / one per CU)
template <class FP, int TH = FFT_THREADS>
__global__ void __launch_bounds__(TH, TH == FFT_THREADS ? ARK_FFT_MIN_WAVES : 1) fft_pass_kernel(const u32* __restrict__ src, u32* __restrict__ dst,
                                                               FftPassArgs a) {
  typedef Fp<FP> F;
  extern __shared__ uint4 lds[];
  const int kp = a.kp, t = a.t, k = a.k;
  const u32 E = 1u << (kp + t);
  uint4* pl0 = lds;
  uint4* pl1 = lds + E;
  const u32 tile = blockIdx.x;
  const int lo_shift = k - a.s0 - kp;
  const u32 T1 = (1u << t) - 1u;
  const u32 Q1 = (1u << kp) - 1u;
  const u32 tid = threadIdx.x;
  u32 mid = 0, hi_bits = 0;
  if (!a.last) {
    mid = tile & ((1u << (lo_shift - t)) - 1u);
    hi_bits = tile >> (lo_shift - t);
  }
  // ---- load tile: issue every global load, then fill LDS ----
  {
    uint4 v0[FFT_MAX_EPT], v1[FFT_MAX_EPT];
    size_t ps[FFT_MAX_EPT];
#pragma unroll
    for (int it = 0; it < FFT_MAX_EPT; it++) {
      u32 e = tid + it * TH;
      if (e < E) {
        size_t pos;
        if (!a.last) {
          u32 q = e >> t, r = e & T1;
          pos = ((size_t)hi_bits << (k - a.s0)) | ((size_t)q << lo_shift) | ((size_t)mid << t) | r;
        } else {
          u32 r = e >> kp, q = e & Q1;
          pos = ((size_t)r << (k - t)) | ((size_t)tile << kp) | q;
        }
        ps[it] = pos;
        const size_t spos = a.zskip ? (pos & (((size_t)1 << (k - a.zskip)) - 1)) : pos;
        const uint4* g = (const uint4*)(src + spos * F::N);
        v0[it] = g[0];
        v1[it] = g[1];
      }
    }
#pragma unroll
    for (int it = 0; it < FFT_MAX_EPT; it++) {
      u32 e = tid + it * TH;
      if (e < E) {
        const size_t pos = ps[it];
        const size_t spos = a.zskip ? (pos & (((size_t)1 << (k - a.zskip)) - 1)) : pos;
        if (a.pre_lo) {
          F x = fft_unpack<F>(v0[it], v1[it]);
          F pw;
          if (a.pre_full)
            pw = F::load(a.pre_full + spos * F::N);
          else
            pw = F::mul(F::load(a.pre_hi + (spos >> PW_LO_BITS) * F::N),
                        F::load(a.pre_lo + (spos & ((1u << PW_LO_BITS) - 1)) * F::N));
          x = F::mul(x, pw);
          fft_pack<F>(x, v0[it], v1[it]);
        }
        if (a.zskip) {
          const u32 blk = (u32)(pos >> (k - a.zskip));
          const size_t ex = (spos * (size_t)bitrev32(blk, a.zskip)) & (((size_t)1 << k) - 1);
          if (ex != 0) {
            const size_t half = (size_t)1 << (k - 1);
            F x = fft_unpack<F>(v0[it], v1[it]);
            F w = F::load(a.roots + (ex & (half - 1)) * F::N);
            x = F::mul(x, w);
            if (ex >= half) x = F::neg(x);  // w^(n/2) = -1
            fft_pack<F>(x, v0[it], v1[it]);
          }
        }
        pl0[e] = v0[it];
        pl1[e] = v1[it];
      }
    }
  }
  __syncthreads();
  // ---- kp butterfly stages in LDS, two at a time ----
  // A lane owns a 4-point group {q0, q0+lg/2, q0+lg, q0+3lg/2} of two consecutive stages (gaps lg, lg/2): one LDS
  // round trip and one barrier per TWO stages, three twiddles per four butterflies (both butterflies of the second
  // stage share theirs).  An odd stage count ends with a single radix-2 stage.
  int ls = 0;
  for (; ls + 1 < kp; ls += 2) {
    const u32 lg = 1u << (kp - 1 - ls), qt = lg >> 1;
    const int s = a.s0 + ls;
    // where stage s and stage s + 1 find their twiddles: the stage's own compact table, or the full table strided
    const size_t nn = (size_t)1 << k;
    const size_t sbase0 = a.compact ? nn - (nn >> s) : 0, sbase1 = a.compact ? nn - (nn >> (s + 1)) : 0;
    const int ssh0 = a.compact ? 0 : s, ssh1 = a.compact ? 0 : s + 1;
    u32 idx[FFT_MAX_EPT / 4][4];
    uint4 wa0[FFT_MAX_EPT / 4][2], wa1[FFT_MAX_EPT / 4][2], wb[FFT_MAX_EPT / 4][2];
    bool ha0[FFT_MAX_EPT / 4], hb[FFT_MAX_EPT / 4];
#pragma unroll
    for (int it = 0; it < FFT_MAX_EPT / 4; it++) {
      const u32 g = tid + it * TH;
      ha0[it] = hb[it] = false;
      if (g < E / 4) {
        u32 gq, r;
        if (!a.last) { r = g & T1; gq = g >> t; }
        else { gq = g & ((1u << (kp - 2)) - 1u); r = g >> (kp - 2); }
        const u32 j0 = gq & (qt - 1u);
        const u32 q0 = ((gq & ~(qt - 1u)) << 2) | j0;
        size_t ta0, ta1, tb;   // twiddle exponents >> stage: entries of the stage's compact table
        if (!a.last) {
          const size_t col = ((size_t)mid << t) | r;
#pragma unroll
          for (int m = 0; m < 4; m++) idx[it][m] = ((q0 + m * qt) << t) | r;
          ta0 = (((size_t)j0) << lo_shift) | col;
          ta1 = (((size_t)(j0 + qt)) << lo_shift) | col;
          tb = ta0;
        } else {
#pragma unroll
          for (int m = 0; m < 4; m++) idx[it][m] = (r << kp) | (q0 + m * qt);
          ta0 = (size_t)j0;
          ta1 = (size_t)(j0 + qt);
          tb = ta0;
        }
        const uint4* g1 = (const uint4*)(a.roots + (sbase0 + (ta1 << ssh0)) * F::N);  // never the trivial twiddle: j0 + lg/2 > 0
        wa1[it][0] = g1[0];
        wa1[it][1] = g1[1];
        if (ta0 != 0) {
          const uint4* g0 = (const uint4*)(a.roots + (sbase0 + (ta0 << ssh0)) * F::N);
          wa0[it][0] = g0[0];
          wa0[it][1] = g0[1];
          ha0[it] = true;
        }
        if (tb != 0) {
          const uint4* g2 = (const uint4*)(a.roots + (sbase1 + (tb << ssh1)) * F::N);
          wb[it][0] = g2[0];
          wb[it][1] = g2[1];
          hb[it] = true;
        }
      }
    }
#pragma unroll
    for (int it = 0; it < FFT_MAX_EPT / 4; it++) {
      const u32 g = tid + it * TH;
      if (g < E / 4) {
        F x0 = fft_unpack<F>(pl0[idx[it][0]], pl1[idx[it][0]]);
        F x1 = fft_unpack<F>(pl0[idx[it][1]], pl1[idx[it][1]]);
        F x2 = fft_unpack<F>(pl0[idx[it][2]], pl1[idx[it][2]]);
        F x3 = fft_unpack<F>(pl0[idx[it][3]], pl1[idx[it][3]]);
        // first stage, gap lg: (x0, x2) and (x1, x3)                    fft.rs:190-198 butterfly_fn_io
        // (tile values are relaxed residues in [0, 2p): the products skip their final conditional subtraction, the
        //  last pass's stores make the results canonical)
        F s0 = F::add_r2(x0, x2), d0 = F::sub_r(x0, x2);
        if (ha0[it]) d0 = F::mul_r1(d0, fft_unpack<F>(wa0[it][0], wa0[it][1]));
        F s1 = F::add_r2(x1, x3), d1 = F::mul_r1(F::sub_r(x1, x3), fft_unpack<F>(wa1[it][0], wa1[it][1]));
        // second stage, gap lg/2: (s0, s1) and (d0, d1), one twiddle for both
        F y0 = F::add_r2(s0, s1), y1 = F::sub_r(s0, s1);
        F y2 = F::add_r2(d0, d1), y3 = F::sub_r(d0, d1);
        if (hb[it]) {
          const F w = fft_unpack<F>(wb[it][0], wb[it][1]);
          y1 = F::mul_r1(y1, w);
          y3 = F::mul_r1(y3, w);
        }
        uint4 o0, o1;
        fft_pack<F>(y0, o0, o1); pl0[idx[it][0]] = o0; pl1[idx[it][0]] = o1;
        fft_pack<F>(y1, o0, o1); pl0[idx[it][1]] = o0; pl1[idx[it][1]] = o1;
        fft_pack<F>(y2, o0, o1); pl0[idx[it][2]] = o0; pl1[idx[it][2]] = o1;
        fft_pack<F>(y3, o0, o1); pl0[idx[it][3]] = o0; pl1[idx[it][3]] = o1;
      }
    }
    __syncthreads();
  }
  for (; ls < kp; ls++) {
    const u32 lg = 1u << (kp - 1 - ls);
    const int s = a.s0 + ls;
    const size_t sbase = a.compact ? ((size_t)1 << k) - (((size_t)1 << k) >> s) : 0;
    const int ssh = a.compact ? 0 : s;
    u32 i0s[FFT_MAX_EPT / 2], i1s[FFT_MAX_EPT / 2];
    uint4 w0[FFT_MAX_EPT / 2], w1[FFT_MAX_EPT / 2];
    bool hasw[FFT_MAX_EPT / 2];
#pragma unroll
    for (int it = 0; it < FFT_MAX_EPT / 2; it++) {
      u32 b = tid + it * TH;
      hasw[it] = false;
      if (b < E / 2) {
        u32 i0, i1;
        size_t tw;
        if (!a.last) {
          u32 r = b & T1, qq = b >> t;
          u32 q = ((qq & ~(lg - 1)) << 1) | (qq & (lg - 1));
          i0 = (q << t) | r;
          i1 = i0 + (lg << t);
          tw = (((size_t)(q & (lg - 1))) << lo_shift) | ((size_t)mid << t) | r;
        } else {
          u32 qq = b & ((1u << (kp - 1)) - 1u), r = b >> (kp - 1);
          u32 q = ((qq & ~(lg - 1)) << 1) | (qq & (lg - 1));
          i0 = (r << kp) | q;
          i1 = i0 + lg;
          tw = (size_t)(q & (lg - 1));
        }
        i0s[it] = i0;
        i1s[it] = i1;
        if (tw != 0) {  // twiddle loads of the whole stage go out before any butterfly waits on one
          const uint4* g = (const uint4*)(a.roots + (sbase + (tw << ssh)) * F::N);
          w0[it] = g[0];
          w1[it] = g[1];
          hasw[it] = true;
        }
      }
    }
#pragma unroll
    for (int it = 0; it < FFT_MAX_EPT / 2; it++) {
      u32 b = tid + it * TH;
      if (b < E / 2) {
        const u32 i0 = i0s[it], i1 = i1s[it];
        F lo = fft_unpack<F>(pl0[i0], pl1[i0]);
        F hi = fft_unpack<F>(pl0[i1], pl1[i1]);
        F sum = F::add_r2(lo, hi);  // fft.rs:190-198 butterfly_fn_io, on relaxed residues
        F dif = F::sub_r(lo, hi);
        if (hasw[it]) dif = F::mul_r1(dif, fft_unpack<F>(w0[it], w1[it]));
        uint4 o0, o1;
        fft_pack<F>(sum, o0, o1);
        pl0[i0] = o0;
        pl1[i0] = o1;
        fft_pack<F>(dif, o0, o1);
        pl0[i1] = o0;
        pl1[i1] = o1;
      }
    }
    __syncthreads();
  }
  // ---- store tile ----
  if (!a.last) {
#pragma unroll
    for (int it = 0; it < FFT_MAX_EPT; it++) {
      u32 e = tid + it * TH;
      if (e < E) {
        u32 q = e >> t, r = e & T1;
        size_t pos = ((size_t)hi_bits << (k - a.s0)) | ((size_t)q << lo_shift) | ((size_t)mid << t) | r;
        uint4* g = (uint4*)(dst + pos * F::N);
        g[0] = pl0[e];
        g[1] = pl1[e];
      }
    }
  } else {
    // position p = r<<(k-t) | tile<<kp | q holds X[bitrev_k(p)]  (replaces derange(), fft.rs:373-380)
    const int tb = k - kp - t;
    const size_t tile_rev = bitrev32(tile, tb);
#pragma unroll
    for (int it = 0; it < FFT_MAX_EPT; it++) {
      u32 e = tid + it * TH;
      if (e < E) {
        u32 r2 = e & T1, q2 = e >> t;  // output-side coordinates
        size_t opos = ((size_t)q2 << (k - kp)) | (tile_rev << t) | r2;
        u32 i = (bitrev32(r2, t) << kp) | bitrev32(q2, kp);
        uint4 v0 = pl0[i], v1 = pl1[i];
        if (a.post_lo || a.post_const) {
          F x = fft_unpack<F>(v0, v1);
          F pw;
          if (a.post_full)
            pw = F::load(a.post_full + opos * F::N);
          else if (a.post_lo)
            pw = F::mul(F::load(a.post_hi + (opos >> PW_LO_BITS) * F::N),
                        F::load(a.post_lo + (opos & ((1u << PW_LO_BITS) - 1)) * F::N));
          else
            pw = F::load(a.post_const);
          x = F::mul(x, pw);  // relaxed x times canonical pw: mul's one conditional subtraction makes it canonical
          fft_pack<F>(x, v0, v1);
        } else {
          F x = fft_unpack<F>(v0, v1);
          x = F::reduce_once(x.l);  // [0, 2p) -> [0, p)
          fft_pack<F>(x, v0, v1);
        }
        uint4* g = (uint4*)(dst + opos * F::N);
        g[0] = v0;
        g[1] = v1;
      }
    }
  }
}

// ---- the same pass on carry-free 9 x 29-bit limbs (fp28.cuh with W = 29) ------------------------------------------
// Why: the saturated Fr product is 128 v_mad_u64_u32 + 128 v_addc_co_u32 (+ moves) = 295 vector instructions; on 9 limbs
// of 29 bits a column of 9 + 9 products fits the 64-bit accumulator: 162 multiply-adds + 44 others = 206
// (profiles/r4_ubench_product_rate_29bit.txt: 155 against 123 G products/s at two waves per SIMD, 177 against 134 at
// eight).  Sums are limb-wise additions, differences add a spread multiple of p (no borrow chain), and only what is never
// multiplied -- the sum outputs -- is swept.
//
// Representation.  DATA keeps the reference's Montgomery residues (x R, R = 2^256) as plain integers in 9 x 29-bit limbs;
// TWIDDLES and coset powers are stored canonical (8 x 32 bits in HBM) as w 2^261 mod p, so that the carry-free product
// a w 2^261 / 2^261 returns data form.  2^261 / p >= 64 for the three scalar fields (LZ_RP), which is the room everything
// below lives in:
//   tile invariant: every element has normalised limbs (< 2^29) and a value below 3.01 p
//   s  = x + x'                         limbs < 2^30, value < 6.02 p
//   d  = (x - x' + 4p) w                difference semi-normalised (limbs < 3 2^29) x canonical twiddle:
//                                       column 9 x 3 2^58 + 9 x 2^58 < 2^63.2; value < 7.01 / 64 + 1 < 1.11
//   y0 = s0 + s1                        limbs < 2^31, value < 12.04 p -> reduce_sweep: q = estimate of floor(y0 / p) from the
//                                       top limb, ONE pass  limb_i = y0_i + q (2^261 - p)_i + carry  -> normalised, < 3.01 p
//   y1 = (s0 - s1 + 7p) w               limbs < 2.5 2^30: column 9 x 2.5 2^59 + 9 x 2^58 = 54 2^58 < 2^64; value < 1.21
//   y2 = d0 + d1                        swept; value < 2.22
//   y3 = (d0 - d1 + 2p) w               value < 1.05
// Between passes the 9 limbs travel as 8 + 1 words (a 32-byte plane and a 4-byte plane of the ping buffer); the first
// pass repacks the caller's canonical input, the last pass's stores end in an exact reduction to [0, p).
template <class FP>
struct Fft29 {
  typedef FpL<FP> L;
  static_assert(L::W == 29 && L::L == 9 && FP::N == 8, "laid out for the 254 / 255-bit scalar fields");
  static_assert(FP::LZ_RP >= 64, "value bounds below assume 2^261 / p >= 64");
  static constexpr u32 MASK = L::MASK;
  static constexpr u32 comp(int i) { return i == 0 ? (1u << 29) - FP::LZ_KP[1][0] : MASK - FP::LZ_KP[1][i]; }   // 2^261 - p
  static constexpr u32 PTOP = FP::LZ_KP[1][8];
  static constexpr u32 QM = (u32)((1ull << 32) / (PTOP + 1));

  ARK_HD static L sum(const L& a, const L& b) { return L::add_lazy(a, b); }
  // a - b + K p, limbs of b below H 2^29: never negative, limbs below (limbs of a) + (H + 1) 2^29
  template <int K, int H>
  ARK_HD static L dif(const L& a, const L& b) {
    L r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = a.l[i] - b.l[i] + L::template kp_spread_any<K, H>(i);
    return r;
  }
  // limbs < 2^31.4, value V < 2^261 -> normalised limbs, the same residue, below 3.01 p.  q = floor(top limb / (p_top + 1))
  // by one multiply-high never exceeds floor(V / p) and falls short of it by at most 2 (the limbs below the top one weigh
  // < 6 units of it, the 9-bit reciprocal costs < 1); V + q (2^261 - p) = (V - q p) + q 2^261: the carry out of the top
  // limb is q itself and is dropped.
  ARK_HD static L reduce_sweep(const L& v) {
    const u32 q = (u32)(((u64)v.l[8] * QM) >> 32);
    L r;
    u32 carry = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
      const u64 acc = (u64)q * comp(i) + (u64)(v.l[i] + carry);
      r.l[i] = (u32)acc & MASK;
      carry = (u32)(acc >> 29);
    }
    return r;
  }
  ARK_HD static L sweep(const L& v) {   // limbs < 2^31 -> normalised, value unchanged (< 2^261)
    L r;
    u32 carry = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const u32 t = v.l[i] + carry;
      r.l[i] = t & MASK;
      carry = t >> 29;
    }
    r.l[8] = v.l[8] + carry;
    return r;
  }
  ARK_HD static L cond_sub_p(const L& r) {   // normalised r: r >= p ? r - p : r
    L t;
    int borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const int d = (int)r.l[i] - (int)FP::LZ_KP[1][i] + borrow;
      t.l[i] = (u32)d & MASK;
      borrow = d >> 29;
    }
    const int top = (int)r.l[8] - (int)FP::LZ_KP[1][8] + borrow;
    t.l[8] = (u32)top;
    L o;
#pragma unroll
    for (int i = 0; i < 9; i++) o.l[i] = top < 0 ? r.l[i] : t.l[i];
    return o;
  }
  ARK_HD static L canon(const L& v) { return cond_sub_p(cond_sub_p(reduce_sweep(v))); }   // -> [0, p)
  ARK_HD static L unpack(const uint4& a, const uint4& b) {   // 8 x 32-bit words -> 9 x 29-bit limbs (the integer unchanged)
    const u32 w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    return L::unpack32(w);
  }
  ARK_HD static L limbs9(const uint4& a, const uint4& b, u32 top) {   // a twiddle as fetched: canonical 8 words, or (A/B
#if ARK_FFT29_UNPACKED_TW                                                // build) an entry of the unpacked table
    L x;
    x.l[0] = a.x; x.l[1] = a.y; x.l[2] = a.z; x.l[3] = a.w;
    x.l[4] = b.x; x.l[5] = b.y; x.l[6] = b.z; x.l[7] = b.w;
    x.l[8] = top;
    return x;
#else
    (void)top;
    return unpack(a, b);
#endif
  }
  ARK_HD static L load_canonical(const u32* g) {
    const uint4* q = (const uint4*)g;
    return unpack(q[0], q[1]);
  }
  ARK_HD static void pack(const L& x, uint4& a, uint4& b) {   // normalised value below 2^256 -> 8 x 32-bit words
    u32 w[8];
    x.pack32(w);
    a = make_uint4(w[0], w[1], w[2], w[3]);
    b = make_uint4(w[4], w[5], w[6], w[7]);
  }
};

struct FftPass29Args {
  FftPassArgs a;
  const u32* roots9; // the twiddle table is stored UNPACKED: a.roots holds limbs 0..7 of w^j 2^261 mod p, roots9 limb 8
                     // (the per-butterfly repack of a canonical entry cost 22 instructions x 3 twiddles per 4-point group)
  const u32* src9;   // 9th limbs of the source (nullptr: the source holds canonical 8-word elements -- the caller's input)
  u32* dst9;         // 9th limbs of the destination (unused by the last pass, which stores canonical elements)
};

#ifndef ARK_FFT29_FENCE
#if defined(__HIP_DEVICE_COMPILE__)
#define ARK_FFT29_FENCE __builtin_amdgcn_sched_barrier(0)
#else
#define ARK_FFT29_FENCE
#endif
#endif
#ifndef ARK_FFT29_PARK
#define ARK_FFT29_PARK 1   // the two sums and the first difference wait in the lane's own LDS slots while the products run: 156 -> 123 VGPRs, four waves per SIMD
#endif
#ifndef ARK_FFT29_MIN_WAVES
#define ARK_FFT29_MIN_WAVES 1   // waves per SIMD the register allocation must leave room for (A/B builds)
#endif
#ifndef ARK_FFT29_UNPACKED_TW
#define ARK_FFT29_UNPACKED_TW 0 // 1: twiddle table stored as 8 + 1 words of 29-bit limbs (A/B builds; measured slower: the 4-byte
#endif                          // gathers of the ninth limbs double the twiddle traffic -- profiles/r4_fft_carry_free_ab.txt)
template <class FP>
__global__ void __launch_bounds__(FFT_THREADS, ARK_FFT29_MIN_WAVES) fft_pass29_kernel(const u32* __restrict__ src, u32* __restrict__ dst,
                                                                 FftPass29Args pa) {
  typedef Fft29<FP> A;
  typedef FpL<FP> L;
  const FftPassArgs& a = pa.a;
  extern __shared__ uint4 lds[];
  const int kp = a.kp, t = a.t, k = a.k;
  const u32 E = 1u << (kp + t);
  uint4* pl0 = lds;            // limbs 0..3
  uint4* pl1 = lds + E;        // limbs 4..7
  u32* pl2 = (u32*)(lds + 2 * E);   // limb 8
  const u32 tile = blockIdx.x;
  const int lo_shift = k - a.s0 - kp;
  const u32 T1 = (1u << t) - 1u;
  const u32 Q1 = (1u << kp) - 1u;
  const u32 tid = threadIdx.x;
  u32 mid = 0, hi_bits = 0;
  if (!a.last) {
    mid = tile & ((1u << (lo_shift - t)) - 1u);
    hi_bits = tile >> (lo_shift - t);
  }
  auto lds_get = [&](u32 i) -> L {
    const uint4 p = pl0[i], q = pl1[i];
    L x;
    x.l[0] = p.x; x.l[1] = p.y; x.l[2] = p.z; x.l[3] = p.w;
    x.l[4] = q.x; x.l[5] = q.y; x.l[6] = q.z; x.l[7] = q.w;
    x.l[8] = pl2[i];
    return x;
  };
  auto lds_put = [&](u32 i, const L& x) {
    pl0[i] = make_uint4(x.l[0], x.l[1], x.l[2], x.l[3]);
    pl1[i] = make_uint4(x.l[4], x.l[5], x.l[6], x.l[7]);
    pl2[i] = x.l[8];
  };
  // ---- load tile: issue every global load, then fill LDS ----
  {
    uint4 v0[FFT_MAX_EPT], v1[FFT_MAX_EPT];
    u32 v8[FFT_MAX_EPT];
    size_t ps[FFT_MAX_EPT];
#pragma unroll
    for (int it = 0; it < FFT_MAX_EPT; it++) {
      u32 e = tid + it * FFT_THREADS;
      v8[it] = 0;
      if (e < E) {
        size_t pos;
        if (!a.last) {
          u32 q = e >> t, r = e & T1;
          pos = ((size_t)hi_bits << (k - a.s0)) | ((size_t)q << lo_shift) | ((size_t)mid << t) | r;
        } else {
          u32 r = e >> kp, q = e & Q1;
          pos = ((size_t)r << (k - t)) | ((size_t)tile << kp) | q;
        }
        ps[it] = pos;
        const size_t spos = a.zskip ? (pos & (((size_t)1 << (k - a.zskip)) - 1)) : pos;
        const uint4* g = (const uint4*)(src + spos * 8);
        v0[it] = g[0];
        v1[it] = g[1];
        if (pa.src9) v8[it] = pa.src9[spos];
      }
    }
#pragma unroll
    for (int it = 0; it < FFT_MAX_EPT; it++) {
      u32 e = tid + it * FFT_THREADS;
      if (e < E) {
        const size_t pos = ps[it];
        const size_t spos = a.zskip ? (pos & (((size_t)1 << (k - a.zskip)) - 1)) : pos;
        L x;
        if (pa.src9) {
          x.l[0] = v0[it].x; x.l[1] = v0[it].y; x.l[2] = v0[it].z; x.l[3] = v0[it].w;
          x.l[4] = v1[it].x; x.l[5] = v1[it].y; x.l[6] = v1[it].z; x.l[7] = v1[it].w;
          x.l[8] = v8[it];
        } else {
          x = A::unpack(v0[it], v1[it]);   // canonical input: below p
        }
        if (a.pre_lo) {   // coset pre-scaling: x *= h^pos (tables hold h^i 2^261)
          const L pw = L::mul(A::load_canonical(a.pre_hi + (spos >> PW_LO_BITS) * 8),
                              A::load_canonical(a.pre_lo + (spos & ((1u << PW_LO_BITS) - 1)) * 8));   // < 1.02
          x = L::mul(x, pw);                                                                           // < 1.02
        }
        if (a.zskip) {
          const u32 blk = (u32)(pos >> (k - a.zskip));
          const size_t ex = (spos * (size_t)bitrev32(blk, a.zskip)) & (((size_t)1 << k) - 1);
          if (ex != 0) {
            const size_t half = (size_t)1 << (k - 1);
            const size_t ri = ex & (half - 1);
            const uint4* rg = (const uint4*)(a.roots + ri * 8);
            x = L::mul(x, A::limbs9(rg[0], rg[1], ARK_FFT29_UNPACKED_TW ? pa.roots9[ri] : 0u));
            if (ex >= half) x = L::template neg<2>(x);  // w^(n/2) = -1: 2p - x, swept (x < 1.02)
          }
        }
        lds_put(e, x);
      }
    }
  }
  __syncthreads();
  // ---- kp butterfly stages in LDS, two at a time (the organisation of fft_pass_kernel) ----
  int ls = 0;
  for (; ls + 1 < kp; ls += 2) {
    const u32 lg = 1u << (kp - 1 - ls), qt = lg >> 1;
    const int s = a.s0 + ls;
    const bool tail = a.last && qt == 1;   // the transform's last two stages: only w^(n/4) is not 1; outputs go to the exact reduction
    u32 idx[FFT_MAX_EPT / 4][4];
    uint4 wa0[FFT_MAX_EPT / 4][2], wa1[FFT_MAX_EPT / 4][2], wb[FFT_MAX_EPT / 4][2];
    u32 wa0t[FFT_MAX_EPT / 4], wa1t[FFT_MAX_EPT / 4], wbt[FFT_MAX_EPT / 4];
#pragma unroll
    for (int it = 0; it < FFT_MAX_EPT / 4; it++) {
      const u32 g = tid + it * FFT_THREADS;
      if (g < E / 4) {
        u32 gq, r;
        if (!a.last) { r = g & T1; gq = g >> t; }
        else { gq = g & ((1u << (kp - 2)) - 1u); r = g >> (kp - 2); }
        const u32 j0 = gq & (qt - 1u);
        const u32 q0 = ((gq & ~(qt - 1u)) << 2) | j0;
        size_t ta0, ta1, tb;
        if (!a.last) {
          const size_t col = ((size_t)mid << t) | r;
#pragma unroll
          for (int m = 0; m < 4; m++) idx[it][m] = ((q0 + m * qt) << t) | r;
          ta0 = ((((size_t)j0) << lo_shift) | col) << s;
          ta1 = ((((size_t)(j0 + qt)) << lo_shift) | col) << s;
          tb = ((((size_t)j0) << lo_shift) | col) << (s + 1);
        } else {
#pragma unroll
          for (int m = 0; m < 4; m++) idx[it][m] = (r << kp) | (q0 + m * qt);
          ta0 = (size_t)j0 << s;
          ta1 = (size_t)(j0 + qt) << s;
          tb = (size_t)j0 << (s + 1);
        }
        const uint4* g1 = (const uint4*)(a.roots + ta1 * 8);
        wa1[it][0] = g1[0];
        wa1[it][1] = g1[1];
        wa1t[it] = ARK_FFT29_UNPACKED_TW ? pa.roots9[ta1] : 0u;
        if (!tail) {   // a trivial twiddle is multiplied like any other (roots[0] = 2^261 mod p): no divergent branch
          const uint4* g0 = (const uint4*)(a.roots + ta0 * 8);
          wa0[it][0] = g0[0];
          wa0[it][1] = g0[1];
          wa0t[it] = ARK_FFT29_UNPACKED_TW ? pa.roots9[ta0] : 0u;
          const uint4* g2 = (const uint4*)(a.roots + tb * 8);
          wb[it][0] = g2[0];
          wb[it][1] = g2[1];
          wbt[it] = ARK_FFT29_UNPACKED_TW ? pa.roots9[tb] : 0u;
        }
      }
    }
#pragma unroll
    for (int it = 0; it < FFT_MAX_EPT / 4; it++) {
      const u32 g = tid + it * FFT_THREADS;
      if (g < E / 4) {
        // first stage, gap lg: (x0, x2) and (x1, x3)                    fft.rs:190-198 butterfly_fn_io
        // (the pairs are read, summed and differenced one after the other -- ARK_FFT29_FENCE keeps the scheduler from hoisting
        // all four elements and three unpacked twiddles into registers at once)
        L s0, s1, e0, e1;
        {
          const L x0 = lds_get(idx[it][0]), x2 = lds_get(idx[it][2]);
          s0 = A::sum(x0, x2);
          e0 = A::template dif<4, 1>(x0, x2);
        }
        ARK_FFT29_FENCE;
        {
          const L x1 = lds_get(idx[it][1]), x3 = lds_get(idx[it][3]);
          s1 = A::sum(x1, x3);
          e1 = A::template dif<4, 1>(x1, x3);
        }
        ARK_FFT29_FENCE;
#if ARK_FFT29_PARK
        // the two sums wait in their own LDS slots (this lane's: no barrier) while the two products run: 18 registers less
        // across the multiply-heavy middle of the block
        lds_put(idx[it][0], s0);
        lds_put(idx[it][1], s1);
        lds_put(idx[it][2], e0);
        asm volatile("" ::: "memory");
#endif
        const L d1 = L::mul(e1, A::limbs9(wa1[it][0], wa1[it][1], wa1t[it]));
        ARK_FFT29_FENCE;
#if ARK_FFT29_PARK
        asm volatile("" ::: "memory");
        e0 = lds_get(idx[it][2]);
#endif
        L y0, y1, y2, y3;
        if (!tail) {
          const L d0 = L::mul(e0, A::limbs9(wa0[it][0], wa0[it][1], wa0t[it]));
          ARK_FFT29_FENCE;
#if ARK_FFT29_PARK
          asm volatile("" ::: "memory");
          s0 = lds_get(idx[it][0]);
          s1 = lds_get(idx[it][1]);
          ARK_FFT29_FENCE;
#endif
          const L w = A::limbs9(wb[it][0], wb[it][1], wbt[it]);
          // second stage, gap lg/2: (s0, s1) and (d0, d1), one twiddle for both
          y0 = A::reduce_sweep(A::sum(s0, s1));
          y1 = L::mul(A::template dif<7, 2>(s0, s1), w);
          y2 = A::sweep(A::sum(d0, d1));
          y3 = L::mul(A::template dif<2, 1>(d0, d1), w);
        } else {
#if ARK_FFT29_PARK
          asm volatile("" ::: "memory");
          s0 = lds_get(idx[it][0]);
          s1 = lds_get(idx[it][1]);
#endif
          const L d0 = e0;                              // value < 7.01, limbs < 3 2^29
          y0 = A::sum(s0, s1);                          // < 12.04
          y1 = A::template dif<7, 2>(s0, s1);           // < 13.02, limbs < 2.5 2^30
          y2 = A::sum(d0, d1);                          // < 8.12
          y3 = A::template dif<2, 1>(d0, d1);           // < 9.01, limbs < 2.5 2^30
        }
        lds_put(idx[it][0], y0);
        lds_put(idx[it][1], y1);
        lds_put(idx[it][2], y2);
        lds_put(idx[it][3], y3);
      }
    }
    __syncthreads();
  }
  for (; ls < kp; ls++) {
    const u32 lg = 1u << (kp - 1 - ls);
    const int s = a.s0 + ls;
    const bool tail = a.last && lg == 1;   // the transform's last stage: every twiddle is 1
    u32 i0s[FFT_MAX_EPT / 2], i1s[FFT_MAX_EPT / 2];
    uint4 w0[FFT_MAX_EPT / 2], w1[FFT_MAX_EPT / 2];
    u32 w8[FFT_MAX_EPT / 2];
#pragma unroll
    for (int it = 0; it < FFT_MAX_EPT / 2; it++) {
      u32 b = tid + it * FFT_THREADS;
      if (b < E / 2) {
        u32 i0, i1;
        size_t tw;
        if (!a.last) {
          u32 r = b & T1, qq = b >> t;
          u32 q = ((qq & ~(lg - 1)) << 1) | (qq & (lg - 1));
          i0 = (q << t) | r;
          i1 = i0 + (lg << t);
          tw = ((((size_t)(q & (lg - 1))) << lo_shift) | ((size_t)mid << t) | r) << s;
        } else {
          u32 qq = b & ((1u << (kp - 1)) - 1u), r = b >> (kp - 1);
          u32 q = ((qq & ~(lg - 1)) << 1) | (qq & (lg - 1));
          i0 = (r << kp) | q;
          i1 = i0 + lg;
          tw = ((size_t)(q & (lg - 1))) << s;
        }
        i0s[it] = i0;
        i1s[it] = i1;
        if (!tail) {
          const uint4* g = (const uint4*)(a.roots + tw * 8);
          w0[it] = g[0];
          w1[it] = g[1];
          w8[it] = ARK_FFT29_UNPACKED_TW ? pa.roots9[tw] : 0u;
        }
      }
    }
#pragma unroll
    for (int it = 0; it < FFT_MAX_EPT / 2; it++) {
      u32 b = tid + it * FFT_THREADS;
      if (b < E / 2) {
        const u32 i0 = i0s[it], i1 = i1s[it];
        const L lo = lds_get(i0), hi = lds_get(i1);
        L sm = A::sum(lo, hi), df = A::template dif<4, 1>(lo, hi);   // < 6.02 / < 7.01
        if (!tail) {
          sm = A::reduce_sweep(sm);
          df = L::mul(df, A::limbs9(w0[it], w1[it], w8[it]));
        }
        lds_put(i0, sm);
        lds_put(i1, df);
      }
    }
    __syncthreads();
  }
  // ---- store tile ----
  if (!a.last) {
#pragma unroll
    for (int it = 0; it < FFT_MAX_EPT; it++) {
      u32 e = tid + it * FFT_THREADS;
      if (e < E) {
        u32 q = e >> t, r = e & T1;
        size_t pos = ((size_t)hi_bits << (k - a.s0)) | ((size_t)q << lo_shift) | ((size_t)mid << t) | r;
        uint4* g = (uint4*)(dst + pos * 8);
        g[0] = pl0[e];
        g[1] = pl1[e];
        pa.dst9[pos] = pl2[e];
      }
    }
  } else {
    // position p = r<<(k-t) | tile<<kp | q holds X[bitrev_k(p)]  (replaces derange(), fft.rs:373-380)
    const int tb = k - kp - t;
    const size_t tile_rev = bitrev32(tile, tb);
#pragma unroll
    for (int it = 0; it < FFT_MAX_EPT; it++) {
      u32 e = tid + it * FFT_THREADS;
      if (e < E) {
        u32 r2 = e & T1, q2 = e >> t;  // output-side coordinates
        size_t opos = ((size_t)q2 << (k - kp)) | (tile_rev << t) | r2;
        u32 i = (bitrev32(r2, t) << kp) | bitrev32(q2, kp);
        L x = lds_get(i);   // the tail stages' raw outputs: limbs < 2.5 2^30, value < 13.02 p
        if (a.post_lo || a.post_const) {
          L pw;
          if (a.post_lo)
            pw = L::mul(A::load_canonical(a.post_hi + (opos >> PW_LO_BITS) * 8),
                        A::load_canonical(a.post_lo + (opos & ((1u << PW_LO_BITS) - 1)) * 8));   // < 1.02
          else
            pw = A::load_canonical(a.post_const);
          x = A::cond_sub_p(L::mul(x, pw));   // column 9 x 2.5 2^59 + 9 x 2^58 < 2^64; value < 13.02 * 1.02 / 64 + 1 < 1.21
        } else {
          x = A::canon(x);
        }
        uint4 o0, o1;
        A::pack(x, o0, o1);
        uint4* g = (uint4*)(dst + opos * 8);
        g[0] = o0;
        g[1] = o1;
      }
    }
  }
}

// ---- G-point transform along the slow axis of a [G][cols] array (multi-GPU exchange step) ----------------
// out[j][c] = sum_i root^(i j) in[i][c], natural order in and out, G in {2,4,8,16}.  One lane per column: the G
// values travel through registers (radix-2 decimation in frequency, then a bit-reversed write-back).
// `pw` holds root^k for k < G/2.
template <class FP, int G>
__global__ void __launch_bounds__(256) fft_axis_kernel(const u32* src, u32* dst, size_t stride, size_t cols,
                                                       const u32* __restrict__ pw) {
  // rows are `stride` elements apart (>= cols: a column slice of a wider array); src may equal dst -- a lane reads its
  // whole column before it writes
  typedef Fp<FP> F;
  size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cols) return;
  F v[G];
#pragma unroll
  for (int i = 0; i < G; i++) v[i] = F::load(src + ((size_t)i * stride + c) * F::N);
#pragma unroll
  for (int gap = G / 2; gap >= 1; gap >>= 1) {
#pragma unroll
    for (int base = 0; base < G; base += 2 * gap) {
#pragma unroll
      for (int j = 0; j < gap; j++) {
        F lo = v[base + j], hi = v[base + j + gap];
        v[base + j] = F::add(lo, hi);
        F d = F::sub(lo, hi);
        const int tw = j * (G / (2 * gap));  // root^(j * G/(2 gap))
        if (tw != 0) d = F::mul(d, F::load(pw + (size_t)tw * F::N));
        v[base + j + gap] = d;
      }
    }
  }
  constexpr int LG = G == 2 ? 1 : (G == 4 ? 2 : (G == 8 ? 3 : 4));
#pragma unroll
  for (int i = 0; i < G; i++) {
    int r = 0;
#pragma unroll
    for (int b = 0; b < LG; b++) r |= ((i >> b) & 1) << (LG - 1 - b);
    v[i].store(dst + ((size_t)r * stride + c) * F::N);  // position i holds X[bitrev(i)]
  }
}

// ---- host side: cached twiddle tables + pass plan ------------------------------------------------
// T_s[i] = T_0[i << s] for every stage s >= 1, packed behind T_0 (entry n - (n >> s) onwards): one thread per entry of
// the tail [n/2, n - 1)
template <class FP>
__global__ void __launch_bounds__(256) fft_compact_tables_kernel(u32* __restrict__ tab, int k) {
  const size_t n = (size_t)1 << k;
  const size_t e = n / 2 + (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // destination entry
  if (e >= n - 1) return;
  // stage of entry e: the s with n - (n >> s) <= e < n - (n >> (s + 1)), i.e. s = number of leading ones of e in k bits
  const size_t gap = n - e;                      // in (n >> (s + 1), n >> s]
  const int s = k - (64 - __clzll((unsigned long long)(gap - 1)));   // gap - 1 in [n >> (s+1), n >> s): top bit k - s - 1
  const size_t i = e - (n - (n >> s));
  const uint4* src = (const uint4*)(tab + (i << s) * 8);
  uint4* dst = (uint4*)(tab + e * 8);
  const uint4 a = src[0], b = src[1];
  dst[0] = a;
  dst[1] = b;
}
struct FftTables {
  DevBuf roots;   // n/2 entries (form 1: limbs 0..7 of the unpacked 9 x 29-bit entries); form 0: n - 1 entries, compact per stage
  DevBuf roots9;  // form 1: limb 8 of every entry
  DevBuf small;   // scratch for the two-level build + generator copy
};
// canonical 8-word entries -> 9 x 29-bit limbs, in place for limbs 0..7 (+ the 9th-limb plane)
template <class FP>
__global__ void __launch_bounds__(256) fft_unpack_table_kernel(u32* __restrict__ tab, size_t count, u32* __restrict__ tab9) {
  const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= count) return;
  const FpL<FP> x = Fft29<FP>::load_canonical(tab + j * 8);
  uint4* g = (uint4*)(tab + j * 8);
  g[0] = make_uint4(x.l[0], x.l[1], x.l[2], x.l[3]);
  g[1] = make_uint4(x.l[4], x.l[5], x.l[6], x.l[7]);
  tab9[j] = x.l[8];
}
struct FftKey {
  int field, k;
  std::array<uint64_t, 4> gen;
  int form = 0;   // 0: w^j R (the saturated kernel's operands); 1: w^j 2^261 (the carry-free kernel's)
  bool operator<(const FftKey& o) const {
    if (field != o.field) return field < o.field;
    if (k != o.k) return k < o.k;
    if (form != o.form) return form < o.form;
    return gen < o.gen;
  }
};
// cached coset power tables h^i = hi[i >> 10] * lo[i & 1023] (optionally with a constant folded into hi) and single
// constants: provers transform over the same coset again and again, and rebuilding them cost two kernels, three small
// uploads and a stream synchronisation per call (coset FFT at 2^16: 92 -> 48 us, at 2^4: 36 -> 7 us)
struct FftPwKey {
  int field, k;                      // k = -1: a single constant
  std::array<uint64_t, 4> base, mul; // mul = 0: none
  int form = 0;                      // as FftKey::form
  bool operator<(const FftPwKey& o) const {
    if (field != o.field) return field < o.field;
    if (k != o.k) return k < o.k;
    if (form != o.form) return form < o.form;
    if (base != o.base) return base < o.base;
    return mul < o.mul;
  }
};
struct FftWorkspace {
  std::map<FftKey, FftTables> tables;  // per (field, log n, root)
  std::map<FftPwKey, DevBuf> powers;   // per (field, log n, offset[, constant]): [lo (1024) | hi (n >> 10)]
  std::map<hipStream_t, DevBuf> tmps;  // ping buffer of the multi-pass transform, one per stream (transforms in flight)
  DevBuf pw;                           // coset power tables + constants
  DevBuf axis_pw;                      // G-point cross transform: root and its first G/2 powers
  uint64_t axis_root[4] = {0, 0, 0, 0};
  DevBuf stage;                        // host-pointer entry: device copy of the data
  hipEvent_t ev[10] = {};              // pass timing (created on first use, reused)
  int kernel_variant = -1;             // -1: environment (ARK_HIP_FFT_LAZY), 0: saturated pass kernel, 1: carry-free
  std::mutex mu;
  void release() {
    for (auto& kv : tables) { kv.second.roots.release(); kv.second.roots9.release(); kv.second.small.release(); }
    tables.clear();
    for (auto& kv : powers) kv.second.release();
    powers.clear();
    for (auto& kv : tmps) kv.second.release();
    tmps.clear();
    pw.release(); stage.release(); axis_pw.release();
    for (auto& e : ev) { if (e) (void)hipEventDestroy(e); e = nullptr; }
  }
};
static inline bool fft_lazy_env() {
  static const bool v = getenv("ARK_HIP_FFT_LAZY") && getenv("ARK_HIP_FFT_LAZY")[0] == '1';
  return v;
}
// ARK_HIP_FFT_FULL_POWERS=0: coset scaling from the two factor tables (two products per element; the round-4 path; A/B only)
static inline bool fft_full_powers_env() {
  static const bool v = [] { const char* e = getenv("ARK_HIP_FFT_FULL_POWERS"); return !(e && e[0] == '0'); }();
  return v;
}
// ARK_HIP_FFT_COMPACT=0: every stage reads the size-n table strided (the round-4 access pattern; A/B only)
static inline bool fft_compact_env() {
  static const bool v = [] { const char* e = getenv("ARK_HIP_FFT_COMPACT"); return !(e && e[0] == '0'); }();
  return v;
}
struct FftTimings { float total = 0; float pass[8] = {0, 0, 0, 0, 0, 0, 0, 0}; int npass = 0; };

// root^k, k < G/2, for fft_axis_launch: built into the workspace's own small buffer (stream-ordered; `root4` is host
// memory and is consumed before the call returns only if the caller synchronises -- pass memory that outlives the stream
// work, e.g. a field of the domain struct copied into the workspace, as fft_axis_prepare does)
template <class FP>
int fft_axis_prepare(FftWorkspace& ws, unsigned G, const uint64_t* root4, hipStream_t stream, const u32** out_pw) {
  typedef Fp<FP> F;
  if (G != 2 && G != 4 && G != 8 && G != 16) return -2;
  if (ws.axis_pw.ensure((1 + 16) * F::BYTES)) return -3;
  u32* d_root = (u32*)ws.axis_pw.p;
  u32* d_pw = d_root + F::N;
  memcpy(ws.axis_root, root4, 32);  // stable host copy for the asynchronous upload
  ARK_HIP_TRY(hipMemcpyAsync(d_root, ws.axis_root, F::BYTES, hipMemcpyHostToDevice, stream));
  hipLaunchKernelGGL((fft_pow_table_kernel<FP>), dim3(1), dim3(256), 0, stream, d_root, (u64)1, (u32)(G / 2),
                     (const u32*)nullptr, d_pw);
  *out_pw = d_pw;
  return 0;
}
template <class FP>
int fft_axis_launch(const void* d_src, void* d_dst, unsigned G, size_t stride, size_t cols, const u32* d_pw,
                    hipStream_t stream) {
  if (cols == 0) return 0;
  const unsigned blocks = (unsigned)((cols + 255) / 256);
  const u32* s = (const u32*)d_src;
  u32* d = (u32*)d_dst;
  switch (G) {
    case 2: hipLaunchKernelGGL((fft_axis_kernel<FP, 2>), dim3(blocks), dim3(256), 0, stream, s, d, stride, cols, d_pw); break;
    case 4: hipLaunchKernelGGL((fft_axis_kernel<FP, 4>), dim3(blocks), dim3(256), 0, stream, s, d, stride, cols, d_pw); break;
    case 8: hipLaunchKernelGGL((fft_axis_kernel<FP, 8>), dim3(blocks), dim3(256), 0, stream, s, d, stride, cols, d_pw); break;
    case 16: hipLaunchKernelGGL((fft_axis_kernel<FP, 16>), dim3(blocks), dim3(256), 0, stream, s, d, stride, cols, d_pw); break;
    default: return -2;
  }
  return 0;
}
// the whole [G][cols] array at once (d_src -> d_dst, may alias); `pw_stream`: the stream the table is built on (the
// launch follows on `stream` behind `after_table`, which is recorded here when the two differ)
template <class FP>
int fft_axis_run(FftWorkspace& ws, const void* d_src, void* d_dst, unsigned G, size_t cols, const uint64_t* root4,
                 hipStream_t stream) {
  std::lock_guard<std::mutex> lock(ws.mu);
  if (G == 1 || cols == 0) {
    if (d_src != d_dst && cols) ARK_HIP_TRY(hipMemcpyAsync(d_dst, d_src, (size_t)G * cols * Fp<FP>::BYTES, hipMemcpyDeviceToDevice, stream));
    return 0;
  }
  const u32* d_pw = nullptr;
  if (int rc = fft_axis_prepare<FP>(ws, G, root4, stream, &d_pw)) return rc;
  if (int rc = fft_axis_launch<FP>(d_src, d_dst, G, cols, cols, d_pw, stream)) return rc;
  ARK_HIP_TRY(hipGetLastError());
  return 0;
}


// form29: entries w^j 2^261 mod p instead of w^j R -- the whole table (or its `hi` factor) starts from the Montgomery residue
// of 2^5, whose integer is 2^261 mod p (FP::LZ_CIN)
template <class FP>
int fft_get_roots(FftWorkspace& ws, int k, const uint64_t* root4, hipStream_t stream, const u32** out, bool form29 = false,
                  const u32** out9 = nullptr) {
  typedef Fp<FP> F;
  FftKey key{FP::ID, k, {root4[0], root4[1], root4[2], root4[3]}, form29 ? 1 : 0};
  auto it = ws.tables.find(key);
  if (it != ws.tables.end()) {
    *out = (const u32*)it->second.roots.p;
    if (out9) *out9 = (const u32*)it->second.roots9.p;
    return 0;
  }
  FftTables tb;  // built here, entered into the cache only once complete (a failed build must not leave a null table)
  struct Guard {
    FftTables* t;
    ~Guard() { if (t) { t->roots.release(); t->roots9.release(); t->small.release(); } }
  } guard{&tb};
  size_t half = k >= 1 ? ((size_t)1 << (k - 1)) : 1;
  // the saturated kernel's table carries every stage's twiddles compactly: T_0 (n/2) | T_1 (n/4) | ... = n - 1 entries
  const bool compact = !form29 && k >= 2;
  if (tb.roots.ensure((compact ? 2 * half : half) * F::BYTES)) return -3;
  if (form29 && ARK_FFT29_UNPACKED_TW && tb.roots9.ensure(half * 4)) return -3;
  const int LB = 11;
  size_t nlo = half < ((size_t)1 << LB) ? half : ((size_t)1 << LB);
  size_t nhi = half >> LB; if (nhi == 0) nhi = 1;
  if (tb.small.ensure((2 + nlo + nhi) * F::BYTES)) return -3;
  u32* d_gen = (u32*)tb.small.p;
  u32* d_cin = d_gen + F::N;
  u32* d_lo = d_cin + F::N;
  u32* d_hi = d_lo + nlo * F::N;
  ARK_HIP_TRY(hipMemcpyAsync(d_gen, root4, F::BYTES, hipMemcpyHostToDevice, stream));
  ARK_HIP_TRY(hipMemcpyAsync(d_cin, FP::LZ_CIN, F::BYTES, hipMemcpyHostToDevice, stream));
  const u32* scale = form29 ? (const u32*)d_cin : (const u32*)nullptr;
  if (half <= ((size_t)1 << LB)) {
    hipLaunchKernelGGL((fft_pow_table_kernel<FP>), dim3((u32)((half + 255) / 256)), dim3(256), 0, stream, d_gen, (u64)1,
                       (u32)half, scale, (u32*)tb.roots.p);
  } else {
    hipLaunchKernelGGL((fft_pow_table_kernel<FP>), dim3((u32)((nlo + 255) / 256)), dim3(256), 0, stream, d_gen, (u64)1,
                       (u32)nlo, (const u32*)nullptr, d_lo);
    hipLaunchKernelGGL((fft_pow_table_kernel<FP>), dim3((u32)((nhi + 255) / 256)), dim3(256), 0, stream, d_gen,
                       (u64)1 << LB, (u32)nhi, scale, d_hi);
    hipLaunchKernelGGL((fft_expand_table_kernel<FP>), dim3((u32)((half + 255) / 256)), dim3(256), 0, stream, d_lo, d_hi,
                       LB, half, (u32*)tb.roots.p);
  }
  if (compact)
    hipLaunchKernelGGL((fft_compact_tables_kernel<FP>), dim3((u32)((half - 1 + 255) / 256)), dim3(256), 0, stream,
                       (u32*)tb.roots.p, k);
  if (form29 && ARK_FFT29_UNPACKED_TW)
    hipLaunchKernelGGL((fft_unpack_table_kernel<FP>), dim3((u32)((half + 255) / 256)), dim3(256), 0, stream, (u32*)tb.roots.p,
                       half, (u32*)tb.roots9.p);
  ARK_HIP_TRY(hipGetLastError());
  ARK_HIP_TRY(hipStreamSynchronize(stream));  // root4 is a caller stack pointer
  FftTables& slot = ws.tables[key];
  slot = tb;          // DevBuf is a plain handle: ownership moves to the cache
  guard.t = nullptr;
  *out = (const u32*)slot.roots.p;
  if (out9) *out9 = (const u32*)slot.roots9.p;
  return 0;
}

// The power-table cache is bounded (a caller cycling through offsets must not grow it for ever).  Eviction happens
// only here, BEFORE a transform looks anything up: a transform takes at most two entries (pre + post), so the pointers
// it has fetched are never freed under it.
static constexpr size_t FFT_POWERS_BUDGET = (size_t)8 << 30;   // expanded power tables are n x 32 B each
static inline int fft_powers_make_room(FftWorkspace& ws) {
  size_t bytes = 0;
  for (auto& kv : ws.powers) bytes += kv.second.cap;
  if (ws.powers.size() + 2 <= 64 && bytes <= FFT_POWERS_BUDGET) return 0;
  ARK_HIP_TRY(hipDeviceSynchronize());  // transforms in flight on any stream may still read the tables
  for (auto& kv : ws.powers) kv.second.release();
  ws.powers.clear();
  return 0;
}

// lo/hi power tables of `base4` for a size-2^k transform (hi optionally multiplied by `mul4`), or -- k < 0 -- the single
// constant `base4`, resident and cached.  base4 / mul4 are host pointers.
// form29: BOTH factors carry 2^261 instead of R (the carry-free kernel multiplies them with its own product, which divides
// by 2^261 once), and so does a single constant.
// full (optional out): the expanded table full[i] = hi[i >> 10] * lo[i & 1023], i < 2^k -- built behind the two factor
// tables for the saturated kernel when 2^k x 32 B is worth a cache slot (k >= 12, at most a quarter of the budget)
template <class FP>
int fft_get_powers(FftWorkspace& ws, int k, const uint64_t* base4, const uint64_t* mul4, hipStream_t stream,
                   const u32** lo, const u32** hi, bool form29 = false, const u32** full = nullptr) {
  typedef Fp<FP> F;
  FftPwKey key{FP::ID, k, {base4[0], base4[1], base4[2], base4[3]}, {0, 0, 0, 0}, form29 ? 1 : 0};
  if (mul4) key.mul = {mul4[0], mul4[1], mul4[2], mul4[3]};
  auto it = ws.powers.find(key);
  if (it == ws.powers.end()) {
    const size_t nlo = k < 0 ? 0 : ((size_t)1 << PW_LO_BITS);
    const size_t nhi = k < 0 ? 0 : (k > PW_LO_BITS ? ((size_t)1 << (k - PW_LO_BITS)) : 1);
    const bool want_full = !form29 && k >= 12 && fft_full_powers_env() && (((size_t)32 << k) <= FFT_POWERS_BUDGET / 4);
    size_t nfull = want_full ? ((size_t)1 << k) : 0;
    DevBuf buf;
    struct Guard {
      DevBuf* b;
      ~Guard() { if (b) b->release(); }
    } guard{&buf};
    // the expanded table is OPTIONAL (it saves one product per element: 3 % of a coset transform): under memory pressure -- a
    // verified cache at its budget, a large prepared table -- the two factor tables (a few hundred KB) are what the transform
    // needs, and the round-4 two-factor path serves (a failed 2^k x 32 B allocation used to fail the transform)
    if (nfull) {
      size_t fr = 0, tot = 0;
      if (hipMemGetInfo(&fr, &tot) == hipSuccess && fr < ((nlo + nhi + 3 + nfull) * F::BYTES) + ((size_t)256 << 20)) nfull = 0;
    }
    if (buf.ensure((nlo + nhi + 3 + nfull) * F::BYTES)) {
      if (!nfull) return -3;
      (void)hipGetLastError();
      nfull = 0;
      if (buf.ensure((nlo + nhi + 3) * F::BYTES)) return -3;
    }
    u32* base = (u32*)buf.p;
    u32* d_c = base + (nlo + nhi) * F::N;  // [base | mul | 2^261 mod p], then the expanded table
    // host-side scaling of the hi factor / the constant: m 2^261 = mont_mul(m R, 2^261 mod p)
    uint64_t hi_mul[4], cin[4];
    memcpy(cin, FP::LZ_CIN, 32);
    const uint64_t* hi_mul_p = mul4;
    if (form29) {
      const F c = F::load(cin);
      const F m = mul4 ? F::mul(F::load(mul4), c) : c;
      m.store(hi_mul);
      hi_mul_p = hi_mul;
    }
    if (k < 0 && form29) {
      const F v = F::mul(F::load(base4), F::load(cin));
      uint64_t tmpc[4];
      v.store(tmpc);
      ARK_HIP_TRY(hipMemcpyAsync(d_c, tmpc, F::BYTES, hipMemcpyHostToDevice, stream));
      ARK_HIP_TRY(hipStreamSynchronize(stream));  // tmpc is a stack array
    } else {
      ARK_HIP_TRY(hipMemcpyAsync(d_c, base4, F::BYTES, hipMemcpyHostToDevice, stream));
    }
    if (hi_mul_p) ARK_HIP_TRY(hipMemcpyAsync(d_c + F::N, hi_mul_p, F::BYTES, hipMemcpyHostToDevice, stream));
    ARK_HIP_TRY(hipMemcpyAsync(d_c + 2 * F::N, FP::LZ_CIN, F::BYTES, hipMemcpyHostToDevice, stream));
    if (k >= 0) {
      hipLaunchKernelGGL((fft_pow_table_kernel<FP>), dim3((u32)(nlo / 256)), dim3(256), 0, stream, d_c, (u64)1, (u32)nlo,
                         form29 ? (const u32*)(d_c + 2 * F::N) : (const u32*)nullptr, base);
      hipLaunchKernelGGL((fft_pow_table_kernel<FP>), dim3((u32)((nhi + 255) / 256)), dim3(256), 0, stream, d_c, (u64)nlo,
                         (u32)nhi, hi_mul_p ? (const u32*)(d_c + F::N) : (const u32*)nullptr, base + nlo * F::N);
      if (nfull)
        hipLaunchKernelGGL((fft_expand_table_kernel<FP>), dim3((u32)((nfull + 255) / 256)), dim3(256), 0, stream,
                           (const u32*)base, (const u32*)(base + nlo * F::N), PW_LO_BITS, nfull, d_c + 3 * F::N);
      ARK_HIP_TRY(hipGetLastError());
    }
    ARK_HIP_TRY(hipStreamSynchronize(stream));  // base4 / mul4 are caller memory
    it = ws.powers.emplace(key, buf).first;     // DevBuf is a plain handle: ownership moves to the cache
    guard.b = nullptr;
  }
  const u32* base = (const u32*)it->second.p;
  if (full) *full = nullptr;
  if (k < 0) {
    *lo = base;  // the constant itself
    if (hi) *hi = nullptr;
  } else {
    *lo = base;
    *hi = base + ((size_t)1 << PW_LO_BITS) * F::N;
    const size_t nlo2 = (size_t)1 << PW_LO_BITS, nhi2 = k > PW_LO_BITS ? ((size_t)1 << (k - PW_LO_BITS)) : 1;
    // (an entry built for the carry-free form, or before the budget allowed it, has no expanded table: its size tells)
    if (full && !form29 && it->second.cap >= (nlo2 + nhi2 + 3 + ((size_t)1 << k)) * F::BYTES && k >= 12 && fft_full_powers_env())
      *full = base + (nlo2 + nhi2 + 3) * F::N;
  }
  return 0;
}

// out[i] = mul * base^i, i < count (mul = nullptr: 1): the per-position scalars of a transform over GROUP elements
// (gfft.cuh: h^i before the stages of a coset transform, size_inv * h^-i after the inverse's).  base4 / mul4: host pointers.
template <class FP>
int fft_scalars_run(FftWorkspace&, const uint64_t* base4, const uint64_t* mul4, size_t count, void* d_out, hipStream_t stream) {
  if (count == 0) return 0;
  if (count > 0xffffffffull) return -2;
  // base and multiplier travel as kernel arguments: no staging buffer, no wait -- the entry stays asynchronous
  FftElemArg b, m;
  memcpy(b.w, base4, sizeof(b.w));
  memcpy(m.w, mul4 ? mul4 : base4, sizeof(m.w));
  hipLaunchKernelGGL((fft_pow_table_val_kernel<FP>), dim3((u32)((count + 255) / 256)), dim3(256), 0, stream, b, m, mul4 ? 1 : 0,
                     (u32)count, (u32*)d_out);
  ARK_HIP_TRY(hipGetLastError());
  return 0;
}
template <class FP>
int fft_roots_run(FftWorkspace& ws, int k, const uint64_t* root4, hipStream_t stream, const u32** out) {
  std::lock_guard<std::mutex> lock(ws.mu);
  return fft_get_roots<FP>(ws, k, root4, stream, out);
}

// d_data: device pointer to 2^k elements (Montgomery, reference layout).
// root4:  group_gen (forward) or group_gen_inv (inverse) of the size-2^k domain, host pointer.
// pre4:   coset offset h (forward coset FFT) or nullptr.        x[i] *= h^i before the transform
// post4:  inverse: h^-1 or nullptr;  postc4: constant multiplier of every output (size_inv) or nullptr
// zlog:   degree-aware forward transform (fft.rs:29-71): only the first 2^(k - zlog) elements of d_data are input
//         (the rest is treated as zero whatever it holds) and the first zlog stages are not executed; 0 = plain.
// more than 64 KiB of dynamic LDS must be granted per function and device, once
static inline hipError_t fft_allow_lds(const void* fn, int bytes) {
  static std::mutex mu;
  static std::map<std::pair<const void*, int>, bool> done;
  int dev = 0;
  if (hipError_t e = hipGetDevice(&dev)) return e;
  std::lock_guard<std::mutex> lk(mu);
  bool& d = done[std::make_pair(fn, dev)];
  if (d) return hipSuccess;
  const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) d = true;
  return e;
}
// tile size (log2 elements) of a transform of kx executed stages on the saturated kernel: see fft_run_device
static inline int fft_default_tile_log(int kx) {
  (void)kx;
  return 10;
}
template <class FP>
int fft_run_device(FftWorkspace& ws, void* d_data, int k, const uint64_t* root4, const uint64_t* pre4,
                   const uint64_t* post4, const uint64_t* postc4, int zlog, hipStream_t stream, FftTimings* tm) {
  typedef Fp<FP> F;
  std::lock_guard<std::mutex> lock(ws.mu);
  if (k < 0 || k > 30 || k > FP::TWO_ADICITY) return -2;
  const size_t n = (size_t)1 << k;
  // which pass kernel: saturated 32-bit limbs (default) or carry-free 9 x 29-bit limbs (ARK_HIP_FFT_LAZY=1 /
  // ark_hip_fft_set_kernel(1)).  The carry-free product is 27 % faster in a loop, the carry-free PASS is not: 18.6 % fewer
  // vector instructions, but 148 registers (three waves per SIMD instead of four), the vector ALU busy 76 % of the time
  // instead of 90 % -- 0.53-0.56 against 0.52-0.53 ms at 2^22 (profiles/r4_fft_carry_free_ab.txt).  Kept, tested, not default.
  const bool lazy29 = ws.kernel_variant < 0 ? fft_lazy_env() : ws.kernel_variant == 1;
  const u32* roots = nullptr;
  const u32* roots9 = nullptr;
  if (k >= 1) {
    int rc = fft_get_roots<FP>(ws, k, root4, stream, &roots, lazy29, &roots9);
    if (rc) return rc;
  }
  // coset scaling tables (cached per offset): x[i] *= h^i on the way in; out[i] *= postc * h^-i on the way out
  if (int rc = fft_powers_make_room(ws)) return rc;
  const u32 *pre_lo = nullptr, *pre_hi = nullptr, *post_lo = nullptr, *post_hi = nullptr, *post_const = nullptr;
  const u32 *pre_full = nullptr, *post_full = nullptr;
  if (pre4) {
    int rc = fft_get_powers<FP>(ws, k, pre4, nullptr, stream, &pre_lo, &pre_hi, lazy29, &pre_full);
    if (rc) return rc;
  }
  if (post4) {
    int rc = fft_get_powers<FP>(ws, k, post4, postc4, stream, &post_lo, &post_hi, lazy29, &post_full);
    if (rc) return rc;
  } else if (postc4) {
    int rc = fft_get_powers<FP>(ws, -1, postc4, nullptr, stream, &post_const, nullptr, lazy29);
    if (rc) return rc;
  }
  if (k == 0) {
    // size-1 domain: X[0] = x[0] (h^0 = 1, n^-1 = 1)
    return 0;
  }
  if (zlog < 0 || zlog >= k) return -1;
  if (zlog && k <= FFT_SINGLE_MAX) {  // tiny transform: zero-fill the tail and run it whole
    ARK_HIP_TRY(hipMemsetAsync((char*)d_data + (n >> zlog) * F::BYTES, 0, (n - (n >> zlog)) * F::BYTES, stream));
    zlog = 0;
  }
  hipEvent_t* ev = ws.ev;
  int nev = 0;
  if (tm) {
    for (int i = 0; i < 10; i++)
      if (!ev[i]) ARK_HIP_TRY(hipEventCreate(&ev[i]));
    ARK_HIP_TRY(hipEventRecord(ev[nev++], stream));
  }
  // pass plan over the kx = k - zlog executed stages
  const int kx = k - zlog;
  int P, kps[8], t;
  // tile size (log2 elements): 10 = the 1024-element tiles of rounds 1-5; 11 / 12 (round 6, saturated kernel only): 2048- /
  // 4096-element tiles, i.e. up to 10 / 11 stages per pass -- 2^22 in TWO passes of 11 stages over 64-byte segments instead of
  // three (8 + 8 + 6).  ARK_HIP_FFT_TILE_LOG forces one; the default follows fft_default_tile_log (measured)
  int tile_log = 10;
  if (!lazy29 && k > FFT_SINGLE_MAX) {
    tile_log = fft_default_tile_log(kx);
    if (const char* e = getenv("ARK_HIP_FFT_TILE_LOG"))
      if (atoi(e) >= 10 && atoi(e) <= 12) tile_log = atoi(e);
  }
  if (k <= FFT_SINGLE_MAX) {
    P = 1; kps[0] = k; t = 0;
  } else {
    int maxkp = FFT_MAX_KP;
    const char* env = getenv("ARK_HIP_FFT_KP");
    if (env && atoi(env) >= 5 && atoi(env) <= 8) maxkp = atoi(env);
    if (tile_log > 10) maxkp = tile_log - 1;   // larger tiles: up to tile_log - 1 stages per pass over >= 2 adjacent columns
    P = (kx + maxkp - 1) / maxkp;
    int base = kx / P, rem = kx % P;
    for (int i = 0; i < P; i++) kps[i] = base + (i < rem ? 1 : 0);
    // stages run two per LDS round trip: a pass with an odd stage count pays a whole round trip for its last stage, so
    // pairs of odd passes trade one stage (8,7,7 -> 8,8,6: 11 round trips instead of 12)
    if (!getenv("ARK_HIP_FFT_BALANCED")) {
      for (int i = 0; i < P; i++)
        for (int j = P - 1; j > i; j--)
          if ((kps[i] & 1) && (kps[j] & 1) && kps[i] < maxkp && kps[j] > 2) {
            kps[i]++;
            kps[j]--;
            break;
          }
    }
    // the SHORT pass first (round 6): the first pass reads the stage-0 .. twiddles, all distinct, and a pass of fewer stages
    // takes more adjacent columns -- 2^22 as 6 + 8 + 8 stages is 0.8 % faster than 8 + 8 + 6 single, 2 % in a batch
    // (profiles/r6_fft_pass_order_ab.txt).  ARK_HIP_FFT_ASCENDING=0: the order of rounds 2-5 (longest first).
    {
      static const bool asc = [] { const char* e = getenv("ARK_HIP_FFT_ASCENDING"); return !(e && e[0] == '0'); }();
      if (asc && P <= 3)   // (four passes, 2^25 and up: measured 1.3 % slower that way -- left longest first)
        for (int i = 1; i < P; i++)   // insertion sort
          for (int j = i; j > 0 && kps[j - 1] > kps[j]; j--) {
            const int tmpk = kps[j];
            kps[j] = kps[j - 1];
            kps[j - 1] = tmpk;
          }
    }
    // an explicit plan (experiments): ARK_HIP_FFT_PLAN=12,10 -- stage counts per pass, each <= tile_log, summing to kx
    if (const char* pe = getenv("ARK_HIP_FFT_PLAN")) {
      int q[8], nq = 0, sum = 0;
      bool ok = true;
      for (const char* c = pe; *c && nq < 8;) {
        q[nq] = atoi(c);
        if (q[nq] < 1 || q[nq] > tile_log) ok = false;
        sum += q[nq++];
        while (*c && *c != ',') c++;
        if (*c == ',') c++;
      }
      if (ok && sum == kx && nq >= 1) {
        P = nq;
        for (int i = 0; i < P; i++) kps[i] = q[i];
      }
    }
    t = FFT_LANE_BITS;
  }
  u32* data = (u32*)d_data;
  u32* tmp = nullptr;
  u32* tmp9 = nullptr;   // carry-free passes: the 9th limbs of the ping buffer's elements (a 4-byte plane behind the 32-byte one)
  if (P > 1 || zlog) {
    DevBuf& tb = ws.tmps[stream];
    if (tb.ensure(n * (F::BYTES + 4))) return -3;
    tmp = (u32*)tb.p;
    tmp9 = tmp + n * F::N;
  }
  int s0 = zlog;
  for (int i = 0; i < P; i++) {
    FftPassArgs a;
    // columns per tile: keep every tile at 1024 elements (32 KiB of LDS, 4 elements per lane) -- passes with
    // fewer stages take more adjacent columns, i.e. longer contiguous segments
    int ti = t;
    if (P > 1) {
      ti = tile_log - kps[i];
      const int min_t = tile_log > 10 ? 0 : FFT_LANE_BITS;
      if (ti < min_t) ti = min_t;
      const int room = (i == P - 1) ? (k - kps[i]) : (k - s0 - kps[i]);  // bits available for columns
      if (ti > room) ti = room;
    } else if (tile_log > 10) {   // ONE executed pass of 9 .. 11 stages on the larger tiles: the columns that fit the tile and exist
      if (kps[i] + ti > tile_log) ti = tile_log - kps[i];
      if (ti > k - kps[i]) ti = k - kps[i];
    }
    a.k = k; a.s0 = s0; a.kp = kps[i]; a.t = ti; a.last = (i == P - 1) ? 1 : 0;
    a.zskip = (i == 0) ? zlog : 0;
    a.roots = roots;
    a.compact = (!lazy29 && k >= 2 && fft_compact_env()) ? 1 : 0;
    a.pre_lo = (i == 0) ? pre_lo : nullptr;
    a.pre_hi = (i == 0) ? pre_hi : nullptr;
    a.post_lo = a.last ? post_lo : nullptr;
    a.post_hi = a.last ? post_hi : nullptr;
    a.post_const = a.last ? post_const : nullptr;
    a.pre_full = (i == 0) ? pre_full : nullptr;
    a.post_full = a.last ? post_full : nullptr;
    const u32* src;
    u32* dst;
    if (P == 1 && zlog) { src = data; dst = tmp; }  // the compact input is read by every tile: not in place
    else if (P == 1) { src = data; dst = data; }
    else if (i == 0) { src = data; dst = tmp; }
    else if (i == P - 1) { src = tmp; dst = data; }
    else { src = tmp; dst = tmp; }
    u32 tiles = (u32)(n >> (kps[i] + ti));
    if (lazy29) {
      FftPass29Args pa;
      pa.a = a;
      pa.roots9 = roots9;
      pa.src9 = src == tmp ? tmp9 : nullptr;    // the caller's buffer holds canonical elements, the ping buffer 9 limbs
      pa.dst9 = dst == tmp ? tmp9 : nullptr;
      if (P == 1 && zlog) {   // single pass into the ping buffer: canonical out (it is the last pass), copied back below
        pa.dst9 = nullptr;
      }
      const size_t lds_bytes = ((size_t)1 << (kps[i] + ti)) * 36;
      hipLaunchKernelGGL((fft_pass29_kernel<FP>), dim3(tiles), dim3(256), lds_bytes, stream, src, dst, pa);
    } else {
      size_t lds_bytes = ((size_t)2 << (kps[i] + ti)) * sizeof(uint4);
      const int tl = kps[i] + ti;
      if (tl <= 10) {
        hipLaunchKernelGGL((fft_pass_kernel<FP, 256>), dim3(tiles), dim3(256), lds_bytes, stream, src, dst, a);
      } else if (tl == 11) {
        ARK_HIP_TRY(fft_allow_lds((const void*)fft_pass_kernel<FP, 512>, 64 << 10));
        hipLaunchKernelGGL((fft_pass_kernel<FP, 512>), dim3(tiles), dim3(512), lds_bytes, stream, src, dst, a);
      } else {
        ARK_HIP_TRY(fft_allow_lds((const void*)fft_pass_kernel<FP, 1024>, 128 << 10));
        hipLaunchKernelGGL((fft_pass_kernel<FP, 1024>), dim3(tiles), dim3(1024), lds_bytes, stream, src, dst, a);
      }
    }
    if (tm) ARK_HIP_TRY(hipEventRecord(ev[nev++], stream));
    s0 += kps[i];
  }
  if (P == 1 && zlog) ARK_HIP_TRY(hipMemcpyAsync(data, tmp, n * F::BYTES, hipMemcpyDeviceToDevice, stream));
  ARK_HIP_TRY(hipGetLastError());
  if (tm) {
    ARK_HIP_TRY(hipStreamSynchronize(stream));
    tm->npass = P;
    for (int i = 0; i < P; i++) (void)hipEventElapsedTime(&tm->pass[i], ev[i], ev[i + 1]);
    (void)hipEventElapsedTime(&tm->total, ev[0], ev[nev - 1]);
  }
  return 0;
}

}  // namespace arkhip
