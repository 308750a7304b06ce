// Pippenger multi-scalar multiplication on one MI355X.
//
// Replaces the reference's CPU bucket method on this path:
//   ec/src/scalar_mul/variable_base/mod.rs:59-85   (msm_unchecked / msm_bigint entry, into_bigint pass)
//   :437-503 msm_bigint_wnaf_parallel  (signed digits -> bucket accumulation -> running-sum
//            bucket reduction -> window combine), :754-794 make_digits,
//   ec/src/models/short_weierstrass/bucket.rs (XYZZ bucket arithmetic, see ec.cuh).
// It is a different algorithm organisation, chosen for the GPU, producing the same group element:
//
//   K1  msm_digits        one lane per scalar: (optional Montgomery->canonical), fold s -> r-s when that is
//                         smaller (negating the base instead; the GPU analogue of the reference's
//                         negative-small-scalar classes, mod.rs:251-285), signed base-2^c recoding exactly as
//                         make_digits: one key (sign | bucket) per (window, scalar).
//   K2  msm_part_*        (msm_sort.cuh) two-pass partition sort of the keys with LDS counters -> `sorted`
//                         point indices grouped by bucket slot, `offsets` = prefix sums of the bucket loads.
//   K3  msm_order_*       bucket slots ordered heaviest load class first (lanes of a wave get equal loads).
//   K4  msm_accumulate    one lane per bucket: gathers its bases (96 B random gathers run at ~3.5 TB/s on this
//                         chip) and sums them with XYZZ mixed additions on relaxed residues ([0, 2p)).
//   K4h msm_heavy_*       buckets too long for one lane (skewed scalars): one wave per 2048-entry chunk.
//   K5a msm_reduce_level  sum_k k*B_k per window, level 0: chunked running sums over <= 32 buckets per lane
//                         (parallel form of mod.rs:478-484).
//   K5b msm_reduce_bits   the rest, bit-sliced: log2(m)+1 independent masked sums per window.
//   host                  one Horner over the bit positions of all bit sums and window sums: <= 256 serial doublings -- a
//                         chain with no parallelism, run on the host in the same templated formulas (~0.2 ms).
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <mutex>
#include <atomic>
#include "hostpool.hpp"
#include <math.h>
#include "curves.cuh"
#include "ec28.cuh"
#include "lazyk.cuh"
#include "msm_sort.cuh"

namespace arkhip {

#define ARK_HIP_TRY(expr)                                                                          \
  do {                                                                                             \
    hipError_t _e = (expr);                                                                        \
    if (_e != hipSuccess) {                                                                        \
      fprintf(stderr, "ark_hip: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return -(int)_e - 1000;                                                                      \
    }                                                                                              \
  } while (0)

static constexpr u32 KEY_NONE = 0xffffffffu;

// Window widths: W windows of c bits, except that the top `narrow` windows are c-1 bits wide, so that the widths
// add up to the scalar's bit length exactly and no window is left with only a few significant bits.
__host__ __device__ __forceinline__ int msm_window_width(int w, int c, int W, int narrow) {
  return w >= W - narrow ? c - 1 : c;
}

// ---- K1: signed-digit recoding -------------------------------------------------------------------
// The magnitude the digits are taken from: v = min(s, r - s) of scalar i (canonical value; from Montgomery form first
// when `mont`), S::N + 1 words with a zero on top; returns the sign flip (bit 31) when r - s was used.
template <class SP>
__host__ __device__ __forceinline__ u32 msm_scalar_magnitude(const u32* __restrict__ scalars, u32 i, int mont, u32* v /*[N+1]*/,
                                                             u32* __restrict__ err) {
  typedef Fp<SP> S;
  S s = S::load(scalars + (size_t)i * S::N);
  if (mont) s = S::from_mont(s);  // mod.rs:60-62 into_bigint
  // msm_bigint accepts any BigInt<4>.  The reference's make_digits (mod.rs:754-794) covers ceil(BITS/c)*c bits, so it is
  // exact (as an integer multiple, hence mod r on the prime-order subgroup) for every s < 2^BITS and drops higher
  // bits in a window-size-dependent way: such scalars are rejected here (ARK_HIP_ERR_SCALAR_RANGE).
  // s in [r, 2^BITS): s -= r once (2^BITS < 2r for the three scalar fields), then the fold below applies.
  if constexpr (SP::BITS < 32 * S::N) {
    if ((s.l[S::N - 1] >> (SP::BITS - 32 * (S::N - 1))) != 0) {
#if defined(__HIP_DEVICE_COMPILE__)
      if (err) atomicOr(err, 1u);
#else
      (void)err;   // (host callers only measure widths)
#endif
      s = S::zero();
    }
  }
  {
    u32 u[S::N];
    u32 borrow = 0;
#pragma unroll
    for (int k = 0; k < S::N; k++) {
      u32 bo;
      u[k] = __builtin_subc(s.l[k], (u32)SP::P[k], borrow, &bo);
      borrow = bo;
    }
    if (!borrow) {
#pragma unroll
      for (int k = 0; k < S::N; k++) s.l[k] = u[k];
    }
  }
  // t = r - s ; use it (and negate the point) when t < s
  u32 t[S::N];
  {
    u32 borrow = 0;
#pragma unroll
    for (int k = 0; k < S::N; k++) {
      u64 x = (u64)SP::P[k] - s.l[k] - borrow;
      t[k] = (u32)x;
      borrow = (u32)(x >> 63);
    }
  }
  bool lt = false;  // t < s ?
#pragma unroll
  for (int k = 0; k < S::N; k++) {
    if (t[k] != s.l[k]) lt = t[k] < s.l[k];
  }
#pragma unroll
  for (int k = 0; k < S::N; k++) v[k] = lt ? t[k] : s.l[k];
  v[S::N] = 0;
  return lt ? 0x80000000u : 0u;
}

// K0: width probe of an msm_bigint call.  Over the sampled scalars (every `stride`-th), with b = the bit length of
// min(s, r - s):  out[0] <- max b,  out[1 + k] += number of scalars in width class k (MSM_WIDTH_TOP below).  A witness
// of 0 / 1 and small values, or the reference's u8 .. u64 bench distributions handed to msm_bigint, occupy the low
// windows only: planned for n full-width scalars they leave most windows empty and most of the 2^(c-1) W buckets -- whose
// reduction does not shrink with the entries -- unused.  The reference sorts the scalars into the same classes and runs
// one MSM per class (msm_signed, variable_base/mod.rs:251-336); here ONE pipeline runs, planned for the measured classes.
static constexpr int MSM_WIDTH_CLASSES = 9;
static constexpr int MSM_WIDTH_TOP[MSM_WIDTH_CLASSES] = {0, 1, 8, 16, 32, 64, 128, 192, 256};   // class k: TOP[k-1] < b <= TOP[k]
struct MsmWidths {
  u32 max_bits;
  u32 count[MSM_WIDTH_CLASSES];
};
// chunk > 0 (the exact pass, stride 1): workgroup b owns the contiguous scalars [b chunk, (b + 1) chunk) and also leaves the
// number of NON-ZERO ones among them in nz_counts[b] -- what the zero-scalar compaction below needs, for free
template <class SP>
__global__ void __launch_bounds__(1024) msm_scalar_bits_kernel(const u32* __restrict__ scalars, u32 n, u32 stride, int mont,
                                                               u32* __restrict__ out /*[1 + MSM_WIDTH_CLASSES]*/,
                                                               u32 chunk = 0, u32* __restrict__ nz_counts = nullptr) {
  typedef Fp<SP> S;
  __shared__ u32 blk[1 + MSM_WIDTH_CLASSES];
  if (threadIdx.x <= MSM_WIDTH_CLASSES) blk[threadIdx.x] = 0;
  __syncthreads();
  u32 bits = 0;
  u32 mine[MSM_WIDTH_CLASSES] = {0, 0, 0, 0, 0, 0, 0, 0, 0};   // lane 0 of the wave: its wave's counts
  const u64 step = chunk ? (u64)blockDim.x : (u64)gridDim.x * blockDim.x;
  const u64 first = chunk ? (u64)blockIdx.x * chunk + threadIdx.x : (u64)blockIdx.x * blockDim.x + threadIdx.x;
  const u64 rounds = chunk ? (u64)chunk / blockDim.x
                           : (((u64)n + stride - 1) / stride + step - 1) / step;   // the same trip count for every lane (ballots below)
  for (u64 it = 0; it < rounds; it++) {
    const u64 t = it * step + first;
    int cls = -1;
    if (t * stride < n) {
      u32 v[S::N + 1];
      (void)msm_scalar_magnitude<SP>(scalars, (u32)(t * stride), mont, v, nullptr);   // out of range: magnitude 0, the digits kernel reports it
      u32 b = 0;
#pragma unroll
      for (int k = 0; k < S::N; k++)
        if (v[k]) b = 32u * (u32)k + 32u - (u32)__builtin_clz(v[k]);
      bits = b > bits ? b : bits;
      cls = 0;
#pragma unroll
      for (int k = 1; k < MSM_WIDTH_CLASSES; k++) cls += b > (u32)MSM_WIDTH_TOP[k - 1] ? 1 : 0;
    }
#pragma unroll
    for (int k = 0; k < MSM_WIDTH_CLASSES; k++) mine[k] += (u32)__popcll(__ballot(cls == k));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const u32 other = (u32)__shfl_xor((int)bits, o);
    bits = other > bits ? other : bits;
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMax(&blk[0], bits);
#pragma unroll
    for (int k = 0; k < MSM_WIDTH_CLASSES; k++)
      if (mine[k]) atomicAdd(&blk[1 + k], mine[k]);
  }
  __syncthreads();
  if (nz_counts && threadIdx.x == 0) {
    u32 seen = 0;
#pragma unroll
    for (int k = 0; k < MSM_WIDTH_CLASSES; k++) seen += blk[1 + k];
    nz_counts[blockIdx.x] = seen - blk[1];   // scalars of this chunk whose magnitude is not zero
  }
  // one same-address atomic per wave serialises (2^18 waves: 3 ms, measured): one per workgroup and counter, and the
  // maximum only where it can still rise
  if (gridDim.x == 1) {   // the sample: one workgroup, plain stores, no zeroing beforehand
    if (threadIdx.x <= MSM_WIDTH_CLASSES) out[threadIdx.x] = blk[threadIdx.x];
  } else if (threadIdx.x == 0) {
    if (blk[0] > __atomic_load_n(out, __ATOMIC_RELAXED)) atomicMax(out, blk[0]);
  } else if (threadIdx.x <= MSM_WIDTH_CLASSES && blk[threadIdx.x]) {
    atomicAdd(out + threadIdx.x, blk[threadIdx.x]);
  }
}

// ---- K0c: zero scalars leave before the sort (round 5) ----------------------------------------------------------------
// A prover's witness is mostly zeros and ones (the bench's witness vector: 60 % zeros; a bool vector: half).  A zero scalar
// has no digit anywhere, yet its W KEY_NONE keys are written, histogrammed and scattered like any others: the sort of a
// 2^24-pair witness MSM moves 2^24 x 15 keys of which 7 % are live (1.35 ms of 6.5).  With the classes measured (K0's exact
// pass: the number of zero scalars is known on the host, and each of its chunks has left its non-zero count), the non-zero
// scalars are copied out in order with their indices, the pipeline runs on n' = n - zeros scalars, and the sort's first
// scatter (msm_part_scatter_kernel) writes base indices into its pairs -- a coalesced read of the index list there.  Same magnitude function as K0 and K1: "zero"
// is s = 0 (mod r); a scalar out of range counts as zero here too and raises the same flag K1 would have raised.
template <class SP>
__global__ void __launch_bounds__(1024) msm_compact_scalars_kernel(const u32* __restrict__ scalars, u32 n, int mont, u32 chunk,
                                                                   const u32* __restrict__ chunk_off, u32* __restrict__ err,
                                                                   u32* __restrict__ out_scalars, u32* __restrict__ out_idx, u32 cap) {
  typedef Fp<SP> S;
  __shared__ u32 wave_cnt[16];
  __shared__ u32 run_s;
  if (threadIdx.x == 0) run_s = chunk_off[blockIdx.x];
  __syncthreads();
  const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const u32 rounds = chunk / blockDim.x;
  for (u32 it = 0; it < rounds; it++) {
    const u64 i = (u64)blockIdx.x * chunk + (u64)it * blockDim.x + threadIdx.x;
    bool keep = false;
    if (i < n) {
      u32 v[S::N + 1];
      (void)msm_scalar_magnitude<SP>(scalars, (u32)i, mont, v, err);
      u32 any = 0;
#pragma unroll
      for (int k = 0; k < S::N; k++) any |= v[k];
      keep = any != 0;
    }
    const unsigned long long m = __ballot(keep);
    if (lane == 0) wave_cnt[wave] = (u32)__popcll(m);
    __syncthreads();
    u32 before = run_s;
    for (u32 w = 0; w < wave; w++) before += wave_cnt[w];
    if (keep) {
      const u32 pos = before + (u32)__popcll(m & ((1ull << lane) - 1ull));
      if (pos < cap) {
        const uint4* src = (const uint4*)(scalars + (size_t)i * S::N);
        uint4* dst = (uint4*)(out_scalars + (size_t)pos * S::N);
        dst[0] = src[0];
        dst[1] = src[1];
        out_idx[pos] = (u32)i;
      } else {
        atomicOr(err, 2u);   // cannot happen while K0 and this kernel agree on "zero"; never write past the buffers if they do not
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      u32 tot = 0;
      for (u32 w = 0; w < (blockDim.x >> 6); w++) tot += wave_cnt[w];
      run_s += tot;
    }
    __syncthreads();
  }
}
// The same classes for HOST scalars, estimated from a spread sample of about 1024 of them (the streamed entries plan
// before the first piece is uploaded and cannot wait for a device pass over all scalars): counts scaled to n, max_bits
// = the field's -- an estimate may choose the window size, never the number of windows.
template <class SP>
void msm_sample_widths_host(const void* h_scalars, size_t n, int mont, MsmWidths* out) {
  typedef Fp<SP> S;
  *out = MsmWidths{};
  out->max_bits = (u32)SP::BITS;
  if (n == 0) return;
  const size_t stride = n > 1024 ? n / 1024 : 1;   // (1024: a Montgomery -> canonical product per sample costs ~30 ns of host time)
  size_t seen = 0;
  u32 cnt[MSM_WIDTH_CLASSES] = {0};
  for (size_t i = 0; i < n; i += stride) {
    u32 v[S::N + 1];
    (void)msm_scalar_magnitude<SP>((const u32*)h_scalars, (u32)i, mont, v, nullptr);
    u32 b = 0;
    for (int k = 0; k < S::N; k++)
      if (v[k]) b = 32u * (u32)k + 32u - (u32)__builtin_clz(v[k]);
    int cls = 0;
    for (int k = 1; k < MSM_WIDTH_CLASSES; k++) cls += b > (u32)MSM_WIDTH_TOP[k - 1] ? 1 : 0;
    cnt[cls]++;
    seen++;
  }
  for (int k = 0; k < MSM_WIDTH_CLASSES; k++) out->count[k] = (u32)((double)cnt[k] * (double)n / (double)seen);
}
// plan of a plain MSM whose width classes were MEASURED over all n scalars (msm_enqueue after K0; ark_hip_msm_plan_widths):
// windows for the widest scalar unless fewer than 8 bits would be saved, window size from the digits the classes have
static inline struct MsmPlan msm_plan_for_widths(size_t n, int field_bits, double mul_cost, bool lazy28, const MsmWidths& w);
// true when the classes say "not n uniform full-width scalars": fewer than half wider than 128 bits
static inline bool msm_widths_skewed(const MsmWidths& w) {
  u64 seen = 0, wide = 0;
  for (int k = 0; k < MSM_WIDTH_CLASSES; k++) {
    seen += w.count[k];
    if (MSM_WIDTH_TOP[k] > 128) wide += w.count[k];
  }
  return 2 * wide < seen;
}

template <class SP>
__global__ void __launch_bounds__(256) msm_digits_kernel(const u32* __restrict__ scalars, u32 n, int mont, int c,
                                                         int W, int narrow, u32* __restrict__ keys,
                                                         u32* __restrict__ err) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  typedef Fp<SP> S;
  u32 v[S::N + 1];
  const u32 flip = msm_scalar_magnitude<SP>(scalars, i, mont, v, err);
  u32 carry = 0;
  for (int w = 0; w < W; w++) {
    const int cw = msm_window_width(w, c, W, narrow);  // this window's width
    const u32 mask = (1u << cw) - 1u;
    const u32 half = 1u << (cw - 1);
    u32 raw = (v[0] & mask) + carry;
    // shift the 256-bit register right by cw (cw < 32)
#pragma unroll
    for (int k = 0; k < S::N; k++) v[k] = (v[k] >> cw) | (v[k + 1] << (32 - cw));
    u32 key = KEY_NONE;
    if (w < W - 1) {
      carry = raw >= half ? 1u : 0u;  // mod.rs:783-786: carry = (digit + radix/2) >> c
    } else {
      carry = 0;
    }
    int d = (int)raw - (int)(carry << cw);
    if (d != 0) {
      u32 mag = d < 0 ? (u32)(-d) : (u32)d;
      if (mag > half) {  // unreachable after the range handling above; kept as a tripwire
        atomicOr(err, 1u);
        mag = half;
      }
      u32 sign = (d < 0 ? 0x80000000u : 0u) ^ flip;
      key = sign | (mag - 1);
    }
    keys[(size_t)w * n + i] = key;
  }
}

// ---- K1n: the same recoding for NARROW unsigned scalars ------------------------------------------------
// VariableBaseMSM::msm_u1 / msm_u8 / msm_u16 / msm_u32 / msm_u64 (variable_base/mod.rs:87-117; CPU bodies :373-434):
// the scalars arrive as BYTES-byte unsigned integers (bool = one byte, 0 / 1) and only the windows their `bits`
// significant bits reach exist at all -- no 32-byte expansion, no empty windows to sort.  No s -> r-s fold (the value is
// far below r/2); the widths add up to bits + 1, so the top window keeps the spare bit signed digits need.
template <int BYTES>
__global__ void __launch_bounds__(256) msm_digits_small_kernel(const unsigned char* __restrict__ scalars, u32 n, int c, int W,
                                                               int narrow, u64 vmask, u32* __restrict__ keys) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u64 v;
  if constexpr (BYTES == 1) v = scalars[i];
  else if constexpr (BYTES == 2) v = ((const unsigned short*)scalars)[i];
  else if constexpr (BYTES == 4) v = ((const u32*)scalars)[i];
  else v = ((const u64*)scalars)[i];
  v &= vmask;  // bits above the caller's bound do not exist (msm_u1: any non-zero byte of a bool is "true" -> see the host side)
  u32 carry = 0;
  for (int w = 0; w < W; w++) {
    const int cw = msm_window_width(w, c, W, narrow);
    const u32 mask = (1u << cw) - 1u;
    const u32 half = 1u << (cw - 1);
    const u32 raw = ((u32)v & mask) + carry;
    v >>= cw;
    carry = (w < W - 1 && raw >= half) ? 1u : 0u;
    const int d = (int)raw - (int)(carry << cw);
    u32 key = KEY_NONE;
    if (d != 0) {
      u32 mag = d < 0 ? (u32)(-d) : (u32)d;
      if (mag > half) mag = half;  // unreachable: the top window has a spare bit
      key = (d < 0 ? 0x80000000u : 0u) | (mag - 1);
    }
    keys[(size_t)w * n + i] = key;
  }
}

// ---- K2: exclusive scan (three small kernels) --------------------------------------------------
static constexpr int SCAN_TILE = 2048;  // elements per block (256 threads x 8)

static __global__ void __launch_bounds__(256) scan_block_sums(const u32* __restrict__ in, size_t m, u32* __restrict__ sums) {
  __shared__ u32 sh[256];
  size_t base = (size_t)blockIdx.x * SCAN_TILE;
  u32 s = 0;
  for (int k = 0; k < 8; k++) {
    size_t j = base + threadIdx.x + (size_t)k * 256;
    if (j < m) s += in[j];
  }
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) sums[blockIdx.x] = sh[0];
}
// single block: exclusive scan of nb block sums in place (u64 not needed: total < 2^32)
static __global__ void __launch_bounds__(1024) scan_sums_inplace(u32* sums, u32 nb) {
  __shared__ u32 sh[1024];
  __shared__ u32 carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (u32 base = 0; base < nb; base += 1024) {
    u32 j = base + threadIdx.x;
    u32 x = j < nb ? sums[j] : 0;
    sh[threadIdx.x] = x;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
      u32 y = (int)threadIdx.x >= o ? sh[threadIdx.x - o] : 0;
      __syncthreads();
      sh[threadIdx.x] += y;
      __syncthreads();
    }
    u32 incl = sh[threadIdx.x];
    u32 cbase = carry_s;
    if (j < nb) sums[j] = cbase + incl - x;
    __syncthreads();
    if (threadIdx.x == 1023) carry_s = cbase + incl;
    __syncthreads();
  }
}
// small arrays (m <= SCAN_SMALL_MAX): the whole exclusive scan in ONE workgroup -- the three-kernel form costs two
// dependent launches more than it computes at n <= 2^18.  offsets[m] = total.
static constexpr u32 SCAN_SMALL_MAX = 16384;
static __global__ void __launch_bounds__(1024) scan_small_kernel(const u32* __restrict__ in, u32 m, u32* __restrict__ offsets) {
  __shared__ u32 sh[1024];
  const u32 per = (m + 1023u) / 1024u;
  const u32 lo = threadIdx.x * per;
  u32 s = 0;
  for (u32 k = 0; k < per; k++)
    if (lo + k < m) s += in[lo + k];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (u32 o = 1; o < 1024; o <<= 1) {
    u32 y = threadIdx.x >= o ? sh[threadIdx.x - o] : 0;
    __syncthreads();
    sh[threadIdx.x] += y;
    __syncthreads();
  }
  u32 run = sh[threadIdx.x] - s;
  for (u32 k = 0; k < per; k++)
    if (lo + k < m) {
      offsets[lo + k] = run;
      run += in[lo + k];
    }
  if (threadIdx.x == 1023) offsets[m] = sh[1023];
}
// exclusive scan of m counters into offsets[0..m] (offsets[m] = total), `sums` = scratch of ceil(m / SCAN_TILE) words
static inline void scan_exclusive(const u32* in, size_t m, u32* sums, u32* offsets, hipStream_t stream);
// each block rescans its tile; writes offsets (exclusive); offsets[m] = total
static __global__ void __launch_bounds__(256) scan_apply(const u32* __restrict__ in, size_t m, const u32* __restrict__ sums,
                                                  u32* __restrict__ offsets) {
  __shared__ u32 sh[256];
  size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)threadIdx.x * 8;
  u32 x[8];
  u32 s = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    size_t j = base + k;
    x[k] = j < m ? in[j] : 0;
    s += x[k];
  }
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    u32 y = (int)threadIdx.x >= o ? sh[threadIdx.x - o] : 0;
    __syncthreads();
    sh[threadIdx.x] += y;
    __syncthreads();
  }
  u32 run = sums[blockIdx.x] + sh[threadIdx.x] - s;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    size_t j = base + k;
    if (j < m) offsets[j] = run;
    run += x[k];
    if (j == m - 1) offsets[m] = run;
  }
}

static inline void scan_exclusive(const u32* in, size_t m, u32* sums, u32* offsets, hipStream_t stream) {
  if (m <= SCAN_SMALL_MAX) {
    hipLaunchKernelGGL(scan_small_kernel, dim3(1), dim3(1024), 0, stream, in, (u32)m, offsets);
    return;
  }
  const u32 nblk = (u32)((m + SCAN_TILE - 1) / SCAN_TILE);
  hipLaunchKernelGGL(scan_block_sums, dim3(nblk), dim3(256), 0, stream, in, m, sums);
  hipLaunchKernelGGL(scan_sums_inplace, dim3(1), dim3(1024), 0, stream, sums, nblk);
  hipLaunchKernelGGL(scan_apply, dim3(nblk), dim3(256), 0, stream, in, m, sums, offsets);
}

// ---- K3b: bucket processing order, heaviest first -------------------------------------------------
// Lanes of one wave should own buckets of (nearly) equal load, otherwise the wave runs for its
// longest bucket (Poisson loads: ~30% of the lanes' time idle at mean 32).  Counting sort of bucket
// ids by load class (load >> shift, clamped to 255), descending; per-block LDS histograms keep the
// global atomics out of it.
static constexpr int ORDER_TILE = 2048;
static constexpr int ORDER_BINS = 256;
// A lane owns `wsum` runs (one per window when the windows share a bucket set, else 1) `wstride` slots apart.
__device__ __forceinline__ u32 msm_lane_load(const u32* __restrict__ offsets, size_t g, int wsum, size_t wstride) {
  u32 cnt = 0;
  for (int w = 0; w < wsum; w++) cnt += offsets[g + w * wstride + 1] - offsets[g + w * wstride];
  return cnt;
}
static __global__ void __launch_bounds__(256) msm_order_hist_kernel(const u32* __restrict__ offsets, size_t nb, int shift,
                                                                    u32 nblocks, int wsum, size_t wstride,
                                                                    u32* __restrict__ block_hist) {
  __shared__ u32 sh[ORDER_BINS];
  sh[threadIdx.x] = 0;
  __syncthreads();
  size_t base = (size_t)blockIdx.x * ORDER_TILE;
  for (int k = 0; k < ORDER_TILE / 256; k++) {
    size_t g = base + threadIdx.x + (size_t)k * 256;
    if (g < nb) {
      u32 cls = msm_lane_load(offsets, g, wsum, wstride) >> shift;
      if (cls > ORDER_BINS - 1) cls = ORDER_BINS - 1;
      atomicAdd(&sh[ORDER_BINS - 1 - cls], 1u);  // class 0 of the output = heaviest
    }
  }
  __syncthreads();
  block_hist[(size_t)threadIdx.x * nblocks + blockIdx.x] = sh[threadIdx.x];  // bin-major
}
static __global__ void __launch_bounds__(256) msm_order_scatter_kernel(const u32* __restrict__ offsets, size_t nb, int shift,
                                                                       u32 nblocks, int wsum, size_t wstride,
                                                                       const u32* __restrict__ block_off,
                                                                       u32* __restrict__ order) {
  __shared__ u32 sh[ORDER_BINS];
  sh[threadIdx.x] = block_off[(size_t)threadIdx.x * nblocks + blockIdx.x];
  __syncthreads();
  size_t base = (size_t)blockIdx.x * ORDER_TILE;
  for (int k = 0; k < ORDER_TILE / 256; k++) {
    size_t g = base + threadIdx.x + (size_t)k * 256;
    if (g < nb) {
      u32 cls = msm_lane_load(offsets, g, wsum, wstride) >> shift;
      if (cls > ORDER_BINS - 1) cls = ORDER_BINS - 1;
      u32 pos = atomicAdd(&sh[ORDER_BINS - 1 - cls], 1u);
      order[pos] = (u32)g;
    }
  }
}

// ---- K4: bucket accumulation --------------------------------------------------------------------
// One lane per bucket.  `order` (optional) maps lane -> bucket id so that lanes of one wave own
// buckets of similar load.  accum != 0: a later piece of a streamed MSM whose pieces share one bucket array
// (MsmPiece): the bucket's previous sum is the starting value instead of infinity.
template <class C>
__global__ void __launch_bounds__(256, C::ACC_MIN_WAVES) msm_accumulate_kernel(const char* __restrict__ bases,
                                                             const u32* __restrict__ sorted,
                                                             const u32* __restrict__ offsets,
                                                             const u32* __restrict__ order, u32 nbuckets,
                                                             const u32* __restrict__ d_thresh, int HB, int LB,
                                                             int accum, char* __restrict__ buckets) {
  typedef typename C::FA F;  // Fp, or Fp2Half: then a lane PAIR owns the bucket (both lanes run the same control flow)
  u32 t = (blockIdx.x * blockDim.x + threadIdx.x) / F::LANES;
  if (t >= nbuckets) return;
  u32 g = order ? order[t] : t;
  u32 j = offsets[g], end = offsets[g + 1];
  if (end - j > *d_thresh) return;  // left to the heavy-bucket kernels
  if (accum && j == end) return;    // nothing to add to the stored sum
  char* cell = buckets + (size_t)msm_slot_to_bucket(g, HB, LB) * XYZZ<F>::BYTES;  // g is a slot (msm_sort.cuh)
  XYZZ<F> acc = accum ? XYZZ<F>::load(cell) : XYZZ<F>::zero();   // stored buckets are canonical: valid relaxed values
  if (j < end) {
    // software pipeline, two deep on the indices: the gather of point t+1 (whose index arrived an iteration ago) and
    // the index of point t+2 are both in flight during the ~10 multiplications of addition t
    u32 e = sorted[j];
    u32 e1 = j + 1 < end ? sorted[j + 1] : 0;
    Affine<F> p = Affine<F>::load(bases + (size_t)(e & 0x7fffffffu) * Affine<F>::BYTES);
    for (;;) {
      u32 e2 = 0;
      Affine<F> p_next = p;
      const bool more = j + 1 < end;
      if (more) {
        p_next = Affine<F>::load(bases + (size_t)(e1 & 0x7fffffffu) * Affine<F>::BYTES);
        if (j + 2 < end) e2 = sorted[j + 2];
      }
      if (!p.is_zero()) {  // identity base contributes nothing (bucket.rs:171-173)
        F y = p.y;  // negative digit: -P.  On relaxed residues 2p - y serves (a multiplication operand; no zero test)
        if constexpr (C::RELAXED_A) y = (e >> 31) != 0 ? F::neg_r(p.y) : p.y;
        else y = F::cond_neg(p.y, (e >> 31) != 0);
        if constexpr (C::RELAXED_A) xyzz_madd_relaxed<F>(acc, p.x, y);
        else xyzz_madd<F>(acc, p.x, y);
      }
      if (!more) break;
      e = e1;
      e1 = e2;
      p = p_next;
      j++;
    }
  }
  if constexpr (C::RELAXED_A) acc = xyzz_canonical<F>(acc);
  acc.store(cell);
}

// ---- K4 on carry-free 28-bit limbs (Fp384 G1 curves: C::LAZY_A) -----------------------------------------------
#ifndef ARK_LAZY_MIN_WAVES
#define ARK_LAZY_MIN_WAVES 2   // waves per SIMD the register allocation must leave room for
#endif
// The same kernel with the accumulator and every product of the mixed addition in fp28.cuh's form: 392 v_mad_u64_u32 per
// product instead of 288 + 288 carry instructions, no conditional subtractions.  Bases are gathered in the reference's
// canonical limbs and only repacked (the radix change is a curve isomorphism, ec28.cuh); buckets leave canonical.
template <class C>
__global__ void __launch_bounds__(256, ARK_LAZY_MIN_WAVES) msm_accumulate_lazy_kernel(const char* __restrict__ bases,
                                                                  const u32* __restrict__ sorted,
                                                                  const u32* __restrict__ offsets,
                                                                  const u32* __restrict__ order, u32 nbuckets,
                                                                  const u32* __restrict__ d_thresh, int HB, int LB,
                                                                  int accum, char* __restrict__ buckets) {
  typedef LazyK<C> K;            // prime-field curves: one lane per bucket; G2: one lane PAIR (uniform control flow)
  typedef typename K::FM F;
  u32 t = (blockIdx.x * blockDim.x + threadIdx.x) / K::LANES;
  if (t >= nbuckets) return;
  u32 g = order ? order[t] : t;
  u32 j = offsets[g], end = offsets[g + 1];
  if (end - j > *d_thresh) return;  // left to the heavy-bucket kernels
  if (accum && j == end) return;    // nothing to add to the stored sum
  char* cell = buckets + (size_t)msm_slot_to_bucket(g, HB, LB) * XYZZ<F>::BYTES;
  typename K::Acc acc = K::inf();
  if (accum) acc = K::from_bucket(XYZZ<F>::load(cell));
  if (j < end) {
    u32 e = sorted[j];
    u32 e1 = j + 1 < end ? sorted[j + 1] : 0;
    Affine<F> p = Affine<F>::load(bases + (size_t)(e & 0x7fffffffu) * Affine<F>::BYTES);
    for (;;) {
      u32 e2 = 0;
      Affine<F> p_next = p;
      const bool more = j + 1 < end;
      if (more) {
        p_next = Affine<F>::load(bases + (size_t)(e1 & 0x7fffffffu) * Affine<F>::BYTES);
        if (j + 2 < end) e2 = sorted[j + 2];
      }
      if (!p.is_zero()) {  // identity base contributes nothing (bucket.rs:171-173)
        if (K::madd(acc, p, (e >> 31) != 0))   // the base equals the accumulated point (duplicate bases): doubling, from a re-read base
          K::mdbl(acc, bases + (size_t)(e & 0x7fffffffu) * Affine<F>::BYTES, (e >> 31) != 0);
      }
      if (!more) break;
      e = e1;
      e1 = e2;
      p = p_next;
      j++;
    }
  }
  K::to_bucket(acc).store(cell);
}
// The same walk with every run cut into `parts` equal pieces, one lane each (narrow scalars at large n: msm_u16's ONE
// window has 2^16 runs of n / 2^16 entries -- half a wave per SIMD walking 1024 entries each at 2^26; i16 half of that).
// Lane (part, t) sums piece `part` of run order[t] into parts[part][bucket]; msm_sum_parts_kernel adds a run's pieces into
// its bucket.  Heavy runs are skipped by both, as in the kernel above.  Parts are chosen on the host (msm_enqueue) only
// where the slots cannot fill the chip's lanes and the runs are long; uniform 255-bit scalars never come here.
template <class C>
__global__ void __launch_bounds__(256, ARK_LAZY_MIN_WAVES) msm_accumulate_parts_kernel(const char* __restrict__ bases,
                                                                  const u32* __restrict__ sorted,
                                                                  const u32* __restrict__ offsets,
                                                                  const u32* __restrict__ order, u32 nbuckets,
                                                                  const u32* __restrict__ d_thresh, int HB, int LB,
                                                                  u32 nparts, char* __restrict__ parts) {
  typedef LazyK<C> K;
  typedef typename K::FM F;
  const u32 gid = (blockIdx.x * blockDim.x + threadIdx.x) / K::LANES;
  const u32 part = gid / nbuckets, t = gid - part * nbuckets;
  if (part >= nparts) return;
  const u32 g = order ? order[t] : t;
  const u32 start = offsets[g], stop = offsets[g + 1];
  const u32 len = stop - start;
  if (len > *d_thresh) return;  // left to the heavy-bucket kernels
  const u32 per = (len + nparts - 1) / nparts;
  const u32 a = part * per, b = a + per;
  u32 j = start + (a < len ? a : len);
  const u32 end = start + (b < len ? b : len);
  char* cell = parts + ((size_t)part * nbuckets + msm_slot_to_bucket(g, HB, LB)) * XYZZ<F>::BYTES;
  typename K::Acc acc = K::inf();
  if (j < end) {
    u32 e = sorted[j];
    u32 e1 = j + 1 < end ? sorted[j + 1] : 0;
    Affine<F> p = Affine<F>::load(bases + (size_t)(e & 0x7fffffffu) * Affine<F>::BYTES);
    for (;;) {
      u32 e2 = 0;
      Affine<F> p_next = p;
      const bool more = j + 1 < end;
      if (more) {
        p_next = Affine<F>::load(bases + (size_t)(e1 & 0x7fffffffu) * Affine<F>::BYTES);
        if (j + 2 < end) e2 = sorted[j + 2];
      }
      if (!p.is_zero()) {
        if (K::madd(acc, p, (e >> 31) != 0))
          K::mdbl(acc, bases + (size_t)(e & 0x7fffffffu) * Affine<F>::BYTES, (e >> 31) != 0);
      }
      if (!more) break;
      e = e1;
      e1 = e2;
      p = p_next;
      j++;
    }
  }
  K::to_bucket(acc).store(cell);
}
template <class C>
__global__ void __launch_bounds__(256) msm_sum_parts_kernel(const u32* __restrict__ offsets, u32 nbuckets,
                                                            const u32* __restrict__ d_thresh, int HB, int LB, u32 nparts,
                                                            const char* __restrict__ parts, int accum,
                                                            char* __restrict__ buckets) {
  typedef LazyK<C> K;
  typedef typename K::FM F;
  const u32 g = (blockIdx.x * blockDim.x + threadIdx.x) / K::LANES;
  if (g >= nbuckets) return;
  const u32 len = offsets[g + 1] - offsets[g];
  if (len > *d_thresh) return;
  if (accum && len == 0) return;
  const size_t bkt = msm_slot_to_bucket(g, HB, LB);
  char* cell = buckets + bkt * XYZZ<F>::BYTES;
  typename K::Acc acc = K::inf();
  if (accum) acc = K::from_bucket(XYZZ<F>::load(cell));
  for (u32 q = 0; q < nparts; q++) K::add(acc, XYZZ<F>::load(parts + ((size_t)q * nbuckets + bkt) * XYZZ<F>::BYTES));
  K::to_bucket(acc).store(cell);
}


template <class C>
__global__ void __launch_bounds__(256, ARK_LAZY_MIN_WAVES) msm_accumulate_shared_lazy_kernel(
    const char* __restrict__ table, size_t wstride, const u32* __restrict__ sorted, const u32* __restrict__ offsets,
    const u32* __restrict__ order, u32 nbuckets, int W, int B, const u32* __restrict__ d_thresh, int HB, int LB,
    char* __restrict__ buckets) {
  typedef LazyK<C> K;
  typedef typename K::FM F;
  u32 t = (blockIdx.x * blockDim.x + threadIdx.x) / K::LANES;
  if (t >= nbuckets) return;
  const u32 s = order ? order[t] : t;
  const u32 heavy_thresh = *d_thresh;
  typename K::Acc acc = K::inf();
  int w = 0;
  u32 na = offsets[s], nb2 = offsets[s + 1];
  u32 j = 0, end = 0;
  const char* wbase = table;
  auto open_next = [&]() -> bool {
    while (w < W) {
      const u32 a = na, b = nb2;
      const char* wb = table + (size_t)w * wstride * Affine<F>::BYTES;
      w++;
      if (w < W) {
        const u32 g = ((u32)w << B) | s;
        na = offsets[g];
        nb2 = offsets[g + 1];
      }
      if (b - a > heavy_thresh) continue;
      if (b > a) {
        j = a;
        end = b;
        wbase = wb;
        return true;
      }
    }
    return false;
  };
  auto next_entry = [&](u32& eo, const char*& wbo) -> bool {
    if (j >= end && !open_next()) return false;
    eo = sorted[j];
    wbo = wbase;
    j++;
    return true;
  };
  u32 e = 0, e1 = 0;
  const char *wb = table, *wb1 = table;
  if (next_entry(e, wb)) {
    Affine<F> p = Affine<F>::load(wb + (size_t)(e & 0x7fffffffu) * Affine<F>::BYTES);
    bool have1 = next_entry(e1, wb1);
    for (;;) {
      u32 e2 = 0;
      const char* wb2 = table;
      bool have2 = false;
      Affine<F> p_next = p;
      if (have1) {
        p_next = Affine<F>::load(wb1 + (size_t)(e1 & 0x7fffffffu) * Affine<F>::BYTES);
        have2 = next_entry(e2, wb2);
      }
      if (!p.is_zero()) {
        if (K::madd(acc, p, (e >> 31) != 0))   // the base equals the accumulated point (duplicate bases): doubling, from a re-read base
          K::mdbl(acc, wb + (size_t)(e & 0x7fffffffu) * Affine<F>::BYTES, (e >> 31) != 0);
      }
      if (!have1) break;
      e = e1;
      p = p_next;
      e1 = e2;
      wb1 = wb2;
      have1 = have2;
    }
  }
  K::to_bucket(acc).store(buckets + (size_t)msm_slot_to_bucket(s, HB, LB) * XYZZ<F>::BYTES);
}

// ---- K4s: bucket accumulation over a PREPARED base set ---------------------------------------------
// With the per-window multiples T_w[i] = 2^(offset_w) P_i precomputed once for a fixed base set (an SRS), every
// window's digit d contributes d * T_w[i] with weight 1: all W windows share ONE set of 2^(c-1) buckets, so the
// bucket reduction shrinks W-fold and c can grow (fewer windows, fewer mixed additions) -- 288 GB of HBM pay for
// the W-fold larger table.  The sort is unchanged (runs per (window, bucket) slot); a lane owns bucket slot s and
// walks its W runs, gathering from the window's table.  Runs too long for one lane (`heavy`) are skipped here,
// summed by the K4h kernels and added to the bucket afterwards (msm_apply_heavy_kernel).
template <class C>
__global__ void __launch_bounds__(256, C::ACC_MIN_WAVES) msm_accumulate_shared_kernel(
    const char* __restrict__ table, size_t wstride, const u32* __restrict__ sorted, const u32* __restrict__ offsets,
    const u32* __restrict__ order, u32 nbuckets, int W, int B, const u32* __restrict__ d_thresh, int HB, int LB,
    char* __restrict__ buckets) {
  typedef typename C::FA F;  // Fp, or Fp2Half: then a lane PAIR owns the bucket
  typedef XYZZ<F> Pt;
  u32 t = (blockIdx.x * blockDim.x + threadIdx.x) / F::LANES;
  if (t >= nbuckets) return;
  const u32 s = order ? order[t] : t;
  const u32 heavy_thresh = *d_thresh;
  Pt acc = Pt::zero();
  int w = 0;                    // next window to open
  u32 na = offsets[s], nb2 = offsets[s + 1];  // bounds of window w's run, fetched one window ahead
  u32 j = 0, end = 0;
  const char* wbase = table;
  auto open_next = [&]() -> bool {  // (j, end, wbase) <- the next non-empty run; heavy runs are folded on the way
    while (w < W) {
      const u32 a = na, b = nb2;
      const char* wb = table + (size_t)w * wstride * Affine<F>::BYTES;
      w++;
      if (w < W) {
        const u32 g = ((u32)w << B) | s;
        na = offsets[g];
        nb2 = offsets[g + 1];
      }
      if (b - a > heavy_thresh) continue;  // summed by the K4h kernels, added by msm_apply_heavy_kernel
      if (b > a) {
        j = a;
        end = b;
        wbase = wb;
        return true;
      }
    }
    return false;
  };
  // the run iterator runs two entries ahead of the additions: (e1, wb1) is entry t+1, whose gather is issued at the
  // top of iteration t with an index that arrived an iteration ago; the index of entry t+2 is fetched meanwhile
  auto next_entry = [&](u32& eo, const char*& wbo) -> bool {
    if (j >= end && !open_next()) return false;
    eo = sorted[j];
    wbo = wbase;
    j++;
    return true;
  };
  u32 e = 0, e1 = 0;
  const char *wb = table, *wb1 = table;
  if (next_entry(e, wb)) {
    Affine<F> p = Affine<F>::load(wb + (size_t)(e & 0x7fffffffu) * Affine<F>::BYTES);
    bool have1 = next_entry(e1, wb1);
    for (;;) {
      u32 e2 = 0;
      const char* wb2 = table;
      bool have2 = false;
      Affine<F> p_next = p;
      if (have1) {
        p_next = Affine<F>::load(wb1 + (size_t)(e1 & 0x7fffffffu) * Affine<F>::BYTES);
        have2 = next_entry(e2, wb2);
      }
      if (!p.is_zero()) {
        F y = p.y;  // negative digit: -P.  On relaxed residues 2p - y serves (a multiplication operand; no zero test)
        if constexpr (C::RELAXED_A) y = (e >> 31) != 0 ? F::neg_r(p.y) : p.y;
        else y = F::cond_neg(p.y, (e >> 31) != 0);
        if constexpr (C::RELAXED_A) xyzz_madd_relaxed<F>(acc, p.x, y);
        else xyzz_madd<F>(acc, p.x, y);
      }
      if (!have1) break;
      e = e1;
      p = p_next;
      e1 = e2;
      wb1 = wb2;
      have1 = have2;
    }
  }
  if constexpr (C::RELAXED_A) acc = xyzz_canonical<F>(acc);
  acc.store(buckets + (size_t)msm_slot_to_bucket(s, HB, LB) * Pt::BYTES);
}

// T_{w+1}[i] = 2^width * T_w[i]: affine in, XYZZ out (msm_prepare_table normalises a whole row with the lane-batched
// inversion of ec.cuh -- one Fermat inversion per 8 points instead of one per point; a one-time cost per base set)
template <class C>
__global__ void __launch_bounds__(128) msm_table_step_kernel(const char* __restrict__ in, char* __restrict__ out, size_t n,
                                                             int width) {
  typedef typename C::F F;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Affine<F> p = Affine<F>::load(in + i * Affine<F>::BYTES);
  XYZZ<F> acc = XYZZ<F>::zero();
  if (!p.is_zero()) {
    acc = xyzz_mdbl<F>(p.x, p.y);
    for (int k = 1; k < width; k++) acc = xyzz_dbl<F>(acc);
  }
  acc.store(out + i * XYZZ<F>::BYTES);
}

// ---- K4h: heavy buckets --------------------------------------------------------------------------
// A bucket far above the mean load (non-uniform scalars: the reference's bool / u8 / ... benches, or
// many equal scalars) would pin one lane for its whole length.  Buckets above `thresh` are skipped by
// the lane-per-bucket kernel, cut into chunks of HEAVY_CHUNK entries, each chunk summed by one
// wave (64 lanes striding + LDS tree), and the chunk partials combined per bucket.
#ifndef ARK_HEAVY_CHUNK
#define ARK_HEAVY_CHUNK 1024          // entries per wave of msm_heavy_partial_kernel: 16 serial additions per lane + the wave's
                                      // LDS tree (2048 / 64 lanes before: bool 1.85 -> 1.71 ms, u8 1.42 -> 1.21, msm_u8 1.20 -> 0.96
                                      // at 2^20 with the 128-lane combine below; 512 / 256 measured between the two -- profiles/r4_heavy_geometry_ab.txt)
#endif
#ifndef ARK_HEAVY_COMBINE_THREADS
#define ARK_HEAVY_COMBINE_THREADS 128 // workgroup of msm_heavy_combine_kernel: one per heavy run
#endif
static constexpr u32 HEAVY_CHUNK = ARK_HEAVY_CHUNK;
struct HeavyEntry { u32 bucket, first_item, items; };
// Entries per chunk for a window group with `total` sorted entries: the wave's tree (6 full additions) and the combine
// kernel's serial walk over a run's partials are fixed costs per chunk, so large inputs take longer chunks -- a 2^23-entry
// run (half of a 2^24 witness being ones) in 1024-entry chunks leaves 8192 partials to ONE 128-lane combine workgroup
// (1.03 ms of 71 dependent additions) after a partial kernel that spent a third of its additions in trees.
__host__ __device__ __forceinline__ u32 msm_heavy_chunk(u32 total) {
  return total >= (1u << 23) ? 4u * HEAVY_CHUNK : total >= (1u << 22) ? 2u * HEAVY_CHUNK : HEAVY_CHUNK;
}

// A run is "heavy" when one lane walking it (~28 us per entry) would outlast the whole accumulate kernel, whose
// throughput-bound duration is (non-zero entries) / 5.5e9 s: threshold = entries / 154 000, at least 64 and at least
// 4 x the mean run.  The entry count is only known after the sort (zero digits are dropped: small scalars leave most
// windows empty), so the threshold is computed on the device.  ctr[2] <- threshold.
__device__ __forceinline__ u32 msm_heavy_threshold(const u32* __restrict__ offsets, u32 nslots, u32 forced) {
  const u32 total = offsets[nslots];
  u32 t = total / 154000u;
  if (t < 64u) t = 64u;
  // ... and at least 4 x the mean run -- where the slots can fill the chip's lanes at all.  With a few thousand slots
  // (narrow scalars in one narrow window: msm_u8 with c = 9 has 256) every run is long, none is 4 x the mean, and 256
  // lanes walking 4096 entries each took 39 ms at 2^20 where the chunked kernels need 0.5 ms.
  const u32 mr = 4u * (total / nslots);
  if (t < mr && nslots >= 32768u) t = mr;
  return forced ? forced : t;
}

static __global__ void __launch_bounds__(256) msm_find_heavy_kernel(const u32* __restrict__ offsets, u32 nbuckets,
                                                                    u32 forced, u32* __restrict__ ctr /*[3]*/,
                                                                    HeavyEntry* __restrict__ list,
                                                                    uint2* __restrict__ items) {
  u32 g = blockIdx.x * blockDim.x + threadIdx.x;
  const u32 thresh = msm_heavy_threshold(offsets, nbuckets, forced);  // the same value in every lane
  if (g == 0) ctr[2] = thresh;                                        // for the kernels that follow
  const u32 chunk = msm_heavy_chunk(offsets[nbuckets]);
  u32 k = 0, first = 0;
  if (g < nbuckets) {
    const u32 cnt = offsets[g + 1] - offsets[g];
    if (cnt > thresh) {
      k = (cnt + chunk - 1) / chunk;
      first = atomicAdd(&ctr[0], k);
      const u32 slot = atomicAdd(&ctr[1], 1u);
      list[slot] = HeavyEntry{g, first, k};
    }
  }
  // the chunk items of a run: a few by its own lane, many (one bucket holding half of a 2^24 witness: 2048 chunks) by the wave
  if (k && k <= 16)
    for (u32 q = 0; q < k; q++) items[first + q] = make_uint2(g, q);
  unsigned long long big = __ballot(k > 16);
  while (big) {
    const int src = __ffsll((long long)big) - 1;
    big &= big - 1;
    const u32 kk = (u32)__shfl((int)k, src), ff = (u32)__shfl((int)first, src), gg = (u32)__shfl((int)g, src);
    for (u32 q = threadIdx.x & 63u; q < kk; q += 64u) items[ff + q] = make_uint2(gg, q);
  }
}

// Point additions of the reduction / heavy-run kernels.  `Pt` is a point as it lives in memory (canonical XYZZ over the
// accumulate field C::FA), `Acc` the form it is summed in:
//   * saturated limbs (BN254, the Fp2 curves): Acc = Pt on relaxed residues where the field allows it, one element per
//     lane (Fp) or per lane PAIR (Fp2Half; index arithmetic below is in "slots" = lanes / LANES, every branch uniform
//     over a pair);
//   * carry-free 28-bit limbs (the Fp384 G1 curves, C::LAZY_A): Acc = XYZZL, memory operands enter by the shifted
//     repack (ec28.cuh: xyzz_add_lazy) -- the full addition's 10 products + 2 squares + one two-product sum at the
//     28-bit product's rate instead of 12 + 2 + 1.5 at the saturated one's.
#ifndef ARK_REDUCE_LAZY
#define ARK_REDUCE_LAZY 1   // 0: the reduction / heavy-run kernels of the Fp384 G1 curves on saturated limbs too (A/B builds)
#endif
template <class C, bool LAZY = (C::LAZY_A && ARK_REDUCE_LAZY != 0)>
struct AccOps;

template <class C>
struct AccOps<C, false> {
  typedef typename C::FA F;
  typedef XYZZ<F> Pt;
  typedef Pt Acc;
  static constexpr u32 LANES = F::LANES;
  static constexpr size_t ACC_BYTES = Pt::BYTES;   // one accumulator parked in LDS
  ARK_DEV static Acc zero() { return Pt::zero(); }
  ARK_DEV static Acc from_pt(const Pt& p) { return p; }
  // acc += (+-) the non-identity base p gathered from `src` (neg: the digit's sign)
  ARK_DEV static void madd(Acc& acc, const Affine<F>& p, bool neg, const char*) {
    if constexpr (C::RELAXED_A) xyzz_madd_relaxed<F>(acc, p.x, neg ? F::neg_r(p.y) : p.y);   // 2p - y: a multiplication operand
    else xyzz_madd<F>(acc, p.x, F::cond_neg(p.y, neg));
  }
  ARK_DEV static void add(Acc& acc, const Pt& b) {
    if constexpr (C::RELAXED_A) xyzz_add_relaxed<F>(acc, b);
    else xyzz_add<F>(acc, b);
  }
  ARK_DEV static void add_acc(Acc& acc, const Acc& b) { add(acc, b); }
  ARK_DEV static Pt fin(const Acc& a) {  // canonical form for memory
    if constexpr (C::RELAXED_A) return xyzz_canonical<F>(a);
    else return a;
  }
  ARK_DEV static void park(const Acc& a, char* slot) { a.store(slot); }
  ARK_DEV static Acc unpark(const char* slot) { return Pt::load(slot); }
  // sum over the `width` slots of an LDS array of per-slot partials (width a power of two); result in slot 0's acc
  ARK_DEV static void tree(Acc& acc, char* sh, u32 slot, u32 width) {
    park(acc, sh + (size_t)slot * ACC_BYTES);
    __syncthreads();
    for (u32 o = width / 2; o > 0; o >>= 1) {
      if (slot < o) {
        add_acc(acc, unpark(sh + (size_t)(slot + o) * ACC_BYTES));
        park(acc, sh + (size_t)slot * ACC_BYTES);
      }
      __syncthreads();
    }
  }
};

template <class C>
struct AccOps<C, true> {
  typedef LazyK<C> K;            // carry-free limbs: ec28.cuh (one lane per point) or ec28x2.cuh (G2: one lane pair)
  typedef typename K::FM F;
  typedef XYZZ<F> Pt;
  typedef typename K::Acc Acc;
  static constexpr u32 LANES = K::LANES;
  static constexpr size_t ACC_BYTES = ((size_t)K::WORDS * 4 + 15) / 16 * 16;   // G1: 4 x 14 limbs + flag = 240 B; G2: 464 B per pair
  ARK_DEV static Acc zero() { return K::inf(); }
  ARK_DEV static Acc from_pt(const Pt& p) { return K::from_bucket(p); }
  // acc += (+-) the non-identity base p gathered from `src`; the doubling of equal points RE-READS the base, as the accumulate
  // kernels do: nothing of the addition has to stay alive for a branch that almost never comes (inlined with the repacked x / y
  // kept alive, the heavy-run kernel fell to one wave per SIMD; out of line, its accumulator lived in scratch memory)
  ARK_DEV static void madd(Acc& acc, const Affine<F>& p, bool neg, const char* src) {
    if (K::madd(acc, p, neg)) K::mdbl(acc, src, neg);
  }
  ARK_DEV static void add(Acc& acc, const Pt& b) { K::add(acc, b); }
  ARK_DEV static void add_acc(Acc& acc, const Acc& b) { K::add_acc(acc, b); }
  ARK_DEV static Pt fin(const Acc& a) { return K::to_bucket(a); }
  ARK_DEV static void park(const Acc& a, char* slot) { K::park(a, slot); }
  ARK_DEV static Acc unpark(const char* slot) { return K::unpark(slot); }
  ARK_DEV static void tree(Acc& acc, char* sh, u32 slot, u32 width) {
    park(acc, sh + (size_t)slot * ACC_BYTES);
    __syncthreads();
    for (u32 o = width / 2; o > 0; o >>= 1) {
      if (slot < o) {
        add_acc(acc, unpark(sh + (size_t)(slot + o) * ACC_BYTES));
        park(acc, sh + (size_t)slot * ACC_BYTES);
      }
      __syncthreads();
    }
  }
};

// one WAVE per chunk: its slots stride over the chunk's entries, then an LDS tree inside the wave's own LDS region.
// All waves of a workgroup run the same number of rounds so the barriers stay uniform.
template <class C>
__global__ void __launch_bounds__(256, (C::LAZY_A && C::FA::LANES == 2) ? 1 : 2) msm_heavy_partial_kernel(const char* __restrict__ bases,
                                                                const u32* __restrict__ sorted,
                                                                const u32* __restrict__ offsets,
                                                                const u32* __restrict__ ctr,
                                                                const uint2* __restrict__ items, size_t wstride, int B,
                                                                u32 nslots, char* __restrict__ partials) {
  typedef AccOps<C> Ops;
  typedef typename Ops::F F;           // the field as it lives in memory (Fp, or the lane-pair Fp2Half)
  typedef typename Ops::Pt Pt;
  constexpr u32 NS = 64 / Ops::LANES;  // slots per wave
  extern __shared__ uint4 heavy_lds[];
  // the wave's index is wave-uniform: told to the compiler, the item, its bounds and its table pointer live in scalar registers
  // (with the doubling expanded in place the kernel needs every vector register it can get to keep two waves per SIMD)
  const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), slot = (threadIdx.x & 63) / Ops::LANES, wpb = blockDim.x >> 6;
  char* sh = (char*)heavy_lds + (size_t)wave * NS * Ops::ACC_BYTES;
  const u32 nitems = ctr[0];
  const u32 chunk = msm_heavy_chunk(offsets[nslots]);   // as msm_find_heavy_kernel cut the runs
  for (u32 first = blockIdx.x * wpb; first < nitems; first += gridDim.x * wpb) {
    const u32 item = first + wave;
    const bool live = item < nitems;
    typename Ops::Acc acc = Ops::zero();
    if (live) {
      uint2 it = items[item];
      u32 lo = offsets[it.x] + it.y * chunk;
      u32 hi = offsets[it.x + 1];
      if (hi > lo + chunk) hi = lo + chunk;
      const char* wb = bases + (size_t)(it.x >> B) * wstride * Affine<F>::BYTES;  // prepared set: the window's table
      for (u32 j = lo + slot; j < hi; j += NS) {
        u32 e = sorted[j];
        const char* src = wb + (size_t)(e & 0x7fffffffu) * Affine<F>::BYTES;
        Affine<F> p = Affine<F>::load(src);
        if (!p.is_zero()) Ops::madd(acc, p, (e >> 31) != 0, src);
      }
    }
    Ops::tree(acc, sh, slot, NS);
    if (live && slot == 0) Ops::fin(acc).store(partials + (size_t)item * Pt::BYTES);
    __syncthreads();
  }
}

// one wave per heavy bucket: its slots stride over the chunk partials, then an LDS tree
// SHARED (prepared base set): the result goes to hfinal[slot] and its index into sorted[run start], where the lane
// that owns the bucket picks it up (msm_apply_heavy_kernel); else straight into the (window, bucket) cell.
template <class C, bool SHARED>
__global__ void __launch_bounds__(ARK_HEAVY_COMBINE_THREADS) msm_heavy_combine_kernel(const u32* __restrict__ ctr,
                                                               const HeavyEntry* __restrict__ list,
                                                               const char* __restrict__ partials, int HB, int LB,
                                                               const u32* __restrict__ offsets, u32* __restrict__ sorted,
                                                               int accum, char* __restrict__ buckets) {
  typedef AccOps<C> Ops;
  typedef typename Ops::Pt Pt;
  constexpr u32 NS = ARK_HEAVY_COMBINE_THREADS / Ops::LANES;
  extern __shared__ uint4 heavy_lds[];
  char* sh = (char*)heavy_lds;
  const u32 slot = threadIdx.x / Ops::LANES;
  const u32 nheavy = ctr[1];
  for (u32 hslot = blockIdx.x; hslot < nheavy; hslot += gridDim.x) {  // uniform for the whole workgroup
    HeavyEntry h = list[hslot];
    typename Ops::Acc acc = Ops::zero();
    for (u32 q = slot; q < h.items; q += NS) {
      Pt x = Pt::load(partials + (size_t)(h.first_item + q) * Pt::BYTES);
      Ops::add(acc, x);
    }
    Ops::tree(acc, sh, slot, NS);
    if (slot == 0) {
      if constexpr (SHARED) {
        Ops::fin(acc).store(buckets + (size_t)hslot * Pt::BYTES);  // `buckets` is the hfinal array here
        if (threadIdx.x == 0) sorted[offsets[h.bucket]] = hslot;
      } else {
        char* cell = buckets + (size_t)msm_slot_to_bucket(h.bucket, HB, LB) * Pt::BYTES;
        if (accum) {  // a later piece of a streamed MSM: add to the sum the earlier pieces left (MsmPiece)
          Pt prev = Pt::load(cell);
          Ops::add(acc, prev);
        }
        Ops::fin(acc).store(cell);
      }
    }
    __syncthreads();
  }
}

// Prepared base set: adds the heavy runs' sums into their (shared) buckets after msm_accumulate_shared_kernel.  One
// slot per heavy run (w, s); the lowest heavy window of bucket slot s owns the bucket and adds every heavy run of
// that slot (their partial's index sits in sorted[run start]), so no two slots touch one bucket.
template <class C>
__global__ void __launch_bounds__(64) msm_apply_heavy_kernel(const u32* __restrict__ ctr, const HeavyEntry* __restrict__ list,
                                                             const u32* __restrict__ offsets,
                                                             const u32* __restrict__ sorted,
                                                             const char* __restrict__ hfinal, int W, int B,
                                                             int HB, int LB, char* __restrict__ buckets) {
  typedef AccOps<C> Ops;
  typedef typename Ops::Pt Pt;
  const u32 nheavy = ctr[1], heavy_thresh = ctr[2];
  const u32 stride = gridDim.x * blockDim.x / Ops::LANES;
  for (u32 h = (blockIdx.x * blockDim.x + threadIdx.x) / Ops::LANES; h < nheavy; h += stride) {
    const u32 g = list[h].bucket;
    const u32 w0 = g >> B, s = g & ((1u << B) - 1u);
    bool owner = true;
    for (u32 w = 0; w < w0; w++) {
      const u32 gg = (w << B) | s;
      if (offsets[gg + 1] - offsets[gg] > heavy_thresh) owner = false;  // a lower window owns this bucket
    }
    if (!owner) continue;
    char* cell = buckets + (size_t)msm_slot_to_bucket(s, HB, LB) * Pt::BYTES;
    typename Ops::Acc acc = Ops::from_pt(Pt::load(cell));
    for (u32 w = w0; w < (u32)W; w++) {
      const u32 gg = (w << B) | s;
      const u32 a2 = offsets[gg];
      if (offsets[gg + 1] - a2 > heavy_thresh) {
        Pt x = Pt::load(hfinal + (size_t)sorted[a2] * Pt::BYTES);
        Ops::add(acc, x);
      }
    }
    Ops::fin(acc).store(cell);
  }
}

// ---- K5a: level 0 of the bucket reduction --------------------------------------------------------------
// Per window the buckets X[0..mwin) carry weights 1..mwin.  A slot folds the chunk of L consecutive buckets
// [tL, (t+1)L) with a running sum:  S_t = sum_r X[tL+r],  A_t = sum_r (r+1) X[tL+r], so that
//   sum_k k B_k = sum_t A_t + L * sum_t t S_t          (parallel form of mod.rs:478-484)
// (two waves per SIMD for the one-lane-per-point curves; the lane-pair G2 form keeps two accumulators of 112 limbs and an
// addition's temporaries per lane: one wave per SIMD, its overflow in AGPRs instead of scratch memory)
template <class C>
__global__ void __launch_bounds__(128, (C::LAZY_A && C::FA::LANES == 2) ? 1 : 2) msm_reduce_level_kernel(const char* __restrict__ in, u32 L, u32 total_out,
                                                               char* __restrict__ outS, char* __restrict__ outA) {
  typedef AccOps<C> Ops;
  typedef typename Ops::Pt Pt;
  const u32 t = (blockIdx.x * blockDim.x + threadIdx.x) / Ops::LANES;
  if (t >= total_out) return;
  const size_t base = (size_t)t * L;
  // the weighted sum A lives in LDS between its updates (one 4-coordinate slot per lane, or per lane pair over Fp2): two
  // accumulators + the loaded bucket + an addition's temporaries do not fit 256 registers (they spilled 16 B to scratch)
  __shared__ uint4 park_lds[(128 / Ops::LANES) * Ops::ACC_BYTES / 16];
  char* slot = (char*)park_lds + (size_t)(threadIdx.x / Ops::LANES) * Ops::ACC_BYTES;
  typename Ops::Acc running = Ops::zero();
  Ops::park(running, slot);
  for (u32 r = L; r-- > 0;) {
    Pt x = Pt::load(in + (base + r) * Pt::BYTES);
    Ops::add(running, x);
    typename Ops::Acc acc = Ops::unpark(slot);
    Ops::add_acc(acc, running);
    Ops::park(acc, slot);
  }
  Ops::fin(running).store(outS + (size_t)t * Pt::BYTES);
  Ops::fin(Ops::unpark(slot)).store(outA + (size_t)t * Pt::BYTES);
}

// K5a with the two chains of a chunk on two WAVES (round 5).  In the kernel above a lane alternates between the running
// sum and the weighted sum: 2 L dependent additions, ~25 us each when a SIMD holds one or two waves -- at 2^16 .. 2^20 pairs
// the whole level is that latency (65 536 lanes at 2^20).  Here wave 0 of a workgroup keeps the running sums of 64 chunks and
// wave 1 their weighted sums one step behind, the running sum crossing through a double-buffered LDS slot: L + 1 dependent
// additions, twice the lanes.  Used where the doubled lane count still fits the chip's resident lanes (msm_enqueue).
template <class C>
__global__ void __launch_bounds__(128, (C::LAZY_A && C::FA::LANES == 2) ? 1 : 2) msm_reduce_level_split_kernel(
    const char* __restrict__ in, u32 L, u32 total_out, char* __restrict__ outS, char* __restrict__ outA) {
  typedef AccOps<C> Ops;
  typedef typename Ops::Pt Pt;
  constexpr u32 PTS = 64 / Ops::LANES;   // chunks per workgroup
  __shared__ uint4 hand_lds[2 * PTS * Ops::ACC_BYTES / 16];
  const u32 role = threadIdx.x >> 6;     // 0: running sum, 1: weighted sum
  const u32 pslot = (threadIdx.x & 63u) / Ops::LANES;
  const u32 t = blockIdx.x * PTS + pslot;
  const bool live = t < total_out;
  const size_t base = (size_t)t * L;
  char* slot0 = (char*)hand_lds + (size_t)pslot * Ops::ACC_BYTES;
  char* slot1 = slot0 + (size_t)PTS * Ops::ACC_BYTES;
  typename Ops::Acc a = Ops::zero();
  for (u32 step = 0; step <= L; step++) {
    if (role == 0) {
      if (step < L && live) {
        Pt x = Pt::load(in + (base + (L - 1 - step)) * Pt::BYTES);
        Ops::add(a, x);
        Ops::park(a, (step & 1u) ? slot1 : slot0);
      }
    } else if (step > 0 && live) {
      Ops::add_acc(a, Ops::unpark(((step - 1) & 1u) ? slot1 : slot0));
    }
    __syncthreads();   // the running sum of this step is visible; the slot read in this step may be overwritten in the next
  }
  if (live) Ops::fin(a).store((role == 0 ? outS : outA) + (size_t)t * Pt::BYTES);
}

// ---- K5b: the rest of the reduction, bit-sliced -------------------------------------------------------
// After level 0 every window holds m pairs (S_j, A_j) with  sum_k k B_k = sum_j A_j + L0 * sum_j j S_j.
// Continuing with chunked running sums costs ~0.4 ms of pure latency per level (a handful of lanes, each a
// serial chain).  Instead  sum_j j S_j = sum_b 2^b U_b  with  U_b = sum_{j : bit b of j set} S_j : log2(m)
// masked plain sums plus the plain sum of the A_j -- all independent, each a strided partial sum + LDS tree.
// grid = (chunks, log2(m) + 1, W): quantity q < nbits -> U_q, q == nbits -> sum A.  The host finishes with
// a Horner over the bits (a few dozen point operations per window, same templated formulas).
template <class C>
__global__ void __launch_bounds__(256) msm_reduce_bits_kernel(const char* __restrict__ S, const char* __restrict__ A,
                                                              u32 m, int nbits, u32 chunk, char* __restrict__ partial) {
  typedef AccOps<C> Ops;
  typedef typename Ops::Pt Pt;
  extern __shared__ uint4 reduce_lds[];
  char* sh = (char*)reduce_lds;
  const u32 q = blockIdx.y, w = blockIdx.z, ch = blockIdx.x;
  const u32 slot = threadIdx.x / Ops::LANES, nslots = blockDim.x / Ops::LANES;
  const bool plain = (int)q == nbits;
  const char* src = plain ? A : S;
  typename Ops::Acc acc = Ops::zero();
  for (u32 e = slot; e < chunk; e += nslots) {
    u32 j = ch * chunk + e;
    if (j < m && (plain || ((j >> q) & 1u))) {
      Pt x = Pt::load(src + ((size_t)w * m + j) * Pt::BYTES);
      Ops::add(acc, x);
    }
  }
  Ops::tree(acc, sh, slot, nslots);
  if (slot == 0) Ops::fin(acc).store(partial + (((size_t)w * gridDim.y + q) * gridDim.x + ch) * Pt::BYTES);
}

// sums the `nchunks` chunk partials of every (window, quantity) pair: one wave per pair, its slots stride over the
// partials, then an LDS tree (a single lane walking 16-32 partials serially cost ~0.5 ms of pure latency)
template <class C>
__global__ void __launch_bounds__(64) msm_sum_chunks_kernel(const char* __restrict__ partial, u32 npairs, u32 nchunks,
                                                            char* __restrict__ out) {
  typedef AccOps<C> Ops;
  typedef typename Ops::Pt Pt;
  constexpr u32 NS = 64 / Ops::LANES;
  extern __shared__ uint4 chunk_lds[];
  char* sh = (char*)chunk_lds;
  const u32 t = blockIdx.x, slot = threadIdx.x / Ops::LANES;
  if (t >= npairs) return;
  typename Ops::Acc acc = Ops::zero();
  for (u32 k = slot; k < nchunks; k += NS) {
    Pt x = Pt::load(partial + ((size_t)t * nchunks + k) * Pt::BYTES);
    Ops::add(acc, x);
  }
  u32 width = 1;
  while (width < nchunks && width < NS) width <<= 1;
  Ops::tree(acc, sh, slot, width);
  if (slot == 0) Ops::fin(acc).store(out + (size_t)t * Pt::BYTES);
}

// ---- host-side plan / workspace -----------------------------------------------------------------
struct MsmPlan {
  int c;          // window bits (widest windows)
  int W;          // windows
  int narrow;     // the top `narrow` windows are c-1 bits wide, so that the widths sum to bits exactly and no
                  // window is left with only a few significant bits (0: uniform widths)
  size_t nb;      // bucket slots of the sort = W << (c-1)
  bool shared;    // prepared base set: the W windows share one set of 2^(c-1) buckets
  unsigned parts = 0;   // small plain MSMs: lanes per (window, bucket) run the accumulate kernel should use (0: its own rule)
  size_t nbuckets() const { return shared ? ((size_t)1 << (c - 1)) : nb; }
  int red_windows() const { return shared ? 1 : W; }
};

// relative cost of one mixed addition (Fp384 G1 = 1): Fp256 ~0.5; Fp2 over Fp384 2.7 (measured: 0.49 ns against
// 0.18 ns per addition at full occupancy)
static inline double msm_mul_cost(int curve_id) { return curve_id == 0 ? 0.5 : (curve_id >= 3 ? 2.7 : 1.0); }
// curves whose additions run on 28-bit limbs (C::LAZY_A: the Fp384 G1 curves)
static inline bool msm_lazy28(int curve_id) { return curve_id <= 2; }   // (BN254 since round 4: 9 x 29 bits -- the device path plans with C::LAZY_A)
// ARK_HIP_MSM_LAZY=0: the saturated kernels on those curves too (A/B measurements)
static inline bool msm_lazy_enabled() {
  static const bool on = [] {
    const char* e = getenv("ARK_HIP_MSM_LAZY");
    return !(e && e[0] == '0');
  }();
  return on;
}
static inline int msm_scalar_bits(int curve_id) {
  switch (curve_id) {
    case 0: return BN254_FR::BITS;
    case 1: case 4: return BLS12_381_FR::BITS;
    default: return BLS12_377_FR::BITS;
  }
}

static inline void msm_window_layout(int c, int bits, int* W, int* narrow) {
  // signed digits of a (bits-1)-bit value (after the s -> r-s fold) need bits significant positions in total
  // (the top window is not recoded and must keep one spare bit).  W windows of c bits, the top `narrow` of them
  // one bit narrower so that the widths add up exactly.
  int w = (bits + c - 1) / c;
  int deficit = w * c - bits;
  if (deficit > w || c < 3) {  // cannot be spread one bit per window: uniform widths, sparse top window
    *W = (bits + 1 + c - 1) / c;
    *narrow = 0;
    return;
  }
  *W = w;
  *narrow = deficit;
}
// bit offset of window w (sum of the widths below it): the weight 2^offset of its digits
static inline int msm_window_offset(int w, int c, int W, int narrow) {
  int off = 0;
  for (int k = 0; k < w; k++) off += msm_window_width(k, c, W, narrow);
  return off;
}

// Window size.  Model (seconds) of the phases that depend on c, from this chip's measured rates
// (profiles/): mixed additions stream at ~5.5e9/s (Fp384; scaled by `mul_cost` for other fields) but a
// single bucket is a serial chain (~14 us per addition on a lightly loaded SIMD), the first reduction
// level costs 2 full additions per bucket, the bit-sliced remainder ~0.5 ms.  With a prepared base set
// (`shared`) only one bucket set is reduced, which moves the optimum to wider windows.
static inline MsmPlan msm_make_plan(size_t n, int bits, double mul_cost, bool shared, bool lazy28 = false,
                                    const MsmWidths* widths = nullptr, bool split_runs = false) {
  const double plain_k = lazy28 ? 0.79 : 1.0;   // plain path only: see the accumulate model below
  int best_c = 3;
  double best = 1e300;
  const char* env = getenv(shared ? "ARK_HIP_MSM_C_PREPARED" : "ARK_HIP_MSM_C");
  if (env && atoi(env) >= 3 && atoi(env) <= 26) {
    best_c = atoi(env);
  } else {
    // c <= bits: a window wider than the scalar only adds empty buckets to sort and reduce (msm_u16 at 2^24: the model's
    // c = 19 took 8.2 ms, c = 17 -- one window of exactly the 2^16 buckets the digits reach -- 4.1 ms;
    // profiles/r4_narrow_scalars.txt)
    const int c_max = shared ? 25 : 23;
    for (int c = 3; c <= (bits < c_max ? (bits < 3 ? 3 : bits) : c_max); c++) {
      int W, narrow;
      msm_window_layout(c, bits, &W, &narrow);
      double nbk = (double)(W - narrow) * (double)(1u << (c - 1)) + (double)narrow * (double)(1u << (c - 2));
      if (shared) nbk = (double)(1u << (c - 1));
      double entries = (double)n * W;
      if (widths) {
        // measured width classes (K0): a scalar of b bits has digits in the windows below bit b + 1 only
        entries = 0.0;
        for (int k = 1; k < MSM_WIDTH_CLASSES; k++) {
          int need = (MSM_WIDTH_TOP[k] + 1 + c - 1) / c;
          if (need > W) need = W;
          entries += (double)widths->count[k] * need;
        }
        if (entries < 1.0) entries = 1.0;
      }
      const double madd = 1.0 / 5.5e9 * mul_cost, fadd = 1.4 / 5.5e9 * mul_cost;
      const bool fp2 = mul_cost > 2.0;  // a lane PAIR per bucket
      // accumulate: throughput-bound when the buckets make several rounds over the chip's resident lanes (2 waves x 4
      // SIMDs x 256 CUs x 64 lanes; a G2 bucket takes a lane pair); with a single round the kernel lasts as long as
      // its most loaded lane (Poisson tail): ~24 us per addition on a fully occupied SIMD, ~14 us with one wave per SIMD
      // (profiles/r2_small_n_sweep.txt)
      double acc = entries * madd;
      if (shared) {
        const double lanes = 131072.0 / (fp2 ? 2.0 : 1.0);
        const double load = entries / nbk, lmax = load + 3.0 * sqrt(load) + 2.0;
        // one dependent addition: a wave alone on its SIMD runs a chain faster; over Fp2 ~30 us whatever the occupancy
        // (BLS12-377 G2 2^16, c = 15 / 16 / 17: 25 / 28 / 31 us -- profiles/r2_msm_sweeps.txt)
        const double per_add = fp2 ? 30e-6 : (nbk <= lanes / 2 ? 14e-6 : 24e-6) * mul_cost;
        const double walk = lmax * per_add + W * 5e-6;                          // + run switches of a shared bucket
        if (nbk <= lanes || walk > acc) acc = walk;
      } else if (fp2) {
        // G2 plain path on the carry-free lane-pair kernels, fitted on BLS12-377 G2 2^16 / 2^20 / 2^22, c = 13 .. 21
        // (profiles/r4_planner_sweeps.txt): 0.40 ns per entry, rising for narrow windows (c = 15: 0.47, c = 14: 0.63); one
        // lane pair walks one (window, bucket) run at 38 us per dependent addition, the most loaded run sets the floor
        acc = entries * 0.40e-9 * (1.0 + ldexp(1.0, 13 - c));
        const double load = entries / nbk;
        const double chain = (load + 3.0 * sqrt(load) + 2.0) * 38e-6;
        if (chain > acc) acc = chain;
      } else {
        // G1 on carry-free limbs (round 3; BN254 since round 4, scaled by its mul_cost): 7.0e9 instead of 5.5e9 mixed
        // additions/s (plain_k = 0.79).
        // One lane walks one (window, bucket) run, and the kernel lasts at least as long as its most loaded lane: measured
        // 19 us per dependent addition for BLS12-381 whatever the occupancy (2^15 / 2^16, mean loads 2 .. 128:
        // profiles/r3_window_sweep.txt), 14-18 us for BN254.
        const double load = entries / nbk;
        // (round 5, profiles/r5_window_sweep_mid_sizes.txt) the first point of a bucket is a copy, not an addition:
        // entries - occupied buckets additions (2^19, c = 16 / 17: 7.86e6 / 6.88e6 additions in 1.13 / 1.00 ms; 2^20, c = 17:
        // 1.475e7 in 2.07 ms), and a launch of only ~2 rounds over the chip's 131 072 resident lanes pays for its ragged last
        // round: c = 15 (2.78e5 lanes) runs 5-9 % over the rate at 2^18 / 2^19 where c = 16 (four rounds) and wider match it
        // That saving is there while the bucket array stays in the 256 MB last-level cache (<= ~1e6 buckets of 192 B); beyond,
        // a bucket's first touch and its store cost what the copy saves (2^22: c = 19 / 20, 2.2e6 / 5.5e6 buckets, run at
        // 6.5e9 entries/s against c = 17's 6.8e9): the credit fades out between 1e6 and 4e6 buckets.
        const double rounds = nbk / 131072.0;
        const double credit = nbk <= 1e6 ? 1.0 : (nbk >= 4e6 ? 0.0 : (4e6 - nbk) / 3e6);
        acc = (entries - credit * nbk * (1.0 - exp(-load))) * madd * plain_k * (rounds >= 1.0 ? 1.0 + 0.3 / (rounds * rounds) : 1.0);
        const double lmax = load + 3.0 * sqrt(load) + 2.0;
        // one dependent addition: 19 us on 14 x 28-bit limbs, 13.5 us on BN254's 9 x 29 (2^17, c = 15 / 16: 0.24 / 0.165 ms)
        const double chain = lmax * (lazy28 ? (mul_cost < 1.0 ? 13.5e-6 : 19e-6) : 32e-6 * mul_cost);
        if (chain > acc) acc = chain;
        if (!widths && (narrow > 0 || W * c == bits)) {
          // the TOP window: scalars below r (folded below r / 2) reach only r / 2^bits of its buckets -- 0.58 for BLS12-377's
          // r = 0x12ab..., 0.76 for BN254's, 0.91 for BLS12-381's -- so its runs are that much longer than the layout says and,
          // sorted to the front, are what the kernel's last waves are still walking: ~11 us per dependent addition once few
          // waves are left (BLS12-377 G1 2^18: accumulate 0.89 ms with c = 15, 0.57 with c = 16, where BLS12-381 takes 0.63 /
          // 0.55; 2^17: 0.51 / 0.32; 2^16, c = 14: 0.47 against 0.36)
          const double frac = bits == 253 ? 0.583 : (bits == 254 ? 0.756 : (bits == 255 ? 0.906 : 1.0));
          const int wt = narrow > 0 ? c - 1 : c;
          const double load_top = (double)n / (frac * ldexp(1.0, wt - 1));
          const double chain_top = (load_top + 3.0 * sqrt(load_top) + 2.0) * 11e-6 * (mul_cost < 1.0 ? 0.7 : 1.0);
          if (chain_top > acc) acc = chain_top;
        }
      }
      // level 0 of the reduction: 2 full additions per bucket; over Fp2 with ONE bucket set (a lane PAIR per bucket: half
      // the lanes, the same chain length) the kernels run at ~40 % of the addition throughput (measured: BLS12-377 G2 2^16, 2^18 buckets 1.7 ms,
      // 2^16 buckets 0.76 ms; 2^22, 2^19 buckets 2.2 ms -- profiles/r2_msm_sweeps.txt)
      double red0 = nbk * 2.0 * fadd * (fp2 && shared ? 2.5 : 1.0);
      const double red0_lat = (fp2 && shared) ? 0.6e-3 : 2.0 * 8.0 * 21e-6 * mul_cost;  // latency floor of the reduction (G2: measured 0.62-0.76 ms for 2^12..2^16 buckets)
      if (red0_lat > red0) red0 = red0_lat;
      double bits_stage = 0.5e-3 * mul_cost;                    // bit-sliced stage + host tail
      if (!shared && fp2) {
        // measured reduction of the plain G2 path: 0.45 ms + 2.6 ns per bucket up to ~10^6 buckets (short level-0 chunks: the
        // bit-sliced stage is almost half of it), 1.25 ns per bucket beyond (same sweeps)
        // (round 5 refit, profiles/r5_g2_window_sweep.txt: the lane-pair kernels of round 4 reduce 1.97e6 buckets in 4.6 ms,
        // 4.98e6 in 10.6, 1.36e7 in 23 -- 1.7 ns per bucket beyond the first 1.2e6, not the 1.25 of the saturated kernels this
        // line was fitted on; the old figure made c = 19 look 1.7 ms cheaper than it is and cost BLS12-377 G2 2^20 14 %)
        // (the L0 = 16 rule of the same round made 2.8e5 .. 9.8e5 buckets 0.3-0.5 ms cheaper than this line says; a refit on that moved
        // 2^20 to c = 17 and lost 8 % on BLS12-381 G2 -- 9.72 against 8.98 ms, profiles/r5_window_sweep_mid_sizes.txt -- so it stays)
        red0 = 0.45e-3 + (nbk < 1.2e6 ? nbk : 1.2e6) * 2.6e-9 + (nbk > 1.2e6 ? nbk - 1.2e6 : 0.0) * 1.7e-9;
        bits_stage = 0.0;
      }
      if (!shared && !fp2) {
        // plain path, fitted on BLS12-381 2^16 .. 2^24 (profiles/r3_window_sweep.txt): 0.2 ms + 0.45 ns per bucket
        // + 1.4 ns per bucket for the first 3e5 (few buckets leave the chip's lanes idle, the chains dominate)
        // (round 5 refit on the split-level kernels, BLS12-381 G1: 1.56e5 buckets 0.43 ms, 2.8e5 0.54, 5.2e5 0.65-0.71, 9.8e5
        // 0.95, 3.7e6 2.1, 6.8e6 3.6; BN254 half of that: 0.40 ms + 0.47 ns per bucket + 0.10 ns for the first 1e6.  The older
        // fit -- 0.2 ms + 0.40 ns + 1.4 ns for the first 3e5 -- was 0.1 ms high between 3e5 and 1e6 buckets, 0.25 ms low at 6.8e6)
        // Counted half way between the occupied slots and W 2^(c-1): the reduction walks every window at full width, and the
        // empty upper half of a narrow one costs it its lanes' launch and barriers but no additions (2^23, c = 19: 2.09 ms with
        // 11 of 14 windows narrow on BLS12-381, 1.67 with 13 of 14 on BLS12-377; c = 18, uniform, 1.26)
        const double nred = 0.5 * (nbk + (double)W * ldexp(1.0, c - 1));
        red0 = (nred * 0.47e-9 + (nred < 1e6 ? nred : 1e6) * 0.10e-9) * mul_cost;
        bits_stage = 0.40e-3 * mul_cost;
      }
      // partition sort: per entry, plus a per-(window, bucket) term.  On the shared path at n >= 2^23 the latter is
      // measured nearly flat up to c = 22 (9 super-bucket bits + 12 bits finished in LDS, msm_part_split); beyond that the
      // super-bucket histogram grows and so do both sort passes (2^25: c = 24 costs +2.8 ms of sort and +7 ms of
      // reduction for -4.7 ms of accumulation; BN254 2^23 / 2^24: c = 22 beats 20 by 6-7 %; profiles/r2_msm_sweeps.txt)
      const double per_bucket = (shared && n >= ((size_t)1 << 23) && c <= 22) ? 1.5e-11 : 1.0e-10;
      const double sort = (double)n * W * 2.0e-11 + (double)W * (double)(1u << (c - 1)) * per_bucket;   // every key is read, live or not
      double cost = acc + red0 + bits_stage + sort;
      if (narrow == W) continue;  // every window one bit narrower: the layout of c - 1 with twice the buckets
      if (narrow == 0) {
        // uniform widths: a top window with only a few significant bits funnels n/2^tb points into each of 2^tb
        // buckets: correct (heavy-bucket path) but measured ~1.4x slower on the plain path and ~2x on a prepared set
        // (BLS12-377 G2 2^16: c = 18 3.7 ms against 1.9 ms at c = 17; 2^24 G1: c = 21, 23)
        const int tb = (bits - 1) - (W - 1) * c;
        if (tb >= (shared ? 0 : 1) && tb <= 5) cost *= shared ? 2.0 : 1.4;
        // narrow scalars (two to five windows): a top window that is more than two bits short of full is a large share
        // of the work and loses more -- msm_u32 with c = 18 (18 + 15 bits) 12.5 ms at 2^24 against 8.8 ms for the exact
        // 17 + 16 layout; only a top window that holds nothing but the carry bit (tb = 0: one heavy bucket) is cheap
        else if (!shared && !fp2 && bits <= 129 && tb > 5 && tb < c - 3) cost *= 1.4;
      }
      if ((size_t)n * (size_t)W >= (1ull << 32)) continue;  // 32-bit sort positions
      static const bool dbg = getenv("ARK_HIP_PLAN_DEBUG") != nullptr;   // the model's terms per candidate, on stderr
      if (dbg)
        fprintf(stderr, "plan n=%zu c=%d W=%d narrow=%d: accumulate %.3f reduce %.3f + %.3f sort %.3f -> %.3f ms\n", n, c, W, narrow,
                acc * 1e3, red0 * 1e3, bits_stage * 1e3, sort * 1e3, cost * 1e3);
      if (cost < best) { best = cost; best_c = c; }
    }
  }
  unsigned parts = 0;
  if (split_runs && !shared && !widths && bits > 128 && !(env && atoi(env) >= 3 && atoi(env) <= 26)) {
    // Small plain MSMs (round 5, profiles/r5_small_n_run_parts.txt).  Below ~2^17 pairs the accumulate kernel lasts as long as
    // its most loaded bucket (a Poisson tail of dependent additions at ~17 us each) and the model above answers with wide
    // windows -- few points per bucket, many buckets to reduce.  Narrower windows with every run walked by 4 or 8 lanes
    // (msm_accumulate_parts_kernel: the chain is cut, the pieces are summed by msm_sum_parts_kernel) win on both sides:
    // BLS12-381 G1 2^16 c = 14 -> 12: reduce 0.43 -> 0.27 ms, accumulate 0.36 -> 0.40, call 1.11 -> 0.99 ms; 2^14 0.91 -> 0.79;
    // 2^12 0.80 -> 0.71; 2^8 0.69 -> 0.60; BN254 2^16 0.67 -> 0.60; BLS12-377 G2 2^14 2.00 -> 1.61, 2^12 1.74 -> 1.39.
    // From 2^17 (G2: 2^16) the model's choice with one lane per run is the faster one again.  BLS12-377 G1 leaves the rule at
    // 2^15 already (1.09 against 1.07 ms, 2^16 1.29 against 1.22; 2^14 0.81 against 0.99): r = 0x12ab... x 2^240 fills only
    // 0.58 of its top window's buckets, whose runs are then twice as long as anybody else's and set the kernel's time.
    static const bool on = [] { const char* e = getenv("ARK_HIP_MSM_SMALL_SPLIT"); return !(e && e[0] == '0'); }();
    const bool fp2 = mul_cost > 2.0;
    int logn = 0;
    while (((size_t)1 << logn) < n) logn++;
    if (on && n >= 256 && n <= (fp2 ? (size_t)24576 : bits == 253 ? (size_t)20480 : (size_t)73728)) {
      const int cap = fp2 ? 11 : 12;
      best_c = logn - 2 < cap ? logn - 2 : cap;
      parts = (!fp2 && (n >> (best_c - 1)) >= 32) ? 8u : 4u;
    }
  }
  MsmPlan p;
  p.c = best_c;
  p.shared = shared;
  p.parts = parts;
  msm_window_layout(best_c, bits, &p.W, &p.narrow);
  if (!shared && getenv("ARK_HIP_MSM_UNIFORM")) {  // A/B knob: the older uniform-width layout
    p.W = (bits + 1 + best_c - 1) / best_c;
    p.narrow = 0;
  }
  p.nb = (size_t)p.W << (best_c - 1);
  return p;
}

static inline MsmPlan msm_plan_for_widths(size_t n, int field_bits, double mul_cost, bool lazy28, const MsmWidths& w) {
  const int slack = 8;
  int bits = field_bits;
  if ((int)w.max_bits + 1 + slack <= field_bits) bits = (w.max_bits ? (int)w.max_bits : 1) + 1;
  return msm_make_plan(n, bits, mul_cost, false, lazy28, &w);
}

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  int ensure(size_t bytes) {
    if (bytes <= cap) return 0;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    size_t want = bytes + bytes / 8 + 256;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {
      fprintf(stderr, "ark_hip: hipMalloc(%zu) failed: %s\n", want, hipGetErrorString(e));
      (void)hipGetLastError();  // callers may fall back to a smaller plan / the streaming path: no sticky error left behind
      return -1;
    }
    cap = want;
    return 0;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
};

struct MsmTimings {  // filled when requested (HIP events on the MSM stream), milliseconds
  float digits = 0, scan = 0, scatter = 0, accumulate = 0, reduce = 0, total = 0;
  int c = 0, W = 0;
};

// One MSM in flight: the device work is enqueued on the stream, its (small) result lands in this slot's pinned
// staging area, `done` fires when it has; the host tail runs in msm_finish().  Device workspaces are shared by
// consecutive MSMs (stream order keeps them apart), the staging area is per slot.
struct MsmJob {
  void* pinned = nullptr;
  size_t pinned_cap = 0;
  hipEvent_t done = nullptr;
  hipEvent_t acc_done = nullptr;   // recorded in front of the bucket reduction: when the host tail's helpers are called in (msm_finish)
  bool busy = false;
  bool empty = false;     // n == 0: identity, nothing enqueued
  bool no_result = false; // a non-final piece of a streamed MSM (MsmPiece): only the scalar-range flag comes back
  MsmPlan pl{};
  int nbits = 0, log2L0 = 0;
  u32 Q = 0;
  size_t npairs = 0;
  const char* d_sums = nullptr;   // the job's part sums in device memory (npairs XYZZ points): read by the sharded combine
  bool short_job = false;         // a couple of milliseconds of GPU work at most: the host tail is a visible share of the call
  bool timing = false;
  hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
};
static constexpr int MSM_JOBS = 4;

struct MsmWorkspace {
  DevBuf hctr, hlist, hitems, hpart, hfinal, keys, part, thist, toff, sorted, offsets, sums, buckets, lvlS[2], lvlA[2],
      order, ohist, ooff, probe, big, parts;
  MsmJob jobs[MSM_JOBS];
  std::mutex mu;
  bool attr_set = false;  // dynamic-LDS opt-in done for this device
  // window groups (msm_enqueue): the second group's sort runs on this stream under the first group's accumulate kernel
  hipStream_t side = nullptr;
  hipEvent_t grp_ev[2] = {nullptr, nullptr};
  u32* probe_host = nullptr;   // pinned word the width probe reads back into (a pageable target costs a staged copy)
  DevBuf nzcnt, cscal, cidx;   // zero-scalar compaction (K0c): per-chunk non-zero counts + their scan, the compacted scalars, their indices
  bool probe_allowed = true;   // set per call by the C ABI: the *_async entries must not wait for the lane's stream to drain
  void release() {
    if (probe_host) (void)hipHostFree(probe_host);
    probe_host = nullptr;
    hctr.release(); hlist.release(); hitems.release(); hpart.release(); hfinal.release();
    nzcnt.release(); cscal.release(); cidx.release();
    keys.release(); part.release(); thist.release(); toff.release(); sorted.release();
    offsets.release(); sums.release();
    order.release(); ohist.release(); ooff.release(); probe.release(); big.release(); parts.release();
    buckets.release();
    if (side) { (void)hipStreamSynchronize(side); (void)hipStreamDestroy(side); side = nullptr; }
    for (auto& e : grp_ev) { if (e) (void)hipEventDestroy(e); e = nullptr; }
    for (int i = 0; i < 2; i++) { lvlS[i].release(); lvlA[i].release(); }
    for (auto& j : jobs) {
      if (j.pinned) (void)hipHostFree(j.pinned);
      j.pinned = nullptr;
      j.pinned_cap = 0;
      if (j.done) (void)hipEventDestroy(j.done);
      if (j.acc_done) (void)hipEventDestroy(j.acc_done);
      j.done = nullptr;
      for (auto& e : j.ev) { if (e) (void)hipEventDestroy(e); e = nullptr; }
      j.busy = false;
    }
  }
};

static inline int msm_job_pinned(MsmJob& j, size_t bytes) {
  if (j.pinned_cap >= bytes) return 0;
  if (j.pinned) (void)hipHostFree(j.pinned);
  j.pinned = nullptr;
  j.pinned_cap = 0;
  ARK_HIP_TRY(hipHostMalloc(&j.pinned, bytes + 256));
  j.pinned_cap = bytes + 256;
  return 0;
}

// One piece of a STREAMED MSM (capi.hip msm_stream: the host-pointer entry points).  The pairs of one MSM arrive in
// pieces (uploads overlapping compute); every piece is digit-recoded and sorted on its own, but all pieces accumulate into
// ONE bucket array laid out by the plan of the whole job, and only the last piece runs the bucket reduction: cutting an
// MSM into independent sub-MSMs would pay the reduction and the sort's latency floors once per piece and force narrower
// windows on each (2^24 as four 2^22 MSMs: +11 ms; as four pieces of one MSM: +1 ms).
struct MsmPiece {
  const MsmPlan* plan;     // plan of the WHOLE job
  void* buckets;           // shared bucket array: plan->nbuckets() XYZZ points
  bool first, last;        // first: buckets are overwritten, otherwise added to; last: the reduction runs
  hipEvent_t after_prev;   // (nullable) recorded behind the previous piece's accumulate kernels, on another stream
  hipEvent_t after_this;   // recorded behind this piece's accumulate kernels
};

// Enqueue one single-GPU MSM with device-resident inputs on `stream`; returns the job slot (>= 0) or a negative code.
//   points / wstride / prepared:  plain call: points = the n bases, wstride = 0, prepared = nullptr;
//                                 prepared base set: points = the [W][wstride] table of per-window multiples and
//                                 `prepared` = the plan it was built for (msm_prepare_table).
template <class C>
int msm_enqueue(MsmWorkspace& ws, const void* d_points, size_t wstride, const MsmPlan* prepared, const void* d_scalars,
                size_t n, int scalars_mont, hipStream_t stream, bool timing, int sbytes = 0, int sbits = 0,
                const MsmPiece* piece = nullptr) {
  // sbytes != 0: narrow unsigned scalars of sbytes bytes with at most sbits significant bits (K1n); never with `prepared`
  // piece != nullptr: see MsmPiece (never with `prepared`; n > 0)
  typedef typename C::F F;
  typedef XYZZ<F> Pt;
  std::lock_guard<std::mutex> lock(ws.mu);
  int slot = -1;
  for (int i = 0; i < MSM_JOBS; i++)
    if (!ws.jobs[i].busy) { slot = i; break; }
  if (slot < 0) return -6;  // ARK_HIP_ERR_BUSY
  MsmJob& job = ws.jobs[slot];
  job.empty = (n == 0);
  job.timing = false;
  job.no_result = false;
  if (piece && (prepared || n == 0)) return -1;
  if (n == 0) {
    job.busy = true;
    return slot;
  }
  if (n >= (1ull << 31)) return -2;
  if (sbytes && (prepared || (sbytes != 1 && sbytes != 2 && sbytes != 4 && sbytes != 8) || sbits < 1 || sbits > 8 * sbytes))
    return -1;
  // Width classes of full-width scalars (K0 above): one 40-byte read-back decides the plan, so the probe runs only where
  // that round trip (~30 us; 2 % of a 2^18 call, measured) is small against the call (n >= 2^19) and nothing else of this workspace is in flight to wait
  // behind (a pipelined caller keeps the plan for n uniform 255-bit scalars).  ARK_HIP_MSM_PROBE=0 turns it off.
  int plan_bits = sbytes ? sbits + 1 : C::S::BITS;
  MsmWidths widths{};
  bool have_widths = false;
  u32 nz_chunk = 0, nz_blocks = 0;   // the exact probe's chunks (K0c)
  bool measured_exact = false;
  if (!sbytes && !piece && n >= ((size_t)1 << 19)) {
    static const bool probe_on = [] {
      const char* e = getenv("ARK_HIP_MSM_PROBE");
      return !(e && atoi(e) == 0);
    }();
    bool idle = true;
    for (int i = 0; i < MSM_JOBS; i++) idle = idle && !ws.jobs[i].busy;
    // (the probe waits for its 40-byte read-back, i.e. for everything already queued on this stream -- a device transform
    //  producing the scalars, say: the asynchronous entry points skip it and keep the uniform-scalar plan)
    if (probe_on && idle && ws.probe_allowed) {
      constexpr size_t PW = (1 + MSM_WIDTH_CLASSES) * 4;
      if (ws.probe.ensure(PW)) return -3;
      if (!ws.probe_host) ARK_HIP_TRY(hipHostMalloc((void**)&ws.probe_host, 64, hipHostMallocDefault));
      auto measure = [&](u32 stride, MsmWidths* w) -> int {
        const u32 cnt = (u32)((n + stride - 1) / stride);
        u32 blocks = cnt <= 8192 ? 1u : (cnt + 1023) / 1024;
        if (blocks > 1024) blocks = 1024;
        if (blocks > 1) ARK_HIP_TRY(hipMemsetAsync(ws.probe.p, 0, PW, stream));
        if (stride == 1 && blocks > 1) {
          // the exact pass in contiguous chunks, one per workgroup, each leaving its non-zero count (K0c)
          nz_chunk = (u32)(((n + blocks - 1) / blocks + 1023) / 1024 * 1024);
          nz_blocks = (u32)((n + nz_chunk - 1) / nz_chunk);
          if (ws.nzcnt.ensure((size_t)(2 * nz_blocks + 2) * 4)) return -3;
          hipLaunchKernelGGL((msm_scalar_bits_kernel<typename C::S>), dim3(nz_blocks), dim3(1024), 0, stream,
                             (const u32*)d_scalars, (u32)n, stride, scalars_mont, (u32*)ws.probe.p, nz_chunk, (u32*)ws.nzcnt.p);
        } else
        hipLaunchKernelGGL((msm_scalar_bits_kernel<typename C::S>), dim3(blocks), dim3(1024), 0, stream,
                           (const u32*)d_scalars, (u32)n, stride, scalars_mont, (u32*)ws.probe.p);
        ARK_HIP_TRY(hipMemcpyAsync(ws.probe_host, ws.probe.p, PW, hipMemcpyDeviceToHost, stream));
        ARK_HIP_TRY(hipStreamSynchronize(stream));
        w->max_bits = ws.probe_host[0];
        for (int k = 0; k < MSM_WIDTH_CLASSES; k++) w->count[k] = ws.probe_host[1 + k];
        return 0;
      };
      // a spread sample of 4096 scalars first: uniform scalars stop here (more than half of the sample wider than 128 bits)
      MsmWidths sample{};
      if (int rc = measure((u32)(n / 4096), &sample)) return rc;
      if (msm_widths_skewed(sample)) {
        if (int rc = measure(1u, &widths)) return rc;
        measured_exact = nz_blocks > 1;
        if (!prepared) {
          have_widths = true;
        } else if (widths.max_bits <= 48) {   // (u32: 8.9 ms prepared, 6.8 plain; u64: 11.0 prepared, 12.4 plain -- at 2^24)
          // A prepared set shares ONE bucket set of 2^(c-1) buckets (c = 22 at 2^24) between all windows: right for
          // n x 12 digits, wrong for scalars that have one to four -- u16 at 2^24: 12.8 ms against the plain path's 3.8.
          // Row 0 of the table is the base set itself (2^0 P_i): scalars no wider than 48 bits take the plain pipeline on it.
          prepared = nullptr;
          wstride = 0;
          have_widths = true;
        }
      }
    }
  }
  const MsmPlan pl = prepared ? *prepared
                     : piece  ? *piece->plan
                     : have_widths ? msm_plan_for_widths(n, C::S::BITS, msm_mul_cost(C::ID), C::LAZY_A, widths)
                                   : msm_make_plan(n, plan_bits, msm_mul_cost(C::ID), false, C::LAZY_A, nullptr,
                                                   C::LAZY_A && msm_lazy_enabled());
  // K0c: a quarter or more of the scalars zero (known exactly from the probe's class 0) -> they leave before the sort.
  // From here on `n` is the number of scalars the pipeline carries; base indices come back through ws.cidx after the sort.
  const size_t n_all = n;
  const void* const d_scalars_all = d_scalars;
  bool compacted = false;
  if (measured_exact && !sbytes && !piece) {
    static const bool compact_env = [] { const char* e = getenv("ARK_HIP_MSM_COMPACT"); return !(e && e[0] == '0'); }();
    const size_t zeros = widths.count[0];
    // what the compaction saves grows with zeros x windows (the keys that are never written, counted and scattered), what it
    // costs with n (one more pass over the scalars): witness-like 2^24 (60 % zeros, 15 windows) 6.52 -> 5.90 ms, a bool vector
    // (half zeros, ONE window) 3.22 -> 3.32 -- so: at least a quarter zeros and zeros x W >= 3 n (profiles/r5_zero_compaction.txt)
    if (compact_env && zeros * 4 >= n && zeros < n && zeros * (size_t)pl.W >= 3 * n) {
      const size_t nz = n - zeros;
      if (ws.cscal.ensure(nz * 32) || ws.cidx.ensure(nz * 4)) return -3;
      compacted = true;
      n = nz;
      d_scalars = ws.cscal.p;
    }
  }
  const int c = pl.c, W = pl.W;
  const size_t nb = pl.nb;                      // sort slots
  const size_t nbk = pl.nbuckets();             // accumulated buckets
  const int Wr = pl.red_windows();
  if ((size_t)n * (size_t)W >= (1ull << 32)) return -2;  // sort positions are 32-bit
  if (prepared && (wstride < n || (size_t)W * wstride >= (1ull << 31))) return -2;
  const size_t mwin = (size_t)1 << (c - 1);

  // bucket-id split for the two-pass partition sort (msm_sort.cuh)
  const int Bbits = c - 1;
  int HB, LB;
  msm_part_split(n, Bbits, &HB, &LB);
  const u32 nsuper = (u32)W << HB;
  const u32 ptile = msm_part_tile(HB);
  const u32 ntiles = (u32)((n + ptile - 1) / ptile);
  const size_t nthist = (size_t)nsuper * ntiles;
  if (ws.keys.ensure((size_t)W * n * 4)) return -3;
  if (ws.sorted.ensure((size_t)W * n * 4)) return -3;
  if (ws.part.ensure((size_t)W * n * 8)) return -3;
  if (ws.thist.ensure(nthist * 4) || ws.toff.ensure((nthist + 2) * 4)) return -3;   // (+1 per window group)
  if (ws.offsets.ensure((nb + 2) * 4)) return -3;
  const u32 ntscan = (u32)((nthist + SCAN_TILE - 1) / SCAN_TILE);
  const u32 noblk = (u32)((nbk + ORDER_TILE - 1) / ORDER_TILE);
  const size_t nohist = (size_t)noblk * ORDER_BINS;
  const u32 noscan = (u32)((nohist + SCAN_TILE - 1) / SCAN_TILE);
  const size_t nsums = (size_t)(ntscan > noscan ? ntscan : noscan) + 2;
  if (ws.sums.ensure(2 * nsums * 4)) return -3;   // one scan scratch per window group
  if (ws.order.ensure(nbk * 4) || ws.ohist.ensure((nohist + ORDER_BINS) * 4) || ws.ooff.ensure((nohist + ORDER_BINS + 2) * 4)) return -3;
  if (!piece && ws.buckets.ensure(nbk * Pt::BYTES)) return -3;
  char* const d_buckets = piece ? (char*)piece->buckets : (char*)ws.buckets.p;
  const int accum = (piece && !piece->first) ? 1 : 0;

  // bucket reduction geometry: level 0 (chunked running sums over L0 buckets per lane), then the bit-sliced sums
  u32 L0 = 32;
  {
    size_t want = (mwin * (size_t)Wr) >> 17;  // keep ~1e5 (S, A) pairs for the bit-sliced stage
    u32 p2 = 1;
    while (p2 < want) p2 <<= 1;
    // few buckets: short chains beat fewer pairs (measured: 2^15 buckets L0 = 2, 2^16..2^17 L0 = 4, profiles/r2_msm_sweeps.txt)
    const size_t nbr = mwin * (size_t)Wr;
    const u32 l0_min = nbr <= ((size_t)1 << 15) ? 2 : (nbr <= ((size_t)1 << 17) ? 4 : 8);
    if (p2 < l0_min) p2 = l0_min;
    if (p2 < L0) L0 = p2;
    // one shared bucket set (prepared base set), measured per bucket count (profiles/r2_msm_sweeps.txt, sessions L0 / AN):
    // 2^18 buckets L0 = 8, 2^19 .. 2^21 L0 = 16 (2^19: reduction 1.32 -> 0.99 ms against L0 = 8)
    if (Wr == 1 && nbr > ((size_t)1 << 18)) L0 = nbr <= ((size_t)1 << 21) ? 16 : 32;
    // (round 5) 2^19 .. 2^20 buckets on the one-lane-per-point curves (c = 17: 2^21 / 2^22 pairs): with L0 = 16 the two-wave
    // level-0 form still fits one round of the chip (2 x 61 440 lanes, 17 steps) and hands the bit-sliced stage half the pairs
    // of L0 = 8, whose 245 760 split lanes do not fit and whose one-wave form walks 16 steps: reduction 1.18 -> 0.95 ms,
    // 2^21 6.35 -> 5.99 ms, 2^22 11.20 -> 10.82 (profiles/r5_reduce_geometry_sweep.txt, session r5rs2)
    if (Wr > 1 && C::FA::LANES == 1 && nbr > ((size_t)1 << 19) && nbr <= ((size_t)1 << 20)) L0 = 16;
    // (round 5) the lane-pair curves (G2) from 2^18 buckets (c >= 15: 2^17 pairs and up): their bit-sliced stage is the
    // expensive half, and twice the chain for half the pairs pays -- BLS12-377 G2 reduction 1.58 -> 1.28 ms at 2.8e5 buckets
    // (2^18: 4.65 -> 4.25 ms), 1.70 -> 1.38 at 5.2e5 (2^19 6.69 -> 6.30, 2^20 10.49 -> 10.10; BLS12-381 G2 2^20 10.13 -> 9.64),
    // 2.93 -> 2.46 at 9.8e5 (2^21 18.04 -> 17.52); at 1.6e5 buckets L0 = 8 stays the best cell (same sweeps, sessions r5g2r / r5g2r2)
    // -- and the same holds further up: 9.8e5 buckets L0 = 32 (2.46 -> 2.13 ms); 3.7e6 (2^22, c = 19) L0 = 64 on BLS12-377 G2 (13 of
    // its 14 windows are narrow: 4.32 -> 3.95 ms, session r5g2l) but 32 on BLS12-381 G2 (11 narrow: 5.85 against 6.5 ms, r5g2l3).
    // One rule covers every cell measured: the power of two at or above buckets / 32 768, from 8 to 32 (BLS12-377: 64).
    if (Wr > 1 && C::FA::LANES == 2 && nbr > ((size_t)1 << 18)) {
      const u32 cap = C::S::BITS == 253 ? 64u : 32u;
      u32 g = 8;
      while (g < cap && (size_t)g * 32768 < nbr) g <<= 1;
      L0 = g;
    }
    if (const char* e0 = getenv("ARK_HIP_MSM_L0")) {  // tuning knob
      int v = atoi(e0);
      if (v == 1 || v == 2 || v == 4 || v == 8 || v == 16 || v == 32 || v == 64) L0 = (u32)v;
    }
    while (L0 > mwin) L0 >>= 1;
  }
  const size_t m = mwin / L0;  // pairs per window after level 0
  int log2L0 = 0;
  while ((1u << log2L0) < L0) log2L0++;
  int nbits = 0;
  while (((size_t)1 << nbits) < m) nbits++;
  const u32 Q = (u32)nbits + 1;
  // chunk of the bit-sliced stage: a workgroup's 256 lanes stride over it (chunk/256 serial additions each) before the
  // 8-step LDS tree -- both pure latency, so chunks are kept short once there are enough of them to fill the chip
  // (few workgroups, e.g. one window of a prepared set at small n: latency only, 2^16 1.39 -> 1.11 ms; many: throughput)
  u32 chunk = (size_t)Wr * Q * ((m + 4095) / 4096) < 512 ? 1024 : 4096;
  if (Wr == 1) chunk = m <= 16384 ? 1024 : (m <= 65536 ? 2048 : 4096);  // measured: m = 2^15 pairs 2048 (0.77 -> 0.70 ms)
  if (const char* ec = getenv("ARK_HIP_MSM_CHUNK")) {
    int v = atoi(ec);
    if (v == 256 || v == 512 || v == 1024 || v == 2048 || v == 4096 || v == 8192 || v == 16384) chunk = (u32)v;
  }
  if (chunk > m) chunk = (u32)m;
  const u32 nchunks = (u32)((m + chunk - 1) / chunk);
  const size_t npart = (size_t)Wr * Q * nchunks;
  const size_t npairs = (size_t)Wr * Q;
  if (ws.lvlS[0].ensure(m * Wr * Pt::BYTES) || ws.lvlA[0].ensure(m * Wr * Pt::BYTES)) return -3;
  if (ws.lvlS[1].ensure(npart * Pt::BYTES)) return -3;
  if (nchunks > 1 && ws.lvlA[1].ensure(npairs * Pt::BYTES)) return -3;
  {
    int rc = msm_job_pinned(job, npairs * Pt::BYTES + 64);
    if (rc) return rc;
  }
  if (!job.done) ARK_HIP_TRY(hipEventCreateWithFlags(&job.done, hipEventDisableTiming));
  if (!job.acc_done) ARK_HIP_TRY(hipEventCreateWithFlags(&job.acc_done, hipEventDisableTiming));
  if (timing) {
    for (auto& e : job.ev)
      if (!e) ARK_HIP_TRY(hipEventCreate(&e));
    ARK_HIP_TRY(hipEventRecord(job.ev[0], stream));
  }

  u32* keys = (u32*)ws.keys.p;
  u32* sorted = (u32*)ws.sorted.p;
  u32* offsets = (u32*)ws.offsets.p;
  u32* thist = (u32*)ws.thist.p;
  u32* toff = (u32*)ws.toff.p;
  uint2* part = (uint2*)ws.part.p;
  u32* order = (u32*)ws.order.p;
  u32* sums = (u32*)ws.sums.p;

  // heavy runs: a lane walks its bucket serially (~28 us per entry with two or three waves per SIMD) and
  // the heaviest buckets start first; a run is "heavy" when its walk would outlast the kernel's
  // throughput-bound duration (entries / 5.5e9 per s).  At 2^24 x 13 windows that is ~1400 entries, so the
  // sparse top window (1024 per bucket) still rides along; skewed scalar distributions do not.
  const size_t total_entries = (size_t)n * W;
  size_t mean_load = total_entries / nbk;        // per lane (a lane of a prepared set walks W runs)
  u32 forced_thresh = 0;  // 0: computed on the device from the number of non-zero entries (msm_thresh_kernel)
  if (const char* hv = getenv("ARK_HIP_MSM_HEAVY")) {
    if (atoi(hv) >= 64) forced_thresh = (u32)atoi(hv);
  }
  const size_t max_heavy = total_entries / 64 + 1;  // the threshold is never below 64
  const size_t max_items = total_entries / HEAVY_CHUNK + max_heavy + 1;
  if (ws.hctr.ensure(64) || ws.hlist.ensure((max_heavy + 1) * sizeof(HeavyEntry)) || ws.hitems.ensure((max_items + 1) * 8) ||
      ws.hpart.ensure((max_items + 1) * Pt::BYTES))
    return -3;
  if (pl.shared && ws.hfinal.ensure(max_heavy * Pt::BYTES)) return -3;
  ARK_HIP_TRY(hipMemsetAsync(ws.hctr.p, 0, 64, stream));  // per window group: [chunk items, heavy runs, threshold, -]; [3]: scalar-range error flag; [8 + g]: super-buckets of group g left to the sliced pass B
  if (compacted) {
    u32* const cnt = (u32*)ws.nzcnt.p;
    u32* const off = cnt + nz_blocks + 1;
    scan_exclusive(cnt, nz_blocks, nullptr, off, stream);   // (nz_blocks <= 1024: the one-workgroup scan)
    hipLaunchKernelGGL((msm_compact_scalars_kernel<typename C::S>), dim3(nz_blocks), dim3(1024), 0, stream,
                       (const u32*)d_scalars_all, (u32)n_all, scalars_mont, nz_chunk, (const u32*)off, (u32*)ws.hctr.p + 3,
                       (u32*)ws.cscal.p, (u32*)ws.cidx.p, (u32)n);
  }
  const u32 nblk = (u32)((n + 255) / 256);
  if (sbytes) {
    const u64 vmask = sbits >= 64 ? ~0ull : ((1ull << sbits) - 1ull);
    const unsigned char* sp = (const unsigned char*)d_scalars;
    switch (sbytes) {
      case 1: hipLaunchKernelGGL((msm_digits_small_kernel<1>), dim3(nblk), dim3(256), 0, stream, sp, (u32)n, c, W, pl.narrow, vmask, keys); break;
      case 2: hipLaunchKernelGGL((msm_digits_small_kernel<2>), dim3(nblk), dim3(256), 0, stream, sp, (u32)n, c, W, pl.narrow, vmask, keys); break;
      case 4: hipLaunchKernelGGL((msm_digits_small_kernel<4>), dim3(nblk), dim3(256), 0, stream, sp, (u32)n, c, W, pl.narrow, vmask, keys); break;
      default: hipLaunchKernelGGL((msm_digits_small_kernel<8>), dim3(nblk), dim3(256), 0, stream, sp, (u32)n, c, W, pl.narrow, vmask, keys); break;
    }
  } else {
    hipLaunchKernelGGL((msm_digits_kernel<typename C::S>), dim3(nblk), dim3(256), 0, stream, (const u32*)d_scalars,
                       (u32)n, scalars_mont, c, W, pl.narrow, keys, (u32*)ws.hctr.p + 3);
  }
  if (timing) ARK_HIP_TRY(hipEventRecord(job.ev[1], stream));

  // ---- window groups -------------------------------------------------------------------------------------------
  // The sort is memory / LDS bound and small (8-28 VGPRs per lane), the accumulate kernel is multiply bound, uses no LDS
  // and leaves 70 of the 512 registers per SIMD lane free: the two overlap well.  A plain MSM is therefore cut into TWO
  // groups of windows: group 0 is sorted and accumulated on `stream`; group 1's sort runs on a side stream UNDER group
  // 0's accumulate kernel and its accumulate kernel follows on `stream`.  Every kernel below already works on a range
  // of windows through its base pointers (keys, pairs, sorted indices, offsets, order and buckets are window-major), so
  // a group is nothing but offsets into the same arrays.  One group: a prepared set (its windows share one bucket set),
  // a piece of a streamed MSM, few windows, and n < 2^25 -- measured (profiles/r3_window_groups_ab.txt): the sort kernels
  // slow the accumulate kernel they run under by about what they hide up to 2^24 (38.6 against 38.5 ms; 2^20 4.27 against
  // 4.14), and win where the sort is a larger share: 2^26 136.4 against 139.7 ms.  ARK_HIP_MSM_GROUPS=1 / =2 force it.
  int ngroups = 1;
  if (!pl.shared && !piece && W >= 6 && n >= ((size_t)1 << 19)) {
    static const int groups_env = [] {
      const char* e = getenv("ARK_HIP_MSM_GROUPS");
      return e ? atoi(e) : 0;
    }();
    if (groups_env == 2 || (groups_env != 1 && n >= ((size_t)1 << 25))) ngroups = 2;
  }
  // Heavy runs (skewed scalars) and the lane-per-bucket kernel touch disjoint buckets: with one window group and a call
  // long enough to pay two event hops, the chunk / combine kernels -- chains of dependent additions on a few hundred
  // waves -- run on the side stream UNDER the accumulate kernel (witness-like 2^24: 1.7 ms of heavy kernels in front of
  // 2.1 ms of accumulation; profiles/r4_skewed_sort_ab.txt).  ARK_HIP_MSM_HEAVY_SIDE=0 keeps them in line.
  static const bool heavy_side_env = [] {
    const char* e = getenv("ARK_HIP_MSM_HEAVY_SIDE");
    return !(e && atoi(e) == 0);
  }();
  static const int heavy_side_min_log = [] {
    const char* e = getenv("ARK_HIP_MSM_HEAVY_SIDE_MIN_LOG");   // tuning knob: smallest log2 n that takes the side stream
    return e ? atoi(e) : 20;
  }();
  const bool heavy_side = heavy_side_env && ngroups == 1 && !piece && n >= ((size_t)1 << heavy_side_min_log);
  if (ngroups == 2 || heavy_side) {
    if (!ws.side) ARK_HIP_TRY(hipStreamCreateWithFlags(&ws.side, hipStreamNonBlocking));
    for (auto& e : ws.grp_ev)
      if (!e) ARK_HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
  const size_t lds_a = ((size_t)8 << HB) + (size_t)ptile * 8;
  if (lds_a > PART_SCATTER_LDS_MAX) return -2;   // 2^14 super-buckets and more: beyond the n W < 2^32 the sort serves anyway
  const u32 stage_cap = msm_part_stage_cap(LB);
  const size_t lds_b = ((size_t)(1 << LB) + 1024 + stage_cap) * 4;
  if (!ws.attr_set) {  // > 64 KiB of dynamic LDS needs the opt-in attribute (once per device; the workspace is per device)
    ARK_HIP_TRY(hipFuncSetAttribute((const void*)msm_part_scatter_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)PART_SCATTER_LDS_MAX));
    ARK_HIP_TRY(hipFuncSetAttribute((const void*)msm_part_finish_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    160 * 1024 - 64));
    ws.attr_set = true;
  }
  // super-buckets too large for one workgroup of pass B (skewed scalars) are finished in slices; below 2^18 keys per
  // window none can exist (PART_BIG = 2^17 and a window of uniform digits is spread over all its super-buckets anyway)
  static const bool big_env = [] {
    const char* e = getenv("ARK_HIP_MSM_BIG_SLICES");
    return !(e && atoi(e) == 0);
  }();
  const bool big_on = big_env && n > (size_t)2 * PART_BIG;
  const size_t big_region = (size_t)nsuper + 2 * nb;   // u32 words per window group (each group's share is smaller)
  if (big_on && ws.big.ensure(2 * big_region * 4)) return -3;
  bool lazy = false;  // Fp384 G1: the accumulate kernels on carry-free 28-bit limbs (fp28.cuh); ARK_HIP_MSM_LAZY=0: saturated
  if constexpr (C::LAZY_A) lazy = msm_lazy_enabled();
  // Few slots with long runs (narrow scalars in one or two windows at large n): each run is walked by `run_parts` lanes
  // (msm_accumulate_parts_kernel) until the lanes make two rounds over the chip's resident ones (2 waves x 4 SIMDs x 256 CUs x 64)
  u32 run_parts = 1;
  size_t parts_region = 0;
  if (lazy && !pl.shared && !piece) {
    static const size_t lanes_target = [] {
      const char* e = getenv("ARK_HIP_MSM_PARTS_LANES");   // 0: never split runs
      return e ? (size_t)atol(e) : (size_t)262144;   // two rounds of the resident lanes: measured better than one (u16 2^24 4.24 -> 3.82 ms,
                                                     // u32 8.85 -> 6.88), four / eight rounds no better (profiles/r4_narrow_scalars.txt)
    }();
    double expect = (double)n * W;   // sorted entries: from the width classes where they were measured
    if (have_widths) {
      expect = 0.0;
      for (int k = 1; k < MSM_WIDTH_CLASSES; k++) {
        int need = (MSM_WIDTH_TOP[k] + 1 + c - 1) / c;
        expect += (double)widths.count[k] * (need > W ? W : need);
      }
    }
    const double mean_run = expect / (double)nb;
    while (run_parts < 8 && nb * (size_t)(2 * run_parts) <= lanes_target && mean_run >= 32.0 * (2 * run_parts)) run_parts *= 2;
    if (pl.parts) run_parts = pl.parts;   // small plain MSMs: the plan's narrow windows count on split runs (msm_make_plan)
    if (const char* e = getenv("ARK_HIP_MSM_RUN_PARTS")) {   // test / tuning knob: force 1, 2, 4 or 8 lanes per run
      const int v = atoi(e);
      if (v == 1 || v == 2 || v == 4 || v == 8) run_parts = (u32)v;
    }
    if (run_parts > 1) {
      parts_region = (size_t)run_parts * nb * Pt::BYTES;
      if (ws.parts.ensure(2 * parts_region)) return -3;
    }
  }
  const int W0 = ngroups == 2 ? (W + 1) / 2 : W;   // windows of group 0
  struct Group {
    int w0, Wg;
    size_t slot0, nslots, nbk_g;   // first sort slot, sort slots, accumulated buckets
    u32 *keys, *sorted, *offsets, *thist, *toff, *order, *sums, *ohist, *ooff, *hctr;
    u32* big;                      // sliced pass B (msm_sort.cuh): [list of super-buckets, counters[nslots], cursors[nslots]]
    u32* big_count;                // their number (hctr[8 + g], zeroed with the other counters)
    uint2* part;
    HeavyEntry* hlist;
    uint2* hitems;
    char *hpart, *buckets;
    size_t max_heavy;
  } grp[2];
  for (int g = 0; g < ngroups; g++) {
    Group& G = grp[g];
    G.w0 = g == 0 ? 0 : W0;
    G.Wg = g == 0 ? W0 : W - W0;
    G.slot0 = (size_t)G.w0 << Bbits;
    G.nslots = (size_t)G.Wg << Bbits;
    G.nbk_g = pl.shared ? nbk : G.nslots;
    G.keys = keys + (size_t)G.w0 * n;
    G.sorted = sorted + (size_t)G.w0 * n;
    G.part = part + (size_t)G.w0 * n;
    G.offsets = offsets + G.slot0 + (size_t)g;                     // nslots + 1 entries each
    G.thist = thist + ((size_t)G.w0 << HB) * ntiles;
    G.toff = toff + ((size_t)G.w0 << HB) * ntiles + (size_t)g;     // (Wg << HB) * ntiles + 1 entries each
    G.order = order + G.slot0;
    G.sums = sums + (size_t)g * nsums;
    const size_t noblk0 = ((size_t)(pl.shared ? nbk : ((size_t)W0 << Bbits)) + ORDER_TILE - 1) / ORDER_TILE;
    G.ohist = (u32*)ws.ohist.p + (size_t)g * noblk0 * ORDER_BINS;
    G.ooff = (u32*)ws.ooff.p + (size_t)g * (noblk0 * ORDER_BINS + 1);
    G.hctr = (u32*)ws.hctr.p + 4 * g;
    G.big = big_on ? (u32*)ws.big.p + (size_t)g * big_region : nullptr;
    G.big_count = (u32*)ws.hctr.p + 8 + g;
    const size_t ent0 = (size_t)n * W0;
    const size_t mh0 = ent0 / 64 + 1, mi0 = ent0 / HEAVY_CHUNK + mh0 + 1;   // group 0's share of the heavy-run arrays
    G.max_heavy = g == 0 ? (ngroups == 2 ? mh0 : max_heavy) : max_heavy - mh0;
    G.hlist = (HeavyEntry*)ws.hlist.p + (g == 0 ? 0 : mh0);
    G.hitems = (uint2*)ws.hitems.p + (g == 0 ? 0 : mi0);
    G.hpart = (char*)ws.hpart.p + (g == 0 ? 0 : mi0) * Pt::BYTES;
    G.buckets = d_buckets + (pl.shared ? 0 : G.slot0 * Pt::BYTES);
  }
  // partition sort of one group: (A) split by the low bucket bits with LDS counters, (B) finish each super-bucket in LDS;
  // then the processing order of its buckets, heaviest load class first
  auto sort_group = [&](const Group& G, hipStream_t st, bool mark) -> int {
    const u32 nsuper_g = (u32)G.Wg << HB;
    const size_t nthist_g = (size_t)nsuper_g * ntiles;
    hipLaunchKernelGGL(msm_part_hist_kernel, dim3(ntiles, G.Wg), dim3(256), (size_t)4 << HB, st, G.keys, (u32)n, HB, LB,
                       ntiles, ptile, G.thist);
    scan_exclusive(G.thist, nthist_g, G.sums, G.toff, st);
    if (mark && timing) ARK_HIP_TRY(hipEventRecord(job.ev[2], st));
    hipLaunchKernelGGL(msm_part_scatter_kernel, dim3(ntiles, G.Wg), dim3(1024), lds_a, st, G.keys, (u32)n, HB, LB, ntiles,
                       ptile, G.toff, G.part, compacted ? (const u32*)ws.cidx.p : (const u32*)nullptr);
    hipLaunchKernelGGL(msm_part_finish_kernel, dim3(nsuper_g), dim3(1024), lds_b, st, G.part, G.toff, ntiles, LB, nsuper_g,
                       stage_cap, big_on ? PART_BIG : 0u, G.big_count, G.big, G.offsets, G.sorted);
    if (big_on) {
      u32* const bigcnt = G.big + nsuper_g;     // counters and cursors of a listed super-bucket: zeroed by the finish kernel
      u32* const cursor = bigcnt + G.nslots;
      hipLaunchKernelGGL(msm_part_big_hist_kernel, dim3(PART_BIG_SLICES, PART_BIG_GRID_Y), dim3(1024), (size_t)4 << LB, st,
                         G.part, G.toff, ntiles, LB, G.big_count, G.big, bigcnt);
      hipLaunchKernelGGL(msm_part_big_place_kernel, dim3(PART_BIG_SLICES, PART_BIG_GRID_Y), dim3(1024),
                         ((size_t)8 << LB) + 4096, st, G.part, G.toff, ntiles, LB, G.big_count, G.big, bigcnt, cursor, G.offsets,
                         G.sorted);
    }
    int shift = 0;  // class width 2^shift so that the mean load falls around class 32..63
    while ((mean_load >> shift) >= 64) shift++;
    const u32 noblk_g = (u32)((G.nbk_g + ORDER_TILE - 1) / ORDER_TILE);
    const int wsum = pl.shared ? W : 1;
    hipLaunchKernelGGL(msm_order_hist_kernel, dim3(noblk_g), dim3(256), 0, st, G.offsets, G.nbk_g, shift, noblk_g, wsum, mwin,
                       G.ohist);
    scan_exclusive(G.ohist, (size_t)noblk_g * ORDER_BINS, G.sums, G.ooff, st);
    hipLaunchKernelGGL(msm_order_scatter_kernel, dim3(noblk_g), dim3(256), 0, st, G.offsets, G.nbk_g, shift, noblk_g, wsum, mwin,
                       G.ooff, G.order);
    return 0;
  };
  // heavy runs + the lane-per-bucket kernel of one group
  auto accumulate_group = [&](const Group& G, hipStream_t st) -> int {   // (st is reassigned below)
    // runs too long for one lane: chunk partials by one wave each, combined per run (empty for uniform scalars)
    hipLaunchKernelGGL(msm_find_heavy_kernel, dim3((u32)((G.nslots + 255) / 256)), dim3(256), 0, st, G.offsets, (u32)G.nslots,
                       forced_thresh, G.hctr, G.hlist, G.hitems);
    constexpr u32 LN = C::FA::LANES;                        // lanes per element of the accumulate field
    constexpr size_t ACCB = AccOps<C>::ACC_BYTES;           // one parked accumulator (the form the kernels sum in)
    const u32 hthreads = ACCB * (256 / LN) > 49152 ? 128 : 256;  // one LDS tree per wave, <= 48 KiB per workgroup
    hipStream_t const acc_st = st;
    if (heavy_side) {
      ARK_HIP_TRY(hipEventRecord(ws.grp_ev[0], st));
      ARK_HIP_TRY(hipStreamWaitEvent(ws.side, ws.grp_ev[0], 0));
      st = ws.side;
    }
    hipLaunchKernelGGL((msm_heavy_partial_kernel<C>), dim3(1024), dim3(hthreads), (hthreads / LN) * ACCB, st,
                       (const char*)d_points, G.sorted, G.offsets, G.hctr, (const uint2*)G.hitems, wstride, Bbits, (u32)G.nslots, G.hpart);
    const u32 combine_grid = G.max_heavy < 16384 ? (u32)G.max_heavy : 16384u;  // grid-stride over the heavy runs
    if (pl.shared)
      hipLaunchKernelGGL((msm_heavy_combine_kernel<C, true>), dim3(combine_grid), dim3(ARK_HEAVY_COMBINE_THREADS), (ARK_HEAVY_COMBINE_THREADS / LN) * ACCB, st, G.hctr,
                         (const HeavyEntry*)G.hlist, (const char*)G.hpart, HB, LB, G.offsets, G.sorted, 0, (char*)ws.hfinal.p);
    else
      hipLaunchKernelGGL((msm_heavy_combine_kernel<C, false>), dim3(combine_grid), dim3(ARK_HEAVY_COMBINE_THREADS), (ARK_HEAVY_COMBINE_THREADS / LN) * ACCB, st, G.hctr,
                         (const HeavyEntry*)G.hlist, (const char*)G.hpart, HB, LB, G.offsets, G.sorted, accum, G.buckets);
    if (heavy_side) {
      ARK_HIP_TRY(hipEventRecord(ws.grp_ev[1], ws.side));
      st = acc_st;
    }
    if (pl.shared) {
      if constexpr (C::LAZY_A) {
        if (lazy)
          hipLaunchKernelGGL((msm_accumulate_shared_lazy_kernel<C>), dim3((u32)((nbk * C::FA::LANES + 255) / 256)), dim3(256), 0, st,
                             (const char*)d_points, wstride, G.sorted, G.offsets, G.order, (u32)nbk, W, Bbits,
                             (const u32*)G.hctr + 2, HB, LB, G.buckets);
      }
      if (!lazy)
        hipLaunchKernelGGL((msm_accumulate_shared_kernel<C>), dim3((u32)((nbk * C::FA::LANES + 255) / 256)), dim3(256), 0, st,
                           (const char*)d_points, wstride, G.sorted, G.offsets, G.order, (u32)nbk, W, Bbits,
                           (const u32*)G.hctr + 2, HB, LB, G.buckets);
      if (heavy_side) ARK_HIP_TRY(hipStreamWaitEvent(st, ws.grp_ev[1], 0));
      hipLaunchKernelGGL((msm_apply_heavy_kernel<C>), dim3(1024), dim3(64), 0, st, (const u32*)G.hctr,
                         (const HeavyEntry*)G.hlist, G.offsets, G.sorted, (const char*)ws.hfinal.p, W, Bbits, HB, LB, G.buckets);
    } else {
      if constexpr (C::LAZY_A) {
        if (lazy && run_parts > 1) {
          char* const pbuf = (char*)ws.parts.p + (size_t)(&G - grp) * parts_region;
          hipLaunchKernelGGL((msm_accumulate_parts_kernel<C>), dim3((u32)((G.nslots * run_parts * C::FA::LANES + 255) / 256)),
                             dim3(256), 0, st, (const char*)d_points, G.sorted, G.offsets, G.order, (u32)G.nslots,
                             (const u32*)G.hctr + 2, HB, LB, run_parts, pbuf);
          hipLaunchKernelGGL((msm_sum_parts_kernel<C>), dim3((u32)((G.nslots * C::FA::LANES + 255) / 256)), dim3(256), 0, st,
                             G.offsets, (u32)G.nslots, (const u32*)G.hctr + 2, HB, LB, run_parts, (const char*)pbuf, accum,
                             G.buckets);
        } else if (lazy) {
          hipLaunchKernelGGL((msm_accumulate_lazy_kernel<C>), dim3((u32)((G.nslots * C::FA::LANES + 255) / 256)), dim3(256), 0, st,
                             (const char*)d_points, G.sorted, G.offsets, G.order, (u32)G.nslots, (const u32*)G.hctr + 2, HB, LB,
                             accum, G.buckets);
        }
      }
      if (!lazy)
        hipLaunchKernelGGL((msm_accumulate_kernel<C>), dim3((u32)((G.nslots * C::FA::LANES + 255) / 256)), dim3(256), 0, st,
                           (const char*)d_points, G.sorted, G.offsets, G.order, (u32)G.nslots, (const u32*)G.hctr + 2, HB, LB,
                           accum, G.buckets);
      if (heavy_side) ARK_HIP_TRY(hipStreamWaitEvent(st, ws.grp_ev[1], 0));
    }
    return 0;
  };
  if (int rc = sort_group(grp[0], stream, true)) return rc;
  if (timing) ARK_HIP_TRY(hipEventRecord(job.ev[3], stream));
  if (ngroups == 2) {   // group 1's sort starts when group 0's accumulate kernel does
    ARK_HIP_TRY(hipEventRecord(ws.grp_ev[0], stream));
    ARK_HIP_TRY(hipStreamWaitEvent(ws.side, ws.grp_ev[0], 0));
    if (int rc = sort_group(grp[1], ws.side, false)) return rc;
    ARK_HIP_TRY(hipEventRecord(ws.grp_ev[1], ws.side));
  }
  if (piece && piece->after_prev) ARK_HIP_TRY(hipStreamWaitEvent(stream, piece->after_prev, 0));  // the buckets' previous writer
  // bucket reduction of the windows [w0, w0 + wg): level 0 (chunked running sums), the bit-sliced sums, the chunk sums
  constexpr u32 LNr = C::FA::LANES;
  const bool two_level = nchunks > 1;
  auto reduce_windows = [&](size_t w0, size_t wg, hipStream_t st) {
    const size_t bo = w0 * m * Pt::BYTES;                       // (S, A) pairs of window w0
    const size_t po = w0 * (size_t)Q * nchunks * Pt::BYTES;     // chunk partials of window w0
    // level 0: the two chains of a chunk on one lane (2 L0 dependent additions) or on two waves (L0 + 1, twice the lanes):
    // the split form where its lanes still fit the chip's resident ones -- 2 waves x 1024 SIMDs (ARK_HIP_MSM_SPLIT_LEVEL=0/1 forces)
    static const int split_env = [] { const char* e = getenv("ARK_HIP_MSM_SPLIT_LEVEL"); return e ? atoi(e) : -1; }();
    // measured (profiles/r5_reduce_split_level.txt, BLS12-381 G1): 2^16 0.48 -> 0.44 ms, 2^18 0.70 -> 0.54, 2^20 0.89 -> 0.71
    // (split lanes <= 131 072: one round of the chip); 2^23 2.31 -> 2.00 (L0 = 32: 33 steps instead of 64 outweigh the 1.75
    // rounds of 229 376 lanes); 2^21 / 2^22 (L0 = 8, 245 760 lanes) 1.15 -> 1.21 / equal; 2^24, 2^26 equal; the lane-pair
    // curves (G2) lose 2-3 % everywhere: off there
    const size_t split_lanes = m * wg * 2;
    const bool split_level = split_env >= 0 ? split_env != 0
                                            : (LNr == 1 && L0 >= 2 && (split_lanes <= 131072 || (L0 >= 16 && split_lanes <= 262144)));
    if (split_level)
      hipLaunchKernelGGL((msm_reduce_level_split_kernel<C>), dim3((u32)((m * wg * LNr + 63) / 64)), dim3(128), 0, st,
                         (const char*)d_buckets + w0 * mwin * Pt::BYTES, L0, (u32)(m * wg), (char*)ws.lvlS[0].p + bo,
                         (char*)ws.lvlA[0].p + bo);
    else
    hipLaunchKernelGGL((msm_reduce_level_kernel<C>), dim3((u32)((m * wg * LNr + 127) / 128)), dim3(128), 0, st,
                       (const char*)d_buckets + w0 * mwin * Pt::BYTES, L0, (u32)(m * wg), (char*)ws.lvlS[0].p + bo,
                       (char*)ws.lvlA[0].p + bo);
    constexpr size_t ACCBr = AccOps<C>::ACC_BYTES;
    const u32 rthreads = ACCBr * (256 / LNr) > 49152 ? 128 : 256;  // LDS tree within 48 KiB
    hipLaunchKernelGGL((msm_reduce_bits_kernel<C>), dim3(nchunks, Q, (u32)wg), dim3(rthreads), (rthreads / LNr) * ACCBr, st,
                       (const char*)ws.lvlS[0].p + bo, (const char*)ws.lvlA[0].p + bo, (u32)m, nbits, chunk,
                       (char*)ws.lvlS[1].p + po);
    if (two_level)
      hipLaunchKernelGGL((msm_sum_chunks_kernel<C>), dim3((u32)(wg * Q)), dim3(64), (64 / LNr) * AccOps<C>::ACC_BYTES, st,
                         (const char*)ws.lvlS[1].p + po, (u32)(wg * Q), nchunks, (char*)ws.lvlA[1].p + w0 * Q * Pt::BYTES);
  };
  const bool whole_job = !piece || piece->last;   // (a non-final piece of a streamed MSM leaves the reduction to the last one)
  // Two window groups: group 0's reduction -- latency-bound chains and LDS trees below ~2^22 pairs -- runs on the side stream
  // UNDER group 1's accumulate kernel (the side stream has finished group 1's sort by then).
  // Measured and OFF by default (ARK_HIP_MSM_SPLIT_REDUCE=1 enables): the reduction's additions compete with the accumulate
  // kernel for the same vector ALU, and what the overlap hides is less than what two half-size launches add -- 2^20 3.82 ->
  // 4.05 ms, 2^22 11.33 -> 11.66, 2^24 36.8 -> 37.9 with both overlaps on (profiles/r4_split_reduce_window_groups_ab.txt).
  static const bool split_env = getenv("ARK_HIP_MSM_SPLIT_REDUCE") && getenv("ARK_HIP_MSM_SPLIT_REDUCE")[0] == '1';
  const bool split_reduce = split_env && ngroups == 2 && whole_job && Wr == W;
  if (int rc = accumulate_group(grp[0], stream)) return rc;
  if (ngroups == 2) {
    if (split_reduce) {
      ARK_HIP_TRY(hipEventRecord(ws.grp_ev[0], stream));
      ARK_HIP_TRY(hipStreamWaitEvent(ws.side, ws.grp_ev[0], 0));
    }
    ARK_HIP_TRY(hipStreamWaitEvent(stream, ws.grp_ev[1], 0));
    if (split_reduce) {
      reduce_windows(0, (size_t)grp[0].Wg, ws.side);
      ARK_HIP_TRY(hipEventRecord(ws.grp_ev[1], ws.side));
    }
    if (int rc = accumulate_group(grp[1], stream)) return rc;
  }
  if (timing) ARK_HIP_TRY(hipEventRecord(job.ev[4], stream));
  if (piece) ARK_HIP_TRY(hipEventRecord(piece->after_this, stream));
  if (piece && !piece->last) {  // the reduction belongs to the last piece: only the scalar-range flag goes back
    ARK_HIP_TRY(hipGetLastError());
    ARK_HIP_TRY(hipMemcpyAsync((char*)job.pinned + npairs * Pt::BYTES, (const u32*)ws.hctr.p + 3, 4, hipMemcpyDeviceToHost, stream));
    if (timing) ARK_HIP_TRY(hipEventRecord(job.ev[5], stream));
    ARK_HIP_TRY(hipEventRecord(job.done, stream));
    job.pl = pl;
    job.npairs = npairs;
    job.timing = timing;
    job.no_result = true;
    job.busy = true;
    return slot;
  }

  ARK_HIP_TRY(hipEventRecord(job.acc_done, stream));   // the accumulate kernels are through: the reduction's 0.4-9 ms remain
  if (split_reduce) {
    reduce_windows((size_t)grp[1].w0, (size_t)grp[1].Wg, stream);
    ARK_HIP_TRY(hipStreamWaitEvent(stream, ws.grp_ev[1], 0));   // group 0's sums
  } else {
    reduce_windows(0, (size_t)Wr, stream);
  }
  const char* d_sums = two_level ? (const char*)ws.lvlA[1].p : (const char*)ws.lvlS[1].p;
  ARK_HIP_TRY(hipGetLastError());
  ARK_HIP_TRY(hipMemcpyAsync(job.pinned, d_sums, npairs * Pt::BYTES, hipMemcpyDeviceToHost, stream));
  ARK_HIP_TRY(hipMemcpyAsync((char*)job.pinned + npairs * Pt::BYTES, (const u32*)ws.hctr.p + 3, 4, hipMemcpyDeviceToHost, stream));
  if (timing) ARK_HIP_TRY(hipEventRecord(job.ev[5], stream));
  ARK_HIP_TRY(hipEventRecord(job.done, stream));
  job.pl = pl;
  job.nbits = nbits;
  job.log2L0 = log2L0;
  job.short_job = (double)n * (double)W <= 6.0e6;   // (up to ~2^18 full-width pairs)
  job.Q = Q;
  job.npairs = npairs;
  job.d_sums = d_sums;
  job.timing = timing;
  job.busy = true;
  return slot;
}

// The caller sits on the critical path of its prover: poll the completion event instead of sleeping on it (a blocked
// host thread was measured to wake up ~1 ms after a 38 ms job on some hosts; short jobs never sleep anyway).
// ARK_HIP_WAIT=block restores hipEventSynchronize (frees the core while the GPU works).
static inline hipError_t msm_wait_event(hipEvent_t ev) {
  static const bool block = [] {
    const char* e = getenv("ARK_HIP_WAIT");
    return e && e[0] == 'b';
  }();
  if (block) return hipEventSynchronize(ev);
  bool polled = false;
  for (;;) {
    const hipError_t e = hipEventQuery(ev);
    if (e != hipErrorNotReady) {
      if (polled) (void)hipGetLastError();  // "not ready" is not an error: do not leave it behind as the thread's last one
      return e;
    }
    polled = true;
#if defined(__x86_64__)
    __builtin_ia32_pause();
#endif
  }
}

// The serial tail of an MSM, on the host:  T_w = sum A + L0 * sum_b 2^b U_b ;  total = sum_w 2^(off_w) T_w  (window
// combine of mod.rs:489-502) -- evaluated as ONE Horner over bit positions: A_w sits at off_w, U_(w,b) at
// off_w + log2 L0 + b, so the chain doubles once per scalar bit (<= 256 doublings) instead of once per bit inside every
// window and again between windows (~2 c W).  parts: [w][q] XYZZ points, q < nbits: U_(w,q), q == nbits: A_w, row
// stride Q; off: Wr + 1 bit offsets.  A prepared base set has a single T (Wr = 1).
// window w's own sum  T_w = A_w + 2^log2L0 sum_b 2^b U_(w,b)
template <class C>
XYZZ<typename C::F> msm_host_window_sum(const char* parts, u32 Q, int nbits, int log2L0, int w) {
  typedef typename C::F F;
  typedef XYZZ<F> Pt;
  auto part_at = [&](u32 q) { return Pt::load(parts + ((size_t)w * Q + q) * Pt::BYTES); };
  Pt h = Pt::zero();
  for (int b = nbits - 1; b >= 0; b--) {
    h = xyzz_dbl<F>(h);
    Pt ub = part_at((u32)b);
    xyzz_add<F>(h, ub);
  }
  for (int i = 0; i < log2L0; i++) h = xyzz_dbl<F>(h);
  Pt asum = part_at((u32)nbits);
  xyzz_add<F>(h, asum);
  return h;
}
// the windows' own sums as a batch of independent tasks for the helper pool (task w -> T[w])
template <class C>
struct MsmWindowSums {
  const char* parts;
  u32 Q;
  int nbits, log2L0;
  XYZZ<typename C::F>* T;
  static void task(void* ctx, int w) {
    const MsmWindowSums& s = *(const MsmWindowSums*)ctx;
    s.T[(size_t)w] = msm_host_window_sum<C>(s.parts, s.Q, s.nbits, s.log2L0, w);
  }
};
// total = sum_w 2^(off_w) T_w: the doublings between the windows, the only serial part of the tail
template <class C>
XYZZ<typename C::F> msm_host_combine_windows(const XYZZ<typename C::F>* T, int Wr, const int* off) {
  typedef typename C::F F;
  XYZZ<F> total = T[(size_t)Wr - 1];
  for (int w = Wr - 2; w >= 0; w--) {
    for (int i = off[w]; i < off[w + 1]; i++) total = xyzz_dbl<F>(total);   // the width of window w
    xyzz_add<F>(total, T[(size_t)w]);
  }
  return total;
}
template <class C>
XYZZ<typename C::F> msm_host_fold(const char* parts, u32 Q, int Wr, int nbits, int log2L0, const int* off) {
  typedef typename C::F F;
  typedef XYZZ<F> Pt;
  auto part_at = [&](int w, u32 q) { return Pt::load(parts + ((size_t)w * Q + q) * Pt::BYTES); };
  // Over Fp2 (G2) a point operation costs the host ~3 x what it costs over Fp384, and the tail's Wr (nbits + 1) additions
  // are half of it (BLS12-377 G2 2^16: 0.9 of 2.6 ms): the windows' own sums T_w = sum A + 2^log2L0 sum_b 2^b U_b are
  // independent -- the process-wide helper pool (hostpool.hpp) computes them together with this thread, and only the 250-odd
  // doublings between the windows (+ Wr additions) stay serial.  One lane per point curves keep the single Horner here (their
  // whole tail is ~0.25 ms); msm_finish splits it for short jobs.
  if (C::FA::LANES == 2 && Wr >= 4 && HostPool::instance().helpers() > 0) {
    std::vector<Pt> T((size_t)Wr);
    MsmWindowSums<C> ws{parts, Q, nbits, log2L0, T.data()};
    HostPool::instance().run_all(&MsmWindowSums<C>::task, &ws, Wr);   // the pool's helpers + this thread; no thread is created here
    return msm_host_combine_windows<C>(T.data(), Wr, off);
  }
  int top = 0;
  for (int w = 0; w < Wr; w++) top = std::max(top, off[w] + log2L0 + nbits - 1);
  Pt total = Pt::zero();
  for (int pos = top; pos >= 0; pos--) {
    total = xyzz_dbl<F>(total);
    for (int w = Wr - 1; w >= 0; w--) {
      const int rel = pos - off[w];
      if (rel < 0) continue;
      if (rel == 0) {
        Pt asum = part_at(w, (u32)nbits);
        xyzz_add<F>(total, asum);
      }
      const int b2 = rel - log2L0;
      if (b2 >= 0 && b2 < nbits) {
        Pt ub = part_at(w, (u32)b2);
        xyzz_add<F>(total, ub);
      }
    }
  }
  return total;
}

// Wait for job `slot` and finish it on the host: out_xyz = Jacobian x|y|z Montgomery limbs (group.rs:34-41);
// identity = (R, R, 0) (group.rs:145-151).  Releases the slot.
template <class C>
int msm_finish(MsmWorkspace& ws, int slot, uint64_t* out_xyz, MsmTimings* tm) {
  typedef typename C::F F;
  typedef XYZZ<F> Pt;
  if (slot < 0 || slot >= MSM_JOBS) return -1;
  MsmJob& job = ws.jobs[slot];
  if (!job.busy) return -1;
  struct Release {
    MsmWorkspace& ws;
    MsmJob& j;
    ~Release() {
      std::lock_guard<std::mutex> lock(ws.mu);
      j.busy = false;
    }
  } release{ws, job};
  if (job.empty) {
    if (out_xyz) Jac<F>::zero().store(out_xyz);
    return 0;
  }
  const MsmPlan& pl = job.pl;
  const int c = pl.c, W = pl.W, Wr = pl.red_windows(), nbits = job.nbits;
  const u32 Q = job.Q;
  // The tail -- ~5400 field products over Fp384, 0.2 ms: a fifth of a 2^16 call, 3 % of a 2^20 one -- is half the windows' own
  // sums, which are independent.  They go to the process-wide helper pool (hostpool.hpp; round 5 created seven threads per
  // call here and let them spin without bound).  The batch is opened BEFORE the wait for the GPU, so the helpers' wake-up hides
  // under the kernels -- for a short job at once, for a longer one when its accumulate kernels are through (job.acc_done: the
  // reduction's 0.4-9 ms remain; a helper spins for at most a millisecond, then sleeps in 50 us steps); the gate opens when the
  // part sums have landed, and this thread claims windows from the same counter -- helpers that are busy elsewhere or asleep
  // cost nothing but their share.  A job whose GPU work has already finished keeps the single Horner.
  // ARK_HIP_HOST_TAIL_THREADS=0: no pool.
  std::vector<Pt> early_T;
  MsmWindowSums<C> early_ws{(const char*)job.pinned, Q, nbits, job.log2L0, nullptr};
  HostPool::Handle early;
  struct EarlyGuard {   // an error return between here and the fold must not leave the batch open
    HostPool::Handle& h;
    ~EarlyGuard() {
      if (h) HostPool::instance().cancel(h);
    }
  } early_guard{early};
  static const bool tail_all = [] { const char* e = getenv("ARK_HIP_HOST_TAIL_ALL"); return !(e && e[0] == '0'); }();   // A/B: 0 = short jobs only
  if (out_xyz && !job.no_result && (job.short_job || tail_all) && Wr >= 8 && HostPool::instance().helpers() > 0 &&
      hipEventQuery(job.done) == hipErrorNotReady) {
    (void)hipGetLastError();
    if (!job.short_job && job.acc_done) ARK_HIP_TRY(msm_wait_event(job.acc_done));
    try {
      early_T.resize((size_t)Wr);
      early_ws.T = early_T.data();
      early = HostPool::instance().open(&MsmWindowSums<C>::task, &early_ws, Wr);
    } catch (...) {   // no memory for Wr points: the single Horner
      early.reset();
    }
  }
  ARK_HIP_TRY(msm_wait_event(job.done));
  const u32 h_err = *(const u32*)((const char*)job.pinned + job.npairs * Pt::BYTES);
  if (h_err) return -4;  // scalar out of range
  if (job.no_result) {  // a non-final piece of a streamed MSM: out_xyz untouched
    if (tm && job.timing) {
      (void)hipEventElapsedTime(&tm->total, job.ev[0], job.ev[5]);
      tm->c = c;
      tm->W = W;
    }
    return 0;
  }

  if (out_xyz) {   // (nullptr: the caller folds sums of its own -- the sharded combine, over all ranks' sums)
    std::vector<int> off((size_t)Wr + 1);
    off[0] = 0;
    for (int w = 0; w < Wr; w++) off[w + 1] = off[w] + msm_window_width(w, c, W, pl.narrow);
    Pt total;
    if (early) {
      HostPool::instance().run(early);   // opens the gate, takes part, returns when every window's sum is in early_T
      early.reset();
      total = msm_host_combine_windows<C>(early_T.data(), Wr, off.data());
    } else {
      total = msm_host_fold<C>((const char*)job.pinned, Q, Wr, nbits, job.log2L0, off.data());
    }
    xyzz_to_jac<F>(total).store(out_xyz);
  }

  if (tm && job.timing) {
    (void)hipEventElapsedTime(&tm->digits, job.ev[0], job.ev[1]);
    (void)hipEventElapsedTime(&tm->scan, job.ev[1], job.ev[2]);
    (void)hipEventElapsedTime(&tm->scatter, job.ev[2], job.ev[3]);
    (void)hipEventElapsedTime(&tm->accumulate, job.ev[3], job.ev[4]);
    (void)hipEventElapsedTime(&tm->reduce, job.ev[4], job.ev[5]);
    (void)hipEventElapsedTime(&tm->total, job.ev[0], job.ev[5]);
    tm->c = c;
    tm->W = W;
  }
  return 0;
}

// Build the table of per-window multiples for a fixed base set: table[w][i] = 2^(offset_w) * bases[i], affine,
// w < pl.W, row stride n.  `table` must hold pl.W * n affine points.  Asynchronous on `stream`.
// ---- one process per GPU: the part sums of all ranks, added on the device --------------------------------------------------
// Every rank's MSM ends in the same array of part sums (per window: the bit-sliced sums U_b and the plain sum A) when the
// ranks share one plan -- equal shard sizes.  The sharded entries all-gather THOSE (npairs XYZZ points, ~40 KB for 2^24
// pairs) behind a 64-byte header instead of finished partial results: no rank runs a host tail before the collective, one
// kernel adds the G copies of every part, and the single host tail runs on the sums of the whole job
// (variable_base/mod.rs:542-557: the chunk sum, moved in front of the window combine -- elliptic-curve addition commutes).
struct MsmSumsHeader {   // 64 bytes in front of a rank's part sums; all ranks must agree on every field but `err`
  int c, W, narrow, shared, nbits, log2L0;
  u32 Q, npairs;
  u32 err;               // this rank's scalar-range flag (msm_digits_kernel)
  u32 pad[7];
};
static_assert(sizeof(MsmSumsHeader) == 64, "header layout");
struct MsmSumsInfo {
  MsmSumsHeader h;
  const char* d_sums;
};
template <class C>
__global__ void __launch_bounds__(128) msm_sum_ranks_kernel(const char* __restrict__ blocks, int world, size_t block_bytes,
                                                            u32 npairs, char* __restrict__ out) {
  typedef AccOps<C> Ops;
  typedef typename Ops::Pt Pt;
  const u32 t = (blockIdx.x * blockDim.x + threadIdx.x) / Ops::LANES;
  if (t >= npairs) return;
  typename Ops::Acc acc = Ops::zero();
  for (int r = 0; r < world; r++) {
    const Pt x = Pt::load(blocks + (size_t)r * block_bytes + sizeof(MsmSumsHeader) + (size_t)t * Pt::BYTES);
    Ops::add(acc, x);
  }
  Ops::fin(acc).store(out + (size_t)t * Pt::BYTES);
}
template <class C>
int msm_sum_ranks(const void* d_blocks, int world, size_t block_bytes, u32 npairs, void* d_out, hipStream_t stream) {
  if (npairs == 0) return 0;
  constexpr u32 LN = C::FA::LANES;
  hipLaunchKernelGGL((msm_sum_ranks_kernel<C>), dim3((npairs * LN + 127) / 128), dim3(128), 0, stream, (const char*)d_blocks,
                     world, block_bytes, npairs, (char*)d_out);
  ARK_HIP_TRY(hipGetLastError());
  return 0;
}
// the host tail over part sums laid out by `h` (host memory): out_xyz = Jacobian limbs
template <class C>
int msm_fold_sums(const MsmSumsHeader& h, const void* h_sums, uint64_t* out_xyz) {
  typedef typename C::F F;
  MsmPlan pl{};
  pl.c = h.c;
  pl.W = h.W;
  pl.narrow = h.narrow;
  pl.shared = h.shared != 0;
  const int Wr = pl.red_windows();
  std::vector<int> off((size_t)Wr + 1);
  off[0] = 0;
  for (int w = 0; w < Wr; w++) off[w + 1] = off[w] + msm_window_width(w, h.c, h.W, h.narrow);
  const XYZZ<F> total = msm_host_fold<C>((const char*)h_sums, h.Q, Wr, h.nbits, h.log2L0, off.data());
  xyzz_to_jac<F>(total).store(out_xyz);
  return 0;
}
// geometry + device pointer of an enqueued job's part sums (valid until the slot is finished)
static inline int msm_job_sums(MsmWorkspace& ws, int slot, MsmSumsInfo* out) {
  if (slot < 0 || slot >= MSM_JOBS) return -1;
  const MsmJob& job = ws.jobs[slot];
  if (!job.busy || job.no_result) return -1;
  *out = MsmSumsInfo{};
  if (job.empty) return 1;   // n == 0: no sums (the caller contributes the identity)
  out->h.c = job.pl.c;
  out->h.W = job.pl.W;
  out->h.narrow = job.pl.narrow;
  out->h.shared = job.pl.shared ? 1 : 0;
  out->h.nbits = job.nbits;
  out->h.log2L0 = job.log2L0;
  out->h.Q = job.Q;
  out->h.npairs = (u32)job.npairs;
  out->d_sums = job.d_sums;
  return 0;
}

template <class C>
int msm_prepare_table(const void* d_bases, size_t n, const MsmPlan& pl, void* d_table, void* d_tmp, hipStream_t stream) {
  typedef typename C::F F;  // d_tmp: n * XYZZ<F>::BYTES of scratch (one unnormalised row)
  const size_t row = n * Affine<F>::BYTES;
  if (n == 0) return 0;
  ARK_HIP_TRY(hipMemcpyAsync(d_table, d_bases, row, hipMemcpyDeviceToDevice, stream));
  for (int w = 0; w + 1 < pl.W; w++) {
    const int width = msm_window_width(w, pl.c, pl.W, pl.narrow);
    hipLaunchKernelGGL((msm_table_step_kernel<C>), dim3((u32)((n + 127) / 128)), dim3(128), 0, stream,
                       (const char*)d_table + (size_t)w * row, (char*)d_tmp, n, width);
    xyzz_to_affine_batched_launch<F>(d_tmp, (char*)d_table + (size_t)(w + 1) * row, n, stream);
  }
  ARK_HIP_TRY(hipGetLastError());
  return 0;
}

}  // namespace arkhip
