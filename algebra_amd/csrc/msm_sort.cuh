// Bucket sort of the (window, scalar) keys for the MSM: replaces one device-scope atomic per key (the
// histogram / scatter of a naive counting sort: 2.2e8 atomics at 2^24 x 13 windows, ~27 G/s on this chip)
// by two partition passes whose counters live in LDS.
//
//   pass A  (msm_part_hist / msm_part_scatter)   split every window's keys by the HIGH bits of the bucket
//           id into 2^HB "super-buckets": per-workgroup LDS histogram -> global exclusive scan over
//           (super-bucket, workgroup) -> second sweep recomputes its local ranks with LDS atomics and
//           writes (low bits | sign, point index) pairs to its reserved slots.
//   pass B  (msm_part_finish)   one workgroup per super-bucket (<= 1024 buckets, ~32 K entries, 256 KiB:
//           L2 resident): LDS histogram of the LOW bits, LDS scan -> bucket offsets, then places the
//           point indices.  Wave-level work only, no cross-workgroup communication.
//
// The super-bucket is taken from the LOW bits of the bucket id and pass B sorts by the HIGH bits: signed
// digits are uniform in their low bits even in the top window (whose high bits are mostly zero), so all
// super-buckets carry the same load.  The price is that buckets come out in a permuted order, "slots":
//     slot = (w << B) | (low HB bits << LB) | (high LB bits),   B = c-1 = HB + LB
// `sorted` (point index, sign in bit 31) is grouped by slot and `offsets` (nb + 1 exclusive prefix sums)
// is indexed by slot; msm_slot_to_bucket() gives the true bucket id where the weights matter (the store
// of the accumulated bucket).
// The order of the points inside a bucket is unspecified -- it only changes the Projective representative,
// never the group element.  (Role in the reference: none -- the CPU walks scalars and adds into
// `buckets[|digit|-1]` directly, ec/src/scalar_mul/variable_base/mod.rs:464-475.)
#pragma once
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "fp.cuh"

namespace arkhip {

static constexpr int PART_LO_BITS = 10;        // buckets per super-bucket = 2^10 by default (2^9 measured no better) ...
static constexpr int PART_LO_BITS_MAX = 12;    // ... up to 2^12 where the window is wide (msm_part_split)
static constexpr u32 PART_LDS_WORDS = (160 * 1024 - 64) / 4;  // dynamic LDS of the finish kernel, u32 words

// Split of the B = c-1 bucket bits into HB super-bucket bits (pass A) and LB bits finished in LDS (pass B).  Pass A
// keeps 2^HB counters per (window, 8192-key tile): its histogram array -- W * 2^HB * n/8192 counters, written and
// scanned in bin-major order -- is what grows with wide windows, so HB is kept as small as pass B allows: a
// super-bucket (n / 2^HB entries on average) must fit the finish kernel's LDS staging area (~32 K entries) and has at
// most 2^12 buckets.
static inline void msm_part_split(size_t n, int B, int* HB, int* LB) {
  int hb = 0;
  if (B > PART_LO_BITS) {
    hb = B - PART_LO_BITS;                       // 2^10 buckets per super-bucket ...
    if (hb > 9) {                                // ... unless that needs more than 2^9 counters per tile
      hb = B - PART_LO_BITS_MAX;
      if (hb < 9) hb = 9;
    }
    int lg = 0;
    while (((size_t)1 << lg) < n) lg++;
    if (hb < lg - 15) hb = lg - 15;              // super-bucket (n / 2^hb entries) within the LDS staging area
    if (hb > B) hb = B;
    if (const char* e = getenv("ARK_HIP_MSM_HB")) {   // tuning knob (tools/): super-bucket bits of pass A
      const int v = atoi(e);
      if (v >= 0 && v <= B && B - v <= PART_LO_BITS_MAX) hb = v;
    }
  } else {
    // few buckets per window (narrow scalars: msm_u8 plans ONE window of 2^8): still split, or a single workgroup of
    // pass B finishes the whole window (2^24 keys through one CU)
    int lg = 0;
    while (((size_t)1 << lg) < n) lg++;
    hb = lg - 15;
    if (hb < 0) hb = 0;
    if (hb > B) hb = B;
  }
  *HB = hb;
  *LB = B - hb;
}
// staging entries of the finish kernel for a given LB
static inline u32 msm_part_stage_cap(int LB) { return PART_LDS_WORDS - 1024u - (1u << LB) - 16u; }
static constexpr int PART_TILE = 8192;         // keys per workgroup in pass A (64 KiB of staged pairs) ...
static constexpr int PART_TILE_BIG = 16384;    // ... 128 KiB where pass A has >= 2^11 super-buckets (n >= 2^26): the per-tile
                                               // histogram array halves and the runs a tile writes per super-bucket double
                                               // (2^26, c = 22: 4 entries = 32 B per run with 8192 keys)
static constexpr size_t PART_SCATTER_LDS_MAX = 160 * 1024 - 4096 - 64;   // dynamic LDS the scatter kernel may ask for
static inline u32 msm_part_tile(int HB) {
  if (const char* e = getenv("ARK_HIP_MSM_TILE")) {   // tuning knob
    const int v = atoi(e);
    if (v == PART_TILE || (v == PART_TILE_BIG && ((size_t)8 << HB) + (size_t)PART_TILE_BIG * 8 <= PART_SCATTER_LDS_MAX)) return (u32)v;
  }
  // the big tile must fit the scatter kernel's LDS beside its 2 x 2^HB counters (HB = 11: 16 + 128 KiB; from HB = 12,
  // i.e. n >= 2^27, it does not -- 32 + 128 KiB -- and the 8192-key tile stays)
  const bool fits = ((size_t)8 << HB) + (size_t)PART_TILE_BIG * 8 <= PART_SCATTER_LDS_MAX;
  return (HB >= 11 && fits) ? (u32)PART_TILE_BIG : (u32)PART_TILE;
}
static constexpr u32 PART_KEY_NONE = 0xffffffffu;
static constexpr int PART_MLP = 8;             // global loads a lane keeps in flight in the sweeps of the sort kernels
static constexpr int PART_OUT_MLP = 4;         // ... in the scatter kernel's output loop (8-byte pairs; 64 VGPRs keep two workgroups per CU)
static constexpr int PART_SCATTER_KEYS = PART_TILE_BIG / 1024;   // keys per lane of the scatter kernel (1024 lanes)

// slot index (sort order) -> bucket index (window-major, weight order)
__host__ __device__ __forceinline__ u32 msm_slot_to_bucket(u32 slot, int HB, int LB) {
  const u32 B = (u32)(HB + LB);
  const u32 w = slot >> B;
  const u32 in = slot & ((1u << B) - 1u);
  const u32 low = in >> LB;                    // low HB bits of the bucket id
  const u32 high = in & ((1u << LB) - 1u);     // high LB bits
  return (w << B) | (high << HB) | low;
}

// Tile of workgroup blockIdx.x, XCD-aware.  Workgroups are dealt to the 8 XCDs round-robin and every XCD has its own L2:
// with tile = blockIdx.x the runs that neighbouring tiles write next to each other -- 128-byte runs of pass A at arbitrary
// 8-byte alignment, 4-byte histogram cells -- meet in eight different L2s and leave each of them as partial lines
// (WRITE_SIZE 2.7 GB for 1.7 GB of pairs, 546 MiB for a 54 MiB histogram: profiles/r3_pmc_fetch_write_size.txt).  Giving
// XCD k the k-th eighth of the tiles, in order, lets one L2 merge them.
static __device__ __forceinline__ u32 part_tile_of_block(u32 b, u32 ntiles) {
  const u32 per = ntiles >> 3;
  return b < (per << 3) ? (b & 7u) * per + (b >> 3) : b;
}

// A1: per-workgroup histogram over the high bits.  grid = (tiles, W); dynamic LDS = 4 << HB bytes.
static __global__ void __launch_bounds__(256) msm_part_hist_kernel(const u32* __restrict__ keys, u32 n, int HB, int LB,
                                                                   u32 ntiles, u32 tile, u32* __restrict__ tile_hist) {
  extern __shared__ u32 part_lds[];
  const u32 nbins = 1u << HB;
  const u32 hmask = nbins - 1u;  // super-bucket = LOW HB bits of the bucket id (see header)
  for (u32 b = threadIdx.x; b < nbins; b += blockDim.x) part_lds[b] = 0;
  __syncthreads();
  const u32 w = blockIdx.y;
  const size_t base = (size_t)w * n;
  const u32 tile_id = part_tile_of_block(blockIdx.x, ntiles);
  const u32 lo = tile_id * tile;
  const u32 hi = lo + tile < n ? lo + tile : n;
  // PART_MLP loads in flight per lane before the first LDS atomic waits on one: a load -> atomic loop leaves one 4-byte
  // load per lane outstanding, ~8 KiB per CU where HBM's latency x bandwidth asks for ~40 KiB (SQ counters of the three
  // sort kernels: 88-91 % of wave time parked in s_waitcnt, LDS bank conflicts 1 %: profiles/r3_pmc_sq_counters.txt)
  for (u32 c0 = lo; c0 < hi; c0 += blockDim.x * PART_MLP) {
    u32 k[PART_MLP];
#pragma unroll
    for (int b = 0; b < PART_MLP; b++) {
      const u32 i = c0 + (u32)b * blockDim.x + threadIdx.x;
      k[b] = i < hi ? keys[base + i] : PART_KEY_NONE;
    }
#pragma unroll
    for (int b = 0; b < PART_MLP; b++)
      if (k[b] != PART_KEY_NONE) atomicAdd(&part_lds[k[b] & hmask], 1u);
  }
  __syncthreads();
  // bin-major: [(w << HB | bin)][tile]
  for (u32 b = threadIdx.x; b < nbins; b += blockDim.x)
    tile_hist[((size_t)((w << HB) | b)) * ntiles + tile_id] = part_lds[b];
}

// A2: second sweep over the same tile.  Pairs are staged in LDS grouped by super-bucket and then written
// out in order, so that each (super-bucket, tile) run leaves as contiguous 8-byte elements.
// dynamic LDS: (2 << HB) counters + `tile` pairs.
static __global__ void __launch_bounds__(1024, 8) msm_part_scatter_kernel(const u32* __restrict__ keys, u32 n, int HB,
                                                                       int LB, u32 ntiles, u32 tile,
                                                                       const u32* __restrict__ tile_off,
                                                                       uint2* __restrict__ part,
                                                                       const u32* __restrict__ point_idx = nullptr) {
  // point_idx (nullable): key position i stands for base point_idx[i] (zero scalars compacted away, msm.cuh K0c): the pair
  // carries the base index from here on -- a coalesced read here instead of a random gather per sorted entry later
  extern __shared__ u32 part_lds[];
  const u32 nbins = 1u << HB;
  u32* cnt = part_lds;                       // per-bin count, then local cursor
  u32* lstart = part_lds + nbins;            // per-bin start inside the staging area
  uint2* stage = (uint2*)(part_lds + 2 * nbins);
  __shared__ u32 wsum[1024];
  const u32 w = blockIdx.y;
  const size_t base = (size_t)w * n;
  const u32 tile_id = part_tile_of_block(blockIdx.x, ntiles);
  const u32 lo = tile_id * tile;
  const u32 hi = lo + tile < n ? lo + tile : n;
  const u32 hmask = nbins - 1u;
  for (u32 b = threadIdx.x; b < nbins; b += blockDim.x) cnt[b] = 0;
  // the lane's keys stay in registers for both sweeps (tile <= PART_SCATTER_KEYS * blockDim.x); all loads are issued
  // before the first LDS atomic waits
  u32 key[PART_SCATTER_KEYS], pidx[PART_SCATTER_KEYS];
#pragma unroll
  for (int b = 0; b < PART_SCATTER_KEYS; b++) {
    const u32 i = lo + (u32)b * blockDim.x + threadIdx.x;
    key[b] = i < hi ? keys[base + i] : PART_KEY_NONE;
    pidx[b] = (point_idx && i < hi) ? point_idx[i] : i;
  }
  __syncthreads();
#pragma unroll
  for (int b = 0; b < PART_SCATTER_KEYS; b++)
    if (key[b] != PART_KEY_NONE) atomicAdd(&cnt[key[b] & hmask], 1u);
  __syncthreads();
  // exclusive scan of cnt[0..nbins) -> lstart ; cnt becomes the cursor
  const u32 per = (nbins + blockDim.x - 1) / blockDim.x;
  u32 s = 0;
  for (u32 q = 0; q < per; q++) {
    u32 b = threadIdx.x * per + q;
    if (b < nbins) s += cnt[b];
  }
  wsum[threadIdx.x] = s;
  __syncthreads();
  for (u32 o = 1; o < blockDim.x; o <<= 1) {
    u32 y = threadIdx.x >= o ? wsum[threadIdx.x - o] : 0;
    __syncthreads();
    wsum[threadIdx.x] += y;
    __syncthreads();
  }
  u32 run = wsum[threadIdx.x] - s;
  for (u32 q = 0; q < per; q++) {
    u32 b = threadIdx.x * per + q;
    if (b < nbins) {
      u32 v = cnt[b];
      lstart[b] = run;
      cnt[b] = run;
      run += v;
    }
  }
  __syncthreads();
  const u32 total = wsum[blockDim.x - 1];
#pragma unroll
  for (int b = 0; b < PART_SCATTER_KEYS; b++) {
    if (key[b] != PART_KEY_NONE) {
      const u32 pos = atomicAdd(&cnt[key[b] & hmask], 1u);
      stage[pos] = make_uint2(key[b], pidx[b]);   // full key kept: the bin is re-derived below
    }
  }
  __syncthreads();
  // output: the run starts (tile_off) of a batch of entries are fetched together, then the pairs are stored
  for (u32 j0 = 0; j0 < total; j0 += blockDim.x * PART_OUT_MLP) {
    uint2 e[PART_OUT_MLP];
    u32 dst[PART_OUT_MLP];
#pragma unroll
    for (int b = 0; b < PART_OUT_MLP; b++) {
      const u32 j = j0 + (u32)b * blockDim.x + threadIdx.x;
      if (j < total) {
        e[b] = stage[j];
        const u32 bin = e[b].x & hmask;
        dst[b] = tile_off[((size_t)((w << HB) | bin)) * ntiles + tile_id] + (j - lstart[bin]);
      }
    }
#pragma unroll
    for (int b = 0; b < PART_OUT_MLP; b++) {
      const u32 j = j0 + (u32)b * blockDim.x + threadIdx.x;
      if (j < total) {
        const u32 bkt = e[b].x & 0x7fffffffu;
        part[dst[b]] = make_uint2((bkt >> HB) | (e[b].x & 0x80000000u), e[b].y);   // remaining (high) bits, < 2^LB
      }
    }
  }
}

// LDS counter increment with the lanes that share the wave's LEADING key combined into one atomic.  Uniform digits
// put one or two lanes of a wave on the same counter and this costs a compare and a ballot; skewed scalars (a
// witness of mostly 0/1 values, the carry bucket of unsigned 16/32-bit scalars) put a whole wave on ONE counter,
// where 64 same-address LDS atomics serialise: the finish kernel of a 2^19-entry single-bucket super-bucket took
// 0.88 ms that way, 1.5 M serialised atomics (profiles/r4_skewed_sort_ab.txt).  Returns the slot the lane received;
// call with all lanes that hold a key (divergent callers are fine: the ballot only sees the active lanes).
__device__ __forceinline__ u32 lds_inc_leading(u32* cnt, u32 key) {
  const u32 k0 = __builtin_amdgcn_readfirstlane(key);
  const unsigned long long same = __ballot(key == k0);
  u32 pos;
  if (key == k0) {
    const u32 rank = __builtin_amdgcn_mbcnt_hi((u32)(same >> 32), __builtin_amdgcn_mbcnt_lo((u32)same, 0u));
    u32 base = 0;
    if (rank == 0) base = atomicAdd(&cnt[k0], (u32)__popcll(same));
    pos = __builtin_amdgcn_readfirstlane(base) + rank;   // the first active lane here is the rank-0 lane
  } else {
    pos = atomicAdd(&cnt[key], 1u);
  }
  return pos;
}

// B: one workgroup per super-bucket sb = (w << HB | low bits): bucket SLOTS sb << LB | high bits.  The sorted indices
// of the super-bucket are assembled in LDS and written out linearly; a super-bucket larger than the
// staging area (skewed scalars) falls back to direct placement.  Dynamic LDS: (2^LB + 1024 + stage_cap) words.
static __global__ void __launch_bounds__(1024) msm_part_finish_kernel(const uint2* __restrict__ part,
                                                                      const u32* __restrict__ tile_off, u32 ntiles,
                                                                      int LB, u32 nsuper, u32 stage_cap, u32 big_thresh,
                                                                      u32* __restrict__ big_count, u32* __restrict__ big,
                                                                      u32* __restrict__ offsets,
                                                                      u32* __restrict__ sorted) {
  extern __shared__ u32 part_lds[];
  const u32 nlow = 1u << LB;
  u32* cnt = part_lds;            // [nlow]
  u32* wsum = part_lds + nlow;    // [1024]
  u32* stage = wsum + 1024;       // [stage_cap]
  const u32 sb = blockIdx.x;
  const u32 start = tile_off[(size_t)sb * ntiles];
  const u32 end = tile_off[(size_t)(sb + 1) * ntiles];  // tile_off has nsuper * ntiles + 1 entries
  if (sb == nsuper - 1 && threadIdx.x == 0) offsets[(size_t)nsuper << LB] = end;
  if (big_thresh && end - start > big_thresh) {   // left to the sliced kernels below: list it, zero its counters and cursors
    if (threadIdx.x == 0) big[atomicAdd(big_count, 1u)] = sb;
    u32* const bigcnt = big + nsuper + ((size_t)sb << LB);
    u32* const cursor = bigcnt + ((size_t)nsuper << LB);
    for (u32 b = threadIdx.x; b < nlow; b += blockDim.x) {
      bigcnt[b] = 0;
      cursor[b] = 0;
    }
    return;
  }
  for (u32 b = threadIdx.x; b < nlow; b += blockDim.x) cnt[b] = 0;
  __syncthreads();
  const u32 lmask = nlow - 1u;
  for (u32 c0 = start; c0 < end; c0 += blockDim.x * PART_MLP) {
    u32 kx[PART_MLP];
#pragma unroll
    for (int b = 0; b < PART_MLP; b++) {
      const u32 j = c0 + (u32)b * blockDim.x + threadIdx.x;
      kx[b] = j < end ? part[j].x : PART_KEY_NONE;   // (a stored .x is < 2^31 + 2^LB: never the marker)
    }
#pragma unroll
    for (int b = 0; b < PART_MLP; b++)
      if (kx[b] != PART_KEY_NONE) lds_inc_leading(cnt, kx[b] & lmask);
  }
  __syncthreads();
  // exclusive scan of cnt[0..nlow): lane t owns `per` consecutive bins (nlow <= 4096, blockDim = 1024)
  const u32 per = nlow > blockDim.x ? nlow / blockDim.x : 1u;
  const u32 b0 = threadIdx.x * per;
  u32 v = 0;
  if (b0 < nlow)
    for (u32 k = 0; k < per; k++) v += cnt[b0 + k];
  wsum[threadIdx.x] = v;
  __syncthreads();
  for (u32 o = 1; o < blockDim.x; o <<= 1) {
    u32 y = threadIdx.x >= o ? wsum[threadIdx.x - o] : 0;
    __syncthreads();
    wsum[threadIdx.x] += y;
    __syncthreads();
  }
  if (b0 < nlow) {
    u32 run = wsum[threadIdx.x] - v;
    for (u32 k = 0; k < per; k++) {
      const u32 c = cnt[b0 + k];
      cnt[b0 + k] = run;  // local placement cursor
      offsets[((size_t)sb << LB) + b0 + k] = start + run;
      run += c;
    }
  }
  __syncthreads();
  const u32 total = end - start;
  if (total <= stage_cap) {
    for (u32 c0 = start; c0 < end; c0 += blockDim.x * PART_MLP) {
      uint2 e[PART_MLP];
#pragma unroll
      for (int b = 0; b < PART_MLP; b++) {
        const u32 j = c0 + (u32)b * blockDim.x + threadIdx.x;
        e[b] = j < end ? part[j] : make_uint2(PART_KEY_NONE, 0u);
      }
#pragma unroll
      for (int b = 0; b < PART_MLP; b++) {
        if (e[b].x != PART_KEY_NONE) {
          const u32 pos = lds_inc_leading(cnt, e[b].x & lmask);
          stage[pos] = e[b].y | (e[b].x & 0x80000000u);
        }
      }
    }
    __syncthreads();
    for (u32 j = threadIdx.x; j < total; j += blockDim.x) sorted[start + j] = stage[j];
  } else {
    for (u32 j = start + threadIdx.x; j < end; j += blockDim.x) {
      uint2 e = part[j];
      u32 pos = lds_inc_leading(cnt, e.x & lmask);
      sorted[start + pos] = e.y | (e.x & 0x80000000u);
    }
  }
}

// ---- pass B for a super-bucket too large for one workgroup ---------------------------------------------------------
// Skewed scalars concentrate a window's keys in a few buckets -- a witness whose values are mostly 1 sends n/2 keys to
// bucket 1 of window 0 -- and the super-bucket that holds such a bucket would stream all of them through ONE CU twice
// (2^23 entries: 5.0 ms at 2^24, the largest kernel of that MSM; profiles/r4_skewed_sort_ab.txt).  Super-buckets above
// PART_BIG entries are listed by the finish kernel (which also zeroes their counters), and each is finished by
// PART_BIG_SLICES workgroups over contiguous slices:
//   big_hist    slice histogram in LDS -> atomicAdd into the super-bucket's global counters
//   big_place   every slice scans the global counters (-> bucket offsets, written by slice 0), counts its slice again,
//               reserves its share of every bucket with one global atomic per non-empty bucket, and places its entries.
// The order inside a bucket depends on the order the slices reserve in -- unspecified already (see the header).
static constexpr u32 PART_BIG = 1u << 17;
static constexpr u32 PART_BIG_SLICES = 16;
static constexpr u32 PART_BIG_GRID_Y = 32;     // listed super-buckets are walked with this stride

// slice s of [start, end)
__device__ __forceinline__ void msm_part_big_slice(u32 start, u32 end, u32* lo, u32* hi) {
  const u32 total = end - start;
  const u32 per = (total + PART_BIG_SLICES - 1) / PART_BIG_SLICES;
  const u32 a = blockIdx.x * per, b = a + per;
  *lo = start + (a < total ? a : total);
  *hi = start + (b < total ? b : total);
}

static __global__ void __launch_bounds__(1024) msm_part_big_hist_kernel(const uint2* __restrict__ part,
                                                                        const u32* __restrict__ tile_off, u32 ntiles, int LB,
                                                                        const u32* __restrict__ big_count,
                                                                        const u32* __restrict__ big,
                                                                        u32* __restrict__ bigcnt) {
  extern __shared__ u32 part_lds[];   // [nlow]
  const u32 nbig = big_count[0];
  const u32 nlow = 1u << LB, lmask = nlow - 1u;
  for (u32 k = blockIdx.y; k < nbig; k += gridDim.y) {
    const u32 sb = big[k];
    u32 lo, hi;
    msm_part_big_slice(tile_off[(size_t)sb * ntiles], tile_off[(size_t)(sb + 1) * ntiles], &lo, &hi);
    for (u32 b = threadIdx.x; b < nlow; b += blockDim.x) part_lds[b] = 0;
    __syncthreads();
    for (u32 c0 = lo; c0 < hi; c0 += blockDim.x * PART_MLP) {
      u32 kx[PART_MLP];
#pragma unroll
      for (int b = 0; b < PART_MLP; b++) {
        const u32 j = c0 + (u32)b * blockDim.x + threadIdx.x;
        kx[b] = j < hi ? part[j].x : PART_KEY_NONE;
      }
#pragma unroll
      for (int b = 0; b < PART_MLP; b++)
        if (kx[b] != PART_KEY_NONE) lds_inc_leading(part_lds, kx[b] & lmask);
    }
    __syncthreads();
    for (u32 b = threadIdx.x; b < nlow; b += blockDim.x)
      if (part_lds[b]) atomicAdd(&bigcnt[((size_t)sb << LB) + b], part_lds[b]);
    __syncthreads();
  }
}

static __global__ void __launch_bounds__(1024) msm_part_big_place_kernel(const uint2* __restrict__ part,
                                                                         const u32* __restrict__ tile_off, u32 ntiles, int LB,
                                                                         const u32* __restrict__ big_count,
                                                                         const u32* __restrict__ big,
                                                                         const u32* __restrict__ bigcnt,
                                                                         u32* __restrict__ cursor, u32* __restrict__ offsets,
                                                                         u32* __restrict__ sorted) {
  extern __shared__ u32 part_lds[];
  const u32 nbig = big_count[0];
  const u32 nlow = 1u << LB, lmask = nlow - 1u;
  u32* cnt = part_lds;            // [nlow]  the super-bucket's counters -> bucket starts
  u32* wsum = part_lds + nlow;    // [1024]
  u32* mine = wsum + 1024;        // [nlow]  this slice's counters -> its placement cursors
  for (u32 k = blockIdx.y; k < nbig; k += gridDim.y) {
    const u32 sb = big[k];
    const u32 start = tile_off[(size_t)sb * ntiles];
    u32 lo, hi;
    msm_part_big_slice(start, tile_off[(size_t)(sb + 1) * ntiles], &lo, &hi);
    for (u32 b = threadIdx.x; b < nlow; b += blockDim.x) {
      cnt[b] = bigcnt[((size_t)sb << LB) + b];
      mine[b] = 0;
    }
    __syncthreads();
    // exclusive scan of cnt[0..nlow), as in msm_part_finish_kernel
    const u32 per = nlow > blockDim.x ? nlow / blockDim.x : 1u;
    const u32 b0 = threadIdx.x * per;
    u32 v = 0;
    if (b0 < nlow)
      for (u32 q = 0; q < per; q++) v += cnt[b0 + q];
    wsum[threadIdx.x] = v;
    __syncthreads();
    for (u32 o = 1; o < blockDim.x; o <<= 1) {
      u32 y = threadIdx.x >= o ? wsum[threadIdx.x - o] : 0;
      __syncthreads();
      wsum[threadIdx.x] += y;
      __syncthreads();
    }
    if (b0 < nlow) {
      u32 run = wsum[threadIdx.x] - v;
      for (u32 q = 0; q < per; q++) {
        const u32 c = cnt[b0 + q];
        cnt[b0 + q] = run;
        if (blockIdx.x == 0) offsets[((size_t)sb << LB) + b0 + q] = start + run;
        run += c;
      }
    }
    __syncthreads();
    for (u32 c0 = lo; c0 < hi; c0 += blockDim.x * PART_MLP) {
      u32 kx[PART_MLP];
#pragma unroll
      for (int b = 0; b < PART_MLP; b++) {
        const u32 j = c0 + (u32)b * blockDim.x + threadIdx.x;
        kx[b] = j < hi ? part[j].x : PART_KEY_NONE;
      }
#pragma unroll
      for (int b = 0; b < PART_MLP; b++)
        if (kx[b] != PART_KEY_NONE) lds_inc_leading(mine, kx[b] & lmask);
    }
    __syncthreads();
    for (u32 b = threadIdx.x; b < nlow; b += blockDim.x) {
      const u32 m = mine[b];
      if (m) mine[b] = cnt[b] + atomicAdd(&cursor[((size_t)sb << LB) + b], m);
    }
    __syncthreads();
    for (u32 c0 = lo; c0 < hi; c0 += blockDim.x * PART_MLP) {
      uint2 e[PART_MLP];
#pragma unroll
      for (int b = 0; b < PART_MLP; b++) {
        const u32 j = c0 + (u32)b * blockDim.x + threadIdx.x;
        e[b] = j < hi ? part[j] : make_uint2(PART_KEY_NONE, 0u);
      }
#pragma unroll
      for (int b = 0; b < PART_MLP; b++) {
        if (e[b].x != PART_KEY_NONE) {
          const u32 pos = lds_inc_leading(mine, e[b].x & lmask);
          sorted[start + pos] = e[b].y | (e[b].x & 0x80000000u);
        }
      }
    }
    __syncthreads();
  }
}


}  // namespace arkhip
