// Fixed-base batch multiplication on one MI355X: out[i] = v[i] * g for ONE group element g.
//
// Replaces ScalarMul::batch_mul / BatchMulPreprocessing (ec/src/scalar_mul/mod.rs:104-251): the reference builds the
// table  T[outer][inner] = inner * 2^(window*outer) * g  (:173-214), computes every product as the sum of one table
// entry per window (windowed_mul, :235-251) and converts the results to affine in one batch (:230-233).
// Same table, dimensioned for the GPU.  The reference sizes its window as ln(num_scalars) (3 below 32 scalars, :222-228)
// because a CPU pays for every table entry; here the entries are computed one per lane, so up to ~10^5 of them cost one
// round of waves whatever the window, and the batch itself costs one mixed addition per window: batchmul_window() keeps
// 12 bits (22 x 4096 affine entries: 8.6 MB for BLS12-381 G1, resident in L2 / Infinity Cache; built in 1.6 ms) for every
// batch size a smaller window would be built for, and moves to 16 bits (16 x 65536 entries, 8 ms to build) from 2^24
// scalars up, where six fewer additions per scalar repay the larger table within ONE batch (new + batch_mul, the way
// ScalarMul::batch_mul runs it: 2^22 18.7 vs 22.6 ms, 2^24 68.2 vs 63.5 ms; profiles/r3_batch_mul.txt).  The window only changes the cost, never the result; the table covers all
// 256 scalar bits, so any BigInt<4> is multiplied exactly.  One lane per scalar: one mixed addition per window from table
// gathers (on the carry-free 28-bit limbs of fp28.cuh for the Fp384 G1 curves), then a lane-batched inversion for the
// affine results (the data-parallel counterpart of the reference's batch inversion).  The 2^(window*o) g chain -- 256
// dependent doublings -- runs on the host in the same templated formulas (a serial chain is ~10x faster there).
#pragma once
#include "msm.cuh"

namespace arkhip {

static constexpr int BATCHMUL_WINDOW_SMALL = 12;   // batches below 2^24 scalars
static constexpr int BATCHMUL_WINDOW_LARGE = 16;
static constexpr int BATCHMUL_MAX_OUTER = 64;      // window >= 4
static inline int batchmul_outer(int window) { return (256 + window - 1) / window; }
static inline int batchmul_window(size_t num_scalars) {
  if (const char* e = getenv("ARK_HIP_BATCHMUL_WINDOW")) {  // measurement knob
    const int w = atoi(e);
    if (w >= 4 && w <= 16) return w;
  }
  return num_scalars >= ((size_t)1 << 24) ? BATCHMUL_WINDOW_LARGE : BATCHMUL_WINDOW_SMALL;
}

// tmp[o << window | i] = i * g_outer[o] in XYZZ form; one lane per entry (double-and-add over the window's bits);
// batchmul_build normalises the whole table afterwards (lane-batched inversion)
template <class C>
__global__ void __launch_bounds__(128) batchmul_table_kernel(const char* __restrict__ g_outer, char* __restrict__ tmp, int window,
                                                             u32 entries) {
  typedef typename C::F F;
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= entries) return;
  const u32 o = t >> window, i = t & ((1u << window) - 1u);
  XYZZ<F> g = XYZZ<F>::load(g_outer + (size_t)o * XYZZ<F>::BYTES);
  XYZZ<F> acc = XYZZ<F>::zero();
  for (int k = window - 1; k >= 0; k--) {
    acc = xyzz_dbl<F>(acc);
    if ((i >> k) & 1u) xyzz_add<F>(acc, g);
  }
  acc.store(tmp + (size_t)t * XYZZ<F>::BYTES);
}

// digit o of a 256-bit scalar (little-endian 32-bit limbs), window <= 16 bits
template <class S>
ARK_DEV u32 batchmul_digit(const S& s, int o, int window) {
  const int bit = o * window;
  const int limb = bit >> 5, sh = bit & 31;
  u32 d = s.l[limb] >> sh;
  if (sh + window > 32 && limb + 1 < S::N) d |= s.l[limb + 1] << (32 - sh);
  return d & ((1u << window) - 1u);
}

// tmp[i] = scalars[i] * g in XYZZ form: windowed_mul (mod.rs:235-251); batchmul_run normalises the batch afterwards
// (normalize_batch, :226 -- lane-batched inversion, ec.cuh)
template <class C>
__global__ void __launch_bounds__(128) batchmul_kernel(const char* __restrict__ table, const u32* __restrict__ scalars, size_t n,
                                                       int mont, int window, int outer, char* __restrict__ out) {
  typedef typename C::F F;
  typedef Fp<typename C::S> S;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  S s = S::load(scalars + i * S::N);
  if (mont) s = S::from_mont(s);  // windowed_mul's into_bigint (:238)
  XYZZ<F> acc = XYZZ<F>::zero();
  for (int o = 0; o < outer; o++) {
    const u32 d = batchmul_digit(s, o, window);
    if (d != 0) {
      Affine<F> p = Affine<F>::load(table + (((size_t)o << window) | d) * Affine<F>::BYTES);
      xyzz_madd<F>(acc, p.x, p.y);
    }
  }
  acc.store(out + i * XYZZ<F>::BYTES);
}

// the same on carry-free 28-bit limbs (fp28.cuh / ec28.cuh), for the curves whose accumulate kernel uses them
template <class C>
__global__ void __launch_bounds__(256, ARK_LAZY_MIN_WAVES) batchmul_lazy_kernel(const char* __restrict__ table,
                                                                                const u32* __restrict__ scalars, size_t n, int mont,
                                                                                int window, int outer, char* __restrict__ out) {
  typedef typename C::F F;
  typedef typename F::P P;
  typedef FpL<P> FL;
  typedef Fp<typename C::S> S;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  S s = S::load(scalars + i * S::N);
  if (mont) s = S::from_mont(s);
  XYZZL<P> acc;
  acc.inf = true;
  acc.x = acc.y = acc.zz = acc.zzz = FL::zero();
  for (int o = 0; o < outer; o++) {
    const u32 d = batchmul_digit(s, o, window);
    if (d == 0) continue;
    const char* entry = table + (((size_t)o << window) | d) * Affine<F>::BYTES;
    Affine<F> p = Affine<F>::load(entry);
    if (p.is_zero()) continue;  // multiples of the identity base
    FL lx, ly;
    lazy_from_affine<P>(p.x, p.y, lx, ly);
    if (xyzz_madd_lazy<P>(acc, lx, ly)) {  // entry == accumulated point (a scalar >= r can wrap onto it): doubling
      XYZZL<P> dbl;
      xyzz_mdbl_lazy<P>(dbl, entry, false);
      acc = dbl;
    }
  }
  lazy_to_bucket<P>(acc).store(out + i * XYZZ<F>::BYTES);
}

// table: batchmul_outer(window) << window affine entries; d_scratch: as many XYZZ points (the unnormalised table, its
// first `outer` cells double as the upload area of the 2^(window o) g chain); h_base_affine: HOST pointer, x | y limbs.
template <class C>
int batchmul_build(const void* h_base_affine, int window, void* d_scratch, void* d_table, hipStream_t stream) {
  typedef typename C::F F;
  const int outer = batchmul_outer(window);
  if (window < 4 || window > 16 || outer > BATCHMUL_MAX_OUTER) return -2;
  const u32 entries = (u32)outer << window;
  // g_outer[o] = 2^(window*o) * g: 256 dependent doublings, on the host (same formulas)
  std::vector<char> chain((size_t)outer * XYZZ<F>::BYTES);
  XYZZ<F> g = XYZZ<F>::from_affine(Affine<F>::load((const char*)h_base_affine));
  for (int o = 0; o < outer; o++) {
    g.store(chain.data() + (size_t)o * XYZZ<F>::BYTES);
    for (int k = 0; k < window; k++) g = xyzz_dbl<F>(g);
  }
  // the chain sits behind the unnormalised table in the scratch area
  char* d_chain = (char*)d_scratch + (size_t)entries * XYZZ<F>::BYTES;
  ARK_HIP_TRY(hipMemcpyAsync(d_chain, chain.data(), chain.size(), hipMemcpyHostToDevice, stream));
  ARK_HIP_TRY(hipStreamSynchronize(stream));  // `chain` is pageable and dies with this frame
  hipLaunchKernelGGL((batchmul_table_kernel<C>), dim3((entries + 127) / 128), dim3(128), 0, stream, (const char*)d_chain,
                     (char*)d_scratch, window, entries);
  xyzz_to_affine_batched_launch<F>(d_scratch, d_table, entries, stream);
  ARK_HIP_TRY(hipGetLastError());
  return 0;
}
// scratch bytes batchmul_build needs for this curve / window
template <class C>
size_t batchmul_build_scratch(int window) {
  const int outer = batchmul_outer(window);
  return (((size_t)outer << window) + (size_t)outer) * XYZZ<typename C::F>::BYTES;
}
template <class C>
int batchmul_run(const void* d_table, int window, const void* d_scalars, size_t n, int mont, void* d_tmp, void* d_out,
                 hipStream_t stream) {  // d_tmp: n XYZZ points of scratch
  if (n == 0) return 0;
  const int outer = batchmul_outer(window);
  bool lazy = false;
  constexpr bool LAZY_BM = C::LAZY_A && C::FA::LANES == 1;   // G2 keeps one whole Fp2 element per lane here (saturated limbs)
  if constexpr (LAZY_BM) lazy = msm_lazy_enabled();
  if constexpr (LAZY_BM) {
    if (lazy)
      hipLaunchKernelGGL((batchmul_lazy_kernel<C>), dim3((u32)((n + 255) / 256)), dim3(256), 0, stream, (const char*)d_table,
                         (const u32*)d_scalars, n, mont, window, outer, (char*)d_tmp);
  }
  if (!lazy)
    hipLaunchKernelGGL((batchmul_kernel<C>), dim3((u32)((n + 127) / 128)), dim3(128), 0, stream, (const char*)d_table,
                       (const u32*)d_scalars, n, mont, window, outer, (char*)d_tmp);
  xyzz_to_affine_batched_launch<typename C::F>(d_tmp, d_out, n, stream);
  ARK_HIP_TRY(hipGetLastError());
  return 0;
}

}  // namespace arkhip
