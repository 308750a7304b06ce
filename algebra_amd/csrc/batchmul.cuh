// Fixed-base batch multiplication on one MI355X: out[i] = v[i] * g for ONE group element g.
//
// Replaces ScalarMul::batch_mul / BatchMulPreprocessing (ec/src/scalar_mul/mod.rs:104-251): the reference builds the
// table  T[outer][inner] = inner * 2^(window*outer) * g  (:173-214), computes every product as the sum of one table
// entry per window (windowed_mul, :235-251) and converts the results to affine in one batch (:230-233).
// Same table, re-dimensioned for the GPU: a fixed 12-bit window (22 x 4096 affine entries: 8.6 MB for BLS12-381 G1,
// resident in L2 / Infinity Cache) instead of the reference's ln(n)-sized one -- the window only changes the cost, never
// the result -- and covering all 256 scalar bits, so any BigInt<4> is multiplied exactly.  One lane per scalar: ~22
// mixed additions from table gathers, then the lane's own inversion for the affine result (the data-parallel
// counterpart of the reference's batch inversion).
#pragma once
#include "msm.cuh"

namespace arkhip {

static constexpr int BATCHMUL_WINDOW = 12;
static constexpr int BATCHMUL_OUTER = (256 + BATCHMUL_WINDOW - 1) / BATCHMUL_WINDOW;  // 22

// g_outer[o] = 2^(window*o) * g as XYZZ: one lane walks the doubling chain (256 doublings, one-time)
template <class C>
__global__ void __launch_bounds__(64) batchmul_outer_kernel(const char* __restrict__ base_affine, char* __restrict__ g_outer) {
  typedef typename C::F F;
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  Affine<F> b = Affine<F>::load(base_affine);
  XYZZ<F> g = XYZZ<F>::from_affine(b);
  for (int o = 0; o < BATCHMUL_OUTER; o++) {
    g.store(g_outer + (size_t)o * XYZZ<F>::BYTES);
    for (int k = 0; k < BATCHMUL_WINDOW; k++) g = xyzz_dbl<F>(g);
  }
}

// T[o][i] = i * g_outer[o], affine; one lane per entry (double-and-add over the window's bits, then one inversion)
template <class C>
__global__ void __launch_bounds__(128) batchmul_table_kernel(const char* __restrict__ g_outer, char* __restrict__ table) {
  typedef typename C::F F;
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (u32)BATCHMUL_OUTER << BATCHMUL_WINDOW) return;
  const u32 o = t >> BATCHMUL_WINDOW, i = t & ((1u << BATCHMUL_WINDOW) - 1u);
  XYZZ<F> g = XYZZ<F>::load(g_outer + (size_t)o * XYZZ<F>::BYTES);
  XYZZ<F> acc = XYZZ<F>::zero();
  for (int k = BATCHMUL_WINDOW - 1; k >= 0; k--) {
    acc = xyzz_dbl<F>(acc);
    if ((i >> k) & 1u) xyzz_add<F>(acc, g);
  }
  F x = F::zero(), y = F::zero();
  if (!acc.is_zero()) {
    F zzzi = F::inverse(acc.zzz);
    F zzi = F::sqr(F::mul(acc.zz, zzzi));  // ZZ^-1 = (ZZ * ZZZ^-1)^2 since ZZ^3 = ZZZ^2
    x = F::mul(acc.x, zzi);
    y = F::mul(acc.y, zzzi);
  }
  x.store(table + (size_t)t * Affine<F>::BYTES);
  y.store(table + (size_t)t * Affine<F>::BYTES + F::BYTES);
}

// tmp[i] = scalars[i] * g in XYZZ form: windowed_mul (mod.rs:235-251); batchmul_run normalises the batch afterwards
// (normalize_batch, :226 -- lane-batched inversion, ec.cuh)
template <class C>
__global__ void __launch_bounds__(128) batchmul_kernel(const char* __restrict__ table, const u32* __restrict__ scalars, size_t n,
                                                       int mont, char* __restrict__ out) {
  typedef typename C::F F;
  typedef Fp<typename C::S> S;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  S s = S::load(scalars + i * S::N);
  if (mont) s = S::from_mont(s);  // windowed_mul's into_bigint (:238)
  XYZZ<F> acc = XYZZ<F>::zero();
  for (int o = 0; o < BATCHMUL_OUTER; o++) {
    const int bit = o * BATCHMUL_WINDOW;
    const int limb = bit >> 5, sh = bit & 31;
    u32 d = s.l[limb] >> sh;
    if (sh + BATCHMUL_WINDOW > 32 && limb + 1 < S::N) d |= s.l[limb + 1] << (32 - sh);
    d &= (1u << BATCHMUL_WINDOW) - 1u;
    if (d != 0) {
      Affine<F> p = Affine<F>::load(table + ((size_t)o << BATCHMUL_WINDOW | d) * Affine<F>::BYTES);
      xyzz_madd<F>(acc, p.x, p.y);
    }
  }
  acc.store(out + i * XYZZ<F>::BYTES);
}

// table: BATCHMUL_OUTER << BATCHMUL_WINDOW affine entries; scratch: BATCHMUL_OUTER XYZZ points
template <class C>
int batchmul_build(const void* d_base_affine, void* d_scratch, void* d_table, hipStream_t stream) {
  hipLaunchKernelGGL((batchmul_outer_kernel<C>), dim3(1), dim3(64), 0, stream, (const char*)d_base_affine, (char*)d_scratch);
  const u32 entries = (u32)BATCHMUL_OUTER << BATCHMUL_WINDOW;
  hipLaunchKernelGGL((batchmul_table_kernel<C>), dim3((entries + 127) / 128), dim3(128), 0, stream, (const char*)d_scratch,
                     (char*)d_table);
  ARK_HIP_TRY(hipGetLastError());
  return 0;
}
template <class C>
int batchmul_run(const void* d_table, const void* d_scalars, size_t n, int mont, void* d_tmp, void* d_out,
                 hipStream_t stream) {  // d_tmp: n XYZZ points of scratch
  if (n == 0) return 0;
  hipLaunchKernelGGL((batchmul_kernel<C>), dim3((u32)((n + 127) / 128)), dim3(128), 0, stream, (const char*)d_table,
                     (const u32*)d_scalars, n, mont, (char*)d_tmp);
  xyzz_to_affine_batched_launch<typename C::F>(d_tmp, d_out, n, stream);
  ARK_HIP_TRY(hipGetLastError());
  return 0;
}

}  // namespace arkhip
