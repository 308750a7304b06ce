// Radix-2 FFT / IFFT whose COEFFICIENTS are group elements (round 5).
//
// Replaces, for T = Projective<P> of a served curve, the generic half of the reference's transform:
//   poly/src/domain/mod.rs:332-362      fft_in_place<T: DomainCoeff<F>> -- T only needs  T + T,  T - T  and  T *= F
//   poly/src/domain/radix2/fft.rs:74-119 in_order_fft / in_order_ifft_in_place (distribute_powers for cosets, the IFFT's
//                                        x[i] *= size_inv * offset_inv^i), :190-210 butterflies, :373-380 derange
//   poly/src/test.rs:57                 the reference's own use with G1Projective (a commitment key moved between bases)
// T *= F on a Projective is a scalar multiplication (group.rs Mul<ScalarField>: mul_bigint of the canonical scalar), so a
// butterfly costs two additions and one 255-bit scalar multiplication -- ~330 point operations: a different kernel family from the
// field transform (fft.cuh: one product per butterfly) and from the MSM (no buckets: every butterfly has its own scalar).
//
// GPU organisation: log2 n decimation-in-frequency stages, one kernel launch per stage, one lane per butterfly on
// XYZZ coordinates (bucket.rs formulas: ec.cuh) in a scratch array; the first kernel converts the caller's Jacobian points
// (and applies the coset's h^i), the last one undoes the bit reversal (derange), applies size_inv * h^-i and converts back.
// Latency-bound below ~2^17 butterflies per stage (a lane walks its 260 doublings alone); against the reference's CPU loop
// -- n/2 log2 n scalar multiplications of ~80 us each -- two to three orders of magnitude at 2^10 .. 2^16.
// Results are group elements: Projective representatives differ from the reference's, into_affine() agrees.
#pragma once
#include "msm.cuh"

namespace arkhip {

// [k] p for a Montgomery-form scalar k of the curve's scalar field: signed 4-bit windows, MSB first.
// k + 0x88...8 has the nibbles n_i with k = sum (n_i - 8) 16^i + carry 16^64, digits in [-8, 7]: 64 windows of four doublings
// and (15 times in 16) one addition of +-[1..8] p from the lane's table -- 260 doublings + ~63 additions against the 255 + ~127
// of the bit-by-bit ladder.  The table ([e][lane], e = 1..8, `nlanes` lanes) lives in device memory: dynamic indexing into
// registers would go through scratch anyway, and a 192-byte load per window is nothing beside its ~70 field products.
template <class C>
ARK_DEV XYZZ<typename C::F> gfft_scalar_mul(const XYZZ<typename C::F>& p, const u32* k_mont, char* tab, size_t lane, size_t nlanes) {
  typedef typename C::F F;
  typedef XYZZ<F> Pt;
  typedef Fp<typename C::S> S;
  static_assert(S::N == 8, "256-bit scalar fields");
  if (p.is_zero()) return Pt::zero();
  const S kc = S::from_mont(S::load(k_mont));   // canonical integer of the field element (group.rs: into_bigint)
  auto slot = [&](int e) { return tab + ((size_t)(e - 1) * nlanes + lane) * Pt::BYTES; };
  p.store(slot(1));
  for (int e = 2; e <= 8; e++) {                // 2p = dbl(p), 3p = 2p + p, 4p = dbl(2p), ... (e is uniform over the wave)
    Pt t;
    if (e & 1) { t = Pt::load(slot(e - 1)); xyzz_add<F>(t, p); }
    else t = xyzz_dbl<F>(Pt::load(slot(e >> 1)));
    t.store(slot(e));
  }
  u32 kw[8];
  u32 carry = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const u64 v = (u64)kc.l[i] + 0x88888888u + carry;
    kw[i] = (u32)v;
    carry = (u32)(v >> 32);
  }
  Pt acc = Pt::zero();
  if (carry) acc = p;
  for (int w = 63; w >= 0; w--) {
    acc = xyzz_dbl<F>(xyzz_dbl<F>(xyzz_dbl<F>(xyzz_dbl<F>(acc))));
    u32 word = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) word = (w >> 3) == i ? kw[i] : word;   // (no dynamic indexing into registers)
    const int d = (int)((word >> ((w & 7) * 4)) & 15u) - 8;
    if (d != 0) {
      Pt q = Pt::load(slot(d < 0 ? -d : d));
      if (d < 0) q = Pt::neg(q);
      xyzz_add<F>(acc, q);
    }
  }
  return acc;
}

// lanes of one launch that multiply by a scalar: the window tables of a launch are GFFT_SLAB * 8 points at most
// (ARK_HIP_GFFT_SLAB_LOG = 6..18 lowers it: the tests run a small transform in several slabs)
static constexpr size_t GFFT_SLAB = (size_t)1 << 18;
static inline size_t gfft_slab() {
  const char* e = getenv("ARK_HIP_GFFT_SLAB_LOG");
  const int l = e ? atoi(e) : 18;
  return l >= 6 && l <= 18 ? (size_t)1 << l : GFFT_SLAB;
}

// Jacobian (x, y, z) -> XYZZ (x, y, z^2, z^3), identity (z = 0) -> zero; optionally times scal[i]
template <class C>
__global__ void __launch_bounds__(64) gfft_load_kernel(const char* __restrict__ jac, char* __restrict__ work, size_t first, size_t n,
                                                       const u32* __restrict__ scal, char* __restrict__ tab, size_t nlanes) {
  typedef typename C::F F;
  typedef XYZZ<F> Pt;
  const size_t lane = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t i = first + lane;
  if (lane >= nlanes || i >= n) return;
  const char* p = jac + i * 3 * F::FULL_BYTES;
  const F z = F::load(p + 2 * F::FULL_BYTES);
  Pt q = Pt::zero();
  if (!z.is_zero()) {
    q.x = F::load(p);
    q.y = F::load(p + F::FULL_BYTES);
    q.zz = F::sqr(z);
    q.zzz = F::mul(q.zz, z);
  }
  if (scal) q = gfft_scalar_mul<C>(q, scal + i * 8, tab, lane, nlanes);
  q.store(work + i * Pt::BYTES);
}

// one DIF stage: (lo, hi) <- (lo + hi, (lo - hi) * w^(j << s)),  gap = n >> (s + 1), j = b mod gap      fft.rs:190-198, 262-293
template <class C>
__global__ void __launch_bounds__(64) gfft_stage_kernel(char* __restrict__ work, size_t first, size_t n, size_t gap, int s,
                                                        const u32* __restrict__ roots, char* __restrict__ tab, size_t nlanes) {
  typedef typename C::F F;
  typedef XYZZ<F> Pt;
  const size_t lane = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t b = first + lane;
  if (lane >= nlanes || b >= n / 2) return;
  const size_t j = b & (gap - 1);
  const size_t i0 = ((b - j) << 1) | j, i1 = i0 + gap;
  Pt lo = Pt::load(work + i0 * Pt::BYTES);
  const Pt hi = Pt::load(work + i1 * Pt::BYTES);
  Pt dif = lo;
  xyzz_add<F>(dif, Pt::neg(hi));
  xyzz_add<F>(lo, hi);
  lo.store(work + i0 * Pt::BYTES);
  if (j != 0) dif = gfft_scalar_mul<C>(dif, roots + (j << s) * 8, tab, lane, nlanes);
  dif.store(work + i1 * Pt::BYTES);
}

// position i holds X[bitrev_k(i)] (derange, fft.rs:373-380): out[j] = [scal[j]] work[bitrev(j)], XYZZ -> Jacobian
template <class C>
__global__ void __launch_bounds__(64) gfft_store_kernel(const char* __restrict__ work, char* __restrict__ jac, size_t first, size_t n,
                                                        int k, const u32* __restrict__ scal, char* __restrict__ tab, size_t nlanes) {
  typedef typename C::F F;
  typedef XYZZ<F> Pt;
  const size_t lane = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t j = first + lane;
  if (lane >= nlanes || j >= n) return;
  const size_t i = k == 0 ? 0 : (size_t)(__brevll((unsigned long long)j) >> (64 - k));
  Pt q = Pt::load(work + i * Pt::BYTES);
  if (scal) q = gfft_scalar_mul<C>(q, scal + j * 8, tab, lane, nlanes);
  xyzz_to_jac<F>(q).store(jac + j * 3 * F::FULL_BYTES);
}

// d_jac: n = 2^k Projective points of curve C in device memory, transformed in place.
// d_roots: w^j, j < n/2, Montgomery residues of the curve's scalar field (the field transform's own table: its first n/2
// entries) for w = group_gen (forward) or group_gen_inv (inverse); d_pre / d_post: n scalars each or nullptr
// (h^i before the stages; size_inv * h^-i after them); d_work: n * XYZZ::BYTES of scratch.
template <class C>
int gfft_run(void* d_jac, int k, const u32* d_roots, const u32* d_pre, const u32* d_post, void* d_work, hipStream_t stream) {
  typedef XYZZ<typename C::F> Pt;
  const size_t n = (size_t)1 << k;
  const size_t cap = gfft_slab();
  const size_t slab = n < cap ? n : cap;                 // lanes per launch (a launch's window tables: slab * 8 points)
  char* work = (char*)d_work;
  char* tab = work + n * Pt::BYTES;
  const unsigned nb = (unsigned)((slab + 63) / 64);
  for (size_t f = 0; f < n; f += slab)
    hipLaunchKernelGGL((gfft_load_kernel<C>), dim3(nb), dim3(64), 0, stream, (const char*)d_jac, work, f, n, d_pre, tab, slab);
  for (int s = 0; s < k; s++) {
    const size_t gap = n >> (s + 1);
    for (size_t f = 0; f < n / 2; f += slab)
      hipLaunchKernelGGL((gfft_stage_kernel<C>), dim3(nb), dim3(64), 0, stream, work, f, n, gap, s, d_roots, tab, slab);
  }
  for (size_t f = 0; f < n; f += slab)
    hipLaunchKernelGGL((gfft_store_kernel<C>), dim3(nb), dim3(64), 0, stream, (const char*)work, (char*)d_jac, f, n, k, d_post, tab,
                       slab);
  return hipGetLastError() == hipSuccess ? 0 : -1000;
}
template <class C>
size_t gfft_work_bytes(int k) {
  const size_t n = (size_t)1 << k;
  return (n + 8 * (n < GFFT_SLAB ? n : GFFT_SLAB)) * XYZZ<typename C::F>::BYTES;   // the points + one launch's window tables
}

}  // namespace arkhip
