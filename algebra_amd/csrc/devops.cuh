// Product kernels over device-resident vectors and point arrays (rounds 2-5 kept them in testops.cuh next to the test hooks):
//   field_op_kernel       pointwise add / sub / mul / sqr / neg / dbl / into_bigint / from_bigint -- what ark_hip_fr_add / sub /
//                         mul / neg_device run (Evaluations +=, -=, *=: poly/src/evaluations/univariate/mod.rs:104-180);
//   fr_scale_kernel       vector times a constant;  fr_div_kernel  pointwise division with lane-batched inversions;
//   sw_add_affine_kernel  P[i] + D for an array of affine points (synthetic base sets);
//   sw_normalize_batch_kernel  CurveGroup::normalize_batch (group.rs:302-319).
#pragma once
#include "curves.cuh"
#include "fp28.cuh"
#include "lazyk.cuh"

namespace arkhip {

// ops 20 / 21 / 22 (reached by the test hooks only): x * y, x^2, x y + x y through the carry-free 28-bit form (fp28.cuh) -- the DEVICE forms of its
// product, square and sum of two products (asm column chains), which the host check of tests/lazy_host_check.hip cannot
// reach: operands enter by the shifted repack, results leave as the accumulate kernels' buckets do.
template <class F> struct LazyFieldOps {
  static constexpr bool OK = false;
  ARK_DEV static F run(int, const F& x, const F&) { return x; }
};
template <class P> struct LazyFieldOps<Fp<P>> {
  static constexpr bool OK = true;
  ARK_DEV static Fp<P> run(int op, const Fp<P>& x, const Fp<P>& y) {
    typedef FpL<P> L;
    const L a = L::unpack32_shl(x.l), b = L::unpack32_shl(y.l);
    L r;
    if (op == 20) r = L::mul(a, b);
    else if (op == 21) r = L::sqr(a);
    else r = L::sop2(a, b, b, a);
    return r.template shr_mod<L::SH>().to_canonical_bits();
  }
};

template <class F, bool IS_PRIME>
__global__ void __launch_bounds__(256) field_op_kernel(int op, const char* a, const char* b, char* r, size_t n) {  // r may alias a or b
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  F x = F::load(a + i * F::BYTES);
  F y = b ? F::load(b + i * F::BYTES) : x;
  F z = x;
  switch (op) {
    case 0: z = F::add(x, y); break;
    case 1: z = F::sub(x, y); break;
    case 2: z = F::mul(x, y); break;
    case 3: z = F::sqr(x); break;
    case 4: z = F::neg(x); break;
    case 5: z = F::dbl(x); break;
    default:
      if constexpr (LazyFieldOps<F>::OK) {
        if (op >= 20 && op <= 22) z = LazyFieldOps<F>::run(op, x, y);
      }
      if constexpr (IS_PRIME) {
        if (op == 7) z = F::from_mont(x);
        if (op == 8) z = F::to_mont(x);
      }
      break;
  }
  z.store(r + i * F::BYTES);
}

template <class F, bool IS_PRIME>
int field_op_launch(int op, const void* a, const void* b, void* r, size_t n, hipStream_t s) {
  if (n == 0) return 0;
  hipLaunchKernelGGL((field_op_kernel<F, IS_PRIME>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, op,
                     (const char*)a, (const char*)b, (char*)r, n);
  return hipGetLastError() == hipSuccess ? 0 : -1000;
}

// r[i] = a[i] * k for a constant k passed by value (DensePolynomial * F, Evaluations scaled by a field element): the
// constant travels in the kernel arguments, so a caller's stack value needs no device copy and no lifetime rule
struct FrConst { u32 l[8]; };
template <class F>
__global__ void __launch_bounds__(256) fr_scale_kernel(const char* a, FrConst k, char* r, size_t n) {  // r may alias a
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  F kk;
#pragma unroll
  for (int j = 0; j < F::N; j++) kk.l[j] = k.l[j];
  F::mul(F::load(a + i * F::BYTES), kk).store(r + i * F::BYTES);
}
template <class F>
int fr_scale_launch(const void* a, const uint64_t* k4, void* r, size_t n, hipStream_t s) {
  static_assert(F::N == 8, "the scalar fields served are 256-bit");
  if (n == 0) return 0;
  FrConst k;
  for (int j = 0; j < 4; j++) { k.l[2 * j] = (u32)k4[j]; k.l[2 * j + 1] = (u32)(k4[j] >> 32); }
  hipLaunchKernelGGL((fr_scale_kernel<F>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const char*)a, k, (char*)r, n);
  return hipGetLastError() == hipSuccess ? 0 : -1000;
}

// r[i] = num[i] / den[i] (num == nullptr: 1 / den[i]); a zero denominator gives zero, as ark_ff::batch_inversion leaves zeros
// in place (ff/src/fields/mod.rs:358-385) and Evaluations::div_assign then multiplies by them (evaluations/univariate/
// mod.rs:153-163).  Montgomery's trick inside every lane over its own 8 values (lane_batch_inverse, ec.cuh): 3 products per
// value + 1/8 of a Fermat inversion.  r may alias num or den: a lane reads an index before it writes it and owns it alone.
template <class F>
__global__ void __launch_bounds__(128) fr_div_kernel(const char* num, const char* den, char* r, size_t n, size_t lanes) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= lanes) return;
  lane_batch_inverse<F, 8>(
      t, lanes, n, [&](size_t i) { return F::load(den + i * F::BYTES); },
      [&](size_t i, const F& zi, bool nonzero) {
        F out = F::zero();
        if (nonzero) out = num ? F::mul(F::load(num + i * F::BYTES), zi) : zi;
        out.store(r + i * F::BYTES);
      });
}
template <class F>
int fr_div_launch(const void* num, const void* den, void* r, size_t n, hipStream_t s) {
  if (n == 0) return 0;
  const size_t lanes = (n + 7) / 8;
  hipLaunchKernelGGL((fr_div_kernel<F>), dim3((unsigned)((lanes + 127) / 128)), dim3(128), 0, s, (const char*)num,
                     (const char*)den, (char*)r, n, lanes);
  return hipGetLastError() == hipSuccess ? 0 : -1000;
}

// out[i] = in[i] + delta, affine in / affine out (one Fermat inversion per point).  Used to grow
// synthetic base sets on the device: P[i + m] = P[i] + (m*b)G continues P_i = (a + i*b)G
// (SURVEY.md 8d synthetic inputs); also the device analogue of normalize_batch's output form.
template <class C>
__global__ void __launch_bounds__(128) sw_add_affine_kernel(const char* in, char* out, size_t n,  // out may alias in
                                                            const char* __restrict__ delta) {
  typedef typename C::F F;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Affine<F> d = Affine<F>::load(delta);
  Affine<F> p = Affine<F>::load(in + i * Affine<F>::BYTES);
  XYZZ<F> acc = XYZZ<F>::from_affine(d);
  if (!p.is_zero()) xyzz_madd<F>(acc, p.x, p.y);
  F x = F::zero(), y = F::zero();
  if (!acc.is_zero()) {
    F zzzi = F::inverse(acc.zzz);
    F zzi = F::sqr(F::mul(acc.zz, zzzi));  // ZZ^-1 = (ZZ * ZZZ^-1)^2 since ZZ^3 = ZZZ^2
    x = F::mul(acc.x, zzi);
    y = F::mul(acc.y, zzzi);
  }
  x.store(out + i * Affine<F>::BYTES);
  y.store(out + i * Affine<F>::BYTES + F::BYTES);
}
template <class C>
int sw_add_affine_launch(const void* in, void* out, size_t n, const void* d_delta, hipStream_t s) {
  if (n == 0) return 0;
  hipLaunchKernelGGL((sw_add_affine_kernel<C>), dim3((unsigned)((n + 127) / 128)), dim3(128), 0, s, (const char*)in,
                     (char*)out, n, (const char*)d_delta);
  return hipGetLastError() == hipSuccess ? 0 : -1000;
}

// CurveGroup::normalize_batch on the device (group.rs:302-319): Jacobian (x, y, z) -> affine (x/z^2, y/z^3), identity
// -> (0, 0).  The reference amortises ONE inversion over the whole batch with Montgomery's trick
// (ff/src/fields/mod.rs:358-385); here every lane does the same over its own 8 (Fp2: 4) points -- lane_batch_inverse, ec.cuh.
template <class C>
__global__ void __launch_bounds__(128) sw_normalize_batch_kernel(const char* __restrict__ in, char* __restrict__ out,
                                                                 size_t n, size_t lanes) {
  typedef typename C::F F;
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= lanes) return;
  lane_batch_inverse<F, LaneBatch<F>::B>(
      t, lanes, n, [&](size_t i) { return F::load(in + i * 3 * F::BYTES + 2 * F::BYTES); },
      [&](size_t i, const F& zi, bool nonzero) {
        F ax = F::zero(), ay = F::zero();
        if (nonzero) {
          const char* p = in + i * 3 * F::BYTES;
          const F zi2 = F::sqr(zi);
          ax = F::mul(F::load(p), zi2);
          ay = F::mul(F::load(p + F::BYTES), F::mul(zi2, zi));
        }
        ax.store(out + i * 2 * F::BYTES);
        ay.store(out + i * 2 * F::BYTES + F::BYTES);
      });
}
template <class C>
int sw_normalize_batch_launch(const void* in, void* out, size_t n, hipStream_t s) {
  if (n == 0) return 0;
  const size_t lanes = (n + LaneBatch<typename C::F>::B - 1) / LaneBatch<typename C::F>::B;
  hipLaunchKernelGGL((sw_normalize_batch_kernel<C>), dim3((unsigned)((lanes + 127) / 128)), dim3(128), 0, s,
                     (const char*)in, (char*)out, n, lanes);
  return hipGetLastError() == hipSuccess ? 0 : -1000;
}

}  // namespace arkhip
