// Elementwise TEST kernels: expose the device point arithmetic to tests/ through the C ABI of libark_hip_test.so, so that
// every formula the MSM kernels use is checked against the oracle on its own (the field-level hooks run the product's own
// pointwise kernel, devops.cuh: field_op_kernel).
#pragma once
#include "devops.cuh"

namespace arkhip {

// acc: XYZZ (kinds 2..6) or affine (kind 7); other: affine (2,3) or XYZZ (4); out: XYZZ, or Jacobian for kind 6
template <class C>
__global__ void __launch_bounds__(128) test_point_op_kernel(int kind, const char* __restrict__ acc_in,
                                                            const char* __restrict__ other, char* __restrict__ out,
                                                            size_t n) {
  typedef typename C::F F;
  typedef XYZZ<F> Pt;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (kind == 7) {
    Affine<F> p = Affine<F>::load(acc_in + i * Affine<F>::BYTES);
    Pt r = p.is_zero() ? Pt::zero() : xyzz_mdbl<F>(p.x, p.y);
    r.store(out + i * Pt::BYTES);
    return;
  }
  Pt acc = Pt::load(acc_in + i * Pt::BYTES);
  if (kind == 2 || kind == 3) {
    Affine<F> p = Affine<F>::load(other + i * Affine<F>::BYTES);
    if (!p.is_zero()) {
      F y = F::cond_neg(p.y, kind == 3);
      xyzz_madd<F>(acc, p.x, y);
    }
    acc.store(out + i * Pt::BYTES);
  } else if (kind == 4) {
    Pt o = Pt::load(other + i * Pt::BYTES);
    xyzz_add<F>(acc, o);
    acc.store(out + i * Pt::BYTES);
  } else if (kind == 5) {
    xyzz_dbl<F>(acc).store(out + i * Pt::BYTES);
  } else if (kind == 6) {
    xyzz_to_jac<F>(acc).store(out + i * Jac<F>::BYTES);
  }
}

// kinds 12..15: the additions of the accumulate / reduction kernels in their carry-free form (LazyK: one lane per point, or
// one lane PAIR over Fp2), stored bucket in -> stored bucket out:
//   12 / 13  acc +/- affine (the mixed addition; equal points double the base)      14  acc += stored bucket (repacked
//   operand)      15  acc += accumulator (both sides small)
template <class C>
__global__ void __launch_bounds__(128) test_lazy_point_op_kernel(int kind, const char* __restrict__ acc_in,
                                                                 const char* __restrict__ other, char* __restrict__ out,
                                                                 size_t n) {
  typedef LazyK<C> K;
  typedef typename K::FM F;
  typedef XYZZ<F> Pt;
  const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / K::LANES;
  if (i >= n) return;
  typename K::Acc acc = K::from_bucket(Pt::load(acc_in + i * Pt::BYTES));
  if (kind == 12 || kind == 13) {
    const char* src = other + i * Affine<F>::BYTES;
    const Affine<F> p = Affine<F>::load(src);
    if (!p.is_zero()) {
      if (K::madd(acc, p, kind == 13)) {
        typename K::Acc d;
        K::mdbl(d, src, kind == 13);
        acc = d;
      }
    }
  } else if (kind == 14) {
    K::add(acc, Pt::load(other + i * Pt::BYTES));
  } else {
    const typename K::Acc o = K::from_bucket(Pt::load(other + i * Pt::BYTES));
    K::add_acc(acc, o);
  }
  K::to_bucket(acc).store(out + i * Pt::BYTES);
}

template <class C>
int test_point_op_launch(int kind, const void* acc, const void* other, void* out, size_t n, hipStream_t s) {
  if (n == 0) return 0;
  if (kind >= 12 && kind <= 15) {
    if constexpr (C::LAZY_A) {
      hipLaunchKernelGGL((test_lazy_point_op_kernel<C>), dim3((unsigned)((n * LazyK<C>::LANES + 127) / 128)), dim3(128), 0, s, kind,
                         (const char*)acc, (const char*)other, (char*)out, n);
      return hipGetLastError() == hipSuccess ? 0 : -1000;
    }
    return -1;
  }
  hipLaunchKernelGGL((test_point_op_kernel<C>), dim3((unsigned)((n + 127) / 128)), dim3(128), 0, s, kind,
                     (const char*)acc, (const char*)other, (char*)out, n);
  return hipGetLastError() == hipSuccess ? 0 : -1000;
}

}  // namespace arkhip
