// C ABI of libark_hip.so, unit 3 of 5: radix-2 domains, transforms over field and group elements, device-resident vectors,
// polynomial multiplication (see include/ark_hip.h).
#include "capi_core.hpp"
#include "capi_hostmath.hpp"
using namespace arkhip;
using namespace arkhip::capi;

extern "C" {

// ---- radix-2 domain / FFT --------------------------------------------------------------------------------
int ark_hip_radix2_domain_new(int field, size_t num_coeffs, ark_hip_radix2_domain* out) {
  if (!out) return ARK_HIP_ERR_ARG;
  switch (field) {
    case ARK_HIP_BN254_FR: return domain_new<BN254_FR>(num_coeffs, out);
    case ARK_HIP_BLS12_381_FR: return domain_new<BLS12_381_FR>(num_coeffs, out);
    case ARK_HIP_BLS12_377_FR: return domain_new<BLS12_377_FR>(num_coeffs, out);
  }
  return ARK_HIP_ERR_ARG;
}
int ark_hip_radix2_domain_get_coset(int field, const ark_hip_radix2_domain* dom, const uint64_t* offset,
                                    ark_hip_radix2_domain* out) {
  if (!dom || !offset || !out) return ARK_HIP_ERR_ARG;
  switch (field) {
    case ARK_HIP_BN254_FR: return domain_coset<BN254_FR>(dom, offset, out);
    case ARK_HIP_BLS12_381_FR: return domain_coset<BLS12_381_FR>(dom, offset, out);
    case ARK_HIP_BLS12_377_FR: return domain_coset<BLS12_377_FR>(dom, offset, out);
  }
  return ARK_HIP_ERR_ARG;
}

static int fft_device_entry(int field, const ark_hip_radix2_domain* dom, void* d, int inverse, size_t num_coeffs) {
  if (!dom || !d) return ARK_HIP_ERR_ARG;
  ARK_SCOPE(sc);
  int zlog = 0;
  if (!inverse && num_coeffs < dom->size) {
    // coefficients beyond num_coeffs are zero by contract (the reference resizes with zeros): make them so up to the
    // power of two the transform reads
    zlog = degree_aware_zlog(dom, num_coeffs);
    const size_t upto = (size_t)dom->size >> zlog;
    if (upto > num_coeffs)
      ARK_HIP_TRY(hipMemsetAsync((char*)d + num_coeffs * 32, 0, (upto - num_coeffs) * 32, sc.c->stream));
  }
  if (int rc = fft_any(sc.c, field, dom, d, inverse, zlog)) return rc;
  return mark_producer(sc.c);
}
static int fft_host_entry(int field, const ark_hip_radix2_domain* dom, uint64_t* data, int inverse, size_t num_coeffs) {
  if (!dom || !data) return ARK_HIP_ERR_ARG;
  ARK_SCOPE(sc);
  Context* c = sc.c;
  const size_t bytes = (size_t)dom->size * 32;
  if (num_coeffs > dom->size) return ARK_HIP_ERR_ARG;
  if (c->stage_a.ensure(bytes)) return ARK_HIP_ERR_NOMEM;
  if (int rc = c->stager.upload(c->stage_a.p, data, num_coeffs * 32, c->stream)) return rc;  // only what is there
  int rc = fft_device_entry(field, dom, c->stage_a.p, inverse, num_coeffs);
  if (rc) return rc;
  ARK_HIP_TRY(hipMemcpyAsync(data, c->stage_a.p, bytes, hipMemcpyDeviceToHost, c->stream));
  ARK_HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}
int ark_hip_fft_in_place(int field, const ark_hip_radix2_domain* dom, uint64_t* data) {
  return fft_host_entry(field, dom, data, 0, dom ? (size_t)dom->size : 0);
}
int ark_hip_ifft_in_place(int field, const ark_hip_radix2_domain* dom, uint64_t* data) {
  return fft_host_entry(field, dom, data, 1, dom ? (size_t)dom->size : 0);
}
int ark_hip_fft_in_place_device(int field, const ark_hip_radix2_domain* dom, void* d) {
  return fft_device_entry(field, dom, d, 0, dom ? (size_t)dom->size : 0);
}
int ark_hip_ifft_in_place_device(int field, const ark_hip_radix2_domain* dom, void* d) {
  return fft_device_entry(field, dom, d, 1, dom ? (size_t)dom->size : 0);
}
int ark_hip_fft_in_place_degree_aware(int field, const ark_hip_radix2_domain* dom, uint64_t* data, size_t num_coeffs) {
  return fft_host_entry(field, dom, data, 0, num_coeffs);
}
int ark_hip_fft_in_place_degree_aware_device(int field, const ark_hip_radix2_domain* dom, void* d, size_t num_coeffs) {
  if (dom && num_coeffs > dom->size) return ARK_HIP_ERR_ARG;
  return fft_device_entry(field, dom, d, 0, num_coeffs);
}

// `count` independent transforms over the same domain, each in place on its own device buffer of dom->size elements.
// Consecutive transforms go to three streams: one transform alone leaves ~20 % of the vector ALU idle around its
// pass boundaries (tail of one kernel, ramp of the next), another one in flight fills it (2^22: 0.53 -> 0.48 ms per
// transform, 2^20: 0.157 -> 0.108, 2^16: 44 -> 18 us; profiles/r2_fft_bench_shapes.txt).  Asynchronous like the single
// transform: later work on the context stream (and ark_hip_synchronize) waits for all of them.
int ark_hip_fft_batch_in_place_device(int field, const ark_hip_radix2_domain* dom, void* const* d_data, size_t count,
                                      int inverse) {
  if (!dom || (count && !d_data)) return ARK_HIP_ERR_ARG;
  for (size_t i = 0; i < count; i++)
    if (!d_data[i]) return ARK_HIP_ERR_ARG;
  ARK_SCOPE(sc);
  Context* c = sc.c;
  if (count == 0) return 0;
  const int lanes = count < 3 ? (int)count : 3;
  for (int j = 0; j < 3; j++)
    if (!c->fft_ev[j]) ARK_HIP_TRY(hipEventCreateWithFlags(&c->fft_ev[j], hipEventDisableTiming));
  for (int j = 0; j + 1 < lanes; j++)
    if (!c->fft_side[j]) ARK_HIP_TRY(hipStreamCreateWithFlags(&c->fft_side[j], hipStreamNonBlocking));
  // the side streams start after whatever is already queued on the context stream (it may produce their inputs)
  ARK_HIP_TRY(hipEventRecord(c->fft_ev[0], c->stream));
  for (int j = 0; j + 1 < lanes; j++) ARK_HIP_TRY(hipStreamWaitEvent(c->fft_side[j], c->fft_ev[0], 0));
  int rc = 0;
  for (size_t i = 0; i < count && rc == 0; i++) {
    const int lane = (int)(i % (size_t)lanes);
    rc = fft_any(c, field, dom, d_data[i], inverse, 0, lane == 0 ? c->stream : c->fft_side[lane - 1]);
  }
  for (int j = 0; j + 1 < lanes; j++) {  // join, also on error: nothing may outlive the call unordered
    (void)hipEventRecord(c->fft_ev[j + 1], c->fft_side[j]);
    (void)hipStreamWaitEvent(c->stream, c->fft_ev[j + 1], 0);
  }
  if (rc == 0) rc = mark_producer(c);
  return rc;
}

// r[i] = a[i] * b[i] over n Fr elements in device memory (Evaluations *= Evaluations,
// poly/src/evaluations/univariate/mod.rs MulAssign; the middle step of DensePolynomial multiplication,
// poly/src/polynomial/univariate/dense.rs:641-656).  Asynchronous on the context stream; r may alias a or b.
int ark_hip_fr_mul_device(int field, const void* d_a, const void* d_b, void* d_r, size_t n) {
  if (n && (!d_a || !d_b || !d_r)) return ARK_HIP_ERR_ARG;
  ARK_SCOPE(sc);
  if (int rc = fr_mul_dispatch(field, d_a, d_b, d_r, n, sc.c->stream)) return rc;
  return mark_producer(sc.c);
}

// `&DensePolynomial * &DensePolynomial` (poly/src/polynomial/univariate/dense.rs:641-656): zero if either factor is zero;
// otherwise evaluate both over the radix-2 domain of size >= na + nb - 1 (evaluate_over_domain_by_ref, univariate/mod.rs:
// 305-360), multiply the evaluations pointwise (Evaluations *=) and interpolate (evaluations/univariate/mod.rs:40-50).
// HOST pointers in and out; in between everything stays on the device: ONE upload of the two coefficient vectors (the zero
// padding is written on the device -- or never read: degree-aware transforms), the two forward transforms in flight
// together, the pointwise product, the inverse transform, ONE download of the na + nb - 1 coefficients.
// out: room for na + nb - 1 elements; *out_len: the product's coefficient count with leading zeros dropped, as
// DensePolynomial::from_coefficients_vec leaves it (0: the zero polynomial).  ARK_HIP_ERR_ARG when the field's 2-adicity
// cannot hold the domain (the reference panics: "field is not smooth enough to construct domain").
int ark_hip_poly_mul(int field, const uint64_t* a, size_t na, const uint64_t* b, size_t nb, uint64_t* out, size_t* out_len) {
  if (!out_len || (na && !a) || (nb && !b)) return ARK_HIP_ERR_ARG;
  *out_len = 0;
  auto all_zero = [](const uint64_t* p, size_t n) {
    for (size_t i = 0; i < 4 * n; i++)
      if (p[i]) return false;
    return true;
  };
  if (na == 0 || nb == 0 || all_zero(a, na) || all_zero(b, nb)) return 0;   // DensePolynomial::is_zero
  if (!out) return ARK_HIP_ERR_ARG;
  const size_t len = na + nb - 1;
  ark_hip_radix2_domain dom;
  if (int rc = ark_hip_radix2_domain_new(field, len, &dom)) return rc;
  const size_t n = (size_t)dom.size;
  ARK_SCOPE(sc);
  Context* c = sc.c;
  if (c->stage_a.cap < n * 32 || c->stage_b.cap < n * 32) {
    if (int rc = sync_compute(c)) return rc;
    if (c->stage_a.ensure(n * 32) || c->stage_b.ensure(n * 32)) return ARK_HIP_ERR_NOMEM;
  }
  if (int rc = c->stager.upload(c->stage_a.p, a, na * 32, c->stream)) return rc;
  if (int rc = c->stager.upload(c->stage_b.p, b, nb * 32, c->stream)) return rc;
  // forward transforms: short inputs take the degree-aware path (their padding is never read beyond the next power of
  // two); the two run on two streams (ark_hip_fft_batch_in_place_device's arrangement)
  auto pad = [&](void* d, size_t have) -> int {
    const int zlog = degree_aware_zlog(&dom, have);
    const size_t upto = n >> zlog;
    if (upto > have) ARK_HIP_TRY(hipMemsetAsync((char*)d + have * 32, 0, (upto - have) * 32, c->stream));
    return zlog;
  };
  const int za = pad(c->stage_a.p, na), zb = pad(c->stage_b.p, nb);
  if (za < 0 || zb < 0) return za < 0 ? za : zb;
  if (!c->fft_ev[0])
    for (int j = 0; j < 3; j++) ARK_HIP_TRY(hipEventCreateWithFlags(&c->fft_ev[j], hipEventDisableTiming));
  if (!c->fft_side[0]) ARK_HIP_TRY(hipStreamCreateWithFlags(&c->fft_side[0], hipStreamNonBlocking));
  ARK_HIP_TRY(hipEventRecord(c->fft_ev[0], c->stream));
  ARK_HIP_TRY(hipStreamWaitEvent(c->fft_side[0], c->fft_ev[0], 0));
  int rc = fft_any(c, field, &dom, c->stage_a.p, 0, za, c->stream);
  if (rc == 0) rc = fft_any(c, field, &dom, c->stage_b.p, 0, zb, c->fft_side[0]);
  (void)hipEventRecord(c->fft_ev[1], c->fft_side[0]);
  (void)hipStreamWaitEvent(c->stream, c->fft_ev[1], 0);
  if (rc == 0) rc = fr_mul_dispatch(field, c->stage_a.p, c->stage_b.p, c->stage_a.p, n, c->stream);
  if (rc == 0) rc = fft_any(c, field, &dom, c->stage_a.p, 1, 0, c->stream);
  if (rc) {
    (void)hipStreamSynchronize(c->stream);
    return rc;
  }
  ARK_HIP_TRY(hipMemcpyAsync(out, c->stage_a.p, len * 32, hipMemcpyDeviceToHost, c->stream));
  ARK_HIP_TRY(hipStreamSynchronize(c->stream));
  size_t top = len;   // truncate_leading_zeros (dense.rs)
  while (top > 0 && !(out[4 * top - 1] | out[4 * top - 2] | out[4 * top - 3] | out[4 * top - 4])) top--;
  *out_len = top;
  return 0;
}

// out = base^exp in Fr (host arithmetic): domain elements / twiddles for hosts without field code of their own
int ark_hip_fr_pow(int field, const uint64_t* base, uint64_t exp, uint64_t* out) {
  if (!base || !out) return ARK_HIP_ERR_ARG;
  uint64_t e[1] = {exp};
  switch (field) {
    case ARK_HIP_BN254_FR: host_pow<BN254_FR>(Fp<BN254_FR>::load(base), e, 1).store(out); return 0;
    case ARK_HIP_BLS12_381_FR: host_pow<BLS12_381_FR>(Fp<BLS12_381_FR>::load(base), e, 1).store(out); return 0;
    case ARK_HIP_BLS12_377_FR: host_pow<BLS12_377_FR>(Fp<BLS12_377_FR>::load(base), e, 1).store(out); return 0;
  }
  return ARK_HIP_ERR_ARG;
}

// G-point transform along the slow axis of a [G][cols] array in device memory: the cross-GPU stage of a
// sharded FFT (algebra_amd/dist.py).  root = primitive G-th root of unity to use (w_n^(n/G) or its inverse).
int ark_hip_fft_axis_device(int field, void* d_data, unsigned G, size_t cols, const uint64_t* root) {
  if (!d_data || !root) return ARK_HIP_ERR_ARG;
  ARK_SCOPE(sc);
  if (int rc = fft_axis_dispatch(field, sc.c->fft, d_data, d_data, G, cols, root, sc.c->stream)) return rc;
  return mark_producer(sc.c);
}

// ---- transform whose coefficients are GROUP elements --------------------------------------------------------------------
// EvaluationDomain::fft_in_place / ifft_in_place for T = Projective<P> (poly/src/domain/mod.rs:332-362 with
// radix2/fft.rs:74-119; the reference's own use: poly/src/test.rs:57): n = dom->size Jacobian points of `curve`, whose scalar
// field must be the domain's field, transformed in place -- forward: X_j = sum_i [(h g^j)^i] P_i; inverse:
// P_i = [n^-1 h^-i] sum_j [g^-ij] X_j.  The caller pads with identities (z = 0) to the domain size as the reference's
// resize does.  gfft.cuh.
static int gfft_entry(Context* c, int curve, const ark_hip_radix2_domain* dom, void* d_jac, int inverse) {
  const int field = CURVES[curve].scalar_field;
  const int k = (int)dom->log_size_of_group;
  if (k < 0 || k > 26 || dom->size != ((uint64_t)1 << k)) return ARK_HIP_ERR_ARG;
  if (!domain_is_of_field(field, dom)) return ARK_HIP_ERR_ARG;   // a domain of another scalar field than the curve's
  const size_t n = (size_t)1 << k;
  const bool coset = !field_is_one(field, dom->offset);
  const uint32_t* roots = nullptr;
  if (k >= 1)
    if (int rc = fft_roots_dispatch(field, c->fft, k, inverse ? dom->group_gen_inv : dom->group_gen, c->stream, &roots)) return rc;
  const size_t wb = gfft_work_bytes_any(curve, k);
  if (wb == 0) return ARK_HIP_ERR_ARG;
  if (c->gfft_work.cap < wb || c->gfft_scal.cap < n * 32) {
    if (int rc = sync_compute(c)) return rc;
    if (c->gfft_work.ensure(wb) || c->gfft_scal.ensure(n * 32)) return ARK_HIP_ERR_NOMEM;
  }
  const uint32_t *pre = nullptr, *post = nullptr;
  if (!inverse && coset) {        // distribute_powers(coeffs, offset)                                   fft.rs:74-79
    if (int rc = fft_scalars_dispatch(field, c->fft, dom->offset, nullptr, n, c->gfft_scal.p, c->stream)) return rc;
    pre = (const uint32_t*)c->gfft_scal.p;
  }
  if (inverse) {                  // x[i] *= size_inv * offset_inv^i  (offset_inv = 1 off a coset)         fft.rs:81-88
    if (int rc = fft_scalars_dispatch(field, c->fft, dom->offset_inv, dom->size_inv, n, c->gfft_scal.p, c->stream)) return rc;
    post = (const uint32_t*)c->gfft_scal.p;
  }
  if (int rc = gfft_run_dispatch(curve, d_jac, k, roots, pre, post, c->gfft_work.p, c->stream)) return rc;
  return mark_producer(c);
}
int ark_hip_fft_group_in_place_device(int curve, const ark_hip_radix2_domain* dom, void* d_jac_points, int inverse) {
  if (curve < 0 || curve > 4 || !dom || !d_jac_points) return ARK_HIP_ERR_ARG;
  ARK_SCOPE(sc);
  return gfft_entry(sc.c, curve, dom, d_jac_points, inverse);
}
int ark_hip_fft_group_in_place(int curve, const ark_hip_radix2_domain* dom, uint64_t* jac_points, int inverse) {
  if (curve < 0 || curve > 4 || !dom || !jac_points) return ARK_HIP_ERR_ARG;
  ARK_SCOPE(sc);
  Context* c = sc.c;
  const size_t bytes = (size_t)dom->size * CURVES[curve].fe_words * 3 * 8;
  if (c->stage_a.cap < bytes) {
    if (int rc = sync_compute(c)) return rc;
    if (c->stage_a.ensure(bytes)) return ARK_HIP_ERR_NOMEM;
  }
  if (int rc = c->stager.upload(c->stage_a.p, jac_points, bytes, c->stream)) return rc;
  if (int rc = gfft_entry(c, curve, dom, c->stage_a.p, inverse)) {
    (void)hipStreamSynchronize(c->stream);
    return rc;   // the device worked on its own copy: the caller's points are intact
  }
  ARK_HIP_TRY(hipMemcpyAsync(jac_points, c->stage_a.p, bytes, hipMemcpyDeviceToHost, c->stream));
  ARK_HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

// The rest of the pointwise algebra on device-resident vectors of Fr (Evaluations +=, -=, negation; a polynomial or an
// evaluation vector times a field element -- poly/src/evaluations/univariate/mod.rs:104-180, polynomial/univariate/
// dense.rs:343-371, :604-622): what a chain evaluate_over_domain -> pointwise -> interpolate needs besides the transforms
// and ark_hip_fr_mul_device to stay on the device between ONE upload and ONE download.  Asynchronous on the context
// stream; r may alias a or b.
int ark_hip_fr_add_device(int field, const void* d_a, const void* d_b, void* d_r, size_t n) {
  if (n && (!d_a || !d_b || !d_r)) return ARK_HIP_ERR_ARG;
  ARK_SCOPE(sc);
  if (int rc = fr_op_dispatch(field, 0, d_a, d_b, d_r, n, sc.c->stream)) return rc;
  return mark_producer(sc.c);
}
int ark_hip_fr_sub_device(int field, const void* d_a, const void* d_b, void* d_r, size_t n) {
  if (n && (!d_a || !d_b || !d_r)) return ARK_HIP_ERR_ARG;
  ARK_SCOPE(sc);
  if (int rc = fr_op_dispatch(field, 1, d_a, d_b, d_r, n, sc.c->stream)) return rc;
  return mark_producer(sc.c);
}
int ark_hip_fr_neg_device(int field, const void* d_a, void* d_r, size_t n) {
  if (n && (!d_a || !d_r)) return ARK_HIP_ERR_ARG;
  ARK_SCOPE(sc);
  if (int rc = fr_op_dispatch(field, 4, d_a, nullptr, d_r, n, sc.c->stream)) return rc;
  return mark_producer(sc.c);
}
// r[i] = a[i] * k; k: one Montgomery element in HOST memory, read before the call returns
int ark_hip_fr_scale_device(int field, const void* d_a, const uint64_t* k, void* d_r, size_t n) {
  if (!k || (n && (!d_a || !d_r))) return ARK_HIP_ERR_ARG;
  ARK_SCOPE(sc);
  if (int rc = fr_scale_dispatch(field, d_a, k, d_r, n, sc.c->stream)) return rc;
  return mark_producer(sc.c);
}
// r[i] = a[i] / b[i] (Evaluations /= Evaluations, evaluations/univariate/mod.rs:142-163) and r[i] = 1 / a[i]
// (ark_ff::batch_inversion, ff/src/fields/mod.rs:358-385): a zero divisor gives zero, as the reference's batch inversion
// leaves zeros in place.  r may alias an operand.
int ark_hip_fr_div_device(int field, const void* d_a, const void* d_b, void* d_r, size_t n) {
  if (n && (!d_a || !d_b || !d_r)) return ARK_HIP_ERR_ARG;
  ARK_SCOPE(sc);
  if (int rc = fr_div_dispatch(field, d_a, d_b, d_r, n, sc.c->stream)) return rc;
  return mark_producer(sc.c);
}
int ark_hip_fr_inverse_device(int field, const void* d_a, void* d_r, size_t n) {
  if (n && (!d_a || !d_r)) return ARK_HIP_ERR_ARG;
  ARK_SCOPE(sc);
  if (int rc = fr_div_dispatch(field, nullptr, d_a, d_r, n, sc.c->stream)) return rc;
  return mark_producer(sc.c);
}
// device-to-device copy / byte fill on the context stream (a device vector's clone() and its zero-extension), ordered with
// the transforms and the pointwise kernels like every other *_device entry
int ark_hip_memcpy_d2d(void* dst_dptr, const void* src_dptr, size_t bytes) {
  if (bytes && (!dst_dptr || !src_dptr)) return ARK_HIP_ERR_ARG;
  ARK_SCOPE(sc);
  if (bytes) ARK_HIP_TRY(hipMemcpyAsync(dst_dptr, src_dptr, bytes, hipMemcpyDeviceToDevice, sc.c->stream));
  return mark_producer(sc.c);
}
int ark_hip_memset_device(void* dptr, int value, size_t bytes) {
  if (bytes && !dptr) return ARK_HIP_ERR_ARG;
  ARK_SCOPE(sc);
  if (bytes) ARK_HIP_TRY(hipMemsetAsync(dptr, value, bytes, sc.c->stream));
  return mark_producer(sc.c);
}

// 0: the saturated pass kernel (default), 1: the carry-free 9 x 29-bit pass kernel, -1: back to the environment's choice
int ark_hip_fft_set_kernel(int variant) {
  if (variant < -1 || variant > 1) return ARK_HIP_ERR_ARG;
  ARK_SCOPE(sc);
  std::lock_guard<std::mutex> lock(sc.c->fft.mu);
  sc.c->fft.kernel_variant = variant;
  return 0;
}
int ark_hip_fft_set_timing(int enable) {
  ARK_SCOPE(sc);
  sc.c->fft_timing = enable != 0;
  return 0;
}
int ark_hip_fft_last_timing(double out[10]) {
  if (!out) return ARK_HIP_ERR_ARG;
  ARK_SCOPE(sc);
  out[0] = sc.c->fft_tm.total;
  out[1] = sc.c->fft_tm.npass;
  for (int i = 0; i < 8; i++) out[2 + i] = sc.c->fft_tm.pass[i];
  return 0;
}

}  // extern "C"
