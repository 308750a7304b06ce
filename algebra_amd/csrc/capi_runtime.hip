// C ABI of libark_hip.so, unit 1 of 5: devices and contexts, device / pinned memory, and the host-side group helpers
// (see include/ark_hip.h for the contract and the reference items replaced).
#include "capi_core.hpp"
#include "capi_hostmath.hpp"
#include "capi_cache.hpp"
using namespace arkhip;
using namespace arkhip::capi;

extern "C" {


int ark_hip_device_count(void) { return device_count_raw(); }

int ark_hip_init(int device) {
  Context* c = nullptr;
  if (device < 0) return ARK_HIP_ERR_ARG;
  int rc = get_ctx(device, &c);
  if (rc) return rc;
  t_dev = device;
  return 0;
}
int ark_hip_set_device(int device) { return ark_hip_init(device); }
int ark_hip_get_device(void) { return t_dev >= 0 ? t_dev : (g_default >= 0 ? g_default : 0); }

void ark_hip_shutdown(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  for (int i = 0; i < MAX_DEV; i++) {
    Context* c = g_ctxs[i];
    if (!c) continue;
    {
      std::lock_guard<std::recursive_mutex> cl(c->mu);  // waits for calls in flight on this device
      (void)hipSetDevice(c->physical);
      (void)hipStreamSynchronize(c->stream);
      if (c->stream_b) (void)hipStreamSynchronize(c->stream_b);
      (void)hipStreamSynchronize(c->copy_stream);
      for (int j = 0; j < 2; j++)
        if (c->fft_side[j]) (void)hipStreamSynchronize(c->fft_side[j]);
      while (!c->base_cache.empty()) cache_drop(c, c->base_cache.size() - 1);
      c->stager.release();
      c->piece_buckets.release();
      for (int j = 0; j < 2; j++)
        if (c->piece_ev[j]) (void)hipEventDestroy(c->piece_ev[j]);
      if (c->lane_ev) (void)hipEventDestroy(c->lane_ev);
      c->msm[0].release();
      c->msm[1].release();
      c->fft.release();
      c->stage_a.release();
      c->stage_b.release();
      c->stage_c.release();
      for (int j = 0; j < 2; j++) {
        c->ring_s[j].release();
        c->ring_b[j].release();
        if (c->ring_free[j]) (void)hipEventDestroy(c->ring_free[j]);
        if (c->ring_up[j]) (void)hipEventDestroy(c->ring_up[j]);
      }
      (void)hipStreamDestroy(c->stream);
      if (c->stream_b) (void)hipStreamDestroy(c->stream_b);
      (void)hipStreamDestroy(c->copy_stream);
      for (int j = 0; j < 2; j++)
        if (c->fft_side[j]) (void)hipStreamDestroy(c->fft_side[j]);
      for (int j = 0; j < 3; j++)
        if (c->fft_ev[j]) (void)hipEventDestroy(c->fft_ev[j]);
    }
    delete c;
    g_ctxs[i] = nullptr;
  }
  g_default = -1;
}

int ark_hip_synchronize(void) {
  ARK_SCOPE(sc);
  ARK_HIP_TRY(hipStreamSynchronize(sc.c->copy_stream));
  if (sc.c->comm && sc.c->comm_world > 1) {   // collectives may be queued (the sharded FFT is asynchronous): a bounded wait
    if (int rc = comm_sync(sc.c, sc.c->stream)) return rc;
    if (sc.c->comm_stream)
      if (int rc = comm_sync(sc.c, sc.c->comm_stream)) return rc;
  }
  if (int rc = sync_compute(sc.c)) return rc;
  return 0;
}

const char* ark_hip_version(void) { return "ark_hip 0.4 (gfx950)"; }
int ark_hip_host_threads(int out[2]) {
  if (!out) return ARK_HIP_ERR_ARG;
  out[0] = HostPool::instance().helpers();
  out[1] = HostPool::instance().threads_created();
  return 0;
}

int ark_hip_curve_info(int curve, int* fe_words, int* scalar_field, int* base_field, int* ext_degree) {
  if (curve < 0 || curve > 4) return ARK_HIP_ERR_ARG;
  if (fe_words) *fe_words = CURVES[curve].fe_words;
  if (scalar_field) *scalar_field = CURVES[curve].scalar_field;
  if (base_field) *base_field = CURVES[curve].base_field;
  if (ext_degree) *ext_degree = CURVES[curve].ext;
  return 0;
}

// ---- device / pinned memory for hosts without their own HIP binding ----
int ark_hip_malloc(size_t bytes, void** out_dptr) {
  if (!out_dptr) return ARK_HIP_ERR_ARG;
  ARK_SCOPE(sc);
  *out_dptr = nullptr;
  if (bytes == 0) return 0;
  if (hipMalloc(out_dptr, bytes) != hipSuccess) return ARK_HIP_ERR_NOMEM;
  return 0;
}
int ark_hip_free(void* dptr) {
  if (!dptr) return 0;
  ARK_SCOPE(sc);
  if (int rc = sync_compute(sc.c)) return rc;
  ARK_HIP_TRY(hipFree(dptr));
  return 0;
}
int ark_hip_memcpy_h2d(void* dst_dptr, const void* src_host, size_t bytes) {
  if (bytes && (!dst_dptr || !src_host)) return ARK_HIP_ERR_ARG;
  ARK_SCOPE(sc);
  ARK_HIP_TRY(hipMemcpyAsync(dst_dptr, src_host, bytes, hipMemcpyHostToDevice, sc.c->stream));
  ARK_HIP_TRY(hipStreamSynchronize(sc.c->stream));
  return 0;
}
int ark_hip_memcpy_d2h(void* dst_host, const void* src_dptr, size_t bytes) {
  if (bytes && (!dst_host || !src_dptr)) return ARK_HIP_ERR_ARG;
  ARK_SCOPE(sc);
  ARK_HIP_TRY(hipMemcpyAsync(dst_host, src_dptr, bytes, hipMemcpyDeviceToHost, sc.c->stream));
  ARK_HIP_TRY(hipStreamSynchronize(sc.c->stream));
  return 0;
}
int ark_hip_host_alloc(size_t bytes, void** out_ptr) {
  if (!out_ptr) return ARK_HIP_ERR_ARG;
  ARK_SCOPE(sc);
  *out_ptr = nullptr;
  if (bytes == 0) return 0;
  if (hipHostMalloc(out_ptr, bytes) != hipSuccess) return ARK_HIP_ERR_NOMEM;
  return 0;
}
int ark_hip_host_free(void* ptr) {
  if (!ptr) return 0;
  ARK_SCOPE(sc);
  ARK_HIP_TRY(hipHostFree(ptr));
  return 0;
}

int ark_hip_curve_generator(int curve, uint64_t* out_xy) {
  if (!out_xy) return ARK_HIP_ERR_ARG;
  const uint64_t* g = nullptr;
  switch (curve) {
    case 0: g = GEN_BN254_G1; break;
    case 1: g = GEN_BLS12_381_G1; break;
    case 2: g = GEN_BLS12_377_G1; break;
    case 3: g = GEN_BLS12_377_G2; break;
    case 4: g = GEN_BLS12_381_G2; break;
    default: return ARK_HIP_ERR_ARG;
  }
  memcpy(out_xy, g, (size_t)CURVES[curve].fe_words * 16);
  return 0;
}

// ---- host-side group helpers (no device involved) ----
int ark_hip_sw_sum(int curve, const uint64_t* jac_points, size_t n, uint64_t* out_xyz) {
  if (!out_xyz || (n && !jac_points)) return ARK_HIP_ERR_ARG;
  switch (curve) {
    case 0: return host_sum<BN254_G1>(jac_points, n, out_xyz);
    case 1: return host_sum<BLS12_381_G1>(jac_points, n, out_xyz);
    case 2: return host_sum<BLS12_377_G1>(jac_points, n, out_xyz);
    case 3: return host_sum<BLS12_377_G2>(jac_points, n, out_xyz);
    case 4: return host_sum<BLS12_381_G2>(jac_points, n, out_xyz);
  }
  return ARK_HIP_ERR_ARG;
}
int ark_hip_sw_into_affine(int curve, const uint64_t* jac_points, size_t n, uint64_t* out_xy) {
  if (n && (!jac_points || !out_xy)) return ARK_HIP_ERR_ARG;
  switch (curve) {
    case 0: return host_into_affine<BN254_G1>(jac_points, n, out_xy);
    case 1: return host_into_affine<BLS12_381_G1>(jac_points, n, out_xy);
    case 2: return host_into_affine<BLS12_377_G1>(jac_points, n, out_xy);
    case 3: return host_into_affine<BLS12_377_G2>(jac_points, n, out_xy);
    case 4: return host_into_affine<BLS12_381_G2>(jac_points, n, out_xy);
  }
  return ARK_HIP_ERR_ARG;
}

// out[i] = in[i] + delta on the device (affine in/out); d_in may equal d_out
int ark_hip_sw_add_affine_device(int curve, const void* d_in, void* d_out, size_t n, const uint64_t* delta_xy) {
  if (curve < 0 || curve > 4 || !delta_xy || (n && (!d_in || !d_out))) return ARK_HIP_ERR_ARG;
  ARK_SCOPE(sc);
  Context* c = sc.c;
  size_t ab = (size_t)CURVES[curve].fe_words * 16;
  if (c->stage_c.ensure(ab)) return ARK_HIP_ERR_NOMEM;
  ARK_HIP_TRY(hipMemcpyAsync(c->stage_c.p, delta_xy, ab, hipMemcpyHostToDevice, c->stream));
  int rc = add_affine_dispatch(curve, d_in, d_out, n, c->stage_c.p, c->stream);
  if (rc) return rc;
  ARK_HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

// CurveGroup::normalize_batch for n Projective points in device memory -> n Affine points (device memory)
int ark_hip_sw_normalize_batch_device(int curve, const void* d_jac, void* d_out_xy, size_t n) {
  if (curve < 0 || curve > 4 || (n && (!d_jac || !d_out_xy))) return ARK_HIP_ERR_ARG;
  ARK_SCOPE(sc);
  int rc = normalize_dispatch(curve, d_jac, d_out_xy, n, sc.c->stream);
  if (rc) return rc;
  ARK_HIP_TRY(hipStreamSynchronize(sc.c->stream));
  return 0;
}

// The same from HOST memory (what the Rust hook behind CurveGroup::normalize_batch hands over, group.rs:302-319): one
// upload of the n Projective points, the lane-batched inversion kernel, one download of the n Affine points.
int ark_hip_sw_normalize_batch(int curve, const uint64_t* jac_points, size_t n, uint64_t* out_xy) {
  if (curve < 0 || curve > 4 || (n && (!jac_points || !out_xy))) return ARK_HIP_ERR_ARG;
  if (n == 0) return 0;
  ARK_SCOPE(sc);
  Context* c = sc.c;
  const size_t fb = (size_t)CURVES[curve].fe_words * 8;
  if (c->stage_a.cap < n * 3 * fb || c->stage_b.cap < n * 2 * fb) {
    if (int rc = sync_compute(c)) return rc;
    if (c->stage_a.ensure(n * 3 * fb) || c->stage_b.ensure(n * 2 * fb)) return ARK_HIP_ERR_NOMEM;
  }
  if (int rc = c->stager.upload(c->stage_a.p, jac_points, n * 3 * fb, c->stream)) return rc;
  if (int rc = normalize_dispatch(curve, c->stage_a.p, c->stage_b.p, n, c->stream)) return rc;
  ARK_HIP_TRY(hipMemcpyAsync(out_xy, c->stage_b.p, n * 2 * fb, hipMemcpyDeviceToHost, c->stream));
  ARK_HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

}  // extern "C"
