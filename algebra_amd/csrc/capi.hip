// C ABI of libark_hip.so (see include/ark_hip.h for the contract and the reference items replaced).
#include "../../include/ark_hip.h"
#include <string.h>
#include <mutex>
#include "msm.cuh"
#include "fft.cuh"
#include "internal.hpp"
#include "curve_consts.hpp"

using namespace arkhip;

namespace {

struct Context {
  int device = -1;
  hipStream_t stream = nullptr;
  MsmWorkspace msm;
  FftWorkspace fft;
  DevBuf stage_a, stage_b, stage_c;  // host-pointer entry points: device copies
  bool msm_timing = false, fft_timing = false;
  MsmTimings msm_tm;
  int msm_c = 0, msm_W = 0;
  FftTimings fft_tm;
};
Context* g_ctx = nullptr;
std::mutex g_mu;

int ensure_ctx() {
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_ctx) {
    // HIP's current device is per host thread: re-select ours on every entry
    return hipSetDevice(g_ctx->device) == hipSuccess ? 0 : ARK_HIP_ERR_NO_DEVICE;
  }
  int cnt = 0;
  if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0) {
    fprintf(stderr, "ark_hip: no HIP device visible -- this library has no CPU fallback\n");
    return ARK_HIP_ERR_NO_DEVICE;
  }
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return ARK_HIP_ERR_NO_DEVICE;
  Context* c = new Context();
  c->device = dev;
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
    delete c;
    return ARK_HIP_ERR_NO_DEVICE;
  }
  g_ctx = c;
  return 0;
}

struct CurveInfo { int fe_words, scalar_field, base_field, ext; };
const CurveInfo CURVES[5] = {
    {4, ARK_HIP_BN254_FR, ARK_HIP_BN254_FQ, 1},       {6, ARK_HIP_BLS12_381_FR, ARK_HIP_BLS12_381_FQ, 1},
    {6, ARK_HIP_BLS12_377_FR, ARK_HIP_BLS12_377_FQ, 1}, {12, ARK_HIP_BLS12_377_FR, ARK_HIP_BLS12_377_FQ, 2},
    {12, ARK_HIP_BLS12_381_FR, ARK_HIP_BLS12_381_FQ, 2}};

int msm_dispatch(int curve, MsmWorkspace& ws, const void* b, const void* s, size_t n, int mont, uint64_t* out,
                 hipStream_t st, MsmTimings* tm) {
  switch (curve) {
#ifndef ARK_HIP_DEV
    case ARK_HIP_BN254_G1: return msm_run_BN254_G1(ws, b, s, n, mont, out, st, tm);
#endif
    case ARK_HIP_BLS12_381_G1: return msm_run_BLS12_381_G1(ws, b, s, n, mont, out, st, tm);
#ifndef ARK_HIP_DEV
    case ARK_HIP_BLS12_377_G1: return msm_run_BLS12_377_G1(ws, b, s, n, mont, out, st, tm);
#endif
#ifndef ARK_HIP_DEV
    case ARK_HIP_BLS12_377_G2: return msm_run_BLS12_377_G2(ws, b, s, n, mont, out, st, tm);
#endif
#ifndef ARK_HIP_DEV
    case ARK_HIP_BLS12_381_G2: return msm_run_BLS12_381_G2(ws, b, s, n, mont, out, st, tm);
#endif
  }
  return ARK_HIP_ERR_ARG;
}

int fft_dispatch(int field, FftWorkspace& ws, void* d, int k, const uint64_t* root, const uint64_t* pre,
                 const uint64_t* post, const uint64_t* postc, hipStream_t st, FftTimings* tm) {
  switch (field) {
#ifndef ARK_HIP_DEV
    case ARK_HIP_BN254_FR: return fft_run_BN254_FR(ws, d, k, root, pre, post, postc, st, tm);
#endif
    case ARK_HIP_BLS12_381_FR: return fft_run_BLS12_381_FR(ws, d, k, root, pre, post, postc, st, tm);
#ifndef ARK_HIP_DEV
    case ARK_HIP_BLS12_377_FR: return fft_run_BLS12_377_FR(ws, d, k, root, pre, post, postc, st, tm);
#endif
  }
  return ARK_HIP_ERR_ARG;
}

// ---- host-side scalar-field arithmetic for the domain constants (same templates as the device) ----
template <class FP>
Fp<FP> host_pow(Fp<FP> b, const uint64_t* e, int words) {
  Fp<FP> r = Fp<FP>::one();
  for (int i = words * 64 - 1; i >= 0; i--) {
    r = Fp<FP>::sqr(r);
    if ((e[i / 64] >> (i % 64)) & 1) r = Fp<FP>::mul(r, b);
  }
  return r;
}
template <class F>
F host_inverse(const F& a) { return F::inverse(a); }

// Projective (Jacobian) -> XYZZ: (X, Y, Z^2, Z^3)
template <class F>
XYZZ<F> jac_to_xyzz(const uint64_t* p) {
  F x = F::load(p), y = F::load((const char*)p + F::BYTES), z = F::load((const char*)p + 2 * F::BYTES);
  if (z.is_zero()) return XYZZ<F>::zero();
  F zz = F::sqr(z);
  return XYZZ<F>{x, y, zz, F::mul(zz, z)};
}
// sum of n Jacobian points on the host (the multi-GPU combine: one partial per rank)
template <class C>
int host_sum(const uint64_t* pts, size_t n, uint64_t* out) {
  typedef typename C::F F;
  XYZZ<F> acc = XYZZ<F>::zero();
  for (size_t i = 0; i < n; i++) {
    XYZZ<F> p = jac_to_xyzz<F>(pts + i * 3 * F::WORDS64);
    xyzz_add<F>(acc, p);
  }
  xyzz_to_jac<F>(acc).store(out);
  return 0;
}
// From<Projective> for Affine (short_weierstrass/affine.rs:374-396): (x/z^2, y/z^3); identity -> (0,0)
template <class C>
int host_into_affine(const uint64_t* pts, size_t n, uint64_t* out) {
  typedef typename C::F F;
  for (size_t i = 0; i < n; i++) {
    const uint64_t* p = pts + i * 3 * F::WORDS64;
    uint64_t* o = out + i * 2 * F::WORDS64;
    F x = F::load(p), y = F::load((const char*)p + F::BYTES), z = F::load((const char*)p + 2 * F::BYTES);
    if (z.is_zero()) {
      F::zero().store(o);
      F::zero().store((char*)o + F::BYTES);
      continue;
    }
    F zi = host_inverse(z);
    F zi2 = F::sqr(zi);
    F::mul(x, zi2).store(o);
    F::mul(y, F::mul(zi2, zi)).store((char*)o + F::BYTES);
  }
  return 0;
}
template <class FP>
bool host_is_one(const uint64_t* x) {
  Fp<FP> a = Fp<FP>::load(x);
  return Fp<FP>::eq(a, Fp<FP>::one());
}

template <class FP>
int domain_new(size_t num_coeffs, ark_hip_radix2_domain* out) {
  typedef Fp<FP> F;
  // usize::next_power_of_two: 0 -> 1
  uint64_t size = 1;
  while (size < num_coeffs) {
    size <<= 1;
    if (size == 0) return ARK_HIP_ERR_SIZE;
  }
  uint32_t lg = 0;
  while (((uint64_t)1 << lg) < size) lg++;
  if ((int)lg > FP::TWO_ADICITY) return ARK_HIP_ERR_SIZE;  // radix2/mod.rs:62-64 -> None
  memset(out, 0, sizeof(*out));
  out->size = size;
  out->log_size_of_group = lg;
  // get_root_of_unity (ff/src/fields/fft_friendly.rs:35-84): TWO_ADIC_ROOT squared (adicity - lg) times
  F g;
  for (int i = 0; i < F::N; i++) g.l[i] = FP::ROOT[i];
  for (int i = (int)lg; i < FP::TWO_ADICITY; i++) g = F::sqr(g);
  // F::from(size): canonical integer -> Montgomery
  F sz = F::zero();
  sz.l[0] = (uint32_t)size;
  sz.l[1] = (uint32_t)(size >> 32);
  sz = F::to_mont(sz);
  F one = F::one();
  sz.store(out->size_as_field_element);
  host_inverse(sz).store(out->size_inv);
  g.store(out->group_gen);
  host_inverse(g).store(out->group_gen_inv);
  one.store(out->offset);
  one.store(out->offset_inv);
  one.store(out->offset_pow_size);
  return 0;
}
template <class FP>
int domain_coset(const ark_hip_radix2_domain* dom, const uint64_t* offset, ark_hip_radix2_domain* out) {
  typedef Fp<FP> F;
  F h = F::load(offset);
  if (h.is_zero()) return ARK_HIP_ERR_ARG;  // inverse() -> None
  ark_hip_radix2_domain d = *dom;
  h.store(d.offset);
  host_inverse(h).store(d.offset_inv);
  uint64_t e[1] = {dom->size};
  host_pow<FP>(h, e, 1).store(d.offset_pow_size);
  *out = d;
  return 0;
}

template <class FP>
int fft_entry(Context* c, const ark_hip_radix2_domain* dom, void* d_data, int inverse) {
  const bool coset = !host_is_one<FP>(dom->offset);
  FftTimings* tm = c->fft_timing ? &c->fft_tm : nullptr;
  int k = (int)dom->log_size_of_group;
  if (dom->size != ((uint64_t)1 << k)) return ARK_HIP_ERR_ARG;
  if (!inverse) {
    // fft.rs:74-79: distribute_powers(offset) then DIF + derange
    return fft_dispatch(FP::ID, c->fft, d_data, k, dom->group_gen, coset ? dom->offset : nullptr, nullptr, nullptr,
                        c->stream, tm);
  }
  // fft.rs:81-88: transform with group_gen_inv, then x[i] *= size_inv * offset_inv^i
  return fft_dispatch(FP::ID, c->fft, d_data, k, dom->group_gen_inv, nullptr, coset ? dom->offset_inv : nullptr,
                      dom->size_inv, c->stream, tm);
}

int fft_any(int field, const ark_hip_radix2_domain* dom, void* d_data, int inverse) {
  if (!dom || !d_data) return ARK_HIP_ERR_ARG;
  int rc = ensure_ctx();
  if (rc) return rc;
  switch (field) {
    case ARK_HIP_BN254_FR: return fft_entry<BN254_FR>(g_ctx, dom, d_data, inverse);
    case ARK_HIP_BLS12_381_FR: return fft_entry<BLS12_381_FR>(g_ctx, dom, d_data, inverse);
    case ARK_HIP_BLS12_377_FR: return fft_entry<BLS12_377_FR>(g_ctx, dom, d_data, inverse);
  }
  return ARK_HIP_ERR_ARG;
}

int fft_host(int field, const ark_hip_radix2_domain* dom, uint64_t* data, int inverse) {
  if (!dom || !data) return ARK_HIP_ERR_ARG;
  int rc = ensure_ctx();
  if (rc) return rc;
  Context* c = g_ctx;
  size_t bytes = (size_t)dom->size * 32;
  if (c->stage_a.ensure(bytes)) return ARK_HIP_ERR_NOMEM;
  ARK_HIP_TRY(hipMemcpyAsync(c->stage_a.p, data, bytes, hipMemcpyHostToDevice, c->stream));
  rc = fft_any(field, dom, c->stage_a.p, inverse);
  if (rc) return rc;
  ARK_HIP_TRY(hipMemcpyAsync(data, c->stage_a.p, bytes, hipMemcpyDeviceToHost, c->stream));
  ARK_HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

size_t field_bytes(int field) { return (field == ARK_HIP_BLS12_381_FQ || field == ARK_HIP_BLS12_377_FQ) ? 48 : 32; }

}  // namespace

extern "C" {

int ark_hip_device_count(void) {
  int cnt = 0;
  if (hipGetDeviceCount(&cnt) != hipSuccess) return 0;
  return cnt;
}

int ark_hip_init(int device) {
  {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_ctx) return g_ctx->device == device ? 0 : ARK_HIP_ERR_ARG;
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0) {
      fprintf(stderr, "ark_hip: no HIP device visible -- this library has no CPU fallback\n");
      return ARK_HIP_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= cnt) return ARK_HIP_ERR_ARG;
    if (hipSetDevice(device) != hipSuccess) return ARK_HIP_ERR_NO_DEVICE;
  }
  return ensure_ctx();
}

void ark_hip_shutdown(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_ctx) return;
  (void)hipStreamSynchronize(g_ctx->stream);
  g_ctx->msm.release();
  g_ctx->fft.release();
  g_ctx->stage_a.release();
  g_ctx->stage_b.release();
  g_ctx->stage_c.release();
  (void)hipStreamDestroy(g_ctx->stream);
  delete g_ctx;
  g_ctx = nullptr;
}

int ark_hip_synchronize(void) {
  if (!g_ctx) return 0;
  ARK_HIP_TRY(hipStreamSynchronize(g_ctx->stream));
  return 0;
}

const char* ark_hip_version(void) { return "ark_hip 0.1 (gfx950)"; }

int ark_hip_curve_info(int curve, int* fe_words, int* scalar_field, int* base_field, int* ext_degree) {
  if (curve < 0 || curve > 4) return ARK_HIP_ERR_ARG;
  if (fe_words) *fe_words = CURVES[curve].fe_words;
  if (scalar_field) *scalar_field = CURVES[curve].scalar_field;
  if (base_field) *base_field = CURVES[curve].base_field;
  if (ext_degree) *ext_degree = CURVES[curve].ext;
  return 0;
}

// ---- device memory for hosts without their own HIP binding (the Rust shim keeps an SRS resident this way) ----
int ark_hip_malloc(size_t bytes, void** out_dptr) {
  if (!out_dptr) return ARK_HIP_ERR_ARG;
  int rc = ensure_ctx();
  if (rc) return rc;
  *out_dptr = nullptr;
  if (bytes == 0) return 0;
  if (hipMalloc(out_dptr, bytes) != hipSuccess) return ARK_HIP_ERR_NOMEM;
  return 0;
}
int ark_hip_free(void* dptr) {
  if (!dptr) return 0;
  int rc = ensure_ctx();
  if (rc) return rc;
  ARK_HIP_TRY(hipStreamSynchronize(g_ctx->stream));
  ARK_HIP_TRY(hipFree(dptr));
  return 0;
}
int ark_hip_memcpy_h2d(void* dst_dptr, const void* src_host, size_t bytes) {
  if (bytes && (!dst_dptr || !src_host)) return ARK_HIP_ERR_ARG;
  int rc = ensure_ctx();
  if (rc) return rc;
  ARK_HIP_TRY(hipMemcpyAsync(dst_dptr, src_host, bytes, hipMemcpyHostToDevice, g_ctx->stream));
  ARK_HIP_TRY(hipStreamSynchronize(g_ctx->stream));
  return 0;
}
int ark_hip_memcpy_d2h(void* dst_host, const void* src_dptr, size_t bytes) {
  if (bytes && (!dst_host || !src_dptr)) return ARK_HIP_ERR_ARG;
  int rc = ensure_ctx();
  if (rc) return rc;
  ARK_HIP_TRY(hipMemcpyAsync(dst_host, src_dptr, bytes, hipMemcpyDeviceToHost, g_ctx->stream));
  ARK_HIP_TRY(hipStreamSynchronize(g_ctx->stream));
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

int ark_hip_msm_sw_device(int curve, const void* d_bases, const void* d_scalars, size_t n, int mont, uint64_t* out_xyz) {
  if (curve < 0 || curve > 4 || !out_xyz || (n && (!d_bases || !d_scalars))) return ARK_HIP_ERR_ARG;
  int rc = ensure_ctx();
  if (rc) return rc;
  Context* c = g_ctx;
  MsmTimings* tm = c->msm_timing ? &c->msm_tm : nullptr;
  rc = msm_dispatch(curve, c->msm, d_bases, d_scalars, n, mont, out_xyz, c->stream, tm);
  if (rc == 0 && tm) {
    MsmPlan pl = msm_make_plan(n ? n : 1, msm_scalar_bits(curve), msm_mul_cost(curve));
    c->msm_c = pl.c;
    c->msm_W = pl.W;
  }
  return rc;
}

int ark_hip_msm_sw(int curve, const uint64_t* bases, const uint64_t* scalars, size_t n, int mont, uint64_t* out_xyz) {
  if (curve < 0 || curve > 4 || !out_xyz || (n && (!bases || !scalars))) return ARK_HIP_ERR_ARG;
  int rc = ensure_ctx();
  if (rc) return rc;
  Context* c = g_ctx;
  size_t bb = n * (size_t)CURVES[curve].fe_words * 2 * 8, sb = n * 32;
  if (n) {
    if (c->stage_a.ensure(bb) || c->stage_b.ensure(sb)) return ARK_HIP_ERR_NOMEM;
    ARK_HIP_TRY(hipMemcpyAsync(c->stage_a.p, bases, bb, hipMemcpyHostToDevice, c->stream));
    ARK_HIP_TRY(hipMemcpyAsync(c->stage_b.p, scalars, sb, hipMemcpyHostToDevice, c->stream));
  }
  return ark_hip_msm_sw_device(curve, c->stage_a.p, c->stage_b.p, n, mont, out_xyz);
}

int ark_hip_msm_set_timing(int enable) {
  int rc = ensure_ctx();
  if (rc) return rc;
  g_ctx->msm_timing = enable != 0;
  return 0;
}
int ark_hip_msm_last_timing(double out[8]) {
  if (!g_ctx || !out) return ARK_HIP_ERR_ARG;
  const MsmTimings& t = g_ctx->msm_tm;
  out[0] = t.digits; out[1] = t.scan; out[2] = t.scatter; out[3] = t.accumulate; out[4] = t.reduce; out[5] = t.total;
  out[6] = g_ctx->msm_c; out[7] = g_ctx->msm_W;
  return 0;
}

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

int ark_hip_fft_in_place(int field, const ark_hip_radix2_domain* dom, uint64_t* data) { return fft_host(field, dom, data, 0); }
int ark_hip_ifft_in_place(int field, const ark_hip_radix2_domain* dom, uint64_t* data) { return fft_host(field, dom, data, 1); }
int ark_hip_fft_in_place_device(int field, const ark_hip_radix2_domain* dom, void* d) { return fft_any(field, dom, d, 0); }
int ark_hip_ifft_in_place_device(int field, const ark_hip_radix2_domain* dom, void* d) { return fft_any(field, dom, d, 1); }

// r[i] = a[i] * b[i] over n Fr elements in device memory (Evaluations *= Evaluations,
// poly/src/evaluations/univariate/mod.rs MulAssign; the middle step of DensePolynomial multiplication,
// poly/src/polynomial/univariate/dense.rs:641-656).  Asynchronous on the context stream.
int ark_hip_fr_mul_device(int field, const void* d_a, const void* d_b, void* d_r, size_t n) {
  if (n && (!d_a || !d_b || !d_r)) return ARK_HIP_ERR_ARG;
  int rc = ensure_ctx();
  if (rc) return rc;
  switch (field) {
#ifndef ARK_HIP_DEV
    case ARK_HIP_BN254_FR: return test_field_op_BN254_FR(2, d_a, d_b, d_r, n, g_ctx->stream);
    case ARK_HIP_BLS12_377_FR: return test_field_op_BLS12_377_FR(2, d_a, d_b, d_r, n, g_ctx->stream);
#endif
    case ARK_HIP_BLS12_381_FR: return test_field_op_BLS12_381_FR(2, d_a, d_b, d_r, n, g_ctx->stream);
  }
  return ARK_HIP_ERR_ARG;
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
  int rc = ensure_ctx();
  if (rc) return rc;
  switch (field) {
#ifndef ARK_HIP_DEV
    case ARK_HIP_BN254_FR: return fft_axis_BN254_FR(g_ctx->fft, d_data, G, cols, root, g_ctx->stream);
    case ARK_HIP_BLS12_377_FR: return fft_axis_BLS12_377_FR(g_ctx->fft, d_data, G, cols, root, g_ctx->stream);
#endif
    case ARK_HIP_BLS12_381_FR: return fft_axis_BLS12_381_FR(g_ctx->fft, d_data, G, cols, root, g_ctx->stream);
  }
  return ARK_HIP_ERR_ARG;
}

int ark_hip_fft_set_timing(int enable) {
  int rc = ensure_ctx();
  if (rc) return rc;
  g_ctx->fft_timing = enable != 0;
  return 0;
}
int ark_hip_fft_last_timing(double out[10]) {
  if (!g_ctx || !out) return ARK_HIP_ERR_ARG;
  out[0] = g_ctx->fft_tm.total;
  out[1] = g_ctx->fft_tm.npass;
  for (int i = 0; i < 8; i++) out[2 + i] = g_ctx->fft_tm.pass[i];
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
  int rc = ensure_ctx();
  if (rc) return rc;
  Context* c = g_ctx;
  size_t ab = (size_t)CURVES[curve].fe_words * 16;
  if (c->stage_c.ensure(ab)) return ARK_HIP_ERR_NOMEM;
  ARK_HIP_TRY(hipMemcpyAsync(c->stage_c.p, delta_xy, ab, hipMemcpyHostToDevice, c->stream));
  switch (curve) {
#ifndef ARK_HIP_DEV
    case 0: rc = sw_add_affine_BN254_G1(d_in, d_out, n, c->stage_c.p, c->stream); break;
#endif
    case 1: rc = sw_add_affine_BLS12_381_G1(d_in, d_out, n, c->stage_c.p, c->stream); break;
#ifndef ARK_HIP_DEV
    case 2: rc = sw_add_affine_BLS12_377_G1(d_in, d_out, n, c->stage_c.p, c->stream); break;
#endif
#ifndef ARK_HIP_DEV
    case 3: rc = sw_add_affine_BLS12_377_G2(d_in, d_out, n, c->stage_c.p, c->stream); break;
#endif
#ifndef ARK_HIP_DEV
    case 4: rc = sw_add_affine_BLS12_381_G2(d_in, d_out, n, c->stage_c.p, c->stream); break;
#endif
  }
  if (rc) return rc;
  ARK_HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

// CurveGroup::normalize_batch for n Projective points in device memory -> n Affine points (device memory)
int ark_hip_sw_normalize_batch_device(int curve, const void* d_jac, void* d_out_xy, size_t n) {
  if (curve < 0 || curve > 4 || (n && (!d_jac || !d_out_xy))) return ARK_HIP_ERR_ARG;
  int rc = ensure_ctx();
  if (rc) return rc;
  Context* c = g_ctx;
  switch (curve) {
#ifndef ARK_HIP_DEV
    case 0: rc = sw_normalize_batch_BN254_G1(d_jac, d_out_xy, n, c->stream); break;
    case 2: rc = sw_normalize_batch_BLS12_377_G1(d_jac, d_out_xy, n, c->stream); break;
    case 3: rc = sw_normalize_batch_BLS12_377_G2(d_jac, d_out_xy, n, c->stream); break;
    case 4: rc = sw_normalize_batch_BLS12_381_G2(d_jac, d_out_xy, n, c->stream); break;
#endif
    case 1: rc = sw_normalize_batch_BLS12_381_G1(d_jac, d_out_xy, n, c->stream); break;
    default: return ARK_HIP_ERR_ARG;
  }
  if (rc) return rc;
  ARK_HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

// ---- test hooks ----
static int run_elementwise(size_t abytes, size_t bbytes, size_t rbytes, const void* a, const void* b, void* r,
                           int (*fn)(int, const void*, const void*, void*, size_t, hipStream_t), int op, size_t n) {
  if (!fn) return ARK_HIP_ERR_ARG;  // curve/field not in this (development) build
  int rc = ensure_ctx();
  if (rc) return rc;
  Context* c = g_ctx;
  if (n == 0) return 0;
  if (c->stage_a.ensure(abytes) || c->stage_b.ensure(bbytes ? bbytes : 16) || c->stage_c.ensure(rbytes)) return ARK_HIP_ERR_NOMEM;
  ARK_HIP_TRY(hipMemcpyAsync(c->stage_a.p, a, abytes, hipMemcpyHostToDevice, c->stream));
  if (b) ARK_HIP_TRY(hipMemcpyAsync(c->stage_b.p, b, bbytes, hipMemcpyHostToDevice, c->stream));
  rc = fn(op, c->stage_a.p, b ? c->stage_b.p : nullptr, c->stage_c.p, n, c->stream);
  if (rc) return rc;
  ARK_HIP_TRY(hipMemcpyAsync(r, c->stage_c.p, rbytes, hipMemcpyDeviceToHost, c->stream));
  ARK_HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

int ark_hip_test_field_op(int field, int op, const uint64_t* a, const uint64_t* b, uint64_t* r, size_t n) {
  if (!a || !r || op < 0 || op > 8 || op == 6) return ARK_HIP_ERR_ARG;
  size_t fb = field_bytes(field);
  int (*fn)(int, const void*, const void*, void*, size_t, hipStream_t) = nullptr;
  switch (field) {
#ifndef ARK_HIP_DEV
    case ARK_HIP_BN254_FR: fn = test_field_op_BN254_FR; break;
#endif
    case ARK_HIP_BLS12_381_FR: fn = test_field_op_BLS12_381_FR; break;
#ifndef ARK_HIP_DEV
    case ARK_HIP_BLS12_377_FR: fn = test_field_op_BLS12_377_FR; break;
#endif
    // base fields: through the G1 curve that lives over them (ops 0..5)
#ifndef ARK_HIP_DEV
    case ARK_HIP_BN254_FQ: fn = test_basefield_op_BN254_G1; break;
#endif
    case ARK_HIP_BLS12_381_FQ: fn = test_basefield_op_BLS12_381_G1; break;
#ifndef ARK_HIP_DEV
    case ARK_HIP_BLS12_377_FQ: fn = test_basefield_op_BLS12_377_G1; break;
#endif
    default: return ARK_HIP_ERR_ARG;
  }
  if ((field == ARK_HIP_BN254_FQ || field == ARK_HIP_BLS12_381_FQ || field == ARK_HIP_BLS12_377_FQ) && op > 5)
    return ARK_HIP_ERR_ARG;
  return run_elementwise(n * fb, b ? n * fb : 0, n * fb, a, b, r, fn, op, n);
}

int ark_hip_test_basefield_op(int curve, int op, const uint64_t* a, const uint64_t* b, uint64_t* r, size_t n) {
  if (curve < 0 || curve > 4 || !a || !r || op < 0 || op > 5) return ARK_HIP_ERR_ARG;
  size_t fb = (size_t)CURVES[curve].fe_words * 8;
  int (*fn)(int, const void*, const void*, void*, size_t, hipStream_t) = nullptr;
  switch (curve) {
#ifndef ARK_HIP_DEV
    case 0: fn = test_basefield_op_BN254_G1; break;
#endif
    case 1: fn = test_basefield_op_BLS12_381_G1; break;
#ifndef ARK_HIP_DEV
    case 2: fn = test_basefield_op_BLS12_377_G1; break;
#endif
#ifndef ARK_HIP_DEV
    case 3: fn = test_basefield_op_BLS12_377_G2; break;
#endif
#ifndef ARK_HIP_DEV
    case 4: fn = test_basefield_op_BLS12_381_G2; break;
#endif
  }
  return run_elementwise(n * fb, b ? n * fb : 0, n * fb, a, b, r, fn, op, n);
}

int ark_hip_test_point_op(int curve, int kind, const uint64_t* acc, const uint64_t* other, uint64_t* out, size_t n) {
  if (curve < 0 || curve > 4 || !acc || !out || kind < 2 || kind > 7) return ARK_HIP_ERR_ARG;
  size_t fb = (size_t)CURVES[curve].fe_words * 8;
  size_t abytes = n * fb * (kind == 7 ? 2 : 4);
  size_t bbytes = (kind == 2 || kind == 3) ? n * fb * 2 : (kind == 4 ? n * fb * 4 : 0);
  size_t rbytes = n * fb * (kind == 6 ? 3 : 4);
  if (bbytes && !other) return ARK_HIP_ERR_ARG;
  int (*fn)(int, const void*, const void*, void*, size_t, hipStream_t) = nullptr;
  switch (curve) {
#ifndef ARK_HIP_DEV
    case 0: fn = test_point_op_BN254_G1; break;
#endif
    case 1: fn = test_point_op_BLS12_381_G1; break;
#ifndef ARK_HIP_DEV
    case 2: fn = test_point_op_BLS12_377_G1; break;
#endif
#ifndef ARK_HIP_DEV
    case 3: fn = test_point_op_BLS12_377_G2; break;
#endif
#ifndef ARK_HIP_DEV
    case 4: fn = test_point_op_BLS12_381_G2; break;
#endif
  }
  return run_elementwise(abytes, bbytes, rbytes, acc, bbytes ? other : nullptr, out, fn, kind, n);
}

}  // extern "C"
