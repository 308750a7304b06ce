// C ABI of libark_hip.so, unit 5 of 5 -- TEST HOOKS, linked into libark_hip_test.so only (the shipped libark_hip.so exports no
// ark_hip_test_* symbol: VERDICT r5 next #8).  Device field / point arithmetic and the host tail, reachable from tests/.
#ifndef ARK_HIP_TEST_HOOKS
#define ARK_HIP_TEST_HOOKS 1
#endif
#include "capi_core.hpp"
#include "capi_hostmath.hpp"
#include "capi_cache.hpp"
using namespace arkhip;
using namespace arkhip::capi;

namespace arkhip {
namespace capi {
int msm_sharded_emulated(int curve, int world, const void* const* d_bases, const void* const* d_scalars, const size_t* n_local,
                         int scalars_are_montgomery, uint64_t* out_xyz, int* path);   // capi_comm.hip
}
}
namespace {
elementwise_fn field_op_fn(int field) {
  switch (field) {
#ifndef ARK_HIP_DEV
    case ARK_HIP_BN254_FR: return field_op_BN254_FR;
    case ARK_HIP_BLS12_377_FR: return field_op_BLS12_377_FR;
    case ARK_HIP_BN254_FQ: return test_basefield_op_BN254_G1;  // base fields: through the G1 curve over them
    case ARK_HIP_BLS12_377_FQ: return test_basefield_op_BLS12_377_G1;
#endif
    case ARK_HIP_BLS12_381_FR: return field_op_BLS12_381_FR;
    case ARK_HIP_BLS12_381_FQ: return test_basefield_op_BLS12_381_G1;
  }
  return nullptr;
}
elementwise_fn basefield_op_fn(int curve) {
  switch (curve) {
#ifndef ARK_HIP_DEV
    case 0: return test_basefield_op_BN254_G1;
    case 2: return test_basefield_op_BLS12_377_G1;
    case 3: return test_basefield_op_BLS12_377_G2;
    case 4: return test_basefield_op_BLS12_381_G2;
#endif
    case 1: return test_basefield_op_BLS12_381_G1;
  }
  return nullptr;
}
elementwise_fn point_op_fn(int curve) {
  switch (curve) {
#ifndef ARK_HIP_DEV
    case 0: return test_point_op_BN254_G1;
    case 2: return test_point_op_BLS12_377_G1;
    case 3: return test_point_op_BLS12_377_G2;
    case 4: return test_point_op_BLS12_381_G2;
#endif
    case 1: return test_point_op_BLS12_381_G1;
  }
  return nullptr;
}
}  // namespace

extern "C" {

// ---- test hooks ----
static int run_elementwise(size_t abytes, size_t bbytes, size_t rbytes, const void* a, const void* b, void* r,
                           elementwise_fn fn, int op, size_t n) {
  if (!fn) return ARK_HIP_ERR_ARG;  // curve/field not in this (development) build
  ARK_SCOPE(sc);
  Context* c = sc.c;
  if (n == 0) return 0;
  if (c->stage_a.ensure(abytes) || c->stage_b.ensure(bbytes ? bbytes : 16) || c->stage_c.ensure(rbytes)) return ARK_HIP_ERR_NOMEM;
  ARK_HIP_TRY(hipMemcpyAsync(c->stage_a.p, a, abytes, hipMemcpyHostToDevice, c->stream));
  if (b) ARK_HIP_TRY(hipMemcpyAsync(c->stage_b.p, b, bbytes, hipMemcpyHostToDevice, c->stream));
  int rc = fn(op, c->stage_a.p, b ? c->stage_b.p : nullptr, c->stage_c.p, n, c->stream);
  if (rc) return rc;
  ARK_HIP_TRY(hipMemcpyAsync(r, c->stage_c.p, rbytes, hipMemcpyDeviceToHost, c->stream));
  ARK_HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

int ark_hip_test_field_op(int field, int op, const uint64_t* a, const uint64_t* b, uint64_t* r, size_t n) {
  const bool lazy_op = op >= 20 && op <= 22;   // the 28-bit-limb device forms (testops.cuh)
  if (!a || !r || op < 0 || (op > 8 && !lazy_op) || op == 6) return ARK_HIP_ERR_ARG;
  size_t fb = field_bytes(field);
  if (field < 0 || field > 5) return ARK_HIP_ERR_ARG;
  if ((field == ARK_HIP_BN254_FQ || field == ARK_HIP_BLS12_381_FQ || field == ARK_HIP_BLS12_377_FQ) && op > 5 && !lazy_op)
    return ARK_HIP_ERR_ARG;
  return run_elementwise(n * fb, b ? n * fb : 0, n * fb, a, b, r, field_op_fn(field), op, n);
}

int ark_hip_test_basefield_op(int curve, int op, const uint64_t* a, const uint64_t* b, uint64_t* r, size_t n) {
  if (curve < 0 || curve > 4 || !a || !r || op < 0 || op > 5) return ARK_HIP_ERR_ARG;
  size_t fb = (size_t)CURVES[curve].fe_words * 8;
  return run_elementwise(n * fb, b ? n * fb : 0, n * fb, a, b, r, basefield_op_fn(curve), op, n);
}

int ark_hip_test_host_basefield_op(int curve, int op, const uint64_t* a, const uint64_t* b, uint64_t* r, size_t n) {
  if (curve < 0 || curve > 4 || !a || !r || op < 0 || op > 5 || (op <= 2 && !b)) return ARK_HIP_ERR_ARG;
  switch (curve) {
#ifndef ARK_HIP_DEV
    case ARK_HIP_BN254_G1: host_field_ops<BN254_G1::F>(op, a, b, r, n); return 0;
    case ARK_HIP_BLS12_377_G1: host_field_ops<BLS12_377_G1::F>(op, a, b, r, n); return 0;
    case ARK_HIP_BLS12_377_G2: host_field_ops<BLS12_377_G2::F>(op, a, b, r, n); return 0;
    case ARK_HIP_BLS12_381_G2: host_field_ops<BLS12_381_G2::F>(op, a, b, r, n); return 0;
#endif
    case ARK_HIP_BLS12_381_G1: host_field_ops<BLS12_381_G1::F>(op, a, b, r, n); return 0;
  }
  return ARK_HIP_ERR_ARG;
}

int ark_hip_test_point_op(int curve, int kind, const uint64_t* acc, const uint64_t* other, uint64_t* out, size_t n) {
  const bool lazy_kind = kind >= 12 && kind <= 15;   // the carry-free forms of kinds 2 / 3 / 4 (testops.cuh)
  if (curve < 0 || curve > 4 || !acc || !out || kind < 2 || (kind > 7 && !lazy_kind)) return ARK_HIP_ERR_ARG;
  size_t fb = (size_t)CURVES[curve].fe_words * 8;
  size_t abytes = n * fb * (kind == 7 ? 2 : 4);
  size_t bbytes = (kind == 2 || kind == 3 || kind == 12 || kind == 13) ? n * fb * 2 : ((kind == 4 || kind >= 14) ? n * fb * 4 : 0);
  size_t rbytes = n * fb * (kind == 6 ? 3 : 4);
  if (bbytes && !other) return ARK_HIP_ERR_ARG;
  return run_elementwise(abytes, bbytes, rbytes, acc, bbytes ? other : nullptr, out, point_op_fn(curve), kind, n);
}


int ark_hip_test_msm_host_fold(int curve, const uint64_t* parts, int windows, int nbits, int log2_l0, const int* widths,
                               uint64_t* out_xyz) {
  if (!parts || !widths || !out_xyz || windows < 1 || windows > 256 || nbits < 0 || nbits > 31 || log2_l0 < 0 || log2_l0 > 16)
    return ARK_HIP_ERR_ARG;
  switch (curve) {
    case 0: return host_fold<BN254_G1>(parts, windows, nbits, log2_l0, widths, out_xyz);
    case 1: return host_fold<BLS12_381_G1>(parts, windows, nbits, log2_l0, widths, out_xyz);
    case 2: return host_fold<BLS12_377_G1>(parts, windows, nbits, log2_l0, widths, out_xyz);
    case 3: return host_fold<BLS12_377_G2>(parts, windows, nbits, log2_l0, widths, out_xyz);
    case 4: return host_fold<BLS12_381_G2>(parts, windows, nbits, log2_l0, widths, out_xyz);
  }
  return ARK_HIP_ERR_ARG;
}
// Test hook (host only, no device): the verified cache's tag of `words` u64 words -- tests/test_capi_host.py checks that
// edits the round-4 hash could not see (a two-word edit built from its published constants) change it.
int ark_hip_test_base_hash(const uint64_t* p, size_t words, uint64_t out[2]) {
  if (!out || (words && !p)) return ARK_HIP_ERR_ARG;
  const Hash128 h = base_hash(p, words);
  out[0] = h.lo;
  out[1] = h.hi;
  return 0;
}
/* test hook: msm_sharded's exchange with the ranks emulated in one process (capi_comm.hip: msm_sharded_emulated) */
int ark_hip_test_msm_sharded_emulated(int curve, int world, const void* const* d_bases, const void* const* d_scalars,
                                      const size_t* n_local, int scalars_are_montgomery, uint64_t* out_xyz, int* path) {
  return arkhip::capi::msm_sharded_emulated(curve, world, d_bases, d_scalars, n_local, scalars_are_montgomery, out_xyz, path);
}
}  // extern "C"
