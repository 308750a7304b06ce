// The device-arithmetic TEST kernels of one curve (testops.cuh): compiled once per curve with -DARK_TEST_CURVE=<name> and linked
// into libark_hip_test.so only -- the shipped libark_hip.so contains neither these kernels nor their entry points.
#include "devops.cuh"
#include "testops.cuh"
#include "internal.hpp"
#ifndef ARK_TEST_CURVE
#error "compile with -DARK_TEST_CURVE=BLS12_381_G1 (or another curve of curves.cuh)"
#endif
#define ARK_CAT2(a, b) a##b
#define ARK_CAT(a, b) ARK_CAT2(a, b)
namespace arkhip {
int ARK_CAT(test_basefield_op_, ARK_TEST_CURVE)(int op, const void* a, const void* b, void* r, size_t n, hipStream_t s) {
  return field_op_launch<ARK_TEST_CURVE::F, false>(op, a, b, r, n, s);
}
int ARK_CAT(test_point_op_, ARK_TEST_CURVE)(int kind, const void* acc, const void* other, void* out, size_t n, hipStream_t s) {
  return test_point_op_launch<ARK_TEST_CURVE>(kind, acc, other, out, n, s);
}
}  // namespace arkhip
