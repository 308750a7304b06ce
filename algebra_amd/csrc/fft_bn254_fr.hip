// Instantiation of the radix-2 FFT and the pointwise field kernels for BN254_FR.
#include "fft.cuh"
#include "devops.cuh"
#include "internal.hpp"
namespace arkhip {
int fft_run_BN254_FR(FftWorkspace& ws, void* d_data, int k, const uint64_t* root4, const uint64_t* pre4, const uint64_t* post4,
               const uint64_t* postc4, int zlog, hipStream_t stream, FftTimings* tm) {
  return fft_run_device<BN254_FR>(ws, d_data, k, root4, pre4, post4, postc4, zlog, stream, tm);
}
int field_op_BN254_FR(int op, const void* a, const void* b, void* r, size_t n, hipStream_t s) {
  return field_op_launch<Fp<BN254_FR>, true>(op, a, b, r, n, s);
}
int fft_roots_BN254_FR(FftWorkspace& ws, int k, const uint64_t* root4, hipStream_t s, const uint32_t** out) {
  return fft_roots_run<BN254_FR>(ws, k, root4, s, out);
}
int fft_scalars_BN254_FR(FftWorkspace& ws, const uint64_t* base4, const uint64_t* mul4, size_t count, void* d_out, hipStream_t s) {
  return fft_scalars_run<BN254_FR>(ws, base4, mul4, count, d_out, s);
}
int fr_div_BN254_FR(const void* num, const void* den, void* r, size_t n, hipStream_t s) {
  return fr_div_launch<Fp<BN254_FR>>(num, den, r, n, s);
}
int fr_scale_BN254_FR(const void* a, const uint64_t* k4, void* r, size_t n, hipStream_t s) {
  return fr_scale_launch<Fp<BN254_FR>>(a, k4, r, n, s);
}
int fft_axis_BN254_FR(FftWorkspace& ws, const void* d_src, void* d_dst, unsigned G, size_t cols, const uint64_t* root4, hipStream_t s) {
  return fft_axis_run<BN254_FR>(ws, d_src, d_dst, G, cols, root4, s);
}
int fft_axis_prepare_BN254_FR(FftWorkspace& ws, unsigned G, const uint64_t* root4, hipStream_t s, const uint32_t** pw) {
  std::lock_guard<std::mutex> lock(ws.mu);
  return fft_axis_prepare<BN254_FR>(ws, G, root4, s, pw);
}
int fft_axis_launch_BN254_FR(const void* d_src, void* d_dst, unsigned G, size_t stride, size_t cols, const uint32_t* pw, hipStream_t s) {
  return fft_axis_launch<BN254_FR>(d_src, d_dst, G, stride, cols, pw, s);
}
}  // namespace arkhip
