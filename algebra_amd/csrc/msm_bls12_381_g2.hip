// Instantiation of the MSM pipeline, the fixed-base batch multiplication and the point-array kernels for BLS12_381_G2 (one TU per curve so
// the five heavy template expansions compile in parallel).
#include "msm.cuh"
#include "batchmul.cuh"
#include "devops.cuh"
#include "gfft.cuh"
#include "internal.hpp"
namespace arkhip {
int msm_enqueue_BLS12_381_G2(MsmWorkspace& ws, const void* d_points, size_t wstride, const MsmPlan* prepared, const void* d_scalars,
                   size_t n, int mont, hipStream_t stream, bool timing, int sbytes, int sbits, const MsmPiece* piece) {
  return msm_enqueue<BLS12_381_G2>(ws, d_points, wstride, prepared, d_scalars, n, mont, stream, timing, sbytes, sbits, piece);
}
int msm_finish_BLS12_381_G2(MsmWorkspace& ws, int slot, uint64_t* out_xyz, MsmTimings* tm) {
  return msm_finish<BLS12_381_G2>(ws, slot, out_xyz, tm);
}
int msm_sum_ranks_BLS12_381_G2(const void* d_blocks, int world, size_t block_bytes, uint32_t npairs, void* d_out, hipStream_t stream) {
  return msm_sum_ranks<BLS12_381_G2>(d_blocks, world, block_bytes, npairs, d_out, stream);
}
int msm_fold_sums_BLS12_381_G2(const MsmSumsHeader& h, const void* h_sums, uint64_t* out_xyz) { return msm_fold_sums<BLS12_381_G2>(h, h_sums, out_xyz); }
void msm_sample_widths_BLS12_381_G2(const void* h_scalars, size_t n, int mont, MsmWidths* out) {
  msm_sample_widths_host<typename BLS12_381_G2::S>(h_scalars, n, mont, out);
}
int msm_prepare_BLS12_381_G2(const void* d_bases, size_t n, const MsmPlan& pl, void* d_table, void* d_tmp, hipStream_t stream) {
  return msm_prepare_table<BLS12_381_G2>(d_bases, n, pl, d_table, d_tmp, stream);
}
int batchmul_build_BLS12_381_G2(const void* h_base_affine, int window, void* d_scratch, void* d_table, hipStream_t stream) {
  return batchmul_build<BLS12_381_G2>(h_base_affine, window, d_scratch, d_table, stream);
}
size_t batchmul_build_scratch_BLS12_381_G2(int window) { return batchmul_build_scratch<BLS12_381_G2>(window); }
int batchmul_run_BLS12_381_G2(const void* d_table, int window, const void* d_scalars, size_t n, int mont, void* d_tmp, void* d_out, hipStream_t s) {
  return batchmul_run<BLS12_381_G2>(d_table, window, d_scalars, n, mont, d_tmp, d_out, s);
}
int sw_add_affine_BLS12_381_G2(const void* in, void* out, size_t n, const void* d_delta, hipStream_t s) {
  return sw_add_affine_launch<BLS12_381_G2>(in, out, n, d_delta, s);
}
int sw_normalize_batch_BLS12_381_G2(const void* in, void* out, size_t n, hipStream_t s) {
  return sw_normalize_batch_launch<BLS12_381_G2>(in, out, n, s);
}
int gfft_run_BLS12_381_G2(void* d_jac, int k, const uint32_t* d_roots, const uint32_t* d_pre, const uint32_t* d_post, void* d_work, hipStream_t s) {
  return gfft_run<BLS12_381_G2>(d_jac, k, d_roots, d_pre, d_post, d_work, s);
}
size_t gfft_work_bytes_BLS12_381_G2(int k) { return gfft_work_bytes<BLS12_381_G2>(k); }
}  // namespace arkhip
