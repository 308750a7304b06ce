// Internal glue between the per-curve / per-field translation units and the C ABI (capi.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace arkhip {
struct MsmWorkspace;
struct MsmTimings;
struct MsmPlan;
struct MsmPiece;
struct MsmSumsHeader;
struct MsmWidths;
struct FftWorkspace;
struct FftTimings;

// one function per curve / field, defined in msm_<curve>.hip / fft_<field>.hip
#define ARK_DECL_CURVE(NAME)                                                                                    \
  int msm_enqueue_##NAME(MsmWorkspace& ws, const void* d_points, size_t wstride, const MsmPlan* prepared,         \
                         const void* d_scalars, size_t n, int mont, hipStream_t stream, bool timing, int sbytes,  \
                         int sbits, const MsmPiece* piece);                                                       \
  int msm_finish_##NAME(MsmWorkspace& ws, int slot, uint64_t* out_xyz, MsmTimings* tm);                          \
  int msm_sum_ranks_##NAME(const void* d_blocks, int world, size_t block_bytes, uint32_t npairs, void* d_out, hipStream_t stream); \
  int msm_fold_sums_##NAME(const MsmSumsHeader& h, const void* h_sums, uint64_t* out_xyz);                         \
  void msm_sample_widths_##NAME(const void* h_scalars, size_t n, int mont, MsmWidths* out);                        \
  int msm_prepare_##NAME(const void* d_bases, size_t n, const MsmPlan& pl, void* d_table, void* d_tmp, hipStream_t stream);   \
  int batchmul_build_##NAME(const void* h_base_affine, int window, void* d_scratch, void* d_table, hipStream_t stream); \
  size_t batchmul_build_scratch_##NAME(int window);                                                               \
  int batchmul_run_##NAME(const void* d_table, int window, const void* d_scalars, size_t n, int mont, void* d_tmp, void* d_out, hipStream_t s); \
  int test_basefield_op_##NAME(int op, const void* d_a, const void* d_b, void* d_r, size_t n, hipStream_t s);    \
  int test_point_op_##NAME(int kind, const void* d_acc, const void* d_other, void* d_out, size_t n, hipStream_t s); \
  int sw_add_affine_##NAME(const void* d_in, void* d_out, size_t n, const void* d_delta, hipStream_t s);        \
  int sw_normalize_batch_##NAME(const void* d_in, void* d_out, size_t n, hipStream_t s);                          \
  int gfft_run_##NAME(void* d_jac, int k, const uint32_t* d_roots, const uint32_t* d_pre, const uint32_t* d_post, \
                      void* d_work, hipStream_t s);                                                                 \
  size_t gfft_work_bytes_##NAME(int k);
ARK_DECL_CURVE(BN254_G1)
ARK_DECL_CURVE(BLS12_381_G1)
ARK_DECL_CURVE(BLS12_377_G1)
ARK_DECL_CURVE(BLS12_377_G2)
ARK_DECL_CURVE(BLS12_381_G2)
#undef ARK_DECL_CURVE

#define ARK_DECL_FIELD(NAME)                                                                                    \
  int fft_run_##NAME(FftWorkspace& ws, void* d_data, int k, const uint64_t* root4, const uint64_t* pre4,         \
                     const uint64_t* post4, const uint64_t* postc4, int zlog, hipStream_t stream, FftTimings* tm);          \
  int field_op_##NAME(int op, const void* d_a, const void* d_b, void* d_r, size_t n, hipStream_t s);          \
  int fr_scale_##NAME(const void* d_a, const uint64_t* k4, void* d_r, size_t n, hipStream_t s);                     \
  int fr_div_##NAME(const void* d_num, const void* d_den, void* d_r, size_t n, hipStream_t s);                      \
  int fft_roots_##NAME(FftWorkspace& ws, int k, const uint64_t* root4, hipStream_t s, const uint32_t** out);        \
  int fft_scalars_##NAME(FftWorkspace& ws, const uint64_t* base4, const uint64_t* mul4, size_t count, void* d_out,  \
                         hipStream_t s);                                                                            \
  int fft_axis_##NAME(FftWorkspace& ws, const void* d_src, void* d_dst, unsigned G, size_t cols, const uint64_t* root4, \
                      hipStream_t s);                                                                               \
  int fft_axis_prepare_##NAME(FftWorkspace& ws, unsigned G, const uint64_t* root4, hipStream_t s, const uint32_t** pw); \
  int fft_axis_launch_##NAME(const void* d_src, void* d_dst, unsigned G, size_t stride, size_t cols, const uint32_t* pw, \
                             hipStream_t s);
ARK_DECL_FIELD(BN254_FR)
ARK_DECL_FIELD(BLS12_381_FR)
ARK_DECL_FIELD(BLS12_377_FR)
#undef ARK_DECL_FIELD
}  // namespace arkhip
