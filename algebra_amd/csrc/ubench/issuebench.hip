// Issue rates of the candidate multiplier / accumulator instructions on gfx950, same loop shape for each: four
// independent dependency chains per lane, 32 instructions per asm statement.  Answers "is there a faster primitive than
// v_mad_u64_u32 for multi-limb products" (DESIGN.md 6): 24-bit multiplies, FP64 fused multiply-adds (the 52-bit-limb
// technique), 64-bit adds, against the plain 32-bit add as the full-rate yardstick.
//   hipcc -O3 --offload-arch=gfx950 issuebench.hip -o issuebench.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef uint32_t u32;
typedef uint64_t u64;

#define R8(X) X X X X X X X X
#define KERNEL32(NAME, I0, I1, I2, I3)                                                                        \
  __global__ void __launch_bounds__(256) NAME(u64* out, int iters) {                                         \
    u32 tid = blockIdx.x * blockDim.x + threadIdx.x;                                                          \
    u32 a0 = tid, a1 = tid + 1, a2 = tid + 2, a3 = tid + 3;                                                   \
    const u32 x = tid * 2654435761u | 0x80000001u, y = (tid * 40503u + 7u) | 0x80000000u;                     \
    for (int k = 0; k < iters; k++)                                                                           \
      asm volatile(R8(I0 "\n\t" I1 "\n\t" I2 "\n\t" I3 "\n\t") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "v"(y) : "vcc"); \
    out[tid] = a0 ^ a1 ^ a2 ^ a3;                                                                             \
  }
#define KERNEL64(NAME, T, TX, I0, I1, I2, I3)                                                                     \
  __global__ void __launch_bounds__(256) NAME(u64* out, int iters) {                                         \
    u32 tid = blockIdx.x * blockDim.x + threadIdx.x;                                                          \
    T a0 = (T)tid, a1 = (T)(tid + 1), a2 = (T)(tid + 2), a3 = (T)(tid + 3);                                   \
    const TX x = (TX)(tid | 1u), y = (TX)((tid * 40503u + 7u) | 0x80000000u);                                    \
    for (int k = 0; k < iters; k++)                                                                           \
      asm volatile(R8(I0 "\n\t" I1 "\n\t" I2 "\n\t" I3 "\n\t") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "v"(y) : "vcc"); \
    T s = a0 + a1 + a2 + a3;                                                                                  \
    out[tid] = *(u64*)&s;                                                                                     \
  }
KERNEL32(k_add32, "v_add_u32 %0, %4, %0", "v_add_u32 %1, %4, %1", "v_add_u32 %2, %4, %2", "v_add_u32 %3, %4, %3")
KERNEL32(k_mad24, "v_mad_u32_u24 %0, %4, %5, %0", "v_mad_u32_u24 %1, %4, %5, %1", "v_mad_u32_u24 %2, %4, %5, %2", "v_mad_u32_u24 %3, %4, %5, %3")
KERNEL32(k_mulhi24, "v_mul_hi_u32_u24 %0, %4, %0", "v_mul_hi_u32_u24 %1, %4, %1", "v_mul_hi_u32_u24 %2, %4, %2", "v_mul_hi_u32_u24 %3, %4, %3")
KERNEL32(k_mullo32, "v_mul_lo_u32 %0, %4, %0", "v_mul_lo_u32 %1, %4, %1", "v_mul_lo_u32 %2, %4, %2", "v_mul_lo_u32 %3, %4, %3")
KERNEL32(k_mulhi32, "v_mul_hi_u32 %0, %4, %0", "v_mul_hi_u32 %1, %4, %1", "v_mul_hi_u32 %2, %4, %2", "v_mul_hi_u32 %3, %4, %3")
KERNEL32(k_addc, "v_add_co_u32 %0, vcc, %4, %0", "v_addc_co_u32 %1, vcc, %5, %1, vcc", "v_add_co_u32 %2, vcc, %4, %2", "v_addc_co_u32 %3, vcc, %5, %3, vcc")
KERNEL64(k_mad64, u64, u32, "v_mad_u64_u32 %0, vcc, %4, %5, %0", "v_mad_u64_u32 %1, vcc, %4, %5, %1", "v_mad_u64_u32 %2, vcc, %4, %5, %2", "v_mad_u64_u32 %3, vcc, %4, %5, %3")
KERNEL64(k_add64, u64, u64, "v_lshl_add_u64 %0, %4, 0, %0", "v_lshl_add_u64 %1, %4, 0, %1", "v_lshl_add_u64 %2, %4, 0, %2", "v_lshl_add_u64 %3, %4, 0, %3")
KERNEL64(k_fma64, double, double, "v_fma_f64 %0, %4, %5, %0", "v_fma_f64 %1, %4, %5, %1", "v_fma_f64 %2, %4, %5, %2", "v_fma_f64 %3, %4, %5, %3")
KERNEL64(k_mul64f, double, double, "v_mul_f64 %0, %4, %0", "v_mul_f64 %1, %4, %1", "v_mul_f64 %2, %4, %2", "v_mul_f64 %3, %4, %3")

typedef void (*kern_t)(u64*, int);
int main() {
  u64* out;
  if (hipMalloc(&out, 256 * 8 * 256 * 8) != hipSuccess) return 1;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  struct { const char* name; kern_t k; } ks[] = {
      {"v_add_u32            ", k_add32},  {"v_mad_u32_u24        ", k_mad24},   {"v_mul_hi_u32_u24     ", k_mulhi24},
      {"v_mul_lo_u32         ", k_mullo32}, {"v_mul_hi_u32         ", k_mulhi32}, {"v_add_co + v_addc_co ", k_addc},
      {"v_mad_u64_u32        ", k_mad64},  {"v_lshl_add_u64       ", k_add64},   {"v_fma_f64            ", k_fma64},
      {"v_mul_f64            ", k_mul64f}};
  for (auto& e : ks) {
    for (int w : {2, 8}) {
      const int b = 256 * w, it = 4000;
      hipLaunchKernelGGL(e.k, dim3(b), dim3(256), 0, 0, out, 10);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      hipLaunchKernelGGL(e.k, dim3(b), dim3(256), 0, 0, out, it);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double rate = (double)b * 256 * it * 32 / (ms * 1e-3);
      printf("%s waves/SIMD=%d  %8.3f ms  %7.2f T lane-ops/s  = %5.2f cycles per wave instruction at 2.4 GHz\n", e.name, w, ms,
             rate * 1e-12, 256.0 * 4 * 64 * 2.4e9 / rate);
    }
  }
  return 0;
}
