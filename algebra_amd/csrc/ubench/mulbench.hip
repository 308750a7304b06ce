// Product-rate microbenchmark of the two multiplications the kernels can run on: fp.cuh's saturated 32-bit Comba
// product against fp28.cuh's carry-free forms (14 x 28 bits for Fp384, 9 x 29 bits for Fp254/255: compiler-scheduled
// and asm columns, dedicated square, sum of two products), back to back in a loop at 1 / 2 / 4 / 8 waves per SIMD.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I.. mulbench.hip -o mulbench.bin            (BLS12-381 Fq)
//   hipcc ... -DMULBENCH_FIELD=BLS12_381_FR mulbench.hip -o mulbench_fr.bin                  (9 x 29 bits)
#include "../fp28.cuh"
#include <stdio.h>
using namespace arkhip;
#ifndef MULBENCH_FIELD
#define MULBENCH_FIELD BLS12_381_FQ
#endif
typedef MULBENCH_FIELD P;
template <int V> __global__ void __launch_bounds__(256) k_lazy(u32* out, int iters) {
  typedef FpL<P> L;
  L a, b;
  u32 tid = blockIdx.x * blockDim.x + threadIdx.x;
  for (int i = 0; i < L::L; i++) { a.l[i] = (tid * 2654435761u * (i + 1)) & L::MASK; b.l[i] = ((tid ^ 77) * 40503u * (i + 3)) & L::MASK; }
  a.l[L::L - 1] &= 0xfff; b.l[L::L - 1] &= 0xfff;
  for (int k = 0; k < iters; k++) {
    if constexpr (V == 0) a = L::mul_c(a, b);
    else if constexpr (V == 1) a = L::mul(a, b);
    else if constexpr (V == 2) a = L::sqr(a);
    else if constexpr (V == 3) a = L::sop2(a, b, b, a);
    else a = L::sqr_c(a);
  }
  u32 r = 0;
  for (int i = 0; i < L::L; i++) r ^= a.l[i];
  out[tid] = r;
}
__global__ void __launch_bounds__(256) k_sat(u32* out, int iters) {
  typedef Fp<P> F;
  F a, b;
  u32 tid = blockIdx.x * blockDim.x + threadIdx.x;
  for (int i = 0; i < F::N; i++) { a.l[i] = (tid * 2654435761u * (i + 1)); b.l[i] = ((tid ^ 77) * 40503u * (i + 3)); }
  a.l[F::N - 1] &= 0x7ffffff; b.l[F::N - 1] &= 0x7ffffff;
  for (int k = 0; k < iters; k++) a = F::mul(a, b);
  u32 r = 0;
  for (int i = 0; i < F::N; i++) r ^= a.l[i];
  out[tid] = r;
}
// issue peak of v_mad_u64_u32: four independent accumulator chains per lane, 32 instructions per asm statement
__global__ void __launch_bounds__(256) k_mad_peak(u64* out, int iters, u32 mask) {
  u32 tid = blockIdx.x * blockDim.x + threadIdx.x;
  u64 a0 = tid, a1 = tid + 1, a2 = tid + 2, a3 = tid + 3;
  const u32 x = (tid * 2654435761u | 0x80000001u) & mask, y = ((tid * 40503u + 7u) | 0x80000000u) & mask;   // operands of `bits` significant bits
#define ARK_M4 "v_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_mad_u64_u32 %1, vcc, %4, %5, %1\n\tv_mad_u64_u32 %2, vcc, %4, %5, %2\n\tv_mad_u64_u32 %3, vcc, %4, %5, %3\n\t"
  for (int k = 0; k < iters; k++)
    asm volatile(ARK_M4 ARK_M4 ARK_M4 ARK_M4 ARK_M4 ARK_M4 ARK_M4 ARK_M4 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "v"(y) : "vcc");
  out[tid] = a0 ^ a1 ^ a2 ^ a3;
}
int main() {
  u32* out;
  if (hipMalloc(&out, 256 * 8 * 256 * 4 * 4 * 2) != hipSuccess) return 1;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000;
  for (int bits : {32, 28, 24, 16}) {
    for (int w : {2, 8}) {
      const int b = 256 * w, it = 4000;
      const u32 mask = bits == 32 ? 0xffffffffu : ((1u << bits) - 1u);
      hipLaunchKernelGGL(k_mad_peak, dim3(b), dim3(256), 0, 0, (u64*)out, 10, mask);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      hipLaunchKernelGGL(k_mad_peak, dim3(b), dim3(256), 0, 0, (u64*)out, it, mask);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      printf("v_mad_u64_u32 alone, 4 chains per lane, %2d-bit operands  waves/SIMD=%d  %8.3f ms  %8.2f T lane-ops/s\n", bits, w, ms,
             (double)b * 256 * it * 32 / (ms * 1e-3) * 1e-12);
    }
  }
  printf("field id %d: %d x %d-bit carry-free limbs against %d saturated 32-bit limbs\n", P::ID, FpL<P>::L, FpL<P>::W, P::N);
  const char* names[6] = {"carry-free mul, compiler-scheduled columns", "carry-free mul  (asm: one chain per column)", "carry-free sqr  (asm)                      ", "carry-free sop2 (asm: a b + c d)           ", "carry-free sqr, compiler-scheduled         ", "sat32  mul  (fp.cuh)                       "};
  for (int w : {1, 2, 4, 8}) {
    for (int v = 0; v < 6; v++) {
      int b = 256 * w;
      auto launch = [&](int it) {
        switch (v) {
          case 0: hipLaunchKernelGGL((k_lazy<0>), dim3(b), dim3(256), 0, 0, out, it); break;
          case 1: hipLaunchKernelGGL((k_lazy<1>), dim3(b), dim3(256), 0, 0, out, it); break;
          case 2: hipLaunchKernelGGL((k_lazy<2>), dim3(b), dim3(256), 0, 0, out, it); break;
          case 3: hipLaunchKernelGGL((k_lazy<3>), dim3(b), dim3(256), 0, 0, out, it); break;
          case 4: hipLaunchKernelGGL((k_lazy<4>), dim3(b), dim3(256), 0, 0, out, it); break;
          default: hipLaunchKernelGGL(k_sat, dim3(b), dim3(256), 0, 0, out, it);
        }
      };
      launch(10);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      launch(iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      printf("%s waves/SIMD=%d  %8.3f ms  %8.2f G/s\n", names[v], w, ms, (double)b * 256 * iters / (ms * 1e-3) * 1e-9);
    }
  }
  return 0;
}
