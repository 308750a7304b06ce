// Instruction-rate microbenchmarks for gfx950 (MI355X).
// Purpose: the local guides do not document integer-multiply issue rates; every MSM/FFT
// throughput estimate hinges on v_mad_u64_u32 vs v_fma_f64 vs add rates (SURVEY.md §7 "Hard parts").
// Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 ubench.hip -o ubench.bin
// Output: one line per test: name, Ginstr/s (lane-instr), cycles per wave-instr per SIMD @2.4GHz.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <string>

#define CHECK(x) do { hipError_t e=(x); if(e!=hipSuccess){fprintf(stderr,"HIP error %s at %s:%d\n",hipGetErrorString(e),__FILE__,__LINE__); exit(1);} } while(0)

typedef uint32_t u32; typedef uint64_t u64;

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

// ---- each kernel: 8 independent dependency chains, ITER iterations x 8 instrs ----
#define KERNEL_BEGIN(name) \
extern "C" __global__ void __launch_bounds__(256) name(u32* out, int iters, u32 s0, u32 s1) { \
  u32 tid = blockIdx.x*blockDim.x+threadIdx.x; \
  u32 a = tid*2654435761u + s0, b = tid*40503u + s1 | 1u;

#define KERNEL_END_U64 \
  u64 r = 0; _Pragma("unroll") for (int i=0;i<8;i++) r ^= x[i]; \
  out[tid] = (u32)r ^ (u32)(r>>32); }

#define KERNEL_END_U32 \
  u32 r = 0; _Pragma("unroll") for (int i=0;i<8;i++) r ^= x[i]; \
  out[tid] = r; }

KERNEL_BEGIN(k_mad_u64_u32)
  u64 x[8]; for (int i=0;i<8;i++) x[i]=a*(i+1);
  for (int it=0; it<iters; it++) {
    #define X(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(x[i]) : "v"(a), "v"(b) : "vcc");
    REP8(X) REP8(X)
    #undef X
  }
KERNEL_END_U64

// mad + addc pair (the Comba inner step): acc(64) += a*b ; top += carry
KERNEL_BEGIN(k_mad_addc)
  u64 x[8]; u32 t[8]; for (int i=0;i<8;i++) {x[i]=a*(i+1); t[i]=i;}
  for (int it=0; it<iters; it++) {
    #define X(i) asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(x[i]), "+v"(t[i]) : "v"(a), "v"(b) : "vcc");
    REP8(X) REP8(X)
    #undef X
  }
  for (int i=0;i<8;i++) x[i]^=t[i];
KERNEL_END_U64

KERNEL_BEGIN(k_mul_lo_u32)
  u32 x[8]; for (int i=0;i<8;i++) x[i]=a*(i+1)|1;
  for (int it=0; it<iters; it++) {
    #define X(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[i]) : "v"(b));
    REP8(X) REP8(X)
    #undef X
  }
KERNEL_END_U32

KERNEL_BEGIN(k_mul_hi_u32)
  u32 x[8]; for (int i=0;i<8;i++) x[i]=a*(i+1)|0x80000000u;
  for (int it=0; it<iters; it++) {
    #define X(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x[i]) : "v"(b|0xf0000000u));
    REP8(X) REP8(X)
    #undef X
  }
KERNEL_END_U32

KERNEL_BEGIN(k_add_u32)
  u32 x[8]; for (int i=0;i<8;i++) x[i]=a*(i+1);
  for (int it=0; it<iters; it++) {
    #define X(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[i]) : "v"(b));
    REP8(X) REP8(X)
    #undef X
  }
KERNEL_END_U32

KERNEL_BEGIN(k_add_co_addc)
  u32 x[8]; u32 y[8]; for (int i=0;i<8;i++) {x[i]=a*(i+1); y[i]=i;}
  for (int it=0; it<iters; it++) {
    #define X(i) asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %2, vcc" : "+v"(x[i]), "+v"(y[i]) : "v"(b) : "vcc");
    REP8(X) REP8(X)
    #undef X
  }
  for (int i=0;i<8;i++) x[i]^=y[i];
KERNEL_END_U32

KERNEL_BEGIN(k_lshl_add_u64)
  u64 x[8]; for (int i=0;i<8;i++) x[i]=a*(i+1);
  u64 bb = ((u64)b<<32)|a;
  for (int it=0; it<iters; it++) {
    #define X(i) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(x[i]) : "v"(bb));
    REP8(X) REP8(X)
    #undef X
  }
KERNEL_END_U64

KERNEL_BEGIN(k_mad_u32_u24)
  u32 x[8]; for (int i=0;i<8;i++) x[i]=a*(i+1);
  for (int it=0; it<iters; it++) {
    #define X(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(x[i]) : "v"(b), "v"(a));
    REP8(X) REP8(X)
    #undef X
  }
KERNEL_END_U32

KERNEL_BEGIN(k_mul_hi_u32_u24)
  u32 x[8]; for (int i=0;i<8;i++) x[i]=a*(i+1)|0x800000u;
  for (int it=0; it<iters; it++) {
    #define X(i) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(x[i]) : "v"(b|0xf00000u));
    REP8(X) REP8(X)
    #undef X
  }
KERNEL_END_U32

KERNEL_BEGIN(k_fma_f64)
  double x[8]; for (int i=0;i<8;i++) x[i]=(double)(a&1023)*(i+1);
  double fa = 1.0000001, fb = 0.5;
  for (int it=0; it<iters; it++) {
    #define X(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[i]) : "v"(fa), "v"(fb));
    REP8(X) REP8(X)
    #undef X
  }
  double r=0; for (int i=0;i<8;i++) r+=x[i];
  out[tid]=(u32)(long long)r; }

KERNEL_BEGIN(k_fma_f32)
  float x[8]; for (int i=0;i<8;i++) x[i]=(float)(a&1023)*(i+1);
  float fa = 1.0000001f, fb = 0.5f;
  for (int it=0; it<iters; it++) {
    #define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(fa), "v"(fb));
    REP8(X) REP8(X)
    #undef X
  }
  float r=0; for (int i=0;i<8;i++) r+=x[i];
  out[tid]=(u32)(int)r; }

KERNEL_BEGIN(k_dot4_u32_u8)
  u32 x[8]; for (int i=0;i<8;i++) x[i]=a*(i+1);
  for (int it=0; it<iters; it++) {
    #define X(i) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(x[i]) : "v"(b), "v"(a));
    REP8(X) REP8(X)
    #undef X
  }
KERNEL_END_U32

KERNEL_BEGIN(k_add3_u32)
  u32 x[8]; for (int i=0;i<8;i++) x[i]=a*(i+1);
  for (int it=0; it<iters; it++) {
    #define X(i) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(b), "v"(a));
    REP8(X) REP8(X)
    #undef X
  }
KERNEL_END_U32

KERNEL_BEGIN(k_mov_b32)
  u32 x[8]; for (int i=0;i<8;i++) x[i]=a*(i+1);
  u32 y[8];
  for (int it=0; it<iters; it++) {
    #define X(i) asm volatile("v_mov_b32 %0, %1" : "=v"(y[i]) : "v"(x[i]));
    REP8(X)
    #undef X
    #define X(i) asm volatile("v_mov_b32 %0, %1" : "=v"(x[i]) : "v"(y[i]));
    REP8(X)
    #undef X
  }
KERNEL_END_U32

KERNEL_BEGIN(k_mul_u64_compiler)   // what hipcc emits for a plain 64x64->64 multiply
  u64 x[8]; for (int i=0;i<8;i++) x[i]=((u64)a<<32|b)*(i+1)|1;
  u64 m = ((u64)b<<32)|a|1;
  for (int it=0; it<iters; it++) {
    #pragma unroll
    for (int r=0;r<2;r++) {
      #pragma unroll
      for (int i=0;i<8;i++) { x[i] = x[i]*m; asm volatile("" : "+v"(x[i])); }
    }
  }
KERNEL_END_U64

// ---------------- memory-side microbenchmarks ----------------
extern "C" __global__ void __launch_bounds__(256) k_hist_atomic(u32* hist, u32 mask, int per_thread, u32 seed) {
  u32 tid = blockIdx.x*blockDim.x+threadIdx.x;
  u32 s = tid*2654435761u + seed;
  for (int i=0;i<per_thread;i++) { s = s*1664525u + 1013904223u; atomicAdd(&hist[(s>>8)&mask], 1u); }
}
// atomic with return value (the scatter step of a counting sort)
extern "C" __global__ void __launch_bounds__(256) k_hist_atomic_ret(u32* hist, u32* sink, u32 mask, int per_thread, u32 seed) {
  u32 tid = blockIdx.x*blockDim.x+threadIdx.x;
  u32 s = tid*2654435761u + seed; u32 acc=0;
  for (int i=0;i<per_thread;i++) { s = s*1664525u + 1013904223u; acc += atomicAdd(&hist[(s>>8)&mask], 1u); }
  sink[tid]=acc;
}
// random 96-byte gathers (6 x 16B) from a large table
extern "C" __global__ void __launch_bounds__(256) k_gather96(const uint4* tbl, u32* sink, u32 nmask, int per_thread, u32 seed) {
  u32 tid = blockIdx.x*blockDim.x+threadIdx.x;
  u32 s = tid*2654435761u + seed; uint4 acc = {0,0,0,0};
  for (int i=0;i<per_thread;i++) {
    s = s*1664525u + 1013904223u; size_t idx = (size_t)((s>>4)&nmask)*6;
    #pragma unroll
    for (int k=0;k<6;k++) { uint4 v = tbl[idx+k]; acc.x^=v.x; acc.y^=v.y; acc.z^=v.z; acc.w^=v.w; }
  }
  sink[tid]=acc.x^acc.y^acc.z^acc.w;
}


// ---------------- Montgomery multiply variants (32-bit limbs) ----------------
#define MAC "v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
#define MAC2 MAC "\n\tv_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
#define MAC3 MAC2 "\n\tv_mad_u64_u32 %0, vcc, %6, %7, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
#define MAC4 MAC3 "\n\tv_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
struct Acc { u64 lo; u32 hi; };
__device__ __forceinline__ void mac1(Acc& c, u32 a0, u32 b0) { asm(MAC : "+v"(c.lo), "+v"(c.hi) : "v"(a0), "v"(b0) : "vcc"); }
__device__ __forceinline__ void mac2(Acc& c, u32 a0, u32 b0, u32 a1, u32 b1) { asm(MAC2 : "+v"(c.lo), "+v"(c.hi) : "v"(a0), "v"(b0), "v"(a1), "v"(b1) : "vcc"); }
__device__ __forceinline__ void mac4(Acc& c, u32 a0, u32 b0, u32 a1, u32 b1, u32 a2, u32 b2, u32 a3, u32 b3) { asm(MAC4 : "+v"(c.lo), "+v"(c.hi) : "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(a2), "v"(b2), "v"(a3), "v"(b3) : "vcc"); }
__device__ __forceinline__ void mac1s(Acc& c, u32 a0, u32 b0) { asm(MAC : "+v"(c.lo), "+v"(c.hi) : "v"(a0), "s"(b0) : "vcc"); }
__device__ __forceinline__ void mac2s(Acc& c, u32 a0, u32 b0, u32 a1, u32 b1) { asm(MAC2 : "+v"(c.lo), "+v"(c.hi) : "v"(a0), "s"(b0), "v"(a1), "s"(b1) : "vcc"); }
__device__ __forceinline__ void mac4s(Acc& c, u32 a0, u32 b0, u32 a1, u32 b1, u32 a2, u32 b2, u32 a3, u32 b3) { asm(MAC4 : "+v"(c.lo), "+v"(c.hi) : "v"(a0), "s"(b0), "v"(a1), "s"(b1), "v"(a2), "s"(b2), "v"(a3), "s"(b3) : "vcc"); }
// sum_{i=lo..hi} x[i]*y[k-i]
template<int LO, int HI, int K, bool S>
__device__ __forceinline__ void col(Acc& c, const u32* x, const u32* y) {
  if constexpr (HI-LO+1 >= 4) {
    if constexpr (S) mac4s(c, x[LO],y[K-LO], x[LO+1],y[K-LO-1], x[LO+2],y[K-LO-2], x[LO+3],y[K-LO-3]);
    else mac4(c, x[LO],y[K-LO], x[LO+1],y[K-LO-1], x[LO+2],y[K-LO-2], x[LO+3],y[K-LO-3]);
    col<LO+4,HI,K,S>(c,x,y);
  } else if constexpr (HI-LO+1 >= 2) {
    if constexpr (S) mac2s(c, x[LO],y[K-LO], x[LO+1],y[K-LO-1]); else mac2(c, x[LO],y[K-LO], x[LO+1],y[K-LO-1]);
    col<LO+2,HI,K,S>(c,x,y);
  } else if constexpr (HI-LO+1 == 1) {
    if constexpr (S) mac1s(c, x[LO],y[K-LO]); else mac1(c, x[LO],y[K-LO]);
  }
}
template<int N> struct Mod { u32 p[N]; u32 inv; };
template<int N, int K>
__device__ __forceinline__ void col_lo(Acc& c, const u32* a, const u32* b, u32* m, const Mod<N>& M) {
  col<0,K,K,false>(c,a,b);
  col<0,K-1,K,true>(c,m,M.p);
  m[K] = (u32)c.lo * M.inv;
  mac1s(c, m[K], M.p[0]);
  c.lo = (c.lo>>32) | ((u64)c.hi<<32); c.hi=0;
  if constexpr (K+1<N) col_lo<N,K+1>(c,a,b,m,M);
}
template<int N, int K>
__device__ __forceinline__ void col_hi(Acc& c, const u32* a, const u32* b, const u32* m, u32* r, const Mod<N>& M) {
  col<K-N+1,N-1,K,false>(c,a,b);
  col<K-N+1,N-1,K,true>(c,m,M.p);
  r[K-N]=(u32)c.lo;
  c.lo = (c.lo>>32) | ((u64)c.hi<<32); c.hi=0;
  if constexpr (K+1<2*N) col_hi<N,K+1>(c,a,b,m,r,M);
}
template<int N>
__device__ __forceinline__ void mont_mul_C(u32* r, const u32* a, const u32* b, const Mod<N>& M) {
  Acc c{0,0}; u32 m[N]; u32 t[N];
  col_lo<N,0>(c,a,b,m,M);
  col_hi<N,N>(c,a,b,m,t,M);
  u32 d[N]; u32 borrow=0;
  #pragma unroll
  for (int i=0;i<N;i++){ u64 x=(u64)t[i]-M.p[i]-borrow; d[i]=(u32)x; borrow=(u32)(x>>63); }
  #pragma unroll
  for (int i=0;i<N;i++) r[i]= borrow? t[i]: d[i];
}
// Variant A: plain CIOS, 32-bit limbs, u64 temporaries
template<int N>
__device__ __forceinline__ void mont_mul_A(u32* r, const u32* a, const u32* b, const u32* p, u32 inv) {
  u32 t[N+2];
  #pragma unroll
  for (int i=0;i<N+2;i++) t[i]=0;
  #pragma unroll
  for (int i=0;i<N;i++) {
    u64 c=0;
    #pragma unroll
    for (int j=0;j<N;j++) { u64 x=(u64)a[j]*b[i]+t[j]+c; t[j]=(u32)x; c=x>>32; }
    u64 x=(u64)t[N]+c; t[N]=(u32)x; t[N+1]=(u32)(x>>32);
    u32 m=t[0]*inv;
    x=(u64)m*p[0]+t[0]; c=x>>32;
    #pragma unroll
    for (int j=1;j<N;j++){ x=(u64)m*p[j]+t[j]+c; t[j-1]=(u32)x; c=x>>32; }
    x=(u64)t[N]+c; t[N-1]=(u32)x; t[N]=t[N+1]+(u32)(x>>32);
  }
  #pragma unroll
  for (int i=0;i<N;i++) r[i]=t[i];
}

template<int N, int VARIANT>
__global__ void __launch_bounds__(256) k_mont(u32* out, int iters, Mod<N> M, u32 seed) {
  u32 a[N], b[N];
  u32 tid = blockIdx.x*blockDim.x+threadIdx.x;
  #pragma unroll
  for (int i=0;i<N;i++){ a[i]=(tid+seed)*2654435761u*(i+1); b[i]=(tid^seed)*40503u*(i+3); }
  a[N-1]&=0x0fffffffu; b[N-1]&=0x0fffffffu;
  for (int k=0;k<iters;k++){
    if constexpr (VARIANT==0) mont_mul_A<N>(a,a,b,M.p,M.inv); else mont_mul_C<N>(a,a,b,M);
  }
  u32 r=0;
  #pragma unroll
  for (int i=0;i<N;i++) r^=a[i];
  out[tid]=r;
}
template<int N, int VARIANT>
static void run_mont(const char* name, u32* out, int cus, Mod<N> M, hipEvent_t e0, hipEvent_t e1) {
  const int iters=2000;
  for (int w : {1,2,4,8}) {
    int b = cus*w;
    hipLaunchKernelGGL((k_mont<N,VARIANT>), dim3(b), dim3(256), 0, 0, out, 10, M, 1u);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_mont<N,VARIANT>), dim3(b), dim3(256), 0, 0, out, iters, M, 1u);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms,e0,e1));
    double rate = (double)b*256*iters/(ms*1e-3);
    printf("%-24s blocks/CU=%d  %8.3f ms  %8.2f Gmul/s\n", name, w, ms, rate*1e-9);
  }
}

typedef void (*kern_t)(u32*, int, u32, u32);
struct Test { const char* name; kern_t k; int instr_per_iter; };

int main(int argc, char** argv) {
  int dev=0; CHECK(hipSetDevice(dev));
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, dev));
  printf("# device: %s  CUs=%d  clock=%d kHz  gcnArch=%s\n", prop.name, prop.multiProcessorCount, prop.clockRate, prop.gcnArchName);
  const int blocks = prop.multiProcessorCount*8, threads=256;   // 8 waves/SIMD
  const int iters = 4096;
  setvbuf(stdout, NULL, _IOLBF, 0);
  u32* out; CHECK(hipMalloc(&out, (size_t)4096*256*4 + (size_t)blocks*threads*4));
  hipEvent_t e0,e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  std::vector<Test> tests = {
    {"v_mad_u64_u32", (kern_t)k_mad_u64_u32, 16}, {"mad_u64_u32+addc (2 instr)", (kern_t)k_mad_addc, 32},
    {"v_mul_lo_u32", (kern_t)k_mul_lo_u32, 16}, {"v_mul_hi_u32", (kern_t)k_mul_hi_u32, 16},
    {"v_add_u32", (kern_t)k_add_u32, 16}, {"add_co+addc (2 instr)", (kern_t)k_add_co_addc, 32},
    {"v_lshl_add_u64", (kern_t)k_lshl_add_u64, 16}, {"v_mad_u32_u24", (kern_t)k_mad_u32_u24, 16},
    {"v_mul_hi_u32_u24", (kern_t)k_mul_hi_u32_u24, 16}, {"v_fma_f64", (kern_t)k_fma_f64, 16},
    {"v_fma_f32", (kern_t)k_fma_f32, 16}, {"v_dot4_u32_u8", (kern_t)k_dot4_u32_u8, 16},
    {"v_add3_u32", (kern_t)k_add3_u32, 16}, {"v_mov_b32", (kern_t)k_mov_b32, 16},
    {"u64*u64 (compiler)", (kern_t)k_mul_u64_compiler, 16},
  };
  double simd_hz = (double)prop.multiProcessorCount*4*2.4e9;
  for (auto& t : tests) {
    hipLaunchKernelGGL(t.k, dim3(blocks), dim3(threads), 0, 0, out, 64, 1u, 3u);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(t.k, dim3(blocks), dim3(threads), 0, 0, out, iters, 1u, 3u);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms,e0,e1));
    double lane_instr = (double)blocks*threads*iters*t.instr_per_iter;
    double rate = lane_instr/(ms*1e-3);
    double wave_instr_per_s = rate/64.0;
    printf("%-28s %8.3f ms  %9.2f Glane-instr/s  %6.2f cyc/wave-instr/SIMD(@2.4GHz)\n", t.name, ms, rate*1e-9, simd_hz/wave_instr_per_s);
  }
  // occupancy sensitivity for mad: 1,2,4 waves/SIMD
  for (int w : {1,2,4}) {
    int b = prop.multiProcessorCount*w;
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_mad_addc, dim3(b), dim3(threads), 0, 0, out, iters, 1u, 3u);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms,e0,e1));
    double rate = (double)b*threads*iters*32/(ms*1e-3);
    printf("mad+addc @%d waves/SIMD       %8.3f ms  %9.2f Glane-instr/s  %6.2f cyc/wave-instr/SIMD\n", w, ms, rate*1e-9, simd_hz/(rate/64));
  }

  {
    Mod<12> M12 = {{0xffffaaab,0xb9feffff,0xb153ffff,0x1eabfffe,0xf6b0f624,0x6730d2a0,0xf38512bf,0x64774b84,0x434bacd7,0x4b1ba7b6,0x397fe69a,0x1a0111ea}, 0xfffcfffd};
    Mod<8> M8 = {{0x00000001,0xffffffff,0xfffe5bfe,0x53bda402,0x09a1d805,0x3339d808,0x299d7d48,0x73eda753}, 0xffffffff};
    run_mont<12,0>("Fp384 CIOS (compiler)", out, prop.multiProcessorCount, M12, e0, e1);
    run_mont<12,1>("Fp384 Comba (asm mac)", out, prop.multiProcessorCount, M12, e0, e1);
    run_mont<8,0>("Fp256 CIOS (compiler)", out, prop.multiProcessorCount, M8, e0, e1);
    run_mont<8,1>("Fp256 Comba (asm mac)", out, prop.multiProcessorCount, M8, e0, e1);
  }
  // ---- memory tests ----
  {
    for (int logb : {10, 16, 20, 24}) {
      u32 nb = 1u<<logb; u32* hist; CHECK(hipMalloc(&hist,(size_t)nb*4)); CHECK(hipMemset(hist,0,(size_t)nb*4));
      int per=64; int b2 = 4096;
      hipLaunchKernelGGL(k_hist_atomic, dim3(b2), dim3(256), 0,0, hist, nb-1, 4, 7u); CHECK(hipDeviceSynchronize());
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_hist_atomic, dim3(b2), dim3(256), 0,0, hist, nb-1, per, 9u);
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms,e0,e1));
      printf("atomicAdd(noret) random into 2^%d bins: %8.3f ms  %8.2f Gatomic/s\n", logb, ms, (double)b2*256*per/(ms*1e-3)*1e-9);
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_hist_atomic_ret, dim3(b2), dim3(256), 0,0, hist, out, nb-1, per, 11u);
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      CHECK(hipEventElapsedTime(&ms,e0,e1));
      printf("atomicAdd(ret)   random into 2^%d bins: %8.3f ms  %8.2f Gatomic/s\n", logb, ms, (double)b2*256*per/(ms*1e-3)*1e-9);
      CHECK(hipFree(hist));
    }
    for (int logn : {20, 24}) {
      size_t n = (size_t)1<<logn; uint4* tbl; CHECK(hipMalloc(&tbl, n*96)); CHECK(hipMemset(tbl,1,n*96));
      int per=32; int b2=4096;
      hipLaunchKernelGGL(k_gather96, dim3(b2), dim3(256), 0,0, tbl, out, (u32)(n-1), 2, 3u); CHECK(hipDeviceSynchronize());
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_gather96, dim3(b2), dim3(256), 0,0, tbl, out, (u32)(n-1), per, 5u);
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms,e0,e1));
      double g = (double)b2*256*per/(ms*1e-3);
      printf("random 96B gathers from 2^%d-entry table (%zu MiB): %8.3f ms  %8.2f Ggather/s  %8.1f GB/s\n", logn, n*96>>20, ms, g*1e-9, g*96e-9);
      CHECK(hipFree(tbl));
    }
  }
  return 0;
}
