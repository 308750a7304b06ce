// Pricing of BATCHED-AFFINE bucket accumulation against the XYZZ mixed addition the accumulate kernel runs (VERDICT r2:
// "the one algorithmic lever not priced").  An affine addition (x3 = l^2 - x1 - x2, y3 = l (x1 - x3) - y1,
// l = (y2 - y1) / (x2 - x1)) costs 2 products + 1 square once 1 / (x2 - x1) is known, and Montgomery's trick
// (ff/src/fields/mod.rs:358-385) turns B inversions into 3 (B - 1) products + ONE inversion (~570 products by Fermat):
// 6 products per addition for a large batch, against 9.5 product-equivalents of the XYZZ mixed addition
// (bucket.rs:168-238).  The catch is memory: a lane that batches B additions must park B running products between its
// forward and its backward sweep -- B x 48 B is far beyond registers or LDS for the B >= 256 that amortises the
// inversion -- so every addition reads its operands TWICE and writes / reads a 48-byte running product and writes a
// 96-byte result: ~480 B of HBM traffic per addition, where the XYZZ kernel moves 100 B (one gathered base + its index).
//
// This program measures exactly that inner structure on n independent additions (operands a[i] + b[i]; no bucket
// bookkeeping, no equal-x handling, no tree levels -- i.e. a LOWER bound on what a real batched-affine K4 would cost):
//   forward  sweep: d_i = x_b - x_a, run *= d_i, store run            (1 product,  96 B read + 48 B written)
//   inversion     : inv = run^-1                                       (~570 products per lane per batch)
//   backward sweep: 1/d_i = inv * run_(i-1), inv *= d_i, l, x3, y3     (4 products + 1 square, 240 B read + 96 B written)
// with lane-interleaved layouts (consecutive lanes own consecutive additions: every sweep is coalesced), operands either
// streamed (the best case: later tree levels) or gathered at random from a 2^27-point table (level 0: the bases).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I.. affbench.hip -o affbench.bin
#include "../ec.cuh"
#include <stdio.h>
#include <vector>
using namespace arkhip;
typedef BLS12_381_FQ P;
typedef Fp<P> F;

// one lane: additions t, t + stride, t + 2 stride, ... (B of them)
template <int GATHER>
__global__ void __launch_bounds__(256) k_batched_affine(const char* __restrict__ pts, const u32* __restrict__ ia,
                                                        const u32* __restrict__ ib, char* __restrict__ run_buf,
                                                        char* __restrict__ out, size_t n, size_t stride, int B) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= stride) return;
  F run = F::one();
  for (int j = 0; j < B; j++) {
    const size_t i = t + (size_t)j * stride;
    if (i >= n) break;
    const size_t a = GATHER ? ia[i] : 2 * i, b = GATHER ? ib[i] : 2 * i + 1;
    const F xa = F::load(pts + a * 96), xb = F::load(pts + b * 96);
    const F d = F::sub(xb, xa);
    run = F::mul(run, d);
    run.store(run_buf + i * 48);
  }
  F inv = F::inverse(run);
  for (int j = B - 1; j >= 0; j--) {
    const size_t i = t + (size_t)j * stride;
    if (i >= n) continue;
    const size_t a = GATHER ? ia[i] : 2 * i, b = GATHER ? ib[i] : 2 * i + 1;
    const Affine<F> pa = Affine<F>::load(pts + a * 96), pb = Affine<F>::load(pts + b * 96);
    const F d = F::sub(pb.x, pa.x);
    F di = inv;
    if (j > 0) di = F::mul(inv, F::load(run_buf + (i - stride) * 48));
    inv = F::mul(inv, d);
    const F l = F::mul(F::sub(pb.y, pa.y), di);
    const F x3 = F::sub(F::sub(F::sqr(l), pa.x), pb.x);
    const F y3 = F::sub(F::mul(l, F::sub(pa.x, x3)), pa.y);
    x3.store(out + i * 96);
    y3.store(out + i * 96 + 48);
  }
}
// the XYZZ mixed addition on the same operand stream, for the ratio: acc_t += point, one accumulator per lane
template <int GATHER>
__global__ void __launch_bounds__(256) k_xyzz(const char* __restrict__ pts, const u32* __restrict__ ia, char* __restrict__ out,
                                              size_t n, size_t stride, int B) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= stride) return;
  XYZZ<F> acc = XYZZ<F>::zero();
  for (int j = 0; j < B; j++) {
    const size_t i = t + (size_t)j * stride;
    if (i >= n) break;
    const size_t a = GATHER ? ia[i] : i;
    const Affine<F> p = Affine<F>::load(pts + a * 96);
    xyzz_madd_relaxed<F>(acc, p.x, p.y);
  }
  xyzz_canonical<F>(acc).store(out + t * 192);
}
__global__ void k_fill(char* pts, size_t n) {   // distinct small "x" values (no equal-x pairs), arbitrary y
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  F x = F::zero(), y = F::zero();
  x.l[0] = (u32)(i * 2654435761u) | 1u; x.l[1] = (u32)i; x.l[2] = 7;
  y.l[0] = (u32)(i * 40503u) | 1u; y.l[1] = (u32)(i >> 3); y.l[3] = 5;
  x.store(pts + i * 96);
  y.store(pts + i * 96 + 48);
}
int main() {
  const size_t n = (size_t)1 << 26, npts = 2 * n;   // 2^26 additions: B = 256 still fills the chip (262 144 lanes)
  char *pts, *run, *out;
  u32 *ia, *ib;
  if (hipMalloc(&pts, npts * 96) || hipMalloc(&run, n * 48) || hipMalloc(&out, n * 96) || hipMalloc(&ia, n * 4) || hipMalloc(&ib, n * 4)) return 1;
  hipLaunchKernelGGL(k_fill, dim3((unsigned)((npts + 255) / 256)), dim3(256), 0, 0, pts, npts);
  std::vector<u32> ha(n), hb(n);
  u32 s = 12345;
  for (size_t i = 0; i < n; i++) {
    s = s * 1664525u + 1013904223u; ha[i] = (s >> 4) & (u32)(npts - 1);
    s = s * 1664525u + 1013904223u; hb[i] = ((s >> 4) & (u32)(npts - 1)) ^ 1u;
    if (hb[i] == ha[i]) hb[i] ^= 2u;
  }
  hipMemcpy(ia, ha.data(), n * 4, hipMemcpyHostToDevice);
  hipMemcpy(ib, hb.data(), n * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  printf("# n = 2^26 additions, BLS12-381 Fq, saturated 32-bit limbs (fp.cuh) on both sides; ns = chip-wide time per addition\n");
  for (int gather = 0; gather < 2; gather++) {
    for (int B : {64, 256, 1024, 4096}) {
      const size_t stride = (n + B - 1) / B;   // lanes
      const unsigned blocks = (unsigned)((stride + 255) / 256);
      for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        if (gather) hipLaunchKernelGGL((k_batched_affine<1>), dim3(blocks), dim3(256), 0, 0, pts, ia, ib, run, out, n, stride, B);
        else hipLaunchKernelGGL((k_batched_affine<0>), dim3(blocks), dim3(256), 0, 0, pts, ia, ib, run, out, n, stride, B);
        hipEventRecord(e1); hipEventSynchronize(e1);
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("batched affine  %-8s B=%5d lanes=%8zu  %8.3f ms  %6.3f ns/add  %7.1f GB/s of the ~480 B/add it moves\n",
             gather ? "gathered" : "streamed", B, stride, ms, ms * 1e6 / n, 480.0 * n / (ms * 1e-3) * 1e-9);
    }
    for (int B : {32, 128}) {   // the XYZZ kernel's own structure: B points into one accumulator per lane
      const size_t stride = (n + B - 1) / B;
      const unsigned blocks = (unsigned)((stride + 255) / 256);
      for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        if (gather) hipLaunchKernelGGL((k_xyzz<1>), dim3(blocks), dim3(256), 0, 0, pts, ia, out, n, stride, B);
        else hipLaunchKernelGGL((k_xyzz<0>), dim3(blocks), dim3(256), 0, 0, pts, ia, out, n, stride, B);
        hipEventRecord(e1); hipEventSynchronize(e1);
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("XYZZ mixed add  %-8s B=%5d lanes=%8zu  %8.3f ms  %6.3f ns/add\n", gather ? "gathered" : "streamed", B, stride, ms,
             ms * 1e6 / n);
    }
  }
  return 0;
}
