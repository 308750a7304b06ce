// Device-side self-check of fp28.cuh: the asm column chains (mul / sqr / sop2) against the portable statements
// (mul_c / sqr_c / sop2_c) limb for limb, and the repack -> product -> shr_mod -> canonical pipeline against fp.cuh's
// saturated product, on every field size.   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I.. lazycheck.hip -o lazycheck.bin
#include "../fp28.cuh"
#include <stdio.h>
using namespace arkhip;
template <class P>
__global__ void k_check(u32* bad, u32* dump, int n) {
  typedef FpL<P> L; typedef Fp<P> F;
  u32 tid = blockIdx.x * blockDim.x + threadIdx.x;
  if ((int)tid >= n) return;
  F x, y;
  u32 s = tid * 2654435761u + 12345u;
  for (int i = 0; i < P::N; i++) { s = s * 1664525u + 1013904223u; x.l[i] = s; s = s * 1664525u + 1013904223u; y.l[i] = s; }
  x.l[P::N - 1] &= (1u << ((P::BITS - 1) % 32)) - 1; y.l[P::N - 1] &= (1u << ((P::BITS - 1) % 32)) - 1;
  const L a = L::unpack32_shl(x.l), b = L::unpack32_shl(y.l);
#if defined(__HIP_DEVICE_COMPILE__)
  const L m1 = L::mul(a, b), m2 = L::mul_c(a, b), q1 = L::sqr(a), q2 = L::sqr_c(a), s1 = L::sop2(a, b, b, a), s2 = L::sop2_c(a, b, b, a);
  u32 e = 0;
  for (int i = 0; i < L::L; i++) { if (m1.l[i] != m2.l[i]) e |= 1; if (q1.l[i] != q2.l[i]) e |= 2; if (s1.l[i] != s2.l[i]) e |= 4; }
  const F full = m1.template shr_mod<L::SH>().to_canonical_bits();
  if (!F::eq(full, F::mul(x, y))) e |= 8;
  const F fullc = m2.template shr_mod<L::SH>().to_canonical_bits();
  if (!F::eq(fullc, F::mul(x, y))) e |= 16;
  if (e) {
    atomicOr(bad, e);
    if (atomicAdd(bad + 1, 1u) == 0) {
      const F sat = F::mul(x, y);
      for (int i = 0; i < L::L; i++) { dump[i] = m1.l[i]; dump[32 + i] = m2.l[i]; }
      for (int i = 0; i < P::N; i++) { dump[64 + i] = x.l[i]; dump[80 + i] = y.l[i]; dump[96 + i] = full.l[i]; dump[112 + i] = sat.l[i]; }
    }
  }
#endif
}
template <class P> int run(const char* name) {
  u32 *bad, *dump; hipMalloc(&bad, 8); hipMalloc(&dump, 128 * 4); hipMemset(bad, 0, 8);
  hipLaunchKernelGGL((k_check<P>), dim3(64), dim3(256), 0, 0, bad, dump, 64 * 256);
  u32 h[2], d[128]; hipMemcpy(h, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(d, dump, 512, hipMemcpyDeviceToHost);
  printf("%-14s L=%d SH=%d  mask=%u (1 mul, 2 sqr, 4 sop2 asm != portable; 8 asm pipeline != saturated; 16 portable pipeline != saturated)  lanes=%u\n", name, FpL<P>::L, FpL<P>::SH, h[0], h[1]);
  if (h[0]) {
    typedef Fp<P> F; typedef FpL<P> L;
    F x, y; for (int i = 0; i < P::N; i++) { x.l[i] = d[64 + i]; y.l[i] = d[80 + i]; }
    const L hm = L::mul_c(L::unpack32_shl(x.l), L::unpack32_shl(y.l));
    const F hfull = hm.template shr_mod<L::SH>().to_canonical_bits(), hsat = F::mul(x, y);
    for (int i = 0; i < L::L; i++) printf("   mul limb %2d  dev asm %08x  dev portable %08x  host %08x\n", i, d[i], d[32 + i], hm.l[i]);
    for (int i = 0; i < P::N; i++) printf("   word %d  x %08x y %08x | dev pipeline %08x host pipeline %08x | dev sat %08x host sat %08x\n", i, d[64 + i], d[80 + i], d[96 + i], hfull.l[i], d[112 + i], hsat.l[i]);
  }
  return h[0] != 0;
}
int main() {
  int b = 0;
  b += run<BN254_FQ>("BN254_FQ"); b += run<BN254_FR>("BN254_FR"); b += run<BLS12_381_FR>("BLS12_381_FR"); b += run<BLS12_377_FR>("BLS12_377_FR");
  b += run<BLS12_381_FQ>("BLS12_381_FQ"); b += run<BLS12_377_FQ>("BLS12_377_FQ");
  printf(b ? "FAIL\n" : "all ok\n");
  return b;
}
