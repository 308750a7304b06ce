// Host-side check of the carry-free 28-bit-limb arithmetic of the bucket-accumulation kernel (algebra_amd/csrc/fp28.cuh,
// ec28.cuh) against the saturated form (fp.cuh / ec.cuh, itself checked against the oracle on the GPU): repacking,
// products, sums of two products, differences, divisions by powers of two, exact zero tests, and random point sequences
// through the curve-isomorphism boundary (lazy_from_affine / lazy_from_bucket / lazy_to_bucket) that hit the doubling and
// infinity branches and the "continue from a stored bucket" path.  The templates are __host__ __device__, so this runs on
// the CPU.  Built and run by tests/test_lazy_host.py.
#include "ec28.cuh"
#include "curves.cuh"
#include <stdio.h>
#include <random>
using namespace arkhip;
template <class P> Fp<P> rnd(std::mt19937_64& g) {
  Fp<P> a;
  for (int i = 0; i < P::N; i++) a.l[i] = (u32)g();
  a.l[P::N - 1] &= (1u << ((P::BITS - 1) % 32)) - 1;   // < 2^(BITS-1) < p
  return a;
}
// the residue a lazy value stands for, as a canonical Fp (radix R): lazy bits b = v R' ; canonical bits = v R = b 2^-8K..
// generic: multiply the repacked integer by (R / R') mod p through one saturated product with the constant R^2 / R'.
template <class P> Fp<P> lazy_value(const FpL<P>& a) {  // value v (mod p) in canonical Montgomery form
  typedef Fp<P> F;
  // reduce the (possibly > p) integer first: pack needs < 2^(32N); all test values are < 9p < 2^(32N)
  u32 w[P::N];
  a.pack32(w);
  F t = F::reduce_full(*(const F*)w);   // integer a mod p, = v R' mod p
  // v R = t * (R / R') = t * 2^-(28L - 32N): repeated halving mod p
  for (int k = 0; k < FpL<P>::W * FpL<P>::L - 32 * P::N; k++) {
    // t/2 mod p
    u32 c = 0;
    F x = t;
    if (x.l[0] & 1) { for (int i = 0; i < P::N; i++) { u64 s = (u64)x.l[i] + P::P[i] + c; x.l[i] = (u32)s; c = (u32)(s >> 32); } }
    for (int i = 0; i < P::N - 1; i++) x.l[i] = (x.l[i] >> 1) | (x.l[i + 1] << 31);
    x.l[P::N - 1] = (x.l[P::N - 1] >> 1) | (c << 31);
    t = x;
  }
  return t;
}
template <class P> FpL<P> to_lazy(const Fp<P>& a) {  // canonical a -> lazy value a (radix R'): bits = a 2^(28L-32N) mod p
  typedef Fp<P> F;
  F t = a;
  for (int k = 0; k < FpL<P>::W * FpL<P>::L - 32 * P::N; k++) t = F::dbl(t);
  return FpL<P>::unpack32(t.l);
}
template <class P> int run(const char* name, const uint64_t* gen) {
  typedef Fp<P> F;
  typedef FpL<P> L;
  std::mt19937_64 g(7);
  int bad = 0;
  for (int it = 0; it < 2000; it++) {
    F a = rnd<P>(g), b = rnd<P>(g), c = rnd<P>(g), d = rnd<P>(g);
    L la = to_lazy<P>(a), lb = to_lazy<P>(b), lc = to_lazy<P>(c), ld = to_lazy<P>(d);
    if (!F::eq(lazy_value<P>(la), a)) { bad++; if (bad < 5) printf("%s roundtrip fail\n", name); }
    if (!F::eq(lazy_value<P>(L::mul(la, lb)), F::mul(a, b))) { bad++; if (bad < 5) printf("%s mul fail\n", name); }
    if (!F::eq(lazy_value<P>(L::mul_c(la, lb)), F::mul(a, b))) { bad++; if (bad < 5) printf("%s mul_c fail\n", name); }
    if (!F::eq(lazy_value<P>(L::sqr(la)), F::mul(a, a))) { bad++; if (bad < 5) printf("%s sqr fail\n", name); }
    if (!F::eq(lazy_value<P>(L::sqr(L::template sub<6>(la, lb))), F::mul(F::sub(a, b), F::sub(a, b)))) { bad++; if (bad < 5) printf("%s sqr(sub<6>) fail\n", name); }
    if (!F::eq(lazy_value<P>(L::sop2(la, lb, lc, ld)), F::add(F::mul(a, b), F::mul(c, d)))) { bad++; if (bad < 5) printf("%s sop2 fail\n", name); }
    if (!F::eq(lazy_value<P>(L::template sub<1>(la, lb)), F::sub(a, b))) { bad++; if (bad < 5) printf("%s sub fail\n", name); }
    if (!F::eq(lazy_value<P>(L::template sub_b_2c<4>(la, lb, lc)), F::sub(F::sub(a, b), F::dbl(c)))) { bad++; if (bad < 5) printf("%s sub_b_2c fail\n", name); }
    if (!F::eq(lazy_value<P>(L::template negsub<4>(la, lb)), F::neg(F::add(a, b)))) { bad++; if (bad < 5) printf("%s negsub fail\n", name); }
    if (!F::eq(lazy_value<P>(L::mul(L::add_lazy(la, lb), lc)), F::mul(F::add(a, b), c))) { bad++; if (bad < 5) printf("%s lazy-add operand fail\n", name); }
    // v 2^-K: compare through 2^K * result == v
    {
      F h = lazy_value<P>(la.template shr_mod<4>());
      for (int k = 0; k < 4; k++) h = F::dbl(h);  // 16 * (a / 16)
      if (!F::eq(h, a)) { bad++; if (bad < 5) printf("%s shr_mod<4> fail\n", name); }
      L h20 = la.template shr_mod<20>(), h16 = la.template shr_mod<16>().template shr_mod<4>();
      if (!F::eq(lazy_value<P>(h20), lazy_value<P>(h16))) { bad++; if (bad < 5) printf("%s shr_mod<20> fail\n", name); }
    }
    // differences without a carry sweep (semi-normalised), as products' operands; the shifted repack
    if constexpr (L::SEMI2) {   // two semi-normalised operands in one product: only where a column has room (14 x 28 bits)
      if (!F::eq(lazy_value<P>(L::mul(L::template sub_semi<2>(la, lb), L::template sub_semi<6>(lc, ld))), F::mul(F::sub(a, b), F::sub(c, d)))) { bad++; if (bad < 5) printf("%s sub_semi fail\n", name); }
      if (!F::eq(lazy_value<P>(L::sqr(L::template sub_semi<6>(la, lb))), F::sqr(F::sub(a, b)))) { bad++; if (bad < 5) printf("%s sqr(sub_semi) fail\n", name); }
    }
    // the mixed addition's own operand classes: sub_op (semi-normalised or swept, by geometry) squared, times a
    // normalised value, and as the normalised-or-semi side of Y3's sum of two products
    if (!F::eq(lazy_value<P>(L::sqr(L::template sub_op<6>(la, lb))), F::sqr(F::sub(a, b)))) { bad++; if (bad < 5) printf("%s sqr(sub_op) fail\n", name); }
    if (!F::eq(lazy_value<P>(L::mul(L::template sub_op<2>(la, lb), lc)), F::mul(F::sub(a, b), c))) { bad++; if (bad < 5) printf("%s mul(sub_op) fail\n", name); }
    if (!F::eq(lazy_value<P>(L::mul(L::template sub_semi<6>(la, lb), lc)), F::mul(F::sub(a, b), c))) { bad++; if (bad < 5) printf("%s mul(sub_semi, n) fail\n", name); }
    if (!L::SEMI2) { L n = L::template sub_op<6>(la, lb); for (int i = 0; i < L::L - 1; i++) if (n.l[i] > L::MASK) { bad++; printf("%s sub_op not normalised\n", name); break; } }
    if (!F::eq(lazy_value<P>(L::mul(L::template neg_semi<2>(la), lb)), F::mul(F::neg(a), b))) { bad++; if (bad < 5) printf("%s neg_semi fail\n", name); }
    if (!F::eq(lazy_value<P>(L::template sub_b_2c_norm<4>(la, lb, lc)), F::sub(F::sub(a, b), F::dbl(c)))) { bad++; if (bad < 5) printf("%s sub_b_2c_norm fail\n", name); }
    { L n = L::template sub_b_2c_norm<4>(la, lb, lc); for (int i = 0; i < L::L - 1; i++) if (n.l[i] > L::MASK) { bad++; printf("%s sub_b_2c_norm not normalised\n", name); break; } }
    if (!F::eq(lazy_value<P>(L::sop2(L::template sub_op<2>(la, lb), L::template sub_semi<6>(lc, ld), L::template neg_semi<2>(la), lb)),
               F::sub(F::mul(F::sub(a, b), F::sub(c, d)), F::mul(a, b)))) { bad++; if (bad < 5) printf("%s sop2(semi) fail\n", name); }
    if (!F::eq(lazy_value<P>(L::mul(L::unpack32_shl(a.l), lb)), F::mul(a, b))) { /* a.l 2^8 = a R': the same residue, unreduced */ bad++; if (bad < 5) printf("%s unpack32_shl fail\n", name); }
    if (!L::mul(L::template sub_semi<1>(la, la), lb).is_zero_or_p()) { bad++; printf("%s is_zero_or_p miss\n", name); }
    if (L::mul(la, lb).is_zero_or_p() && !(F::eq(a, F::zero()) || F::eq(b, F::zero()))) { bad++; printf("%s is_zero_or_p false hit\n", name); }
    L z = L::template sub<3>(la, la);
    if (!z.is_zero_mod_p()) { bad++; printf("zero test fail\n"); }
    if (L::template sub<2>(la, lb).is_zero_mod_p() && !F::eq(a, b)) { bad++; printf("false zero\n"); }
  }
  // point sequences: multiples of the generator as affine points of E (canonical form, as they sit in HBM)
  F gx = F::load(gen), gy = F::load((const char*)gen + F::BYTES);
  const int NP = 40;
  F px[NP], py[NP];
  XYZZ<F> acc = XYZZ<F>::zero();
  for (int k = 0; k < NP; k++) {
    xyzz_madd<F>(acc, gx, gy);
    F zi = F::inverse(acc.zzz);
    F zzi = F::sqr(F::mul(acc.zz, zi));
    px[k] = F::mul(acc.x, zzi);
    py[k] = F::mul(acc.y, zi);
  }
  for (int trial = 0; trial < 300; trial++) {
    XYZZ<F> c = XYZZ<F>::zero();
    XYZZL<P> lz;
    lz.inf = true;
    lz.x = lz.y = lz.zz = lz.zzz = L::zero();
    int len = 1 + g() % 12;
    for (int s = 0; s < len; s++) {
      int k = g() % NP;
      bool neg = g() & 1;
      F y = F::cond_neg(py[k], neg);
      // what the kernel does with one sorted entry: sign onto the canonical y, repack, add; "equal points" comes back
      // to the caller, which doubles the base re-read from memory (here: a two-coordinate buffer in the bases' layout)
      auto add_entry = [&](bool ng) {
        F yy = F::cond_neg(py[k], ng);
        L lx, ly;
        lazy_from_affine<P>(px[k], yy, lx, ly);
        xyzz_madd<F>(c, px[k], yy);
        if (xyzz_madd_lazy<P>(lz, lx, ly)) {
          char raw[2 * F::BYTES];
          px[k].store(raw);
          py[k].store(raw + F::BYTES);
          XYZZL<P> dbl;
          xyzz_mdbl_lazy<P>(dbl, raw, ng);
          lz = dbl;
        }
      };
      (void)y;
      add_entry(neg);
      if (trial % 3 == 0) add_entry(neg);    // the same point again: the doubling branch
      if (trial % 7 == 0) add_entry(!neg);   // then its inverse: cancels (possibly to infinity)
      if (trial % 4 == 1 && s == len / 2) {  // a streamed MSM's later piece: store the bucket, load it back, go on
        XYZZ<F> stored = lazy_to_bucket<P>(lz);  // what reaches HBM is canonical
        lz = lazy_from_bucket<P>(stored);
      }
    }
    XYZZ<F> d = lazy_to_bucket<P>(lz);
    // the stored bucket must be canonical (every coordinate < p) and the same group element
    bool canon = true;
    F* cs[4] = {&d.x, &d.y, &d.zz, &d.zzz};
    for (F* f : cs) { F r = F::reduce_once(f->l); if (!F::eq(r, *f)) canon = false; }
    bool same;
    if (c.is_zero() || d.is_zero()) same = c.is_zero() && d.is_zero();
    else same = F::eq(F::mul(c.x, d.zz), F::mul(d.x, c.zz)) && F::eq(F::mul(c.y, d.zzz), F::mul(d.y, c.zzz)) &&
                F::eq(F::mul(F::sqr(d.zz), d.zz), F::sqr(d.zzz));  // and a valid XYZZ: ZZ^3 = ZZZ^2
    if (!same || !canon) { bad++; if (bad < 8) printf("%s point seq mismatch trial %d (same=%d canon=%d)\n", name, trial, (int)same, (int)canon); }
  }
  // full additions (the reduction / heavy-run kernels): accumulator += stored bucket (canonical operand, repacked) and
  // accumulator += accumulator, incl. the identity on either side, equal points (doubling) and inverse points
  for (int trial = 0; trial < 200; trial++) {
    auto build = [&](int len, XYZZ<F>& c, XYZZL<P>& l) {
      c = XYZZ<F>::zero();
      l.inf = true;
      l.x = l.y = l.zz = l.zzz = L::zero();
      for (int s2 = 0; s2 < len; s2++) {
        const int k = g() % NP;
        const bool ng = g() & 1;
        F yy = F::cond_neg(py[k], ng);
        L lx, ly;
        lazy_from_affine<P>(px[k], yy, lx, ly);
        xyzz_madd<F>(c, px[k], yy);
        if (xyzz_madd_lazy<P>(l, lx, ly)) { XYZZL<P> d; xyzz_mdbl_lazy_xy<P>(d, lx, ly); l = d; }
      }
    };
    XYZZ<F> ca, cb;
    XYZZL<P> la, lb;
    build(trial % 5 == 0 ? 0 : 1 + g() % 6, ca, la);
    build(trial % 7 == 0 ? 0 : 1 + g() % 6, cb, lb);
    if (trial % 3 == 1) { cb = ca; lb = la; }                                   // equal points: doubling
    if (trial % 11 == 2) { cb = ca; cb.y = F::neg(cb.y); lb = lazy_from_bucket<P>(cb); }   // inverse points: infinity
    XYZZ<F> want = ca;
    xyzz_add<F>(want, cb);
    XYZZL<P> v1 = la, v2 = la;
    xyzz_add_lazy<P>(v1, lb.x, lb.y, lb.zz, lb.zzz, lb.inf);                    // += accumulator
    const XYZZ<F> stored = lazy_to_bucket<P>(lb);                               // += stored bucket (what sits in HBM)
    const XYZZOperands<P> o = lazy_operands_of<P>(stored);
    xyzz_add_lazy<P>(v2, o.x, o.y, o.zz, o.zzz, o.inf);
    for (int v = 0; v < 2; v++) {
      const XYZZ<F> d = lazy_to_bucket<P>(v ? v2 : v1);
      bool same;
      if (want.is_zero() || d.is_zero()) same = want.is_zero() && d.is_zero();
      else same = F::eq(F::mul(want.x, d.zz), F::mul(d.x, want.zz)) && F::eq(F::mul(want.y, d.zzz), F::mul(d.y, want.zzz)) &&
                  F::eq(F::mul(F::sqr(d.zz), d.zz), F::sqr(d.zzz));
      if (!same) { bad++; if (bad < 8) printf("%s full-add mismatch trial %d variant %d\n", name, trial, v); }
    }
  }
  printf("%s: %s (%d bad)\n", name, bad ? "FAIL" : "ok", bad);
  return bad;
}
#include "fft.cuh"
// The arithmetic of the carry-free FFT pass (fft.cuh Fft29, 9 x 29-bit limbs) on the host: reduce_sweep / sweep / canon /
// dif at the EXTREMES the kernel's bounds allow (limbs up to 2.5 2^30, values up to 13.02 p), which random transforms
// do not reach, and one 4-point group of two stages against the saturated arithmetic.
template <class FP> int run_fft29(const char* name) {
  typedef Fp<FP> F;
  typedef FpL<FP> L;
  typedef Fft29<FP> A;
  std::mt19937_64 g(11);
  int bad = 0;
  auto value_of = [&](const L& x) -> F {   // integer value of the limbs mod p, as a canonical Fp (plain integer, no Montgomery factor)
    // sum l_i 2^(29 i) mod p by Horner over canonical adds / doublings
    F acc = F::zero();
    for (int i = 8; i >= 0; i--) {
      for (int b = 0; b < 29; b++) acc = F::dbl(acc);
      u32 w[8] = {x.l[i], 0, 0, 0, 0, 0, 0, 0};
      F li = F::reduce_full(*(const F*)w);
      acc = F::add(acc, li);
    }
    return acc;
  };
  auto lt_kp = [&](const L& x, int k) -> bool {   // normalised x < k p ?
    for (int i = 8; i >= 0; i--) {
      const u32 kp = L::kp_limb(k, i);
      if (x.l[i] != kp) return x.l[i] < kp;
    }
    return false;
  };
  for (int it = 0; it < 20000; it++) {
    // an un-normalised value as the sum outputs have them: up to four tile elements (normalised limbs, values c + j p
    // below 3.01 p) added limb-wise: limbs below 2^31, value below 12.04 p
    L v = L::zero();
    const int terms = 1 + (int)(g() % 4);
    for (int t = 0; t < terms; t++) {
      F r = rnd<FP>(g);
      if (it % 7 == 0) for (int i = 0; i < 8; i++) r.l[i] = FP::P[i] - (i == 0 ? 1 + (u32)(g() % 3) : 0);   // p - 1, p - 2, ..
      const L rl = L::unpack32(r.l);
      const int j = it % 5 == 0 ? 2 : (int)(g() % 3);
      L e;
      for (int i = 0; i < 9; i++) e.l[i] = rl.l[i] + L::kp_limb(j, i);
      e = A::sweep(e);
      for (int i = 0; i < 9; i++) v.l[i] += e.l[i];
    }
    const F want = value_of(v);
    const L rs = A::reduce_sweep(v);
    bool norm = true;
    for (int i = 0; i < 9; i++) norm &= rs.l[i] <= L::MASK;
    if (!norm || !F::eq(value_of(rs), want) || !lt_kp(rs, 4)) { bad++; if (bad < 5) printf("%s reduce_sweep fail (terms %d)\n", name, terms); }
    if (!lt_kp(rs, 3) && bad < 5) printf("%s note: reduce_sweep output >= 3p\n", name);
    const L cn = A::canon(v);
    u32 w8[8];
    cn.pack32(w8);
    if (!F::eq(*(const F*)w8, want) || !lt_kp(cn, 1)) { bad++; if (bad < 5) printf("%s canon fail\n", name); }
    {
      const L sw = A::sweep(v);
      bool n2 = true;
      for (int i = 0; i < 8; i++) n2 &= sw.l[i] <= L::MASK;
      if (!n2 || !F::eq(value_of(sw), want)) { bad++; if (bad < 5) printf("%s sweep fail\n", name); }
    }
  }
  // one 4-point group of two DIF stages with random twiddles against the saturated arithmetic, inputs at the invariant's
  // edge (normalised limbs, values up to 3 p: x = c + j p)
  for (int it = 0; it < 3000; it++) {
    F c[4], w[3];
    L x[4], wl[3];
    for (int m = 0; m < 4; m++) {
      c[m] = rnd<FP>(g);
      x[m] = L::unpack32(c[m].l);
      const int j = (int)(g() % 3);   // + j p, re-normalised
      L t;
      for (int i = 0; i < 9; i++) t.l[i] = x[m].l[i] + L::kp_limb(j, i);
      x[m] = A::sweep(t);
    }
    for (int m = 0; m < 3; m++) {
      w[m] = rnd<FP>(g);
      // twiddle form: w 2^261 mod p, canonical: mont_mul(w R, 2^261 mod p) with w[m] read as w R
      u32 cin[8];
      for (int i = 0; i < 8; i++) cin[i] = FP::LZ_CIN[i];
      wl[m] = L::unpack32(F::mul(w[m], *(const F*)cin).l);
    }
    const L s0 = A::sum(x[0], x[2]), s1 = A::sum(x[1], x[3]);
    const L d0 = L::mul(A::template dif<4, 1>(x[0], x[2]), wl[0]);
    const L d1 = L::mul(A::template dif<4, 1>(x[1], x[3]), wl[1]);
    const L y[4] = {A::reduce_sweep(A::sum(s0, s1)), L::mul(A::template dif<7, 2>(s0, s1), wl[2]), A::sweep(A::sum(d0, d1)),
                    L::mul(A::template dif<2, 1>(d0, d1), wl[2])};
    const F fs0 = F::add(c[0], c[2]), fs1 = F::add(c[1], c[3]);
    const F fd0 = F::mul(F::sub(c[0], c[2]), w[0]), fd1 = F::mul(F::sub(c[1], c[3]), w[1]);
    const F fy[4] = {F::add(fs0, fs1), F::mul(F::sub(fs0, fs1), w[2]), F::add(fd0, fd1), F::mul(F::sub(fd0, fd1), w[2])};
    for (int m = 0; m < 4; m++) {
      bool norm = true;
      for (int i = 0; i < 9; i++) norm &= y[m].l[i] <= L::MASK;
      const L cn = A::canon(y[m]);
      u32 w8[8];
      cn.pack32(w8);
      if (!norm || !lt_kp(y[m], 4) || !F::eq(*(const F*)w8, fy[m])) { bad++; if (bad < 8) printf("%s butterfly group output %d fail\n", name, m); }
    }
    // the tail stages' raw outputs through the exact reduction
    const L raw1 = A::template dif<7, 2>(s0, s1), raw3 = A::template dif<2, 1>(A::template dif<4, 1>(x[0], x[2]), d1);
    u32 w8[8];
    A::canon(raw1).pack32(w8);
    if (!F::eq(*(const F*)w8, F::sub(fs0, fs1))) { bad++; if (bad < 8) printf("%s raw y1 fail\n", name); }
    A::canon(raw3).pack32(w8);
    if (!F::eq(*(const F*)w8, F::sub(F::sub(c[0], c[2]), fd1))) { bad++; if (bad < 8) printf("%s raw y3 fail\n", name); }
  }
  printf("%s fft29: %s (%d bad)\n", name, bad ? "FAIL" : "ok", bad);
  return bad;
}
#include "curve_consts.hpp"
int main() {
  int b = 0;
  b += run<BLS12_381_FQ>("BLS12_381_FQ", GEN_BLS12_381_G1);
  b += run<BLS12_377_FQ>("BLS12_377_FQ", GEN_BLS12_377_G1);
  b += run<BN254_FQ>("BN254_FQ", GEN_BN254_G1);   // 9 x 29-bit limbs
  b += run_fft29<BLS12_381_FR>("BLS12_381_FR");
  b += run_fft29<BN254_FR>("BN254_FR");
  b += run_fft29<BLS12_377_FR>("BLS12_377_FR");
  return b != 0;
}
