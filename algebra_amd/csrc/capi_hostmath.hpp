// Host-side field / group arithmetic of the C ABI: the domain constants, the host forms of sum / into_affine / the MSM tail, the
// transform's entry (same templates as the device code, compiled for the host).  Included by the units that need it.
#pragma once
#include "capi_core.hpp"
namespace arkhip {
namespace capi {

// ---- host-side scalar-field arithmetic for the domain constants (same templates as the device) ----
template <class FP>
Fp<FP> host_pow(Fp<FP> b, const uint64_t* e, int words) {
  Fp<FP> r = Fp<FP>::one();
  for (int i = words * 64 - 1; i >= 0; i--) {
    r = Fp<FP>::sqr(r);
    if ((e[i / 64] >> (i % 64)) & 1) r = Fp<FP>::mul(r, b);
  }
  return r;
}
template <class F>
F host_inverse(const F& a) { return F::inverse(a); }

// Projective (Jacobian) -> XYZZ: (X, Y, Z^2, Z^3)
template <class F>
XYZZ<F> jac_to_xyzz(const uint64_t* p) {
  F x = F::load(p), y = F::load((const char*)p + F::BYTES), z = F::load((const char*)p + 2 * F::BYTES);
  if (z.is_zero()) return XYZZ<F>::zero();
  F zz = F::sqr(z);
  return XYZZ<F>{x, y, zz, F::mul(zz, z)};
}
// The HOST builds of the base field's arithmetic (fp.cuh: 64-bit limbs; what the MSM's serial tail runs on), element by
// element on the calling thread: no GPU involved.  op as ark_hip_test_basefield_op (0 add, 1 sub, 2 mul, 3 sqr, 4 neg, 5 dbl).
template <class F>
inline void host_field_ops(int op, const uint64_t* a, const uint64_t* b, uint64_t* r, size_t n) {
  constexpr size_t W = F::BYTES / 8;
  for (size_t i = 0; i < n; i++) {
    const F x = F::load(a + i * W);
    const F y = b ? F::load(b + i * W) : F::zero();
    F z;
    switch (op) {
      case 0: z = F::add(x, y); break;
      case 1: z = F::sub(x, y); break;
      case 2: z = F::mul(x, y); break;
      case 3: z = F::sqr(x); break;
      case 4: z = F::neg(x); break;
      default: z = F::dbl(x); break;
    }
    z.store(r + i * W);
  }
}
// the MSM's host tail on caller-supplied bit sums (ark_hip_test_msm_host_fold)
template <class C>
int host_fold(const uint64_t* parts, int windows, int nbits, int log2_l0, const int* widths, uint64_t* out_xyz) {
  std::vector<int> off((size_t)windows + 1);
  off[0] = 0;
  for (int w = 0; w < windows; w++) {
    if (widths[w] < 1 || widths[w] > 32) return ARK_HIP_ERR_ARG;
    off[w + 1] = off[w] + widths[w];
  }
  const XYZZ<typename C::F> t = msm_host_fold<C>((const char*)parts, (u32)(nbits + 1), windows, nbits, log2_l0, off.data());
  xyzz_to_jac<typename C::F>(t).store(out_xyz);
  return 0;
}
// sum of n Jacobian points on the host (the multi-GPU combine: one partial per rank)
template <class C>
int host_sum(const uint64_t* pts, size_t n, uint64_t* out) {
  typedef typename C::F F;
  XYZZ<F> acc = XYZZ<F>::zero();
  for (size_t i = 0; i < n; i++) {
    XYZZ<F> p = jac_to_xyzz<F>(pts + i * 3 * F::WORDS64);
    xyzz_add<F>(acc, p);
  }
  xyzz_to_jac<F>(acc).store(out);
  return 0;
}
// From<Projective> for Affine (short_weierstrass/affine.rs:374-396): (x/z^2, y/z^3); identity -> (0,0)
template <class C>
int host_into_affine(const uint64_t* pts, size_t n, uint64_t* out) {
  typedef typename C::F F;
  for (size_t i = 0; i < n; i++) {
    const uint64_t* p = pts + i * 3 * F::WORDS64;
    uint64_t* o = out + i * 2 * F::WORDS64;
    F x = F::load(p), y = F::load((const char*)p + F::BYTES), z = F::load((const char*)p + 2 * F::BYTES);
    if (z.is_zero()) {
      F::zero().store(o);
      F::zero().store((char*)o + F::BYTES);
      continue;
    }
    F zi = host_inverse(z);
    F zi2 = F::sqr(zi);
    F::mul(x, zi2).store(o);
    F::mul(y, F::mul(zi2, zi)).store((char*)o + F::BYTES);
  }
  return 0;
}
template <class FP>
bool host_is_one(const uint64_t* x) {
  Fp<FP> a = Fp<FP>::load(x);
  return Fp<FP>::eq(a, Fp<FP>::one());
}

inline bool field_is_one(int field, const uint64_t* x) {
  switch (field) {
#ifndef ARK_HIP_DEV
    case ARK_HIP_BN254_FR: return host_is_one<BN254_FR>(x);
    case ARK_HIP_BLS12_377_FR: return host_is_one<BLS12_377_FR>(x);
#endif
    case ARK_HIP_BLS12_381_FR: return host_is_one<BLS12_381_FR>(x);
  }
  return false;
}
// Does `dom` belong to field FP?  A domain struct carries no field id; its constants are 32-byte residues of SOME field.  The
// generator of a size-2^k domain satisfies g^(2^k) = 1 and g * g_inv = 1 in its own field, and (with overwhelming probability) in
// no other: k squarings and one product on the host (a BN254 domain handed to a BLS12-381 curve's transform over
// group elements gave garbage points silently).  A residue that is not even canonical in FP (>= p) fails as well.
template <class FP>
bool host_domain_is_of_field(const ark_hip_radix2_domain* dom) {
  typedef Fp<FP> F;
  const F g = F::load(dom->group_gen), gi = F::load(dom->group_gen_inv);
  if (!F::eq(g, g.canonical()) || !F::eq(gi, gi.canonical())) return false;
  F t = g;
  for (uint32_t i = 0; i < dom->log_size_of_group; i++) t = F::sqr(t);
  return F::eq(t, F::one()) && F::eq(F::mul(g, gi), F::one());
}
inline bool domain_is_of_field(int field, const ark_hip_radix2_domain* dom) {
  switch (field) {
#ifndef ARK_HIP_DEV
    case ARK_HIP_BN254_FR: return host_domain_is_of_field<BN254_FR>(dom);
    case ARK_HIP_BLS12_377_FR: return host_domain_is_of_field<BLS12_377_FR>(dom);
#endif
    case ARK_HIP_BLS12_381_FR: return host_domain_is_of_field<BLS12_381_FR>(dom);
  }
  return false;
}
template <class FP>
int domain_new(size_t num_coeffs, ark_hip_radix2_domain* out) {
  typedef Fp<FP> F;
  // usize::next_power_of_two: 0 -> 1
  uint64_t size = 1;
  while (size < num_coeffs) {
    size <<= 1;
    if (size == 0) return ARK_HIP_ERR_SIZE;
  }
  uint32_t lg = 0;
  while (((uint64_t)1 << lg) < size) lg++;
  if ((int)lg > FP::TWO_ADICITY) return ARK_HIP_ERR_SIZE;  // radix2/mod.rs:62-64 -> None
  memset(out, 0, sizeof(*out));
  out->size = size;
  out->log_size_of_group = lg;
  // get_root_of_unity (ff/src/fields/fft_friendly.rs:35-84): TWO_ADIC_ROOT squared (adicity - lg) times
  F g;
  for (int i = 0; i < F::N; i++) g.l[i] = FP::ROOT[i];
  for (int i = (int)lg; i < FP::TWO_ADICITY; i++) g = F::sqr(g);
  // F::from(size): canonical integer -> Montgomery
  F sz = F::zero();
  sz.l[0] = (uint32_t)size;
  sz.l[1] = (uint32_t)(size >> 32);
  sz = F::to_mont(sz);
  F one = F::one();
  sz.store(out->size_as_field_element);
  host_inverse(sz).store(out->size_inv);
  g.store(out->group_gen);
  host_inverse(g).store(out->group_gen_inv);
  one.store(out->offset);
  one.store(out->offset_inv);
  one.store(out->offset_pow_size);
  return 0;
}
template <class FP>
int domain_coset(const ark_hip_radix2_domain* dom, const uint64_t* offset, ark_hip_radix2_domain* out) {
  typedef Fp<FP> F;
  F h = F::load(offset);
  if (h.is_zero()) return ARK_HIP_ERR_ARG;  // inverse() -> None
  ark_hip_radix2_domain d = *dom;
  h.store(d.offset);
  host_inverse(h).store(d.offset_inv);
  uint64_t e[1] = {dom->size};
  host_pow<FP>(h, e, 1).store(d.offset_pow_size);
  *out = d;
  return 0;
}


template <class FP>
int fft_entry(Context* c, const ark_hip_radix2_domain* dom, void* d_data, int inverse, int zlog, hipStream_t st) {
  const bool coset = !host_is_one<FP>(dom->offset);
  FftTimings* tm = (c->fft_timing && st == c->stream) ? &c->fft_tm : nullptr;
  int k = (int)dom->log_size_of_group;
  if (dom->size != ((uint64_t)1 << k)) return ARK_HIP_ERR_ARG;
  if (!inverse) {
    // fft.rs:74-79: distribute_powers(offset) then DIF + derange
    return fft_dispatch(FP::ID, c->fft, d_data, k, dom->group_gen, coset ? dom->offset : nullptr, nullptr, nullptr, zlog,
                        st, tm);
  }
  // fft.rs:81-88: transform with group_gen_inv, then x[i] *= size_inv * offset_inv^i
  return fft_dispatch(FP::ID, c->fft, d_data, k, dom->group_gen_inv, nullptr, coset ? dom->offset_inv : nullptr,
                      dom->size_inv, 0, st, tm);
}

inline int fft_any(Context* c, int field, const ark_hip_radix2_domain* dom, void* d_data, int inverse, int zlog,
            hipStream_t st = nullptr) {
  if (!st) st = c->stream;
  switch (field) {
    case ARK_HIP_BN254_FR: return fft_entry<BN254_FR>(c, dom, d_data, inverse, zlog, st);
    case ARK_HIP_BLS12_381_FR: return fft_entry<BLS12_381_FR>(c, dom, d_data, inverse, zlog, st);
    case ARK_HIP_BLS12_377_FR: return fft_entry<BLS12_377_FR>(c, dom, d_data, inverse, zlog, st);
  }
  return ARK_HIP_ERR_ARG;
}

// Stage skipping of the degree-aware path (radix2/mod.rs:141, fft.rs:29-71): with num_coeffs * 4 <= size the first
// log2(size / next_pow2(num_coeffs)) stages only copy; returns that count (0: plain transform).
inline int degree_aware_zlog(const ark_hip_radix2_domain* dom, size_t num_coeffs) {
  if (num_coeffs == 0 || num_coeffs * 4 > dom->size) return 0;
  size_t d = 2;  // at least two input elements are read (the last stage is always executed)
  while (d < num_coeffs) d <<= 1;
  int z = 0;
  while ((d << z) < dom->size) z++;
  return z;
}

inline size_t field_bytes(int field) { return (field == ARK_HIP_BLS12_381_FQ || field == ARK_HIP_BLS12_377_FQ) ? 48 : 32; }

}  // namespace capi
}  // namespace arkhip
