// Exercises include/ark_hip.hpp (the C++ mirror of VariableBaseMSM / Radix2EvaluationDomain) on the GPU and checks
// it against the oracle.  Mirrors the reference's own tests: test_var_base_msm (test-templates/src/msm.rs:17-32),
// the Err(min_len) contract, test_fft_correctness / round trips (poly/src/domain/radix2/mod.rs:351-391).
// Built and run by tests/test_gpu_cpp_mirror.py (g++, links libark_hip.so and the oracle).
#include <cstdio>
#include <cstring>
#include "ark_hip.hpp"
extern "C" {
#include "ark_oracle.h"
}
using namespace ark_hip;

static int fails = 0;
#define EXPECT(c, msg) do { if (!(c)) { std::printf("FAIL: %s\n", msg); fails++; } } while (0)

template <class Curve>
void check_msm(const char* name, size_t n) {
  using M = VariableBaseMSM<Curve>;
  const uint64_t a4[4] = {0xA11CE, 1, 2, 0}, b4[4] = {0xB0B, 3, 0, 0};
  std::vector<typename Curve::AffineT> bases(n);
  ark_oracle_gen_bases(Curve::ID, a4, b4, n, reinterpret_cast<uint64_t*>(bases.data()));
  std::vector<BigInt4> big(n);
  std::vector<Fr> mont(n);
  ark_oracle_gen_scalars(Curve::SCALAR_FIELD, 17, n, 0, reinterpret_cast<uint64_t*>(big.data()));
  ark_oracle_field_op(Curve::SCALAR_FIELD, 8, reinterpret_cast<const uint64_t*>(big.data()), nullptr,
                      reinterpret_cast<uint64_t*>(mont.data()), n);  // from_bigint
  typename Curve::ProjectiveT ref;
  ark_oracle_msm(Curve::ID, reinterpret_cast<const uint64_t*>(bases.data()), reinterpret_cast<const uint64_t*>(big.data()),
                 n, 2, 4, reinterpret_cast<uint64_t*>(&ref));
  typename Curve::AffineT ref_aff;
  ark_oracle_to_affine(Curve::ID, reinterpret_cast<const uint64_t*>(&ref), reinterpret_cast<uint64_t*>(&ref_aff), 1);

  auto r1 = M::msm(bases, mont);
  EXPECT(r1.ok && M::into_affine(r1.value) == ref_aff, name);
  EXPECT(M::into_affine(M::msm_bigint(bases, big)) == ref_aff, name);
  // length mismatch -> Err(min_len), nothing computed
  std::vector<Fr> shorter(mont.begin(), mont.begin() + n / 2);
  auto r2 = M::msm(bases, shorter);
  EXPECT(!r2.ok && r2.min_len == n / 2, "Err(min_len)");
  // msm_unchecked truncates
  typename Curve::ProjectiveT ref2;
  ark_oracle_msm(Curve::ID, reinterpret_cast<const uint64_t*>(bases.data()), reinterpret_cast<const uint64_t*>(big.data()),
                 n / 2, 2, 4, reinterpret_cast<uint64_t*>(&ref2));
  typename Curve::AffineT ref2_aff;
  ark_oracle_to_affine(Curve::ID, reinterpret_cast<const uint64_t*>(&ref2), reinterpret_cast<uint64_t*>(&ref2_aff), 1);
  EXPECT(M::into_affine(M::msm_unchecked(bases, shorter)) == ref2_aff, "msm_unchecked truncation");
  // empty -> identity
  EXPECT(M::into_affine(M::msm_bigint({}, {})).is_zero(), "empty msm");
  // the same sums over a prepared (resident) base set, synchronous and as two jobs in flight
  {
    PreparedBases<Curve> pb(bases);
    auto p1 = pb.msm(mont);
    EXPECT(p1.ok && M::into_affine(p1.value) == ref_aff, "prepared msm");
    EXPECT(M::into_affine(pb.msm_bigint(big)) == ref_aff, "prepared msm_bigint");
    auto p2 = pb.msm(shorter);
    EXPECT(!p2.ok && p2.min_len == n / 2, "prepared Err(min_len)");
    EXPECT(M::into_affine(pb.msm_unchecked(shorter)) == ref2_aff, "prepared msm_unchecked truncation");
    std::vector<BigInt4> half(big.begin(), big.begin() + n / 2);
    auto j1 = pb.msm_bigint_async(big);
    auto j2 = pb.msm_bigint_async(half);
    EXPECT(M::into_affine(j2.wait()) == ref2_aff, "job 2");
    EXPECT(M::into_affine(j1.wait()) == ref_aff, "job 1");
  }
  // msm_chunks == msm (test-templates style: streams of equal length), small steps
  EXPECT(M::into_affine(M::msm_chunks(bases, mont, 300)) == ref_aff, "msm_chunks");
  // ChunkedPippenger (test_chunked_pippenger, test-templates/src/msm.rs:112-133)
  {
    auto cp = ChunkedPippenger<Curve>::with_size(n / 7 + 1);
    for (size_t i = 0; i < n; i++) cp.add(bases[i], big[i]);
    EXPECT(M::into_affine(cp.finalize()) == ref_aff, "ChunkedPippenger");
    EXPECT(M::into_affine(ChunkedPippenger<Curve>(4).finalize()).is_zero(), "empty ChunkedPippenger");
  }
  // HashMapPippenger (test_hashmap_pippenger, msm.rs:135-156): every base added twice with split scalars
  {
    auto add_fr = [](const Fr& a, const Fr& b) {
      Fr r;
      ark_oracle_field_op(Curve::SCALAR_FIELD, 0, a.limbs.data(), b.limbs.data(), r.limbs.data(), 1);  // host Fr addition
      return r;
    };
    HashMapPippenger<Curve, decltype(add_fr)> hp(n / 3 + 1, add_fr);
    std::vector<Fr> part(n);
    ark_oracle_gen_scalars(Curve::SCALAR_FIELD, 99, n, 1, reinterpret_cast<uint64_t*>(part.data()));
    std::vector<Fr> rest(n);
    ark_oracle_field_op(Curve::SCALAR_FIELD, 1, reinterpret_cast<const uint64_t*>(mont.data()),
                        reinterpret_cast<const uint64_t*>(part.data()), reinterpret_cast<uint64_t*>(rest.data()), n);  // mont - part
    for (size_t i = 0; i < n; i++) hp.add(bases[i], part[i]);
    for (size_t i = 0; i < n; i++) hp.add(bases[i], rest[i]);
    EXPECT(M::into_affine(hp.finalize()) == ref_aff, "HashMapPippenger");
  }
  std::printf("%s msm n=%zu checked\n", name, n);
}

template <int FIELD>
void check_fft(const char* name, unsigned log_n) {
  using D = Radix2EvaluationDomain<FIELD>;
  size_t n = (size_t)1 << log_n;
  auto dom = D::new_(n - 3);  // next power of two
  EXPECT(dom && dom->size() == n && dom->log_size_of_group() == log_n, "domain size");
  std::vector<Fr> x(n - 3);
  ark_oracle_gen_scalars(FIELD, 5, n - 3, 1, reinterpret_cast<uint64_t*>(x.data()));
  std::vector<Fr> padded(x);
  padded.resize(n);
  std::vector<Fr> ref(padded);
  ark_oracle_fft(FIELD, reinterpret_cast<uint64_t*>(ref.data()), log_n, nullptr, 0, 4);
  auto y = dom->fft(x);  // zero-extended like the reference
  EXPECT(y.size() == n && std::memcmp(y.data(), ref.data(), n * 32) == 0, name);
  auto back = dom->ifft(y);
  EXPECT(std::memcmp(back.data(), padded.data(), n * 32) == 0, "ifft(fft(x)) == x");
  // coset with offset = GENERATOR (poly/benches/fft.rs:107)
  Fr gen;
  ark_oracle_field_const(FIELD, 3, gen.limbs.data());
  auto coset = dom->get_coset(gen);
  EXPECT(coset.has_value(), "get_coset");
  std::vector<Fr> cref(padded);
  ark_oracle_fft(FIELD, reinterpret_cast<uint64_t*>(cref.data()), log_n, gen.limbs.data(), 0, 4);
  auto cy = coset->fft(x);
  EXPECT(std::memcmp(cy.data(), cref.data(), n * 32) == 0, "coset fft");
  Fr zero;
  EXPECT(!dom->get_coset(zero).has_value(), "zero offset -> None");
  // short input: degree-aware path (radix2/mod.rs:141), same output as the zero-padded transform
  std::vector<Fr> sx(x.begin(), x.begin() + n / 8 + 1), spad(sx);
  spad.resize(n);
  ark_oracle_fft(FIELD, reinterpret_cast<uint64_t*>(spad.data()), log_n, nullptr, 0, 4);
  auto sy = dom->fft(sx);
  EXPECT(sy.size() == n && std::memcmp(sy.data(), spad.data(), n * 32) == 0, "degree-aware fft");
  std::printf("%s fft 2^%u checked\n", name, log_n);
}

// evaluate_over_domain -> pointwise -> interpolate on DEVICE-RESIDENT vectors: one upload per factor,
// one download, against the oracle's transforms and host field arithmetic
template <int FIELD>
void check_device_chain(const char* name, unsigned log_n) {
  using D = Radix2EvaluationDomain<FIELD>;
  using V = DeviceVec<FIELD>;
  const size_t n = (size_t)1 << log_n;
  auto dom = D::new_(n);
  EXPECT(dom && dom->size() == n, "domain");
  const size_t la = n / 2 - 5, lb = n / 8 + 3;   // b takes the degree-aware path, a * b fits the domain
  std::vector<Fr> a(la), b(lb);
  ark_oracle_gen_scalars(FIELD, 31, la, 1, reinterpret_cast<uint64_t*>(a.data()));
  ark_oracle_gen_scalars(FIELD, 32, lb, 1, reinterpret_cast<uint64_t*>(b.data()));
  Fr k;
  ark_oracle_gen_scalars(FIELD, 33, 1, 1, k.limbs.data());
  // oracle: ea = FFT(a), eb = FFT(b); e = ((ea * eb + eb) * k - ea); c = IFFT(e); and -c
  std::vector<Fr> ea(a), eb(b);
  ea.resize(n);
  eb.resize(n);
  ark_oracle_fft(FIELD, reinterpret_cast<uint64_t*>(ea.data()), log_n, nullptr, 0, 4);
  ark_oracle_fft(FIELD, reinterpret_cast<uint64_t*>(eb.data()), log_n, nullptr, 0, 4);
  std::vector<Fr> e(n), kk(n, k);
  auto u = [](std::vector<Fr>& v) { return reinterpret_cast<uint64_t*>(v.data()); };
  ark_oracle_field_op(FIELD, 2, u(ea), u(eb), u(e), n);
  ark_oracle_field_op(FIELD, 0, u(e), u(eb), u(e), n);
  ark_oracle_field_op(FIELD, 2, u(e), u(kk), u(e), n);
  ark_oracle_field_op(FIELD, 1, u(e), u(ea), u(e), n);
  std::vector<Fr> c(e);
  ark_oracle_fft(FIELD, u(c), log_n, nullptr, 1, 4);
  // device: two uploads, everything else resident, one download
  auto da = evaluate_over_domain(V::from_vec(a), *dom);
  auto db = evaluate_over_domain(V::from_vec(b), *dom);
  EXPECT(da.evals.len() == n && db.evals.len() == n, "evaluations have the domain's size");
  auto keep_a = da.evals.clone();
  da *= db;
  da += db;
  da.evals *= k;
  DeviceEvaluations<FIELD> ea_dev{std::move(keep_a), *dom};
  da -= ea_dev;
  auto e_dev = da.evals.clone();
  {  // Evaluations /= Evaluations against the oracle's inversion (mod.rs:153-163)
    auto q = e_dev.clone();
    q /= db.evals;
    std::vector<Fr> inv(n), want(n);
    ark_oracle_field_op(FIELD, 6, u(eb), nullptr, u(inv), n);
    ark_oracle_field_op(FIELD, 2, u(e), u(inv), u(want), n);
    auto got_q = q.to_vec();
    EXPECT(std::memcmp(got_q.data(), want.data(), n * 32) == 0, "pointwise division on the device == oracle");
  }
  auto coeffs = std::move(da).interpolate();
  auto got_e = e_dev.to_vec();
  EXPECT(std::memcmp(got_e.data(), e.data(), n * 32) == 0, "pointwise chain on the device == oracle");
  auto got_c = coeffs.to_vec();
  EXPECT(got_c.size() == n && std::memcmp(got_c.data(), c.data(), n * 32) == 0, "interpolate on the device == oracle");
  coeffs.negate();
  std::vector<Fr> negc(n), zero(n);
  ark_oracle_field_op(FIELD, 1, u(zero), u(c), u(negc), n);
  auto got_n = coeffs.to_vec();
  EXPECT(std::memcmp(got_n.data(), negc.data(), n * 32) == 0, "negate");
  // Vec semantics: resize with zeros keeps the prefix, truncation keeps it too; a fresh vector is zero
  coeffs.resize_zeroed(n + 7);
  auto grown = coeffs.to_vec();
  EXPECT(grown.size() == n + 7 && std::memcmp(grown.data(), negc.data(), n * 32) == 0 && grown[n + 6].is_zero() && grown[n].is_zero(),
         "resize_zeroed grows with zeros");
  coeffs.resize_zeroed(5);
  EXPECT(coeffs.to_vec().size() == 5, "resize_zeroed truncates");
  V z(9);
  auto zv = z.to_vec();
  bool allz = true;
  for (auto& x : zv) allz = allz && x.is_zero();
  EXPECT(allz, "DeviceVec(len) is zero");
  bool threw = false;
  try { V w(3); w += z; } catch (const Error&) { threw = true; }
  EXPECT(threw, "unequal lengths are refused");
  std::printf("%s device-resident chain 2^%u checked\n", name, log_n);
}

// fft_in_place for T = Projective (poly/src/test.rs:57): coefficients P_i = [a + i b] G, so FFT(P)_j = [FFT_F(a + i b)_j] G --
// the field transform and the scalar multiplications from the oracle
void check_group_fft() {
  using Curve = Bls12_381G1;
  using M = VariableBaseMSM<Curve>;
  using D = Radix2EvaluationDomain<ARK_HIP_BLS12_381_FR>;
  const unsigned log_n = 4;
  const size_t n = (size_t)1 << log_n, have = n - 3;   // resized to the domain size with identities
  uint64_t a4[4] = {0xA11CE, 1, 2, 0}, b4[4] = {0xB0B, 3, 0, 0};
  std::vector<Curve::AffineT> aff(have);
  ark_oracle_gen_bases(Curve::ID, a4, b4, have, reinterpret_cast<uint64_t*>(aff.data()));
  uint64_t one[6];
  ark_oracle_field_const(ARK_HIP_BLS12_381_FQ, 1, one);
  std::vector<Curve::ProjectiveT> pts(have);
  for (size_t i = 0; i < have; i++) {
    pts[i].x = aff[i].x;
    pts[i].y = aff[i].y;
    for (int k = 0; k < 6; k++) pts[i].z.limbs[k] = one[k];
  }
  // scalars a + i b as canonical integers (small enough here: no reduction), zero beyond `have`
  std::vector<Fr> canon(n), mont(n), out(n);
  for (size_t i = 0; i < have; i++) {
    unsigned __int128 c = 0;
    for (int k = 0; k < 4; k++) {
      c += (unsigned __int128)a4[k] + (unsigned __int128)b4[k] * i;
      canon[i].limbs[k] = (uint64_t)c;
      c >>= 64;
    }
  }
  ark_oracle_field_op(ARK_HIP_BLS12_381_FR, 8, reinterpret_cast<uint64_t*>(canon.data()), nullptr, reinterpret_cast<uint64_t*>(mont.data()), n);
  ark_oracle_fft(ARK_HIP_BLS12_381_FR, reinterpret_cast<uint64_t*>(mont.data()), log_n, nullptr, 0, 2);
  ark_oracle_field_op(ARK_HIP_BLS12_381_FR, 7, reinterpret_cast<uint64_t*>(mont.data()), nullptr, reinterpret_cast<uint64_t*>(out.data()), n);
  uint64_t gen[12];
  ark_oracle_curve_generator(Curve::ID, gen);
  auto dom = D::new_(n);
  EXPECT(dom.has_value(), "domain");
  dom->fft_group_in_place<Curve>(pts);
  EXPECT(pts.size() == n, "resized to the domain");
  bool ok = true;
  for (size_t j = 0; j < n; j++) {
    Curve::ProjectiveT ref;
    ark_oracle_scalar_mul(Curve::ID, gen, out[j].limbs.data(), reinterpret_cast<uint64_t*>(&ref));
    ok = ok && (M::into_affine(pts[j]) == M::into_affine(ref));
  }
  EXPECT(ok, "group fft == [field fft of the discrete logs] G");
  std::printf("BLS12_381_G1 group fft 2^%u checked\n", log_n);
}

// the library's RCCL communicator from a compiled host: a world of one on the test box's single GPU (dlopen of librccl,
// ncclCommInitRank, the sharded MSM and FFT entries through it)
void check_communicator() {
  using M = VariableBaseMSM<Bls12_381G1>;
  const size_t n = 1500;
  std::vector<Bls12_381G1::AffineT> bases(n);
  std::vector<BigInt4> big(n);
  uint64_t a4[4] = {0xA11CE, 1, 2, 0}, b4[4] = {0xB0B, 3, 0, 0};
  ark_oracle_gen_bases(Bls12_381G1::ID, a4, b4, n, reinterpret_cast<uint64_t*>(bases.data()));
  ark_oracle_gen_scalars(Bls12_381G1::SCALAR_FIELD, 21, n, 0, reinterpret_cast<uint64_t*>(big.data()));
  auto ref_aff = M::into_affine(M::msm_bigint(bases, big));
  void *db = nullptr, *ds = nullptr, *dx = nullptr;
  check(ark_hip_malloc(n * sizeof(bases[0]), &db), "malloc");
  check(ark_hip_malloc(n * 32, &ds), "malloc");
  check(ark_hip_memcpy_h2d(db, bases.data(), n * sizeof(bases[0])), "h2d");
  check(ark_hip_memcpy_h2d(ds, big.data(), n * 32), "h2d");
  {
    Communicator comm(Communicator::unique_id(), 0, 1);
    EXPECT(comm.rank() == 0 && comm.world() == 1, "communicator geometry");
    auto got = comm.msm_bigint_sharded<Bls12_381G1>(db, ds, n);
    EXPECT(M::into_affine(got) == ref_aff, "sharded msm through the communicator");
    using D = Radix2EvaluationDomain<ARK_HIP_BLS12_381_FR>;
    auto dom = D::new_(1 << 11);
    std::vector<Fr> x(1 << 11), ref;
    ark_oracle_gen_scalars(ARK_HIP_BLS12_381_FR, 8, x.size(), 1, reinterpret_cast<uint64_t*>(x.data()));
    ref = x;
    ark_oracle_fft(ARK_HIP_BLS12_381_FR, reinterpret_cast<uint64_t*>(ref.data()), 11, nullptr, 0, 2);
    check(ark_hip_malloc(x.size() * 32, &dx), "malloc");
    check(ark_hip_memcpy_h2d(dx, x.data(), x.size() * 32), "h2d");
    dom->fft_sharded_in_place_device(dx);
    check(ark_hip_synchronize(), "sync");
    check(ark_hip_memcpy_d2h(x.data(), dx, x.size() * 32), "d2h");
    EXPECT(std::memcmp(x.data(), ref.data(), x.size() * 32) == 0, "sharded fft through the communicator");
  }
  int r = -1, w = -1;
  check(ark_hip_comm_info(&r, &w), "info");
  EXPECT(r == 0 && w == 1, "communicator gone after scope");
  check(ark_hip_free(db), "free");
  check(ark_hip_free(ds), "free");
  check(ark_hip_free(dx), "free");
  std::printf("communicator (world of one) checked\n");
}

int main() {
  if (ark_hip_device_count() <= 0) { std::printf("no GPU\n"); return 2; }
  check_msm<Bls12_381G1>("BLS12_381_G1", 2000);
  check_msm<Bn254G1>("BN254_G1", 1024);
  check_msm<Bls12_377G2>("BLS12_377_G2", 300);
  check_fft<ARK_HIP_BLS12_381_FR>("BLS12_381_FR", 12);
  check_communicator();
  check_fft<ARK_HIP_BN254_FR>("BN254_FR", 9);
  check_group_fft();
  check_device_chain<ARK_HIP_BLS12_381_FR>("BLS12_381_FR", 20);
  check_device_chain<ARK_HIP_BLS12_377_FR>("BLS12_377_FR", 11);
  check_device_chain<ARK_HIP_BN254_FR>("BN254_FR", 5);
  EXPECT(!Radix2EvaluationDomain<ARK_HIP_BN254_FR>::new_(((size_t)1 << 28) + 1).has_value(), "too large -> None");
  std::printf(fails ? "FAILED (%d)\n" : "all ok\n", fails);
  return fails ? 1 : 0;
}
