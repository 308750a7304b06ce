// ark_hip.hpp -- C++17 host-side mirror of the two reference interfaces that libark_hip.so replaces, written over
// the C ABI of ark_hip.h (header only).  The reference is Rust; this image has no Rust toolchain, so this header is
// the compiled-language host side (the same surface exists as Rust source in rust/ and as a Python mirror in
// algebra_amd/).  Names, argument meaning and error behaviour follow the reference:
//
//   ark_hip::VariableBaseMSM<Curve>      <- ark_ec::VariableBaseMSM for Projective<P>
//        msm(bases, scalars)   -> Result: Err(min_len) when the lengths differ   (ec/src/scalar_mul/variable_base/mod.rs:67-78,
//                                                                                  short_weierstrass/mod.rs:112-119)
//        msm_unchecked(...)    -> truncates to the shorter input                   (mod.rs:59-64)
//        msm_bigint(...)       -> scalars are canonical BigInt<4>                  (mod.rs:80-85)
//        msm_chunks(...)       -> streams aligned at their end, 2^20-pair steps    (mod.rs:119-150)
//   ark_hip::ChunkedPippenger<Curve>, ark_hip::HashMapPippenger<Curve>             (stream_pippenger.rs:10-128)
//   ark_hip::PreparedBases<Curve>        a fixed base set resident on the GPU (ark_hip_msm_bases_*): msm / msm_unchecked /
//                                        msm_bigint over it, synchronous or as ark_hip::MsmJob (asynchronous)
//   ark_hip::Radix2EvaluationDomain<F>   <- ark_poly::Radix2EvaluationDomain<F> / EvaluationDomain<F>
//        new_(num_coeffs) -> optional (None when log2(size) > TWO_ADICITY)         (poly/src/domain/radix2/mod.rs:55-83)
//        get_coset, size, log_size_of_group, size_inv, group_gen, group_gen_inv, coset_offset, coset_offset_inv,
//        coset_offset_pow_size, fft, fft_in_place, ifft, ifft_in_place             (radix2/mod.rs:85-153, domain/mod.rs:92-112)
//   ark_hip::DeviceVec<F>, ark_hip::DeviceEvaluations<F>, ark_hip::evaluate_over_domain   (round 5)
//        coefficient / evaluation vectors that stay in HBM: evaluate_over_domain -> pointwise +, -, * -> interpolate
//        (polynomial/univariate/mod.rs:305-360, evaluations/univariate/mod.rs:40-50, :104-180) with ONE upload and ONE
//        download instead of two PCIe crossings per transform
//
// Element types are plain limb arrays in the reference's in-memory layout (Montgomery, little-endian u64 limbs).
#pragma once
#include <array>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <map>
#include <optional>
#include <stdexcept>
#include <utility>
#include <vector>
#include "ark_hip.h"

namespace ark_hip {

// ---- element types ------------------------------------------------------------------------------------------------
template <int WORDS>
struct FieldElement {  // Fp (WORDS = 4 or 6) or Fp2 (WORDS = 12): c0 | c1
  std::array<uint64_t, WORDS> limbs{};
  bool operator==(const FieldElement& o) const { return limbs == o.limbs; }
  bool is_zero() const {
    for (auto l : limbs)
      if (l) return false;
    return true;
  }
};
using Fr = FieldElement<4>;      // scalar-field element (Montgomery)
using BigInt4 = FieldElement<4>; // canonical 256-bit integer (PrimeField::BigInt)

template <int WORDS>
struct Affine {  // short_weierstrass::Affine with ZeroFlag = (): identity is (0, 0)
  FieldElement<WORDS> x, y;
  bool is_zero() const { return x.is_zero() && y.is_zero(); }
  bool operator==(const Affine& o) const { return x == o.x && y == o.y; }
};
template <int WORDS>
struct Projective {  // Jacobian (x, y, z)
  FieldElement<WORDS> x, y, z;
};

struct Error : std::runtime_error {
  int code;
  Error(int c, const char* what) : std::runtime_error(what), code(c) {}
};
inline void check(int rc, const char* what) {
  if (rc != 0) throw Error(rc, what);
}

// ---- curve tags -----------------------------------------------------------------------------------------------------
template <int CURVE_ID, int FE_WORDS, int SCALAR_FIELD_ID>
struct CurveTag {
  static constexpr int ID = CURVE_ID;
  static constexpr int WORDS = FE_WORDS;
  static constexpr int SCALAR_FIELD = SCALAR_FIELD_ID;
  using AffineT = Affine<FE_WORDS>;
  using ProjectiveT = Projective<FE_WORDS>;
};
using Bn254G1 = CurveTag<ARK_HIP_BN254_G1, 4, ARK_HIP_BN254_FR>;
using Bls12_381G1 = CurveTag<ARK_HIP_BLS12_381_G1, 6, ARK_HIP_BLS12_381_FR>;
using Bls12_377G1 = CurveTag<ARK_HIP_BLS12_377_G1, 6, ARK_HIP_BLS12_377_FR>;
using Bls12_377G2 = CurveTag<ARK_HIP_BLS12_377_G2, 12, ARK_HIP_BLS12_377_FR>;
using Bls12_381G2 = CurveTag<ARK_HIP_BLS12_381_G2, 12, ARK_HIP_BLS12_381_FR>;

// VariableBaseMSM<..>::msm / msm_bigint are functions of their two spans (a cached device copy of the bases is
// re-validated on every call by a hash of the span's full content: base_cache_config below).
// ResidentBases pins a base vector on the GPU for its lifetime: while it lives the caller does not modify the vector, and
// every msm / msm_bigint / msm_u* whose bases lie inside it runs against the resident copy (ark_hip_msm_bases_pin).
template <class Curve>
class ResidentBases {
 public:
  ResidentBases(const typename Curve::AffineT* bases, size_t n) : p_(bases), n_(n) {
    check(ark_hip_msm_bases_pin(Curve::ID, reinterpret_cast<const uint64_t*>(bases), n), "ark_hip_msm_bases_pin");
  }
  explicit ResidentBases(const std::vector<typename Curve::AffineT>& bases) : ResidentBases(bases.data(), bases.size()) {}
  ResidentBases(const ResidentBases&) = delete;
  ResidentBases& operator=(const ResidentBases&) = delete;
  ~ResidentBases() { (void)ark_hip_msm_bases_unpin(Curve::ID, reinterpret_cast<const uint64_t*>(p_), n_); }

 private:
  const typename Curve::AffineT* p_;
  size_t n_;
};
// The verified cache (ark_hip_msm_cache_* in ark_hip.h; on by default): base vectors passed again at the same address stay
// on the GPU, validated on every call by a hash of their full content.
struct BaseCacheStats { uint64_t entries, bytes, hits, misses, refreshed, evicted, pinned, pinned_hits; };
inline void base_cache_config(long long budget_bytes = -1, int auto_prepare_after = -1) {
  check(ark_hip_msm_cache_config(budget_bytes, auto_prepare_after), "ark_hip_msm_cache_config");
}
inline void base_cache_clear() { check(ark_hip_msm_cache_clear(), "ark_hip_msm_cache_clear"); }
inline BaseCacheStats base_cache_stats() {
  uint64_t o[8];
  check(ark_hip_msm_cache_stats(o), "ark_hip_msm_cache_stats");
  return BaseCacheStats{o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7]};
}

// Result<Projective, usize> of VariableBaseMSM::msm
template <class T>
struct MsmResult {
  bool ok;
  T value;         // valid when ok
  size_t min_len;  // the reference's Err payload when !ok
};

template <class Curve>
struct VariableBaseMSM {
  using A = typename Curve::AffineT;
  using P = typename Curve::ProjectiveT;
  static_assert(sizeof(A) == 2 * Curve::WORDS * 8 && sizeof(P) == 3 * Curve::WORDS * 8, "layout must be packed limbs");

  // msm: length check, then msm_unchecked (scalars are Fr elements in Montgomery form)
  static MsmResult<P> msm(const std::vector<A>& bases, const std::vector<Fr>& scalars) {
    if (bases.size() != scalars.size()) return {false, P{}, bases.size() < scalars.size() ? bases.size() : scalars.size()};
    return {true, msm_unchecked(bases, scalars), 0};
  }
  static P msm_unchecked(const std::vector<A>& bases, const std::vector<Fr>& scalars) {
    return run(bases.data(), reinterpret_cast<const uint64_t*>(scalars.data()),
               bases.size() < scalars.size() ? bases.size() : scalars.size(), 1);
  }
  static P msm_bigint(const std::vector<A>& bases, const std::vector<BigInt4>& bigints) {
    return run(bases.data(), reinterpret_cast<const uint64_t*>(bigints.data()),
               bases.size() < bigints.size() ? bases.size() : bigints.size(), 0);
  }
  // msm_chunks: Fr scalars, streams aligned at their END, steps of 2^20 pairs (0) or `step`
  static P msm_chunks(const std::vector<A>& bases, const std::vector<Fr>& scalars, size_t step = 0) {
    if (scalars.size() > bases.size()) throw Error(ARK_HIP_ERR_ARG, "scalars_stream.len() <= bases_stream.len()");
    P out;
    check(ark_hip_msm_sw_chunks(Curve::ID, reinterpret_cast<const uint64_t*>(bases.data()), bases.size(),
                                reinterpret_cast<const uint64_t*>(scalars.data()), scalars.size(), step,
                                reinterpret_cast<uint64_t*>(&out)),
          "ark_hip_msm_sw_chunks");
    return out;
  }
  // Projective: Sum (group.rs:659-663)
  static P sum(const std::vector<P>& pts) {
    P out;
    check(ark_hip_sw_sum(Curve::ID, reinterpret_cast<const uint64_t*>(pts.data()), pts.size(),
                         reinterpret_cast<uint64_t*>(&out)),
          "ark_hip_sw_sum");
    return out;
  }
  // CurveGroup::into_affine
  static A into_affine(const P& p) {
    A out;
    check(ark_hip_sw_into_affine(Curve::ID, reinterpret_cast<const uint64_t*>(&p), 1, reinterpret_cast<uint64_t*>(&out)),
          "ark_hip_sw_into_affine");
    return out;
  }

 private:
  static P run(const A* bases, const uint64_t* scalars, size_t n, int montgomery) {
    P out;
    check(ark_hip_msm_sw(Curve::ID, reinterpret_cast<const uint64_t*>(bases), scalars, n, montgomery,
                         reinterpret_cast<uint64_t*>(&out)),
          "ark_hip_msm_sw");
    return out;
  }
};

// ---- an MSM in flight (ark_hip_msm_job) -------------------------------------------------------------------------------
template <class Curve>
class MsmJob {
 public:
  using P = typename Curve::ProjectiveT;
  explicit MsmJob(ark_hip_msm_job* j) : j_(j) {}
  MsmJob(MsmJob&& o) noexcept : j_(o.j_) { o.j_ = nullptr; }
  MsmJob(const MsmJob&) = delete;
  ~MsmJob() {
    if (j_) (void)ark_hip_msm_wait(j_, nullptr);  // never leak a device job slot
  }
  P wait() {
    P out;
    ark_hip_msm_job* j = j_;
    j_ = nullptr;
    check(ark_hip_msm_wait(j, reinterpret_cast<uint64_t*>(&out)), "ark_hip_msm_wait");
    return out;
  }

 private:
  ark_hip_msm_job* j_;
};

// ---- a fixed base set (SRS) resident on the GPU with its per-window multiples -----------------------------------------
template <class Curve>
class PreparedBases {
 public:
  using A = typename Curve::AffineT;
  using P = typename Curve::ProjectiveT;
  explicit PreparedBases(const std::vector<A>& bases) : n_(bases.size()) {
    check(ark_hip_msm_bases_prepare(Curve::ID, reinterpret_cast<const uint64_t*>(bases.data()), bases.size(), &h_),
          "ark_hip_msm_bases_prepare");
  }
  PreparedBases(const PreparedBases&) = delete;
  ~PreparedBases() { (void)ark_hip_msm_bases_free(h_); }
  size_t size() const { return n_; }
  MsmResult<P> msm(const std::vector<Fr>& scalars) const {
    if (scalars.size() != n_) return {false, P{}, scalars.size() < n_ ? scalars.size() : n_};
    return {true, run(reinterpret_cast<const uint64_t*>(scalars.data()), n_, 1), 0};
  }
  P msm_unchecked(const std::vector<Fr>& scalars) const {
    return run(reinterpret_cast<const uint64_t*>(scalars.data()), scalars.size() < n_ ? scalars.size() : n_, 1);
  }
  P msm_bigint(const std::vector<BigInt4>& bigints) const {
    return run(reinterpret_cast<const uint64_t*>(bigints.data()), bigints.size() < n_ ? bigints.size() : n_, 0);
  }
  // the scalars must outlive the job
  MsmJob<Curve> msm_bigint_async(const std::vector<BigInt4>& bigints) const {
    ark_hip_msm_job* j = nullptr;
    check(ark_hip_msm_prepared_async(h_, reinterpret_cast<const uint64_t*>(bigints.data()),
                                     bigints.size() < n_ ? bigints.size() : n_, 0, &j),
          "ark_hip_msm_prepared_async");
    return MsmJob<Curve>(j);
  }

 private:
  P run(const uint64_t* scalars, size_t n, int montgomery) const {
    P out;
    check(ark_hip_msm_prepared(h_, scalars, n, montgomery, reinterpret_cast<uint64_t*>(&out)), "ark_hip_msm_prepared");
    return out;
  }
  ark_hip_msm_bases* h_ = nullptr;
  size_t n_;
};

// ---- streaming accumulators (stream_pippenger.rs) ---------------------------------------------------------------------
template <class Curve>
class ChunkedPippenger {  // stream_pippenger.rs:10-66
 public:
  using A = typename Curve::AffineT;
  using P = typename Curve::ProjectiveT;
  explicit ChunkedPippenger(size_t max_msm_buffer) : buf_size_(max_msm_buffer) {
    bases_.reserve(max_msm_buffer);
    scalars_.reserve(max_msm_buffer);
  }
  static ChunkedPippenger with_size(size_t buf_size) { return ChunkedPippenger(buf_size); }
  void add(const A& base, const BigInt4& scalar) {
    bases_.push_back(base);
    scalars_.push_back(scalar);
    if (scalars_.size() == buf_size_) flush();
  }
  P finalize() {
    if (!scalars_.empty()) flush();
    return VariableBaseMSM<Curve>::sum(partials_);  // empty -> identity
  }

 private:
  void flush() {
    partials_.push_back(VariableBaseMSM<Curve>::msm_bigint(bases_, scalars_));
    bases_.clear();
    scalars_.clear();
  }
  size_t buf_size_;
  std::vector<A> bases_;
  std::vector<BigInt4> scalars_;
  std::vector<P> partials_;
};

// HashMapPippenger (stream_pippenger.rs:68-128): scalars of equal bases are added in Fr before the MSM.  The modular
// addition is supplied by the host (`AddFr`: Fr x Fr -> Fr on Montgomery residues), as the reference uses its own
// field type; a flush runs one MSM over the distinct bases with the Fr -> BigInt conversion on the device.
template <class Curve, class AddFr>
class HashMapPippenger {
 public:
  using A = typename Curve::AffineT;
  using P = typename Curve::ProjectiveT;
  HashMapPippenger(size_t max_msm_buffer, AddFr add) : buf_size_(max_msm_buffer), add_(std::move(add)) {}
  void add(const A& base, const Fr& scalar) {
    Key k;
    std::memcpy(k.data(), &base, sizeof(A));
    auto it = map_.find(k);
    if (it == map_.end()) map_.emplace(k, scalar);
    else it->second = add_(it->second, scalar);
    if (map_.size() == buf_size_) flush();
  }
  P finalize() {
    if (!map_.empty()) flush();
    return VariableBaseMSM<Curve>::sum(partials_);
  }

 private:
  using Key = std::array<unsigned char, sizeof(A)>;
  void flush() {
    std::vector<A> bases;
    std::vector<Fr> scalars;
    for (auto& kv : map_) {
      A a;
      std::memcpy(&a, kv.first.data(), sizeof(A));
      bases.push_back(a);
      scalars.push_back(kv.second);
    }
    partials_.push_back(VariableBaseMSM<Curve>::msm_unchecked(bases, scalars));
    map_.clear();
  }
  size_t buf_size_;
  AddFr add_;
  std::map<Key, Fr> map_;
  std::vector<P> partials_;
};

// ---- one process per GPU: the library's RCCL communicator ------------------------------------------------------------
// Rank 0 creates the id, the host ships its bytes to the other ranks (MPI_Bcast, a socket ...), every rank constructs a
// Communicator on its device (collective).  While it lives, msm_sharded / fft_sharded_in_place_device are collective
// calls; the reference's counterpart of the split is its own chunking by base range (variable_base/mod.rs:521-557).
class Communicator {
 public:
  using Id = std::array<unsigned char, ARK_HIP_COMM_ID_BYTES>;
  static Id unique_id() {
    Id id{};
    check(ark_hip_comm_unique_id(id.data()), "ark_hip_comm_unique_id");
    return id;
  }
  Communicator(const Id& id, int rank, int world) { check(ark_hip_comm_init(id.data(), rank, world), "ark_hip_comm_init"); }
  ~Communicator() { (void)ark_hip_comm_destroy(); }
  Communicator(const Communicator&) = delete;
  Communicator& operator=(const Communicator&) = delete;
  int rank() const { int r = 0, w = 1; check(ark_hip_comm_info(&r, &w), "ark_hip_comm_info"); return r; }
  int world() const { int r = 0, w = 1; check(ark_hip_comm_info(&r, &w), "ark_hip_comm_info"); return w; }
  // the sum over ALL ranks' (base, scalar) shards; d_bases / d_scalars: this rank's shard in device memory
  template <class Curve>
  typename Curve::ProjectiveT msm_bigint_sharded(const void* d_bases, const void* d_scalars, size_t n_local) const {
    typename Curve::ProjectiveT out;
    check(ark_hip_msm_sw_device_sharded(Curve::ID, d_bases, d_scalars, n_local, 0, reinterpret_cast<uint64_t*>(&out)),
          "ark_hip_msm_sw_device_sharded");
    return out;
  }
};

// ---- Radix2EvaluationDomain ------------------------------------------------------------------------------------------
template <int FIELD_ID>
class Radix2EvaluationDomain {
 public:
  // EvaluationDomain::new
  static std::optional<Radix2EvaluationDomain> new_(size_t num_coeffs) {
    Radix2EvaluationDomain d;
    int rc = ark_hip_radix2_domain_new(FIELD_ID, num_coeffs, &d.s_);
    if (rc == ARK_HIP_ERR_SIZE) return std::nullopt;
    check(rc, "ark_hip_radix2_domain_new");
    return d;
  }
  std::optional<Radix2EvaluationDomain> get_coset(const Fr& offset) const {
    Radix2EvaluationDomain d;
    int rc = ark_hip_radix2_domain_get_coset(FIELD_ID, &s_, offset.limbs.data(), &d.s_);
    if (rc == ARK_HIP_ERR_ARG) return std::nullopt;
    check(rc, "ark_hip_radix2_domain_get_coset");
    return d;
  }
  static std::optional<size_t> compute_size_of_domain(size_t num_coeffs) {
    auto d = new_(num_coeffs);
    if (!d) return std::nullopt;
    return d->size();
  }
  size_t size() const { return (size_t)s_.size; }
  uint64_t log_size_of_group() const { return s_.log_size_of_group; }
  Fr size_as_field_element() const { return fe(s_.size_as_field_element); }
  Fr size_inv() const { return fe(s_.size_inv); }
  Fr group_gen() const { return fe(s_.group_gen); }
  Fr group_gen_inv() const { return fe(s_.group_gen_inv); }
  Fr coset_offset() const { return fe(s_.offset); }
  Fr coset_offset_inv() const { return fe(s_.offset_inv); }
  Fr coset_offset_pow_size() const { return fe(s_.offset_pow_size); }

  // fft_in_place / ifft_in_place: the Vec is resized to the domain size (zero-extended) first, as in radix2/mod.rs:140-153
  // (a vector of at most size/4 coefficients takes the degree-aware path, radix2/mod.rs:141: only the coefficients
  // are uploaded and the leading butterfly stages are skipped)
  void fft_in_place(std::vector<Fr>& coeffs) const {
    const size_t len = coeffs.size();
    resize(coeffs);
    check(ark_hip_fft_in_place_degree_aware(FIELD_ID, &s_, reinterpret_cast<uint64_t*>(coeffs.data()), len),
          "ark_hip_fft_in_place_degree_aware");
  }
  void ifft_in_place(std::vector<Fr>& evals) const {
    resize(evals);
    check(ark_hip_ifft_in_place(FIELD_ID, &s_, reinterpret_cast<uint64_t*>(evals.data())), "ark_hip_ifft_in_place");
  }
  std::vector<Fr> fft(const std::vector<Fr>& coeffs) const {
    std::vector<Fr> v(coeffs);
    fft_in_place(v);
    return v;
  }
  std::vector<Fr> ifft(const std::vector<Fr>& evals) const {
    std::vector<Fr> v(evals);
    ifft_in_place(v);
    return v;
  }
  // Several polynomials over this domain at once, each size() elements in THIS GPU's memory, transformed in place with up
  // to three transforms in flight (no counterpart in the reference: its callers loop over fft_in_place).  Asynchronous:
  // ark_hip_synchronize() before the results are read.
  void fft_batch_in_place_device(const std::vector<void*>& d_polys, bool inverse = false) const {
    check(ark_hip_fft_batch_in_place_device(FIELD_ID, &s_, d_polys.data(), d_polys.size(), inverse ? 1 : 0),
          "ark_hip_fft_batch_in_place_device");
  }
  // One process per GPU (Communicator below): this rank's m = size() / world elements, in THIS GPU's memory, of ONE
  // transform over all ranks -- forward: d_local[i2] = x[rank + world i2] in, d_local[j1 sub + t] = X[j1 m + rank sub + t]
  // out (sub = m / world); inverse: the other way round.  One all-to-all over RCCL inside the library.  Collective;
  // asynchronous on the device (ark_hip_synchronize() before the results are read).
  void fft_sharded_in_place_device(void* d_local, bool inverse = false) const {
    check(ark_hip_fft_sharded_device(FIELD_ID, &s_, d_local, inverse ? 1 : 0), "ark_hip_fft_sharded_device");
  }
  // fft_in_place / ifft_in_place for T = Projective<P> (poly/src/domain/mod.rs:332-362; poly/src/test.rs:57): points of a
  // curve whose scalar field is this domain's; resized to the domain size with identities (z = 0) first
  template <class Curve>
  void fft_group_in_place(std::vector<typename Curve::ProjectiveT>& pts, bool inverse = false) const {
    static_assert(Curve::SCALAR_FIELD == FIELD_ID, "the domain must be over the curve's scalar field");
    if (pts.size() > size()) throw Error(ARK_HIP_ERR_ARG, "more coefficients than the domain size");
    typename Curve::ProjectiveT zero{};   // all-zero limbs: z = 0, the identity the transform expects
    pts.resize(size(), zero);
    check(ark_hip_fft_group_in_place(Curve::ID, &s_, reinterpret_cast<uint64_t*>(pts.data()), inverse ? 1 : 0),
          "ark_hip_fft_group_in_place");
  }
  const ark_hip_radix2_domain& raw() const { return s_; }

 private:
  ark_hip_radix2_domain s_{};
  static Fr fe(const uint64_t* p) {
    Fr r;
    for (int i = 0; i < 4; i++) r.limbs[i] = p[i];
    return r;
  }
  void resize(std::vector<Fr>& v) const {
    if (v.size() > size()) throw Error(ARK_HIP_ERR_ARG, "more coefficients than the domain size");
    v.resize(size());
  }
};

// ---- device-resident vectors of Fr: DensePolynomial coefficients / Evaluations that stay in HBM -----------------------
// The reference's callers chain transforms: DensePolynomial::evaluate_over_domain (polynomial/univariate/mod.rs:305-360:
// zero-extend + fft_in_place -> Evaluations), the pointwise +, -, * of Evaluations over one domain
// (evaluations/univariate/mod.rs:104-180) and Evaluations::interpolate (mod.rs:40-50: ifft_in_place -> coefficients).
// Through the host-pointer entries every link of such a chain crosses PCIe twice (2 x 128 MiB at 2^22: 5.3 ms around a
// 0.5 ms transform).  DeviceVec owns ark_hip_malloc memory; the chain below costs ONE upload and ONE download.
// All operations are asynchronous on the library's stream of the current device and ordered with each other; to_vec()
// waits.  A DeviceVec belongs to the device that was current when it was made.
template <int FIELD_ID>
class DeviceVec {
 public:
  DeviceVec() = default;
  explicit DeviceVec(size_t len) { alloc(len); zero_from(0); }                       // vec![F::zero(); len]
  static DeviceVec from_slice(const Fr* x, size_t len) {
    DeviceVec v;
    v.alloc(len);
    check(ark_hip_memcpy_h2d(v.p_, x, len * 32), "ark_hip_memcpy_h2d");
    return v;
  }
  static DeviceVec from_vec(const std::vector<Fr>& x) { return from_slice(x.data(), x.size()); }
  DeviceVec(DeviceVec&& o) noexcept : p_(o.p_), len_(o.len_), cap_(o.cap_), dev_(o.dev_) { o.p_ = nullptr; o.len_ = o.cap_ = 0; }
  DeviceVec& operator=(DeviceVec&& o) noexcept {
    if (this != &o) { release(); p_ = o.p_; len_ = o.len_; cap_ = o.cap_; dev_ = o.dev_; o.p_ = nullptr; o.len_ = o.cap_ = 0; }
    return *this;
  }
  DeviceVec(const DeviceVec&) = delete;
  DeviceVec& operator=(const DeviceVec&) = delete;
  ~DeviceVec() { release(); }
  DeviceVec clone() const {
    here();
    DeviceVec v;
    v.alloc(len_);
    check(ark_hip_memcpy_d2d(v.p_, p_, len_ * 32), "ark_hip_memcpy_d2d");
    return v;
  }
  std::vector<Fr> to_vec() const {
    here();
    std::vector<Fr> out(len_);
    check(ark_hip_memcpy_d2h(out.data(), p_, len_ * 32), "ark_hip_memcpy_d2h");
    return out;
  }
  size_t len() const { return len_; }
  bool is_empty() const { return len_ == 0; }
  void* device_ptr() { return p_; }
  const void* device_ptr() const { return p_; }
  // Vec::resize(new_len, F::zero()) / Vec::truncate
  void resize_zeroed(size_t new_len) {
    here();
    if (new_len > cap_) {
      DeviceVec v;
      v.alloc(new_len);
      check(ark_hip_memcpy_d2d(v.p_, p_, len_ * 32), "ark_hip_memcpy_d2d");
      const size_t keep = len_;
      *this = std::move(v);
      len_ = keep;
    }
    const size_t old = len_;
    len_ = new_len;
    if (new_len > old) zero_from(old);
  }
  // pointwise: self = self (op) other, element by element (lengths must agree, as Evaluations over one domain do)
  DeviceVec& operator+=(const DeviceVec& o) { same(o); check(ark_hip_fr_add_device(FIELD_ID, p_, o.p_, p_, len_), "ark_hip_fr_add_device"); return *this; }
  DeviceVec& operator-=(const DeviceVec& o) { same(o); check(ark_hip_fr_sub_device(FIELD_ID, p_, o.p_, p_, len_), "ark_hip_fr_sub_device"); return *this; }
  DeviceVec& operator*=(const DeviceVec& o) { same(o); check(ark_hip_fr_mul_device(FIELD_ID, p_, o.p_, p_, len_), "ark_hip_fr_mul_device"); return *this; }
  DeviceVec& operator/=(const DeviceVec& o) { same(o); check(ark_hip_fr_div_device(FIELD_ID, p_, o.p_, p_, len_), "ark_hip_fr_div_device"); return *this; }
  void batch_inverse() { here(); check(ark_hip_fr_inverse_device(FIELD_ID, p_, p_, len_), "ark_hip_fr_inverse_device"); }   // zeros stay zero
  DeviceVec& operator*=(const Fr& k) { here(); check(ark_hip_fr_scale_device(FIELD_ID, p_, k.limbs.data(), p_, len_), "ark_hip_fr_scale_device"); return *this; }
  void negate() { here(); check(ark_hip_fr_neg_device(FIELD_ID, p_, p_, len_), "ark_hip_fr_neg_device"); }

 private:
  void* p_ = nullptr;
  size_t len_ = 0, cap_ = 0;
  int dev_ = -1;
  void alloc(size_t len) {
    dev_ = ark_hip_get_device();
    len_ = cap_ = len;
    if (len) check(ark_hip_malloc(len * 32, &p_), "ark_hip_malloc");
  }
  void zero_from(size_t from) {
    if (len_ > from) check(ark_hip_memset_device((char*)p_ + from * 32, 0, (len_ - from) * 32), "ark_hip_memset_device");
  }
  // every operation runs on the stream of the thread's CURRENT device: refuse a vector that lives elsewhere
  void here() const {
    if (p_ && ark_hip_get_device() != dev_) throw Error(ARK_HIP_ERR_ARG, "the device vector lives on another device than the current one");
  }
  void same(const DeviceVec& o) const {
    here();
    o.here();
    if (o.len_ != len_) throw Error(ARK_HIP_ERR_ARG, "domains are unequal");   // the reference's assert_eq!(self.domain, other.domain)
  }
  void release() {
    if (!p_) return;
    const int cur = ark_hip_get_device();
    if (dev_ >= 0 && cur != dev_) (void)ark_hip_set_device(dev_);   // freed on the device that owns it
    (void)ark_hip_free(p_);
    if (dev_ >= 0 && cur != dev_ && cur >= 0) (void)ark_hip_set_device(cur);
    p_ = nullptr;
  }
};

// Evaluations<F, Radix2EvaluationDomain<F>> resident on the device (evaluations/univariate/mod.rs:18-29)
template <int FIELD_ID>
struct DeviceEvaluations {
  DeviceVec<FIELD_ID> evals;
  Radix2EvaluationDomain<FIELD_ID> domain;
  // Evaluations::interpolate (mod.rs:47-50): the coefficients, still on the device (size() of them; the reference's
  // from_coefficients_vec drops leading zeros on the host -- do that after to_vec() if a DensePolynomial is wanted)
  DeviceVec<FIELD_ID> interpolate() && {
    check(ark_hip_ifft_in_place_device(FIELD_ID, &domain.raw(), evals.device_ptr()), "ark_hip_ifft_in_place_device");
    return std::move(evals);
  }
  DeviceEvaluations& operator+=(const DeviceEvaluations& o) { evals += o.evals; return *this; }
  DeviceEvaluations& operator-=(const DeviceEvaluations& o) { evals -= o.evals; return *this; }
  DeviceEvaluations& operator*=(const DeviceEvaluations& o) { evals *= o.evals; return *this; }
  DeviceEvaluations& operator/=(const DeviceEvaluations& o) { evals /= o.evals; return *this; }
};
// DensePolynomial::evaluate_over_domain (polynomial/univariate/mod.rs:305-360) for coefficients already on the device:
// zero-extension and transform in place; at most size/4 coefficients take the degree-aware path (radix2/mod.rs:141).
// More coefficients than the domain holds is the reference's folding case (mod.rs:330-352): not served here.
template <int FIELD_ID>
DeviceEvaluations<FIELD_ID> evaluate_over_domain(DeviceVec<FIELD_ID>&& coeffs, const Radix2EvaluationDomain<FIELD_ID>& domain) {
  const size_t have = coeffs.len();
  if (have > domain.size()) throw Error(ARK_HIP_ERR_ARG, "more coefficients than the domain size");
  coeffs.resize_zeroed(domain.size());
  check(ark_hip_fft_in_place_degree_aware_device(FIELD_ID, &domain.raw(), coeffs.device_ptr(), have),
        "ark_hip_fft_in_place_degree_aware_device");
  return DeviceEvaluations<FIELD_ID>{std::move(coeffs), domain};
}

}  // namespace ark_hip
