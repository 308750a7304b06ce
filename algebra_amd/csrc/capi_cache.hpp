// The verified resident-base cache of the host-pointer MSM entries: the keyed 128-bit tag, its hashing pass on the helper
// pool, lookup / eviction / auto-prepare, and msm_with_bases (the speculative run against a cached copy).
#pragma once
#include "capi_core.hpp"
namespace arkhip {
namespace capi {

// ---- base-set cache of the host-pointer entry points ----------------------------------------------------
inline void free_prepared(PreparedBases* pb) {  // the caller has made sure no job in flight reads the table
  pb->table.release();
  delete pb;
}

// Hash of a base slice's FULL content (verified-cache entries): a keyed 128-bit universal hash (round 5; the round-4
// hash was 64 bits of unkeyed multiply-rotate lanes -- a collision was constructible from this source).
//   key      drawn once per process from the OS (getrandom; /dev/urandom; address-space noise as the last resort) and
//            expanded to 64 KiB of key words: nothing in the source or in another process predicts it
//   block    64 KiB of the slice under UMAC's NH with 64-bit words,
//                NH_K(m) = sum_i (m_2i + K_2i mod 2^64) * (m_2i+1 + K_2i+1 mod 2^64)   mod 2^128
//            for ANY two different equal-length blocks, Pr_K[NH_K(m) = NH_K(m')] <= 2^-64 (Black, Halevi, Krawczyk,
//            Krovetz, Rogaway: "UMAC", Crypto '99, thm 4.2 with w = 64); one 64 x 64 -> 128 multiply per two words
//   slice    the 128-bit block values (with the block index and the length folded in) under NH again with a second key
//            stream: a 128-bit tag; two different slices of one length collide with probability <= 2^-63 over the key
// The tag guards a cached device copy against in-place edits of the host slice, accidental OR crafted by a party that does
// not hold the process's key; it is not a MAC against a party that can read this process's memory.
// Blocks are dealt to host threads (ARK_HIP_HASH_THREADS, default 8).
constexpr size_t HASH_BLOCK_WORDS = 8192;
struct HashKey {
  uint64_t k[HASH_BLOCK_WORDS + 2];   // block key (one word per message word, + 2 for the index / length words)
  uint64_t seed2;                     // seed of the second-level key stream
};
inline uint64_t splitmix64(uint64_t& x) {
  uint64_t z = (x += 0x9e3779b97f4a7c15ull);
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}
inline const HashKey& hash_key() {
  static const HashKey* key = [] {
    HashKey* hk = new HashKey;
    uint64_t seed[2] = {0, 0};
    bool ok = false;
    if (FILE* f = fopen("/dev/urandom", "rb")) {
      ok = fread(seed, 1, sizeof(seed), f) == sizeof(seed);
      fclose(f);
    }
    if (!ok) {   // no OS entropy source: time, the heap's and the stack's address (ASLR) -- weaker, still per process
      seed[0] = (uint64_t)std::chrono::high_resolution_clock::now().time_since_epoch().count() ^ (uint64_t)(uintptr_t)hk;
      seed[1] = (uint64_t)(uintptr_t)&seed ^ ((uint64_t)getpid() << 32);
    }
    uint64_t st = seed[0];
    for (size_t i = 0; i < HASH_BLOCK_WORDS + 2; i++) hk->k[i] = splitmix64(st) ^ seed[1];
    st ^= seed[1] * 0xd6e8feb86659fd93ull;
    hk->seed2 = splitmix64(st);
    return hk;
  }();
  return *key;
}
// NH over `words` words of p (odd tail: zero-padded) + two trailing words (a, b), key offset 0
inline Hash128 nh_block(const uint64_t* p, size_t words, uint64_t a, uint64_t b, const uint64_t* k) {
  unsigned __int128 acc0 = 0, acc1 = 0;
  size_t i = 0;
  for (; i + 4 <= words; i += 4) {   // two independent accumulators: the multiplier pipelines
    acc0 += (unsigned __int128)(p[i] + k[i]) * (p[i + 1] + k[i + 1]);
    acc1 += (unsigned __int128)(p[i + 2] + k[i + 2]) * (p[i + 3] + k[i + 3]);
  }
  for (; i + 2 <= words; i += 2) acc0 += (unsigned __int128)(p[i] + k[i]) * (p[i + 1] + k[i + 1]);
  if (i < words) acc0 += (unsigned __int128)(p[i] + k[i]) * k[i + 1];
  acc1 += (unsigned __int128)(a + k[HASH_BLOCK_WORDS]) * (b + k[HASH_BLOCK_WORDS + 1]);
  const unsigned __int128 r = acc0 + acc1;
  return Hash128{(uint64_t)r, (uint64_t)(r >> 64)};
}
// The first level runs as ranges of blocks on the process-wide helper pool (hostpool.hpp) together with the calling thread
// -- rounds 4-5 created up to eight threads per pass.  ARK_HIP_HASH_THREADS (default 8) caps the number of ranges in flight
// by cutting the pass into that many ranges per "wave"; the pool's size bounds the threads whatever it says.
inline int hash_threads() {
  static int nt = -1;
  if (nt < 0) {
    const char* e = getenv("ARK_HIP_HASH_THREADS");
    int v = e ? atoi(e) : 8;
    if (v > 64) v = 64;
    nt = v < 1 ? 1 : v;
  }
  return nt;
}
struct HashPass {   // one pass over a host slice: first-level block hashes by ranges, then the second level
  const uint64_t* p = nullptr;
  size_t words = 0, nblocks = 0, per = 0;
  int nranges = 0;
  std::vector<Hash128> bh;
  std::atomic<int> left{0};
  std::chrono::steady_clock::time_point t0, t1;
  HostPool::Handle batch;
  HashPass() = default;
  HashPass(const HashPass&) = delete;
  ~HashPass() {   // a pass that was begun is always joined before its block table goes away (helpers may be inside a task)
    if (batch) HostPool::instance().run(batch);
  }
  static void task(void* ctx, int r) {
    HashPass& h = *(HashPass*)ctx;
    const HashKey& hk = hash_key();
    const size_t b0 = (size_t)r * h.per, b1 = b0 + h.per < h.nblocks ? b0 + h.per : h.nblocks;
    for (size_t b = b0; b < b1; b++) {
      const size_t off = b * HASH_BLOCK_WORDS;
      h.bh[b] = nh_block(h.p + off, h.words - off < HASH_BLOCK_WORDS ? h.words - off : HASH_BLOCK_WORDS, (uint64_t)b,
                         (uint64_t)h.words, hk.k);
    }
    if (h.left.fetch_sub(1, std::memory_order_acq_rel) == 1) h.t1 = std::chrono::steady_clock::now();
  }
  // lay the pass out and let the pool's helpers start on it; the caller goes on with its own work
  void begin(const uint64_t* p_, size_t words_) {
    p = p_;
    words = words_;
    nblocks = (words + HASH_BLOCK_WORDS - 1) / HASH_BLOCK_WORDS;
    bh.resize(nblocks);
    // ranges of >= 16 blocks (1 MiB), several per thread so that a late helper still finds work
    size_t nr = nblocks / 16;
    const size_t cap = (size_t)hash_threads() * 4;
    if (nr > cap) nr = cap;
    if (nr < 1 || hash_threads() <= 1) nr = 1;
    nranges = (int)nr;
    per = (nblocks + nr - 1) / nr;
    if (per < 1) per = 1;
    nranges = (int)((nblocks + per - 1) / per);
    if (nranges < 1) nranges = 1;
    left.store(nranges);
    t0 = t1 = std::chrono::steady_clock::now();
    batch = HostPool::instance().open(&HashPass::task, this, nranges);
    if (batch) HostPool::instance().start_async(batch);
  }
  // join: finish what is left on this thread, then the second level.  ms: duration of the pass (start to last block).
  Hash128 end(double* ms) {
    if (batch) {
      HostPool::instance().run(batch);
      batch.reset();
    } else {   // no pool: the pass runs here, after the work it could not hide under (its own duration is what is reported)
      t0 = std::chrono::steady_clock::now();
      for (int r = 0; r < nranges; r++) task(this, r);
    }
    if (ms) *ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    const HashKey& hk = hash_key();
    // second level: NH over the block values with a key stream of its own (generated on the fly: 2 words per block)
    unsigned __int128 acc = 0;
    uint64_t st = hk.seed2;
    for (size_t b = 0; b < nblocks; b++) {
      const uint64_t k0 = splitmix64(st), k1 = splitmix64(st);
      acc += (unsigned __int128)(bh[b].lo + k0) * (bh[b].hi + k1);
    }
    const uint64_t k0 = splitmix64(st), k1 = splitmix64(st);
    acc += (unsigned __int128)((uint64_t)words + k0) * ((uint64_t)nblocks + k1);
    return Hash128{(uint64_t)acc, (uint64_t)(acc >> 64)};
  }
};
inline Hash128 base_hash(const uint64_t* p, size_t words) {
  HashPass h;
  h.begin(p, words);
  return h.end(nullptr);
}

inline void cache_configure(Context* c) {
  if (c->cache_budget < 0) {
    // default: 16 GiB, never more than a quarter of the device memory -- room for a 2^26-point G1 SRS or a 2^24-point G2 one; a
    // library loaded into someone's prover does not claim 72 GB by itself (round 5's default: a quarter of HBM).  Larger sets, or
    // prepared tables kept with them (auto-prepare, off by default), are opt-in: ARK_HIP_BASE_CACHE_MB / ark_hip_msm_cache_config
    // (0 turns the cache off).
    // The cache never changes a result: every hit is validated against a hash of the slice's full content, which host
    // threads compute while the device works (2^24 BLS12-381 G1: 40.2 ms per call cached, 40.2 pinned, 47.8 streamed,
    // 36.4 resident -- profiles/r4_trait_modes_sessionA.txt)
    long long budget = -1;
    if (const char* e = getenv("ARK_HIP_BASE_CACHE_MB")) budget = atoll(e) * (1ll << 20);
    if (budget < 0) {
      size_t fr = 0, tot = 0;
      budget = hipMemGetInfo(&fr, &tot) == hipSuccess ? (long long)(tot / 4) : (8ll << 30);
      if (budget > (16ll << 30)) budget = 16ll << 30;
    }
    c->cache_budget = budget;
  }
  if (c->auto_prepare < 0) {
    const char* e = getenv("ARK_HIP_AUTO_PREPARE");
    c->auto_prepare = e ? atoi(e) : 0;
    if (c->auto_prepare < 0) c->auto_prepare = 0;
  }
}
inline long long cache_entry_bytes(const BaseCacheEntry& e) {   // device bytes an entry holds: the copy + its prepared table
  return (long long)e.dev.cap + (e.prepared ? (long long)e.prepared->table.cap : 0);
}
inline void cache_drop(Context* c, size_t idx) {
  BaseCacheEntry& e = c->base_cache[idx];
  if (e.prepared) free_prepared(e.prepared);
  e.dev.release();
  c->base_cache.erase(c->base_cache.begin() + (long)idx);
}
// drops the transparent entries (pinned sets stay until their unpin)
inline int cache_clear(Context* c) {
  bool any = false;
  for (auto& e : c->base_cache) any |= e.pins == 0;
  if (!any) return 0;
  if (int rc = sync_compute(c)) return rc;  // a job in flight may still read a cached copy
  for (size_t i = c->base_cache.size(); i-- > 0;)
    if (c->base_cache[i].pins == 0) cache_drop(c, i);
  return 0;
}
// a pinned set that CONTAINS [bases, bases + n points): index, and the offset in points; -1 if none
inline long cache_find_pinned(Context* c, int curve, const uint64_t* bases, size_t n, size_t* off_points) {
  const size_t ab = (size_t)CURVES[curve].fe_words * 16;
  for (size_t i = 0; i < c->base_cache.size(); i++) {
    const BaseCacheEntry& e = c->base_cache[i];
    if (e.pins <= 0 || e.curve != curve) continue;
    const char* lo = (const char*)e.host;
    const char* q = (const char*)bases;
    if (q < lo || q + n * ab > lo + e.n * ab || (size_t)(q - lo) % ab) continue;
    *off_points = (size_t)(q - lo) / ab;
    return (long)i;
  }
  return -1;
}
inline long cache_find_exact(Context* c, int curve, const void* bases, size_t n, bool pinned) {
  for (size_t i = 0; i < c->base_cache.size(); i++) {
    const BaseCacheEntry& e = c->base_cache[i];
    if (e.curve == curve && e.host == bases && e.n == n && (e.pins > 0) == pinned) return (long)i;
  }
  return -1;
}
// room for `bytes` more under the transparent budget: least recently used transparent entries go first
// returns false if the set cannot be cached
inline bool cache_make_room(Context* c, long long bytes, long keep = -1) {
  if (bytes > c->cache_budget) return false;
  long long used = 0;
  for (auto& e : c->base_cache)
    if (e.pins == 0) used += cache_entry_bytes(e);
  bool synced = false;
  while (used + bytes > c->cache_budget) {
    long lru = -1;
    for (size_t i = 0; i < c->base_cache.size(); i++) {
      const BaseCacheEntry& e = c->base_cache[i];
      if (e.pins > 0 || (long)i == keep) continue;
      if (lru < 0 || e.last_use < c->base_cache[(size_t)lru].last_use) lru = (long)i;
    }
    if (lru < 0) return false;
    if (!synced) {
      if (sync_compute(c)) return false;
      synced = true;
    }
    used -= cache_entry_bytes(c->base_cache[(size_t)lru]);
    cache_drop(c, (size_t)lru);
    if (keep > lru) keep--;
    c->cache_stats.evicted++;
  }
  return true;
}
// an entry whose upload did not complete must not be found again
inline void cache_forget(Context* c, int curve, const void* host, size_t n, bool pinned) {
  const long i = cache_find_exact(c, curve, host, n, pinned);
  if (i < 0) return;
  (void)sync_compute(c);
  (void)hipStreamSynchronize(c->copy_stream);
  cache_drop(c, (size_t)i);
}
// after `auto_prepare` hits on the WHOLE set: build its per-window table (a failed build leaves the plain path in place)
inline void cache_maybe_prepare(Context* c, BaseCacheEntry& e) {
  if (c->auto_prepare <= 0 || e.prepared || e.no_prepare || e.hits < (unsigned)c->auto_prepare) return;
  if (e.pins == 0) {   // transparent entries: the table counts against the budget
    const MsmPlan pl = msm_make_plan(e.n, msm_scalar_bits(e.curve), msm_mul_cost(e.curve), true, msm_lazy28(e.curve));
    const long long table = (long long)pl.W * (long long)e.n * CURVES[e.curve].fe_words * 16;
    const void* host = e.host;
    const size_t n = e.n;
    const int curve = e.curve;
    const long self = cache_find_exact(c, curve, host, n, false);
    if (!cache_make_room(c, table + table / 8, self)) {
      c->base_cache[(size_t)cache_find_exact(c, curve, host, n, false)].no_prepare = true;
      return;
    }
    BaseCacheEntry& e2 = c->base_cache[(size_t)cache_find_exact(c, curve, host, n, false)];  // eviction moved entries
    ark_hip_msm_bases* pb = nullptr;
    if (ark_hip_msm_bases_prepare_device(curve, e2.dev.p, n, &pb) == 0) e2.prepared = (PreparedBases*)pb;
    else e2.no_prepare = true;
    return;
  }
  ark_hip_msm_bases* pb = nullptr;
  if (ark_hip_msm_bases_prepare_device(e.curve, e.dev.p, e.n, &pb) == 0) e.prepared = (PreparedBases*)pb;
  else e.no_prepare = true;
}

// Runs one host-pointer MSM against whatever device copy of `bases` the context may use:
//   run(d_bases, fill, entry)   d_bases == nullptr: no resident copy -- stream the bases with the scalars;
//                               fill: d_bases is reserved but EMPTY -- upload `bases` into it on the way;
//                               entry: the cache entry when the WHOLE set is the operand (prepared table), else nullptr.
// Pinned range: used as is.  Verified-cache entry (the cache is ON by default, ark_hip.h): the run is speculative -- the slice's full-content hash is
// computed on host threads meanwhile and the result only stands if it matches the hash of what the device copy holds.
template <class Run>
int msm_with_bases(Context* c, int curve, const uint64_t* bases, size_t n, Run run) {
  const size_t wpp = (size_t)CURVES[curve].fe_words * 2, bytes = n * wpp * 8;
  size_t off = 0;
  const long pi = cache_find_pinned(c, curve, bases, n, &off);
  if (pi >= 0) {
    BaseCacheEntry& e = c->base_cache[(size_t)pi];
    const bool whole = off == 0 && n == e.n;
    if (whole) {
      e.hits++;
      cache_maybe_prepare(c, e);
    }
    c->cache_stats.pinned_hits++;
    return run((const void*)((const char*)e.dev.p + off * wpp * 8), false, whole ? &e : nullptr);
  }
  cache_configure(c);
  // sets below 64 KiB are not worth a device copy and a hashing pass: they stream with their scalars
  if (c->cache_budget <= 0 || (long long)bytes > c->cache_budget || bytes < ((size_t)64 << 10)) return run(nullptr, false, nullptr);
  c->cache_clock++;
  long idx = cache_find_exact(c, curve, bases, n, false);
  typedef std::chrono::steady_clock Clk;
  auto ms_since = [](Clk::time_point t) { return std::chrono::duration<double, std::milli>(Clk::now() - t).count(); };
  // A hit costs one keyed pass over the host slice on hash_threads() host threads, hidden under the MSM -- when the host has
  // the cores to spare.  Inside a prover whose thread pool already saturates them it does not hide: the pass is measured on
  // every call (bytes per ms, smoothed), and while the predicted pass exceeds 1.5 x what streaming this slice over PCIe took
  // when the copy was filled (sets of 64 MiB and more: below, the pass is noise), the call streams instead (the copy stays;
  // every eighth call hashes again to notice an idle host).
  if (idx >= 0) {
    const BaseCacheEntry& e0 = c->base_cache[(size_t)idx];
    const double rate = c->cache_stats.hash_bytes_per_ms;
    const char* ad = getenv("ARK_HIP_HASH_ADAPTIVE");   // "0": always validate by hashing (tests that count hits)
    if (!(ad && ad[0] == '0') && bytes >= ((size_t)64 << 20) && e0.fill_ms > 0 && rate > 0 &&
        (double)bytes / rate > 1.5 * e0.fill_ms && (c->cache_clock & 7) != 0) {
      c->cache_stats.busy_streamed++;
      return run(nullptr, false, nullptr);
    }
  }
  const Clk::time_point t_call = Clk::now();
  Hash128 h;
  double hash_ms = 0;
  HashPass pass;   // the pool's helpers hash the slice while this thread runs the MSM; join_hash() finishes what is left
  pass.begin(bases, n * wpp);
  auto join_hash = [&]() { h = pass.end(&hash_ms); };
  auto hash_done = [&]() {   // after join_hash(): fold this pass into the smoothed rate
    c->cache_stats.last_hash_ms = hash_ms;
    if (hash_ms > 0) {
      const double r = (double)bytes / hash_ms, old = c->cache_stats.hash_bytes_per_ms;
      c->cache_stats.hash_bytes_per_ms = old > 0 ? 0.5 * old + 0.5 * r : r;
    }
  };
  if (idx >= 0) {
    {
      BaseCacheEntry& e = c->base_cache[(size_t)idx];
      e.last_use = c->cache_clock;
      cache_maybe_prepare(c, e);
    }
    idx = cache_find_exact(c, curve, bases, n, false);
    int rc = run(c->base_cache[(size_t)idx].dev.p, false, &c->base_cache[(size_t)idx]);   // speculative
    join_hash();
    hash_done();
    BaseCacheEntry& e = c->base_cache[(size_t)idx];
    if (h == e.hash) {
      e.hits++;
      c->cache_stats.hits++;
      return rc;
    }
    // the slice changed under the same address and length: refresh the copy and run again
    if (int rc2 = sync_compute(c)) return rc2;
    if (e.prepared) free_prepared(e.prepared);
    e.prepared = nullptr;
    e.no_prepare = false;
    e.hash = h;
    e.hits = 0;
    c->cache_stats.refreshed++;
    const Clk::time_point t_fill = Clk::now();
    rc = run(e.dev.p, true, nullptr);
    e.fill_ms = ms_since(t_fill);
    if (rc) cache_forget(c, curve, bases, n, false);
    return rc;
  }
  // miss: make room (least recently used first), then fill under the call's own kernels
  BaseCacheEntry ne;
  ne.curve = curve;
  ne.host = bases;
  ne.n = n;
  ne.last_use = c->cache_clock;
  if (!cache_make_room(c, (long long)(bytes + bytes / 8 + 256)) || ne.dev.ensure(bytes)) {
    join_hash();   // no room on the device right now: not an error, the bases stream instead
    hash_done();
    return run(nullptr, false, nullptr);
  }
  c->cache_stats.misses++;
  c->base_cache.push_back(ne);
  const int rc = run(ne.dev.p, true, nullptr);
  const double fill_ms = ms_since(t_call);   // bases + scalars over PCIe under the call's kernels: the streamed call's cost
  join_hash();
  hash_done();
  if (rc) {
    cache_forget(c, curve, bases, n, false);
    return rc;
  }
  BaseCacheEntry& filled = c->base_cache[(size_t)cache_find_exact(c, curve, bases, n, false)];
  filled.hash = h;
  filled.fill_ms = fill_ms;
  return 0;
}

}  // namespace capi
}  // namespace arkhip
