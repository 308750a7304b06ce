// C ABI of libark_hip.so, unit 2 of 5: the MSM entry points -- plain / narrow / prepared / streamed / multi-device, pinned and
// cached base sets, fixed-base batch multiplication (see include/ark_hip.h).
#include "capi_core.hpp"
#include "capi_cache.hpp"
#include "capi_stream.hpp"
using namespace arkhip;
using namespace arkhip::capi;

extern "C" {

extern "C++" {
// ---- synchronous MSM entries never fail for want of a job slot ------------------------------------------------------------
// A device keeps at most MSM_JOBS jobs in flight; the *_async entries report ARK_HIP_ERR_BUSY beyond that.  The SYNCHRONOUS
// entries are what SWCurveConfig::msm & co. call from however many rayon threads the prover runs (SURVEY 8(b) "Threading"):
// there BUSY is not an answer -- rounds 2-5 returned it to the fifth concurrent caller.  Every synchronous entry is the body
// (`*_once`: BUSY leaves nothing behind) retried after another thread's job has left its slot; the wait holds no lock.
static int msm_sw_device_once(int curve, const void* d_bases, const void* d_scalars, size_t n, int mont, uint64_t* out_xyz);
static int msm_sw_once(int curve, const uint64_t* bases, const uint64_t* scalars, size_t n, int mont, uint64_t* out_xyz);
static int msm_sw_small_device_once(int curve, const void* d_bases, const void* d_scalars, size_t n, int scalar_bytes, int max_bits,
                                uint64_t* out_xyz);
static int msm_sw_small_once(int curve, const uint64_t* bases, const void* scalars, size_t n, int scalar_bytes, int max_bits,
                         uint64_t* out_xyz);
static int msm_prepared_device_once(const ark_hip_msm_bases* bases, const void* d_scalars, size_t n, int mont,
                                uint64_t* out_xyz);
static int msm_prepared_once(const ark_hip_msm_bases* bases, const uint64_t* scalars, size_t n, int mont, uint64_t* out_xyz);
static int msm_prepared_small_device_once(const ark_hip_msm_bases* bases, const void* d_scalars, size_t n, int scalar_bytes,
                                      int max_bits, uint64_t* out_xyz);
static int msm_sw_chunks_once(int curve, const uint64_t* bases, size_t n_bases, const uint64_t* scalars, size_t n_scalars,
                          size_t step, uint64_t* out_xyz);
static int msm_jobs_in_flight(Context* c) {
  int busy = 0;
  for (int l = 0; l < 2; l++) {
    std::lock_guard<std::mutex> lock(c->msm[l].mu);
    for (const auto& j : c->msm[l].jobs) busy += j.busy ? 1 : 0;
  }
  return busy;
}
template <class Body>
static int retry_while_busy(int logical, Body body) {
  for (;;) {
    Context* c = nullptr;
    uint64_t gen = 0;
    if (get_ctx(logical == -2 ? t_dev : logical, &c) == 0) {
      // callers that outnumber the slots queue HERE, before the body: a body that ends in BUSY has already looked its bases up,
      // started a hashing pass or staged an upload for nothing (32 mixed callers: 30 calls/s before this check, see
      // tools/thread_soak.py), and every finished job would wake all of them to do it again
      std::unique_lock<std::mutex> lk(c->slot_mu);
      c->slot_cv.wait_for(lk, std::chrono::milliseconds(20), [&]() { return msm_jobs_in_flight(c) < MSM_JOBS; });
      gen = c->slot_gen.load(std::memory_order_acquire);
    }
    const int rc = body();
    if (rc != ARK_HIP_ERR_BUSY || !c) return rc;
    std::unique_lock<std::mutex> lk(c->slot_mu);
    c->slot_cv.wait_for(lk, std::chrono::milliseconds(20), [&]() { return c->slot_gen.load(std::memory_order_acquire) != gen; });
  }
}
}
// ---- MSM ------------------------------------------------------------------------------------------------
// may_block: the caller is about to wait for the result anyway (the synchronous entry), so the width probe's read-back may
// drain the lane's stream; the public *_async entries never block on queued device work
static int msm_sw_device_enqueue(int curve, const void* d_bases, const void* d_scalars, size_t n, int mont,
                                 ark_hip_msm_job** out_job, bool may_block) {
  if (curve < 0 || curve > 4 || !out_job || (n && (!d_bases || !d_scalars))) return ARK_HIP_ERR_ARG;
  ARK_SCOPE(sc);
  const int lane = msm_pick_lane(sc.c);
  if (lane < 0) return lane;
  int slot = msm_enqueue_ctx(sc.c, curve, d_bases, 0, nullptr, d_scalars, n, mont, lane, 0, 0, nullptr, may_block);
  if (slot < 0) return slot;
  *out_job = (ark_hip_msm_job*)new MsmJobHandle{sc.c->logical, curve, slot};
  return 0;
}
int ark_hip_msm_sw_device_async(int curve, const void* d_bases, const void* d_scalars, size_t n, int mont,
                                ark_hip_msm_job** out_job) {
  return msm_sw_device_enqueue(curve, d_bases, d_scalars, n, mont, out_job, false);
}

int ark_hip_msm_wait(ark_hip_msm_job* job, uint64_t* out_xyz) {
  if (!job) return ARK_HIP_ERR_ARG;
  MsmJobHandle* h = (MsmJobHandle*)job;
  // no context lock at all: the event wait and the host tail need the lane's job table only (get_ctx also makes the job's
  // device this thread's current one).  A wait therefore never queues behind another thread's entry point -- whose streaming
  // body may itself be waiting for THIS job's slot.
  Context* c = nullptr;
  int rc = get_ctx(h->logical, &c);
  if (rc == 0) {
    uint64_t scratch[36];
    rc = msm_finish_ctx(c, h->curve, h->slot, out_xyz ? out_xyz : scratch);
  }
  delete h;
  return rc;
}

static int msm_sw_device_once(int curve, const void* d_bases, const void* d_scalars, size_t n, int mont, uint64_t* out_xyz) {
  if (!out_xyz) return ARK_HIP_ERR_ARG;
  ark_hip_msm_job* job = nullptr;
  int rc = msm_sw_device_enqueue(curve, d_bases, d_scalars, n, mont, &job, true);
  if (rc) return rc;
  return ark_hip_msm_wait(job, out_xyz);
}

// The entry SWCurveConfig::msm / the msm_bigint hook land in (rust/ark-hip/src/msm.rs, patches/0001): host slices in,
// Projective out -- a function of the two slices.  A base slice inside a PINNED range (ark_hip_msm_bases_pin) or found in
// the verified cache (ON by default with a quarter of the device memory; every hit is validated against a keyed hash of the
// slice's full content, msm_with_bases) uses the resident copy and uploads only its scalars.  With the cache off (budget 0)
// or a slice that does not fit it, bases and scalars stream over PCIe in pieces under the previous piece's kernels
// (msm_stream) and nothing is retained.
static int msm_sw_once(int curve, const uint64_t* bases, const uint64_t* scalars, size_t n, int mont, uint64_t* out_xyz) {
  if (curve < 0 || curve > 4 || !out_xyz || (n && (!bases || !scalars))) return ARK_HIP_ERR_ARG;
  ARK_SCOPE(sc);
  Context* c = sc.c;
  if (n == 0) return msm_stream(c, curve, nullptr, nullptr, nullptr, 0, mont, 0, out_xyz);
  return msm_with_bases(c, curve, bases, n, [&](const void* d_bases, bool fill, BaseCacheEntry* ce) -> int {
    if (!d_bases)   // no resident copy: bases and scalars both stream through the ring, the tail in shrinking pieces
      return msm_stream(c, curve, nullptr, bases, scalars, n, mont, msm_stream_step(n), out_xyz, true, false, true);
    if (fill)       // first call with this set: its bases cross PCIe with the scalars, piece k+1 under piece k's kernels
      return msm_stream(c, curve, d_bases, bases, scalars, n, mont, msm_stream_step(n), out_xyz, true, false, true);
    if (ce && ce->prepared) return msm_prepared_once((const ark_hip_msm_bases*)ce->prepared, scalars, n, mont, out_xyz);
    const char* eg = getenv("ARK_HIP_STREAM_GROWING");   // =0: equal pieces (msm_stream_step); a forced piece count also
    const bool growing = !(eg && eg[0] == '0') && !getenv("ARK_HIP_STREAM_PIECES");
    return msm_stream(c, curve, d_bases, nullptr, scalars, n, mont, msm_stream_step(n), out_xyz, true, growing);
  });
}

// ---- pinned base sets ----
// ark_hip_msm_bases_pin: the caller declares bases[0 .. n) immutable until the matching unpin; the set is uploaded now
// and every host-pointer MSM whose base slice lies inside it (sub-slices at point boundaries included: msm_unchecked's
// truncation, msm_chunks / ChunkedPippenger steps) runs against the resident copy.  Pins nest (a count per
// (curve, address, n)).  Pinned sets are outside the transparent cache's budget and are never evicted.
int ark_hip_msm_bases_pin(int curve, const uint64_t* bases, size_t n) {
  if (curve < 0 || curve > 4 || !bases || n == 0) return ARK_HIP_ERR_ARG;
  ARK_SCOPE(sc);
  Context* c = sc.c;
  long i = cache_find_exact(c, curve, bases, n, true);
  if (i >= 0) {
    c->base_cache[(size_t)i].pins++;
    return 0;
  }
  const size_t bytes = n * (size_t)CURVES[curve].fe_words * 16;
  BaseCacheEntry ne;
  ne.curve = curve;
  ne.host = bases;
  ne.n = n;
  ne.pins = 1;
  ne.last_use = ++c->cache_clock;
  if (ne.dev.ensure(bytes)) return ARK_HIP_ERR_NOMEM;
  int rc = c->stager.upload(ne.dev.p, bases, bytes, c->copy_stream);
  if (rc == 0 && hipStreamSynchronize(c->copy_stream) != hipSuccess) rc = -1000;
  if (rc) {
    ne.dev.release();
    return rc;
  }
  c->base_cache.push_back(ne);
  return 0;
}
int ark_hip_msm_bases_unpin(int curve, const uint64_t* bases, size_t n) {
  if (curve < 0 || curve > 4 || !bases) return ARK_HIP_ERR_ARG;
  ARK_SCOPE(sc);
  Context* c = sc.c;
  const long i = cache_find_exact(c, curve, bases, n, true);
  if (i < 0) return ARK_HIP_ERR_ARG;
  if (--c->base_cache[(size_t)i].pins > 0) return 0;
  if (int rc = sync_compute(c)) return rc;   // a job in flight may still read the copy
  cache_drop(c, (size_t)i);
  return 0;
}

// ---- narrow scalars: VariableBaseMSM::msm_u1 / msm_u8 / msm_u16 / msm_u32 / msm_u64 (variable_base/mod.rs:87-117) ----
// scalars: n unsigned integers of scalar_bytes (1, 2, 4, 8) bytes each, of which the low max_bits (0 = all) may be set
// (msm_u1: one byte per bool, max_bits = 1).  Only ceil((max_bits + 1) / c) windows exist: nothing is expanded to 32
// bytes and no empty window is sorted.
static int msm_sw_small_device_once(int curve, const void* d_bases, const void* d_scalars, size_t n, int scalar_bytes, int max_bits,
                                uint64_t* out_xyz) {
  if (curve < 0 || curve > 4 || !out_xyz || (n && (!d_bases || !d_scalars))) return ARK_HIP_ERR_ARG;
  if (scalar_bytes != 1 && scalar_bytes != 2 && scalar_bytes != 4 && scalar_bytes != 8) return ARK_HIP_ERR_ARG;
  if (max_bits == 0) max_bits = 8 * scalar_bytes;
  if (max_bits < 1 || max_bits > 8 * scalar_bytes) return ARK_HIP_ERR_ARG;
  ARK_SCOPE(sc);
  const int lane = msm_pick_lane(sc.c);
  if (lane < 0) return lane;
  int slot = msm_enqueue_ctx(sc.c, curve, d_bases, 0, nullptr, d_scalars, n, 0, lane, scalar_bytes, max_bits);
  if (slot < 0) return slot;
  return msm_finish_ctx(sc.c, curve, slot, out_xyz);
}
static int msm_sw_small_once(int curve, const uint64_t* bases, const void* scalars, size_t n, int scalar_bytes, int max_bits,
                         uint64_t* out_xyz) {
  if (curve < 0 || curve > 4 || !out_xyz || (n && (!bases || !scalars))) return ARK_HIP_ERR_ARG;
  if (scalar_bytes != 1 && scalar_bytes != 2 && scalar_bytes != 4 && scalar_bytes != 8) return ARK_HIP_ERR_ARG;
  ARK_SCOPE(sc);
  Context* c = sc.c;
  if (n == 0) return msm_sw_small_device_once(curve, nullptr, nullptr, 0, scalar_bytes, max_bits, out_xyz);
  // the base set is looked up like ark_hip_msm_sw's (pinned range / verified cache, on by default); the scalars are small:
  // one upload.  Uploads ride the copy stream and are complete before the MSM is enqueued on either lane.
  const size_t bb = n * (size_t)CURVES[curve].fe_words * 16, sb = n * (size_t)scalar_bytes;
  return msm_with_bases(c, curve, bases, n, [&](const void* d_res, bool fill, BaseCacheEntry*) -> int {
    const void* d_bases = d_res;
    if (!d_bases) {
      if (c->stage_a.cap < bb) {
        if (int rc = sync_compute(c)) return rc;
        if (c->stage_a.ensure(bb)) return ARK_HIP_ERR_NOMEM;
      }
      d_bases = c->stage_a.p;
    }
    if (!d_res || fill)
      if (int rc = c->stager.upload((void*)d_bases, bases, bb, c->copy_stream)) return rc;
    if (c->stage_b.cap < sb) {
      if (int rc = sync_compute(c)) return rc;
      if (c->stage_b.ensure(sb)) return ARK_HIP_ERR_NOMEM;
    }
    if (int rc = c->stager.upload(c->stage_b.p, scalars, sb, c->copy_stream)) return rc;
    ARK_HIP_TRY(hipStreamSynchronize(c->copy_stream));   // staged uploads leave their last slices in flight
    return msm_sw_small_device_once(curve, d_bases, c->stage_b.p, n, scalar_bytes, max_bits, out_xyz);
  });
}

// The narrow entries against a PREPARED base set: the per-window table was laid out for 255-bit scalars (its wide windows
// would leave a u32 vector with one dense and one sparse window); row 0 of the table IS the base set, so narrow scalars
// run as a plain MSM over it with a plan of their own.
static int msm_prepared_small_device_once(const ark_hip_msm_bases* bases, const void* d_scalars, size_t n, int scalar_bytes,
                                      int max_bits, uint64_t* out_xyz) {
  if (!bases || !out_xyz) return ARK_HIP_ERR_ARG;
  const PreparedBases* pb = (const PreparedBases*)bases;
  if (n > pb->n) return ARK_HIP_ERR_ARG;
  Scope sc;
  if (int rc = sc.enter(pb->logical)) return rc;
  return msm_sw_small_device_once(pb->curve, pb->table.p, d_scalars, n, scalar_bytes, max_bits, out_xyz);
}

// ---- resident-base cache control ----
int ark_hip_msm_cache_config(long long budget_bytes, int auto_prepare_after) {
  ARK_SCOPE(sc);
  if (budget_bytes == -2) sc.c->cache_budget = -1;   // back to the default (environment / a quarter of the device memory)
  cache_configure(sc.c);
  if (budget_bytes >= 0) {
    sc.c->cache_budget = budget_bytes;
    if (budget_bytes == 0) {
      if (int rc = cache_clear(sc.c)) return rc;
    }
  }
  if (auto_prepare_after >= 0) sc.c->auto_prepare = auto_prepare_after;
  return 0;
}
int ark_hip_msm_cache_clear(void) {
  ARK_SCOPE(sc);
  return cache_clear(sc.c);
}
int ark_hip_msm_cache_stats(uint64_t out[8]) {
  if (!out) return ARK_HIP_ERR_ARG;
  ARK_SCOPE(sc);
  uint64_t bytes = 0, entries = 0, pinned = 0;
  for (auto& e : sc.c->base_cache) {
    if (e.pins > 0) {
      pinned++;
    } else {
      entries++;
      bytes += (uint64_t)cache_entry_bytes(e);
    }
  }
  out[0] = entries;
  out[1] = bytes;
  out[2] = sc.c->cache_stats.hits;
  out[3] = sc.c->cache_stats.misses;
  out[4] = sc.c->cache_stats.refreshed;
  out[5] = sc.c->cache_stats.evicted;
  out[6] = pinned;
  out[7] = sc.c->cache_stats.pinned_hits;
  return 0;
}
// the validation pass itself: [0] calls that streamed their bases although a copy was cached because the host was too busy
// to hash the slice in the time streaming takes, [1] the latest pass in microseconds, [2] its smoothed rate in MB/s,
// [3] host threads per pass
int ark_hip_msm_cache_hash_stats(uint64_t out[4]) {
  if (!out) return ARK_HIP_ERR_ARG;
  ARK_SCOPE(sc);
  out[0] = sc.c->cache_stats.busy_streamed;
  out[1] = (uint64_t)(sc.c->cache_stats.last_hash_ms * 1e3);
  out[2] = (uint64_t)(sc.c->cache_stats.hash_bytes_per_ms / 1e3);
  out[3] = (uint64_t)hash_threads();
  return 0;
}

// the window plan the library would use (host arithmetic only: no GPU needed)
int ark_hip_msm_plan(int curve, size_t n, int prepared, int* window_bits, int* windows) {
  if (curve < 0 || curve > 4) return ARK_HIP_ERR_ARG;
  const MsmPlan pl = msm_make_plan(n ? n : 1, msm_scalar_bits(curve), msm_mul_cost(curve), prepared != 0, msm_lazy28(curve), nullptr,
                                   prepared == 0 && msm_lazy_enabled());
  if (window_bits) *window_bits = pl.c;
  if (windows) *windows = pl.W;
  return 0;
}

// the plan of a plain MSM whose scalars have the given width classes (what ark_hip_msm_sw_device does after its probe)
int ark_hip_msm_plan_widths(int curve, size_t n, uint32_t max_bits, const uint32_t counts[9], int* window_bits, int* windows) {
  if (curve < 0 || curve > 4 || !counts) return ARK_HIP_ERR_ARG;
  static_assert(MSM_WIDTH_CLASSES == 9, "the header documents nine classes");
  MsmWidths w{};
  w.max_bits = max_bits;
  for (int k = 0; k < MSM_WIDTH_CLASSES; k++) w.count[k] = counts[k];
  const MsmPlan pl = msm_plan_for_widths(n ? n : 1, msm_scalar_bits(curve), msm_mul_cost(curve), msm_lazy28(curve), w);
  if (window_bits) *window_bits = pl.c;
  if (windows) *windows = pl.W;
  return 0;
}

int ark_hip_msm_set_timing(int enable) {
  ARK_SCOPE(sc);
  sc.c->msm_timing = enable != 0;
  return 0;
}
int ark_hip_msm_last_timing(double out[8]) {
  if (!out) return ARK_HIP_ERR_ARG;
  ARK_SCOPE(sc);
  MsmTimings t;
  {
    std::lock_guard<std::mutex> lk(sc.c->slot_mu);
    t = sc.c->msm_tm;
  }
  out[0] = t.digits; out[1] = t.scan; out[2] = t.scatter; out[3] = t.accumulate; out[4] = t.reduce; out[5] = t.total;
  out[6] = t.c; out[7] = t.W;
  return 0;
}

// ---- prepared base sets (fixed SRS) ----
int ark_hip_msm_bases_prepare_device(int curve, const void* d_bases, size_t n, ark_hip_msm_bases** out) {
  if (curve < 0 || curve > 4 || !out || (n && !d_bases)) return ARK_HIP_ERR_ARG;
  ARK_SCOPE(sc);
  Context* c = sc.c;
  PreparedBases* pb = new PreparedBases();
  pb->curve = curve;
  pb->logical = c->logical;
  pb->n = n;
  pb->plan = msm_make_plan(n ? n : 1, msm_scalar_bits(curve), msm_mul_cost(curve), true);
  const size_t row = n * (size_t)CURVES[curve].fe_words * 16;
  if (n) {
    if ((size_t)pb->plan.W * n >= (1ull << 31) || pb->table.ensure((size_t)pb->plan.W * row)) {
      delete pb;
      return ARK_HIP_ERR_NOMEM;
    }
    DevBuf tmp;  // one unnormalised (XYZZ) row: twice an affine row, released once the table stands
    int rc = tmp.ensure(2 * row) ? ARK_HIP_ERR_NOMEM : 0;
    if (rc == 0) rc = msm_prepare_dispatch(curve, d_bases, n, pb->plan, pb->table.p, tmp.p, c->stream);
    if (rc == 0 && hipStreamSynchronize(c->stream) != hipSuccess) rc = -1000;
    else if (rc) (void)hipStreamSynchronize(c->stream);
    tmp.release();
    if (rc) {
      pb->table.release();
      delete pb;
      return rc;
    }
  }
  *out = (ark_hip_msm_bases*)pb;
  return 0;
}
int ark_hip_msm_bases_prepare(int curve, const uint64_t* bases, size_t n, ark_hip_msm_bases** out) {
  if (curve < 0 || curve > 4 || !out || (n && !bases)) return ARK_HIP_ERR_ARG;
  ARK_SCOPE(sc);
  Context* c = sc.c;
  const size_t bb = n * (size_t)CURVES[curve].fe_words * 16;
  if (n) {
    if (c->stage_a.ensure(bb)) return ARK_HIP_ERR_NOMEM;
    if (int rc = c->stager.upload(c->stage_a.p, bases, bb, c->stream)) return rc;
  }
  return ark_hip_msm_bases_prepare_device(curve, c->stage_a.p, n, out);
}
int ark_hip_msm_bases_free(ark_hip_msm_bases* bases) {
  if (!bases) return 0;
  PreparedBases* pb = (PreparedBases*)bases;
  Scope sc;
  if (int rc = sc.enter(pb->logical)) return rc;
  if (int rc = sync_compute(sc.c)) return rc;  // a job in flight on either lane may still read the table
  free_prepared(pb);
  return 0;
}
int ark_hip_msm_bases_info(const ark_hip_msm_bases* bases, size_t* n, int* window_bits, int* windows, size_t* table_bytes) {
  if (!bases) return ARK_HIP_ERR_ARG;
  const PreparedBases* pb = (const PreparedBases*)bases;
  if (n) *n = pb->n;
  if (window_bits) *window_bits = pb->plan.c;
  if (windows) *windows = pb->plan.W;
  if (table_bytes) *table_bytes = (size_t)pb->plan.W * pb->n * (size_t)CURVES[pb->curve].fe_words * 16;
  return 0;
}
static int msm_prepared_device_enqueue(const ark_hip_msm_bases* bases, const void* d_scalars, size_t n, int mont,
                                       ark_hip_msm_job** out_job, bool may_block) {
  if (!bases || !out_job) return ARK_HIP_ERR_ARG;
  const PreparedBases* pb = (const PreparedBases*)bases;
  if (n > pb->n || (n && !d_scalars)) return ARK_HIP_ERR_ARG;
  Scope sc;
  if (int rc = sc.enter(pb->logical)) return rc;
  const int lane = msm_pick_lane(sc.c);
  if (lane < 0) return lane;
  int slot = msm_enqueue_ctx(sc.c, pb->curve, pb->table.p, pb->n, &pb->plan, d_scalars, n, mont, lane, 0, 0, nullptr, may_block);
  if (slot < 0) return slot;
  *out_job = (ark_hip_msm_job*)new MsmJobHandle{pb->logical, pb->curve, slot};
  return 0;
}
int ark_hip_msm_prepared_device_async(const ark_hip_msm_bases* bases, const void* d_scalars, size_t n, int mont,
                                      ark_hip_msm_job** out_job) {
  return msm_prepared_device_enqueue(bases, d_scalars, n, mont, out_job, false);
}
static int msm_prepared_device_once(const ark_hip_msm_bases* bases, const void* d_scalars, size_t n, int mont,
                                uint64_t* out_xyz) {
  if (!out_xyz) return ARK_HIP_ERR_ARG;
  ark_hip_msm_job* job = nullptr;
  int rc = msm_prepared_device_enqueue(bases, d_scalars, n, mont, &job, true);
  if (rc) return rc;
  return ark_hip_msm_wait(job, out_xyz);
}
// Host scalars: uploaded on the copy stream into a two-slot ring, so that the upload of the next MSM's scalars
// overlaps the previous MSM's kernels (the steady state of a prover that commits to one polynomial after another
// against a resident SRS).  Pinned host memory (ark_hip_host_alloc) makes the copy truly asynchronous.
static int msm_prepared_host_enqueue(const ark_hip_msm_bases* bases, const uint64_t* scalars, size_t n, int mont,
                                     ark_hip_msm_job** out_job, bool may_block) {
  if (!bases || !out_job) return ARK_HIP_ERR_ARG;
  const PreparedBases* pb = (const PreparedBases*)bases;
  if (n > pb->n || (n && !scalars)) return ARK_HIP_ERR_ARG;
  Scope sc;
  if (int rc = sc.enter(pb->logical)) return rc;
  Context* c = sc.c;
  const int lane = msm_pick_lane(c);  // first: a BUSY return must not leave a copy from caller memory in flight
  if (lane < 0) return lane;
  hipStream_t compute;
  if (int rc = msm_lane_stream(c, lane, &compute)) return rc;
  int k = 0;
  if (int rc = ring_acquire(c, &k)) return rc;
  if (n) {
    if (c->ring_s[k].cap < n * 32) {
      if (int rc = sync_compute(c)) return rc;  // growing frees memory an enqueued MSM may still read
      if (c->ring_s[k].ensure(n * 32)) return ARK_HIP_ERR_NOMEM;
    }
    // page-locked scalars (ark_hip_host_alloc) are read in place by the DMA engine -- the caller keeps them valid until
    // the wait returns; ordinary memory goes through the pinned staging ring and has been read when this returns
    if (int rc = c->stager.upload(c->ring_s[k].p, scalars, n * 32, c->copy_stream, true)) {
      (void)hipStreamSynchronize(c->copy_stream);
      return rc;
    }
  }
  int rc = ring_publish(c, k, compute);
  int slot = rc ? rc : msm_enqueue_ctx(c, pb->curve, pb->table.p, pb->n, &pb->plan, c->ring_s[k].p, n, mont, lane, 0, 0, nullptr,
                                       may_block);
  (void)ring_release(c, k, compute);
  if (slot < 0) {
    (void)hipStreamSynchronize(c->copy_stream);  // nothing reads caller memory once an error has been returned
    return slot;
  }
  *out_job = (ark_hip_msm_job*)new MsmJobHandle{pb->logical, pb->curve, slot};
  return 0;
}
int ark_hip_msm_prepared_async(const ark_hip_msm_bases* bases, const uint64_t* scalars, size_t n, int mont,
                               ark_hip_msm_job** out_job) {
  return msm_prepared_host_enqueue(bases, scalars, n, mont, out_job, false);
}
static int msm_prepared_once(const ark_hip_msm_bases* bases, const uint64_t* scalars, size_t n, int mont, uint64_t* out_xyz) {
  if (!out_xyz) return ARK_HIP_ERR_ARG;
  ark_hip_msm_job* job = nullptr;
  int rc = msm_prepared_host_enqueue(bases, scalars, n, mont, &job, true);
  if (rc) return rc;
  return ark_hip_msm_wait(job, out_xyz);
}

// VariableBaseMSM::msm_chunks (variable_base/mod.rs:119-150): Fr scalars, streams aligned at their END (the first
// n_bases - n_scalars bases are skipped), steps of `step` pairs (the reference hard-codes 2^20; 0 selects it), each
// step an msm_bigint whose result is added up.  Here step k+1's bases and scalars upload on the copy stream while
// step k's kernels run.
static int msm_sw_chunks_once(int curve, const uint64_t* bases, size_t n_bases, const uint64_t* scalars, size_t n_scalars,
                          size_t step, uint64_t* out_xyz) {
  if (curve < 0 || curve > 4 || !out_xyz || n_scalars > n_bases || (n_scalars && (!bases || !scalars)))
    return ARK_HIP_ERR_ARG;
  if (step == 0) step = (size_t)1 << 20;
  ARK_SCOPE(sc);
  const size_t ab = (size_t)CURVES[curve].fe_words * 16;
  const uint64_t* b0 = bases + (n_bases - n_scalars) * (ab / 8);
  return msm_stream(sc.c, curve, nullptr, b0, scalars, n_scalars, 1, step, out_xyz, false);
}

// fn(g) for every device g < n_gpus, each on its own persistent host thread (DeviceThreads, hostpool.hpp); device 0's share on
// the calling thread.  Without threads to be had the shares run one after another.
extern "C++" {
template <class Fn>
static void for_each_device(int n_gpus, Fn&& fn) {
  struct Tr {
    Fn* f;
    static void call(void* ctx, int g) { (*((Tr*)ctx)->f)(g); }
  } tr{&fn};
  if (!DeviceThreads::instance().run(n_gpus, &Tr::call, &tr))
    for (int g = 0; g < n_gpus; g++) fn(g);
}
}
// One MSM over the GPUs of this node from ONE host process: base-range shards (the reference's own split,
// variable_base/mod.rs:521-557), one host thread and one context per device, partials (3 field elements each) summed
// on the host.  The data path has no collective: RCCL would add nothing to a 144-byte exchange inside one process.
int ark_hip_msm_sw_multi_device(int curve, int n_gpus, const void* const* d_bases, const void* const* d_scalars,
                                const size_t* n_per_gpu, int mont, uint64_t* out_xyz) {
  if (curve < 0 || curve > 4 || n_gpus < 1 || n_gpus > MAX_DEV || !d_bases || !d_scalars || !n_per_gpu || !out_xyz)
    return ARK_HIP_ERR_ARG;
  const size_t pw = (size_t)CURVES[curve].fe_words * 3;
  std::vector<uint64_t> partials((size_t)n_gpus * pw);
  std::vector<int> rcs((size_t)n_gpus, 0);
  const int caller_dev = ark_hip_get_device();
  for_each_device(n_gpus, [&](int g) {
    int rc = ark_hip_set_device(g);
    if (rc == 0) rc = ark_hip_msm_sw_device(curve, d_bases[g], d_scalars[g], n_per_gpu[g], mont, &partials[(size_t)g * pw]);
    rcs[(size_t)g] = rc;
  });
  if (caller_dev >= 0) (void)ark_hip_set_device(caller_dev);   // share 0 ran on this thread
  for (int g = 0; g < n_gpus; g++)
    if (rcs[(size_t)g]) return rcs[(size_t)g];
  return ark_hip_sw_sum(curve, partials.data(), (size_t)n_gpus, out_xyz);
}
int ark_hip_msm_sw_multi(int curve, int n_gpus, const uint64_t* bases, const uint64_t* scalars, size_t n, int mont,
                         uint64_t* out_xyz) {
  if (curve < 0 || curve > 4 || n_gpus < 1 || n_gpus > MAX_DEV || !out_xyz || (n && (!bases || !scalars)))
    return ARK_HIP_ERR_ARG;
  const size_t pw = (size_t)CURVES[curve].fe_words * 3, aw = (size_t)CURVES[curve].fe_words * 2;
  std::vector<uint64_t> partials((size_t)n_gpus * pw);
  std::vector<int> rcs((size_t)n_gpus, 0);
  const int caller_dev = ark_hip_get_device();
  for_each_device(n_gpus, [&](int g) {
    const size_t q = n / (size_t)n_gpus, r = n % (size_t)n_gpus;
    const size_t lo = (size_t)g * q + ((size_t)g < r ? (size_t)g : r), cnt = q + ((size_t)g < r ? 1 : 0);
    int rc = ark_hip_set_device(g);
    if (rc == 0) rc = ark_hip_msm_sw(curve, bases + lo * aw, scalars + lo * 4, cnt, mont, &partials[(size_t)g * pw]);
    rcs[(size_t)g] = rc;
  });
  if (caller_dev >= 0) (void)ark_hip_set_device(caller_dev);   // share 0 ran on this thread
  for (int g = 0; g < n_gpus; g++)
    if (rcs[(size_t)g]) return rcs[(size_t)g];
  return ark_hip_sw_sum(curve, partials.data(), (size_t)n_gpus, out_xyz);
}

// The same split over PREPARED shards (a fixed SRS, one prepared base set per GPU): one host thread per device, each
// through the pinned-ring upload of the prepared entry; the shard sizes cut the scalar vector in order.
int ark_hip_msm_prepared_multi(int n_gpus, const ark_hip_msm_bases* const* shards, const uint64_t* scalars, size_t n, int mont,
                               uint64_t* out_xyz) {
  if (n_gpus < 1 || n_gpus > MAX_DEV || !shards || !out_xyz || (n && !scalars)) return ARK_HIP_ERR_ARG;
  size_t total = 0;
  for (int g = 0; g < n_gpus; g++) {
    if (!shards[g]) return ARK_HIP_ERR_ARG;
    const PreparedBases* pb = (const PreparedBases*)shards[g];
    if (pb->curve != ((const PreparedBases*)shards[0])->curve) return ARK_HIP_ERR_ARG;
    total += pb->n;
  }
  if (n > total) return ARK_HIP_ERR_ARG;   // like msm_unchecked: the scalars may be fewer than the bases, never more
  const int curve = ((const PreparedBases*)shards[0])->curve;
  const size_t pw = (size_t)CURVES[curve].fe_words * 3;
  std::vector<uint64_t> partials((size_t)n_gpus * pw);
  std::vector<int> rcs((size_t)n_gpus, 0);
  std::vector<size_t> los((size_t)n_gpus, 0);
  for (int g = 1; g < n_gpus; g++) los[(size_t)g] = los[(size_t)g - 1] + ((const PreparedBases*)shards[g - 1])->n;
  for_each_device(n_gpus, [&](int g) {
    const PreparedBases* pb = (const PreparedBases*)shards[g];
    const size_t lo = los[(size_t)g];
    const size_t cnt = lo >= n ? 0 : (n - lo < pb->n ? n - lo : pb->n);
    rcs[(size_t)g] = ark_hip_msm_prepared(shards[g], scalars + lo * 4, cnt, mont, &partials[(size_t)g * pw]);  // runs on the shard's device
  });
  for (int g = 0; g < n_gpus; g++)
    if (rcs[(size_t)g]) return rcs[(size_t)g];
  return ark_hip_sw_sum(curve, partials.data(), (size_t)n_gpus, out_xyz);
}

// ---- fixed-base batch multiplication (ScalarMul::batch_mul / BatchMulPreprocessing, ec/src/scalar_mul/mod.rs:104-251) ----
int ark_hip_batch_mul_table_new(int curve, const uint64_t* base_xyz, size_t num_scalars, ark_hip_batch_mul_table** out) {
  // the reference sizes its window from num_scalars (:222-228); so does the device table, by its own cost rule
  // (batchmul.cuh batchmul_window)
  if (curve < 0 || curve > 4 || !base_xyz || !out) return ARK_HIP_ERR_ARG;
  ARK_SCOPE(sc);
  Context* c = sc.c;
  const size_t ab = (size_t)CURVES[curve].fe_words * 16;
  uint64_t aff[24];
  int rc = ark_hip_sw_into_affine(curve, base_xyz, 1, aff);
  if (rc) return rc;
  BatchMulTable* t = new BatchMulTable();
  t->curve = curve;
  t->logical = c->logical;
  t->window = batchmul_window(num_scalars);
  const size_t entries = (size_t)batchmul_outer(t->window) << t->window;
  if (t->table.ensure(entries * ab) || c->stage_c.ensure((size_t)batchmul_build_scratch_dispatch(curve, t->window))) {
    delete t;
    return ARK_HIP_ERR_NOMEM;
  }
  rc = batchmul_build_dispatch(curve, aff, t->window, c->stage_c.p, t->table.p, c->stream);
  if (rc == 0 && hipStreamSynchronize(c->stream) != hipSuccess) rc = -1000;
  if (rc) {
    (void)hipStreamSynchronize(c->stream);
    t->table.release();
    delete t;
    return rc;
  }
  *out = (ark_hip_batch_mul_table*)t;
  return 0;
}
int ark_hip_batch_mul_table_free(ark_hip_batch_mul_table* table) {
  if (!table) return 0;
  BatchMulTable* t = (BatchMulTable*)table;
  Scope sc;
  if (int rc = sc.enter(t->logical)) return rc;
  ARK_HIP_TRY(hipStreamSynchronize(sc.c->stream));
  t->table.release();
  delete t;
  return 0;
}
int ark_hip_batch_mul_device(const ark_hip_batch_mul_table* table, const void* d_scalars, size_t n, int mont, void* d_out_xy) {
  if (!table || (n && (!d_scalars || !d_out_xy))) return ARK_HIP_ERR_ARG;
  const BatchMulTable* t = (const BatchMulTable*)table;
  Scope sc;
  if (int rc = sc.enter(t->logical)) return rc;
  if (n == 0) return 0;
  if (sc.c->stage_c.ensure(n * 2 * (size_t)CURVES[t->curve].fe_words * 16)) return ARK_HIP_ERR_NOMEM;  // XYZZ scratch
  int rc = batchmul_run_dispatch(t->curve, t->table.p, t->window, d_scalars, n, mont, sc.c->stage_c.p, d_out_xy, sc.c->stream);
  if (rc) return rc;
  ARK_HIP_TRY(hipStreamSynchronize(sc.c->stream));
  return 0;
}
int ark_hip_batch_mul(const ark_hip_batch_mul_table* table, const uint64_t* scalars, size_t n, int mont, uint64_t* out_xy) {
  if (!table || (n && (!scalars || !out_xy))) return ARK_HIP_ERR_ARG;
  const BatchMulTable* t = (const BatchMulTable*)table;
  Scope sc;
  if (int rc = sc.enter(t->logical)) return rc;
  Context* c = sc.c;
  const size_t ab = (size_t)CURVES[t->curve].fe_words * 16;
  if (n == 0) return 0;
  if (c->stage_a.ensure(n * 32) || c->stage_b.ensure(n * ab) || c->stage_c.ensure(n * 2 * ab)) return ARK_HIP_ERR_NOMEM;
  ARK_HIP_TRY(hipMemcpyAsync(c->stage_a.p, scalars, n * 32, hipMemcpyHostToDevice, c->stream));
  int rc = batchmul_run_dispatch(t->curve, t->table.p, t->window, c->stage_a.p, n, mont, c->stage_c.p, c->stage_b.p, c->stream);
  if (rc) return rc;
  ARK_HIP_TRY(hipMemcpyAsync(out_xy, c->stage_b.p, n * ab, hipMemcpyDeviceToHost, c->stream));
  ARK_HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

// the public synchronous entries (see retry_while_busy above)
int ark_hip_msm_sw_device(int curve, const void* d_bases, const void* d_scalars, size_t n, int mont, uint64_t* out_xyz) {
  return retry_while_busy(-2, [&]() { return msm_sw_device_once(curve, d_bases, d_scalars, n, mont, out_xyz); });
}
int ark_hip_msm_sw(int curve, const uint64_t* bases, const uint64_t* scalars, size_t n, int mont, uint64_t* out_xyz) {
  return retry_while_busy(-2, [&]() { return msm_sw_once(curve, bases, scalars, n, mont, out_xyz); });
}
int ark_hip_msm_sw_small_device(int curve, const void* d_bases, const void* d_scalars, size_t n, int scalar_bytes, int max_bits,
                                uint64_t* out_xyz) {
  return retry_while_busy(-2, [&]() { return msm_sw_small_device_once(curve, d_bases, d_scalars, n, scalar_bytes, max_bits, out_xyz); });
}
int ark_hip_msm_sw_small(int curve, const uint64_t* bases, const void* scalars, size_t n, int scalar_bytes, int max_bits,
                         uint64_t* out_xyz) {
  return retry_while_busy(-2, [&]() { return msm_sw_small_once(curve, bases, scalars, n, scalar_bytes, max_bits, out_xyz); });
}
int ark_hip_msm_prepared_device(const ark_hip_msm_bases* bases, const void* d_scalars, size_t n, int mont,
                                uint64_t* out_xyz) {
  if (!bases) return ARK_HIP_ERR_ARG;
  return retry_while_busy(((const PreparedBases*)bases)->logical, [&]() { return msm_prepared_device_once(bases, d_scalars, n, mont, out_xyz); });
}
int ark_hip_msm_prepared(const ark_hip_msm_bases* bases, const uint64_t* scalars, size_t n, int mont, uint64_t* out_xyz) {
  if (!bases) return ARK_HIP_ERR_ARG;
  return retry_while_busy(((const PreparedBases*)bases)->logical, [&]() { return msm_prepared_once(bases, scalars, n, mont, out_xyz); });
}
int ark_hip_msm_prepared_small_device(const ark_hip_msm_bases* bases, const void* d_scalars, size_t n, int scalar_bytes,
                                      int max_bits, uint64_t* out_xyz) {
  if (!bases) return ARK_HIP_ERR_ARG;
  return retry_while_busy(((const PreparedBases*)bases)->logical, [&]() { return msm_prepared_small_device_once(bases, d_scalars, n, scalar_bytes, max_bits, out_xyz); });
}
int ark_hip_msm_sw_chunks(int curve, const uint64_t* bases, size_t n_bases, const uint64_t* scalars, size_t n_scalars,
                          size_t step, uint64_t* out_xyz) {
  return retry_while_busy(-2, [&]() { return msm_sw_chunks_once(curve, bases, n_bases, scalars, n_scalars, step, out_xyz); });
}
}  // extern "C"
