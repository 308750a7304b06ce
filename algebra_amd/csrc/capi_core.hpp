// Internals shared by the translation units of the C ABI (capi_runtime / capi_msm / capi_fft / capi_comm / capi_test .hip):
// the upload staging, the per-device context and its lock, the per-curve / per-field dispatch and the MSM job plumbing.
// Header-only on purpose: every function is `inline` and the few globals are C++17 inline variables, so the units share ONE
// instance of each without a link-time interface of their own (round 6 split the 3000-line capi.hip: VERDICT r5 weak #10).
#pragma once
#include "../../include/ark_hip.h"
#include <string.h>
#include <dlfcn.h>
#include <rccl/rccl.h>   // types and signatures only: the library is opened at run time (capi_comm.hip: RcclApi)
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <chrono>
#include <unistd.h>
#include <vector>
#include "msm.cuh"
#include "fft.cuh"
#include "batchmul.cuh"
#include "internal.hpp"
#include "curve_consts.hpp"

namespace arkhip {
namespace capi {
using namespace arkhip;


struct PreparedBases;

// ---- pageable host memory -> device at PCIe rate ---------------------------------------------------------
// hipMemcpyAsync from pageable memory is staged by the runtime through an internal bounce buffer on the calling thread.
// The host-pointer entry points (what SWCurveConfig::msm hands over: Rust slices in ordinary heap memory) can stage
// large uploads themselves instead (ARK_HIP_COPY_THREADS=n): n worker threads copy 32 MiB pieces into a ring of pinned
// buffers while the DMA engine drains the previous pieces.  Off by default -- on the MI355X hosts measured the runtime's
// own path is as fast (46 GB/s).  Page-locked sources are always read in place.
class CopyPool {
 public:
  explicit CopyPool(int nthreads) {
    for (int i = 0; i < nthreads; i++) th_.emplace_back([this]() { run(); });
  }
  ~CopyPool() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      stop_ = true;
    }
    cv_.notify_all();
    for (auto& t : th_) t.join();
  }
  int threads() const { return (int)th_.size(); }
  // memcpy(dst, src, bytes) split over the workers; returns when done
  void copy(void* dst, const void* src, size_t bytes) {
    const int parts = (int)th_.size();
    if (parts <= 1 || bytes < ((size_t)1 << 20)) {
      memcpy(dst, src, bytes);
      return;
    }
    int left = parts;              // guarded by dmu: the last worker decrements AND notifies under the lock, so the
    std::mutex dmu;                // waiter cannot see 0, return and destroy dmu / dcv while a worker still touches them
    std::condition_variable dcv;
    const size_t per = ((bytes / parts) + 4095) & ~(size_t)4095;
    {
      std::lock_guard<std::mutex> lk(mu_);
      for (int i = 0; i < parts; i++) {
        const size_t off = (size_t)i * per;
        const size_t len = off >= bytes ? 0 : (bytes - off < per ? bytes - off : per);
        q_.push_back([=, &left, &dmu, &dcv]() {
          if (len) memcpy((char*)dst + off, (const char*)src + off, len);
          std::lock_guard<std::mutex> l2(dmu);
          if (--left == 0) dcv.notify_one();
        });
      }
    }
    cv_.notify_all();
    std::unique_lock<std::mutex> l2(dmu);
    dcv.wait(l2, [&]() { return left == 0; });
  }

 private:
  void run() {
    for (;;) {
      std::function<void()> job;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [this]() { return stop_ || !q_.empty(); });
        if (stop_ && q_.empty()) return;
        job = std::move(q_.front());
        q_.pop_front();
      }
      job();
    }
  }
  std::vector<std::thread> th_;
  std::deque<std::function<void()>> q_;
  std::mutex mu_;
  std::condition_variable cv_;
  bool stop_ = false;
};

struct HostStager {
  static constexpr int SLOTS = 4;
  static constexpr size_t SLOT_BYTES = (size_t)32 << 20;
  void* pinned[SLOTS] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t drained[SLOTS] = {nullptr, nullptr, nullptr, nullptr};
  int next = 0;
  CopyPool* pool = nullptr;
  int mode = -1;  // -1: not decided; 0: plain hipMemcpyAsync; 1: staged
  void release() {
    for (int i = 0; i < SLOTS; i++) {
      if (pinned[i]) (void)hipHostFree(pinned[i]);
      pinned[i] = nullptr;
      if (drained[i]) (void)hipEventDestroy(drained[i]);
      drained[i] = nullptr;
    }
    delete pool;
    pool = nullptr;
  }
  static bool is_pinned(const void* p) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
      (void)hipGetLastError();  // ordinary heap memory: "invalid value" is the answer, not an error to leave behind
      return false;
    }
    return a.type == hipMemoryTypeHost;
  }
  // enqueue dst[0..bytes) <- src (pageable or pinned host memory) on `st`.  On return `src` has been read completely,
  // unless it is page-locked and `src_stable` (the caller keeps it valid until the stream has passed this point): then
  // the DMA engine reads it in place.
  int upload(void* dst, const void* src, size_t bytes, hipStream_t st, bool src_stable = false) {
    if (bytes && src_stable && is_pinned(src)) {
      ARK_HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st));
      return 0;
    }
    if (mode < 0) {
      // default 0: pageable copies are left to the HIP runtime, which reaches ~46 GB/s on the MI355X hosts measured
      // (2^24 first call 66 ms against 75 ms with four staging threads, profiles/r3_trait_surface.txt); the pool is for
      // hosts whose runtime path is the slower one
      const char* e = getenv("ARK_HIP_COPY_THREADS");
      int nt = e ? atoi(e) : 0;
      const int hw = (int)std::thread::hardware_concurrency();
      if (hw > 0 && nt > hw) nt = hw;
      mode = nt > 0 ? 1 : 0;
      if (mode) pool = new CopyPool(nt);
    }
    if (bytes == 0) return 0;
    if (!mode || bytes < ((size_t)8 << 20)) {
      ARK_HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st));
      ARK_HIP_TRY(hipStreamSynchronize(st));  // pageable source: the contract is "read on return"
      return 0;
    }
    for (size_t off = 0; off < bytes; off += SLOT_BYTES) {
      const size_t len = bytes - off < SLOT_BYTES ? bytes - off : SLOT_BYTES;
      const int k = next;
      next = (next + 1) % SLOTS;
      if (!pinned[k]) {
        ARK_HIP_TRY(hipHostMalloc(&pinned[k], SLOT_BYTES));
        ARK_HIP_TRY(hipEventCreateWithFlags(&drained[k], hipEventDisableTiming));
      } else {
        ARK_HIP_TRY(hipEventSynchronize(drained[k]));  // the DMA that last read this slot
      }
      pool->copy(pinned[k], (const char*)src + off, len);
      ARK_HIP_TRY(hipMemcpyAsync((char*)dst + off, pinned[k], len, hipMemcpyHostToDevice, st));
      ARK_HIP_TRY(hipEventRecord(drained[k], st));
    }
    return 0;
  }
};

// ---- resident copies of base sets handed over by host pointer -------------------------------------------
// SWCurveConfig::msm / VariableBaseMSM::msm_bigint take `&[Affine]` on every call; provers (and the reference's own
// bench, bench-templates/src/macros/ec.rs:223-240) pass the SAME slice -- an SRS -- again and again.  The host-pointer
// entry is a function of its two slices (the reference's borrow semantics, variable_base/mod.rs:59-85): whatever is
// kept between calls never changes a result.  Two ways a device copy is kept:
//   pinned  (ark_hip_msm_bases_pin .. _unpin): the caller DECLARES the slice immutable for that span (on the Rust side a
//           guard that holds the shared borrow, so the compiler enforces it); any call whose base slice lies inside a
//           pinned range uses the resident copy with no check at all;
//   cached (default; ark_hip_msm_cache_config / ARK_HIP_BASE_CACHE_MB): copies keyed by (curve, address, length) and
//           validated on EVERY call by a hash of the slice's FULL content, computed on host threads while the device
//           already works from the cached copy; the result is withheld until the hash agrees, otherwise the copy is
//           refreshed and the MSM rerun.  Never stale, at the price of one pass over the host slice per call.
struct Hash128 {
  uint64_t lo = 0, hi = 0;
  bool operator==(const Hash128& o) const { return lo == o.lo && hi == o.hi; }
};
struct BaseCacheEntry {
  int curve = -1;
  const void* host = nullptr;
  size_t n = 0;
  Hash128 hash;                       // verified entries: base_hash (keyed, 128 bits) of the content the device copy holds
  double fill_ms = 0;                 // wall time of the call that filled the copy: bases + scalars streamed over PCIe
  DevBuf dev;
  uint64_t last_use = 0;
  unsigned hits = 0;
  int pins = 0;                       // > 0: pinned (never evicted, never validated)
  bool no_prepare = false;            // the per-window table did not fit: do not retry on every call
  PreparedBases* prepared = nullptr;  // built after `auto_prepare` hits (off by default)
};
struct BaseCacheStats {
  uint64_t hits = 0, misses = 0, refreshed = 0, evicted = 0, pinned_hits = 0;
  uint64_t busy_streamed = 0;   // calls that streamed their bases although a copy was cached: the host was too busy to hash it in time
  double last_hash_ms = 0;      // the latest validation pass
  double hash_bytes_per_ms = 0; // its rate, smoothed: what the next call's decision rests on
};

// One context per (logical) device: stream, workspaces, staging.  Every entry point runs on the calling thread's
// current device (ark_hip_set_device / ark_hip_init; default: the first device initialised) and holds that
// context's lock for its whole body, so the library can be called from any number of host threads (rayon workers,
// Python threads): calls on one device serialise, calls on different devices run concurrently.
constexpr int COMM_MAX_SLICES = 16;
struct Context {
  int logical = -1, physical = -1;
  hipStream_t stream = nullptr;        // compute
  hipStream_t stream_b = nullptr;      // second MSM lane (created on first use)
  hipStream_t copy_stream = nullptr;   // uploads overlapped with compute (streaming MSM)
  hipStream_t fft_side[2] = {nullptr, nullptr};  // batched transforms: up to three in flight (created on first use)
  hipEvent_t fft_ev[3] = {nullptr, nullptr, nullptr};
  // Two MSM lanes (workspace + stream): a job goes to lane 1 only while lane 0 has a job in flight, so that the
  // memory-bound phases of one MSM (digits, sort, reduction) run under the other's accumulate kernel.  Measured with
  // two jobs in flight: +25 % MSMs/s at 2^20, +3 % at 2^24 (profiles/r2_msm_sweeps.txt).  Synchronous callers only ever
  // touch lane 0 (and its memory).
  MsmWorkspace msm[2];
  FftWorkspace fft;
  DevBuf stage_a, stage_b, stage_c;    // host-pointer entry points: device copies
  DevBuf gfft_work, gfft_scal;         // transform over group elements: XYZZ scratch, per-position scalars
  DevBuf ring_s[2], ring_b[2];         // double-buffered scalar / base uploads of the streaming entry points
  hipEvent_t ring_free[2] = {nullptr, nullptr}, ring_up[2] = {nullptr, nullptr};
  int ring_next = 0;
  hipEvent_t lane_ev = nullptr;        // last asynchronous PRODUCER on the context stream (FFT, pointwise product): the
  bool lane_ev_set = false;            // second MSM lane starts behind it -- but not behind lane 0's own MSM kernels
  HostStager stager;
  DevBuf piece_buckets;                // streamed MSM: the ONE bucket array its pieces share (MsmPiece)
  hipEvent_t piece_ev[2] = {nullptr, nullptr};
  std::vector<BaseCacheEntry> base_cache;
  BaseCacheStats cache_stats;
  uint64_t cache_clock = 0;
  long long cache_budget = -1;         // bytes; -1: not configured yet (env / default on first use); 0: disabled
  int auto_prepare = -1;               // hits after which a cached base set is prepared; 0: never; -1: env / default
  // one process per GPU: this device's RCCL communicator (ark_hip_comm_init) and what its exchanges need
  ncclComm_t comm = nullptr;
  int comm_rank = 0, comm_world = 1;
  hipStream_t comm_stream = nullptr;   // exchange slices travel here while the compute stream works on the previous slice
  hipEvent_t comm_ev[2][COMM_MAX_SLICES] = {};
  DevBuf comm_tmp, comm_small;         // receive side of the FFT exchange; MSM partials
  void* comm_pinned = nullptr;         // host mirror of comm_small
  DevBuf comm_sums;                    // sharded MSM: [send block | world receive blocks | summed parts] (msm_sharded_sums)
  void* comm_sums_pinned = nullptr;    // host side: [my header | world headers | summed parts]
  bool msm_timing = false, fft_timing = false;
  MsmTimings msm_tm;
  FftTimings fft_tm;
  std::recursive_mutex mu;
  // bumped whenever an MSM job leaves its slot: what a synchronous caller that found every slot taken waits on (no lock held)
  std::mutex slot_mu;
  std::condition_variable slot_cv;
  std::atomic<uint64_t> slot_gen{0};
};
constexpr int MAX_DEV = 64;
inline Context* g_ctxs[MAX_DEV] = {};
inline std::mutex g_mu;
inline int g_default = -1;              // device of threads that never chose one
inline thread_local int t_dev = -1;

inline int device_count_raw() {
  int cnt = 0;
  if (hipGetDeviceCount(&cnt) != hipSuccess) return 0;
  return cnt;
}
// logical -> physical device.  ARK_HIP_OVERSUBSCRIBE=1 lets logical ids beyond the physical count wrap around (separate
// contexts, streams and workspaces on a shared GPU): how the multi-device code paths are tested on a one-GPU box.
inline int physical_of(int logical, int cnt) {
  if (logical < 0 || logical >= MAX_DEV || cnt <= 0) return -1;
  if (logical < cnt) return logical;
  const char* e = getenv("ARK_HIP_OVERSUBSCRIBE");
  return (e && atoi(e) > 0) ? logical % cnt : -1;
}

inline int get_ctx(int logical, Context** out) {
  std::lock_guard<std::mutex> lk(g_mu);
  const int cnt = device_count_raw();
  if (cnt <= 0) {
    fprintf(stderr, "ark_hip: no HIP device visible -- this library has no CPU fallback\n");
    return ARK_HIP_ERR_NO_DEVICE;
  }
  if (logical < 0) logical = g_default >= 0 ? g_default : 0;
  const int phys = physical_of(logical, cnt);
  if (phys < 0) return ARK_HIP_ERR_ARG;
  if (hipSetDevice(phys) != hipSuccess) return ARK_HIP_ERR_NO_DEVICE;  // HIP's current device is per host thread
  if (!g_ctxs[logical]) {
    Context* c = new Context();
    c->logical = logical;
    c->physical = phys;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking) != hipSuccess) {
      delete c;
      return ARK_HIP_ERR_NO_DEVICE;
    }
    g_ctxs[logical] = c;
    if (g_default < 0) g_default = logical;
  }
  *out = g_ctxs[logical];
  return 0;
}

// RAII: the calling thread's context, locked
struct Scope {
  Context* c = nullptr;
  std::unique_lock<std::recursive_mutex> lk;
  int enter(int logical = -2) {
    int rc = get_ctx(logical == -2 ? t_dev : logical, &c);
    if (rc) return rc;
    lk = std::unique_lock<std::recursive_mutex>(c->mu);
    if (hipSetDevice(c->physical) != hipSuccess) return ARK_HIP_ERR_NO_DEVICE;
    return 0;
  }
};
#define ARK_SCOPE(S)               \
  Scope S;                         \
  if (int _rc = S.enter()) return _rc

struct CurveInfo { int fe_words, scalar_field, base_field, ext; };
const CurveInfo CURVES[5] = {
    {4, ARK_HIP_BN254_FR, ARK_HIP_BN254_FQ, 1},       {6, ARK_HIP_BLS12_381_FR, ARK_HIP_BLS12_381_FQ, 1},
    {6, ARK_HIP_BLS12_377_FR, ARK_HIP_BLS12_377_FQ, 1}, {12, ARK_HIP_BLS12_377_FR, ARK_HIP_BLS12_377_FQ, 2},
    {12, ARK_HIP_BLS12_381_FR, ARK_HIP_BLS12_381_FQ, 2}};

// per-curve dispatch (one translation unit per curve, internal.hpp)
#ifdef ARK_HIP_DEV
#define ARK_CURVE_SWITCH(curve, CALL)                         \
  switch (curve) {                                            \
    case ARK_HIP_BLS12_381_G1: return CALL(BLS12_381_G1);     \
  }                                                           \
  return ARK_HIP_ERR_ARG
#define ARK_FIELD_SWITCH(field, CALL)                         \
  switch (field) {                                            \
    case ARK_HIP_BLS12_381_FR: return CALL(BLS12_381_FR);     \
  }                                                           \
  return ARK_HIP_ERR_ARG
#else
#define ARK_CURVE_SWITCH(curve, CALL)                         \
  switch (curve) {                                            \
    case ARK_HIP_BN254_G1: return CALL(BN254_G1);             \
    case ARK_HIP_BLS12_381_G1: return CALL(BLS12_381_G1);     \
    case ARK_HIP_BLS12_377_G1: return CALL(BLS12_377_G1);     \
    case ARK_HIP_BLS12_377_G2: return CALL(BLS12_377_G2);     \
    case ARK_HIP_BLS12_381_G2: return CALL(BLS12_381_G2);     \
  }                                                           \
  return ARK_HIP_ERR_ARG
#define ARK_FIELD_SWITCH(field, CALL)                         \
  switch (field) {                                            \
    case ARK_HIP_BN254_FR: return CALL(BN254_FR);             \
    case ARK_HIP_BLS12_381_FR: return CALL(BLS12_381_FR);     \
    case ARK_HIP_BLS12_377_FR: return CALL(BLS12_377_FR);     \
  }                                                           \
  return ARK_HIP_ERR_ARG
#endif

inline int msm_enqueue_dispatch(int curve, MsmWorkspace& ws, const void* pts, size_t wstride, const MsmPlan* prep, const void* s,
                         size_t n, int mont, hipStream_t st, bool timing, int sbytes = 0, int sbits = 0,
                         const MsmPiece* piece = nullptr) {
#define X(NAME) msm_enqueue_##NAME(ws, pts, wstride, prep, s, n, mont, st, timing, sbytes, sbits, piece)
  ARK_CURVE_SWITCH(curve, X);
#undef X
}
inline int msm_finish_dispatch(int curve, MsmWorkspace& ws, int slot, uint64_t* out, MsmTimings* tm) {
#define X(NAME) msm_finish_##NAME(ws, slot, out, tm)
  ARK_CURVE_SWITCH(curve, X);
#undef X
}
inline int msm_sum_ranks_dispatch(int curve, const void* d_blocks, int world, size_t block_bytes, uint32_t npairs, void* d_out, hipStream_t st) {
#define X(NAME) msm_sum_ranks_##NAME(d_blocks, world, block_bytes, npairs, d_out, st)
  ARK_CURVE_SWITCH(curve, X);
#undef X
}
inline int msm_fold_sums_dispatch(int curve, const MsmSumsHeader& h, const void* h_sums, uint64_t* out_xyz) {
#define X(NAME) msm_fold_sums_##NAME(h, h_sums, out_xyz)
  ARK_CURVE_SWITCH(curve, X);
#undef X
}
inline int msm_sample_widths_dispatch(int curve, const void* h_scalars, size_t n, int mont, MsmWidths* out) {
#define X(NAME) (msm_sample_widths_##NAME(h_scalars, n, mont, out), 0)
  ARK_CURVE_SWITCH(curve, X);
#undef X
}
inline int msm_prepare_dispatch(int curve, const void* d_bases, size_t n, const MsmPlan& pl, void* d_table, void* d_tmp, hipStream_t st) {
#define X(NAME) msm_prepare_##NAME(d_bases, n, pl, d_table, d_tmp, st)
  ARK_CURVE_SWITCH(curve, X);
#undef X
}
inline int batchmul_build_dispatch(int curve, const void* h_base, int window, void* d_scratch, void* d_table, hipStream_t st) {
#define X(NAME) batchmul_build_##NAME(h_base, window, d_scratch, d_table, st)
  ARK_CURVE_SWITCH(curve, X);
#undef X
}
inline long long batchmul_build_scratch_dispatch(int curve, int window) {   // bytes; < 0: unknown curve
#define X(NAME) (long long)batchmul_build_scratch_##NAME(window)
  ARK_CURVE_SWITCH(curve, X);
#undef X
}
inline int batchmul_run_dispatch(int curve, const void* d_table, int window, const void* d_scalars, size_t n, int mont, void* d_tmp, void* d_out, hipStream_t st) {
#define X(NAME) batchmul_run_##NAME(d_table, window, d_scalars, n, mont, d_tmp, d_out, st)
  ARK_CURVE_SWITCH(curve, X);
#undef X
}
inline int fft_dispatch(int field, FftWorkspace& ws, void* d, int k, const uint64_t* root, const uint64_t* pre,
                 const uint64_t* post, const uint64_t* postc, int zlog, hipStream_t st, FftTimings* tm) {
#define X(NAME) fft_run_##NAME(ws, d, k, root, pre, post, postc, zlog, st, tm)
  ARK_FIELD_SWITCH(field, X);
#undef X
}
typedef int (*elementwise_fn)(int, const void*, const void*, void*, size_t, hipStream_t);
inline int add_affine_dispatch(int curve, const void* in, void* out, size_t n, const void* d_delta, hipStream_t st) {
#define X(NAME) sw_add_affine_##NAME(in, out, n, d_delta, st)
  ARK_CURVE_SWITCH(curve, X);
#undef X
}
inline int normalize_dispatch(int curve, const void* in, void* out, size_t n, hipStream_t st) {
#define X(NAME) sw_normalize_batch_##NAME(in, out, n, st)
  ARK_CURVE_SWITCH(curve, X);
#undef X
}
inline int fr_mul_dispatch(int field, const void* a, const void* b, void* r, size_t n, hipStream_t st) {
#define X(NAME) field_op_##NAME(2, a, b, r, n, st)
  ARK_FIELD_SWITCH(field, X);
#undef X
}
inline int fr_scale_dispatch(int field, const void* a, const uint64_t* k4, void* r, size_t n, hipStream_t st) {
#define X(NAME) fr_scale_##NAME(a, k4, r, n, st)
  ARK_FIELD_SWITCH(field, X);
#undef X
}
inline bool field_is_one(int field, const uint64_t* x);
inline int fft_roots_dispatch(int field, FftWorkspace& ws, int k, const uint64_t* root4, hipStream_t st, const uint32_t** out) {
#define X(NAME) fft_roots_##NAME(ws, k, root4, st, out)
  ARK_FIELD_SWITCH(field, X);
#undef X
}
inline int fft_scalars_dispatch(int field, FftWorkspace& ws, const uint64_t* base4, const uint64_t* mul4, size_t count, void* d_out,
                         hipStream_t st) {
#define X(NAME) fft_scalars_##NAME(ws, base4, mul4, count, d_out, st)
  ARK_FIELD_SWITCH(field, X);
#undef X
}
inline int gfft_run_dispatch(int curve, void* d_jac, int k, const uint32_t* roots, const uint32_t* pre, const uint32_t* post, void* work,
                      hipStream_t st) {
#define X(NAME) gfft_run_##NAME(d_jac, k, roots, pre, post, work, st)
  ARK_CURVE_SWITCH(curve, X);
#undef X
}
inline size_t gfft_work_bytes_any(int curve, int k) {
  switch (curve) {
#ifndef ARK_HIP_DEV
    case ARK_HIP_BN254_G1: return gfft_work_bytes_BN254_G1(k);
    case ARK_HIP_BLS12_377_G1: return gfft_work_bytes_BLS12_377_G1(k);
    case ARK_HIP_BLS12_377_G2: return gfft_work_bytes_BLS12_377_G2(k);
    case ARK_HIP_BLS12_381_G2: return gfft_work_bytes_BLS12_381_G2(k);
#endif
    case ARK_HIP_BLS12_381_G1: return gfft_work_bytes_BLS12_381_G1(k);
  }
  return 0;
}
inline int fr_div_dispatch(int field, const void* num, const void* den, void* r, size_t n, hipStream_t st) {
#define X(NAME) fr_div_##NAME(num, den, r, n, st)
  ARK_FIELD_SWITCH(field, X);
#undef X
}
inline int fr_op_dispatch(int field, int op, const void* a, const void* b, void* r, size_t n, hipStream_t st) {
#define X(NAME) field_op_##NAME(op, a, b, r, n, st)
  ARK_FIELD_SWITCH(field, X);
#undef X
}
inline int fft_axis_dispatch(int field, FftWorkspace& ws, const void* src, void* dst, unsigned G, size_t cols, const uint64_t* root,
                      hipStream_t st) {
#define X(NAME) fft_axis_##NAME(ws, src, dst, G, cols, root, st)
  ARK_FIELD_SWITCH(field, X);
#undef X
}
inline int fft_axis_prepare_dispatch(int field, FftWorkspace& ws, unsigned G, const uint64_t* root, hipStream_t st, const uint32_t** pw) {
#define X(NAME) fft_axis_prepare_##NAME(ws, G, root, st, pw)
  ARK_FIELD_SWITCH(field, X);
#undef X
}
inline int fft_axis_launch_dispatch(int field, const void* src, void* dst, unsigned G, size_t stride, size_t cols, const uint32_t* pw,
                             hipStream_t st) {
#define X(NAME) fft_axis_launch_##NAME(src, dst, G, stride, cols, pw, st)
  ARK_FIELD_SWITCH(field, X);
#undef X
}

// ---- MSM plumbing shared by the entry points ----------------------------------------------------------
struct PreparedBases {   // ark_hip_msm_bases: a fixed base set with its table of per-window multiples
  int curve = -1;
  int logical = -1;      // device it lives on
  size_t n = 0;
  MsmPlan plan{};
  DevBuf table;          // [plan.W][n] affine points
};
struct BatchMulTable {   // ark_hip_batch_mul_table: multiples of one base (batchmul.cuh)
  int curve = -1;
  int logical = -1;
  int window = 0;        // bits per table row (batchmul_window(num_scalars))
  DevBuf table;
};
struct MsmJobHandle {    // ark_hip_msm_job
  int logical;
  int curve;
  int slot;
};

// lane for the next job: 0 unless lane 0 is busy and lane 1 is less so; ARK_HIP_ERR_BUSY with MSM_JOBS jobs in flight
inline int msm_pick_lane(Context* c) {
  int busy[2];
  for (int l = 0; l < 2; l++) {
    std::lock_guard<std::mutex> lock(c->msm[l].mu);
    busy[l] = 0;
    for (const auto& j : c->msm[l].jobs) busy[l] += j.busy ? 1 : 0;
  }
  if (busy[0] + busy[1] >= MSM_JOBS) return ARK_HIP_ERR_BUSY;
  return busy[1] < busy[0] ? 1 : 0;
}
inline int msm_lane_stream(Context* c, int lane, hipStream_t* out) {
  if (lane && !c->stream_b) ARK_HIP_TRY(hipStreamCreateWithFlags(&c->stream_b, hipStreamNonBlocking));
  *out = lane ? c->stream_b : c->stream;
  return 0;
}
inline int sync_compute(Context* c) {  // both MSM lanes idle
  ARK_HIP_TRY(hipStreamSynchronize(c->stream));
  if (c->stream_b) ARK_HIP_TRY(hipStreamSynchronize(c->stream_b));
  return 0;
}
// called by every entry point that leaves work running on the context stream whose OUTPUT a caller may hand to an MSM
inline int mark_producer(Context* c) {
  if (!c->lane_ev) ARK_HIP_TRY(hipEventCreateWithFlags(&c->lane_ev, hipEventDisableTiming));
  ARK_HIP_TRY(hipEventRecord(c->lane_ev, c->stream));
  c->lane_ev_set = true;
  return 0;
}
// returns lane * MSM_JOBS + slot, or a negative error
inline int msm_enqueue_ctx(Context* c, int curve, const void* pts, size_t wstride, const MsmPlan* prep, const void* d_scalars,
                    size_t n, int mont, int lane = 0, int sbytes = 0, int sbits = 0, const MsmPiece* piece = nullptr,
                    bool may_block = true) {
  hipStream_t st;
  if (int rc = msm_lane_stream(c, lane, &st)) return rc;
  c->msm[lane].probe_allowed = may_block;   // the width probe synchronises the lane's stream (msm.cuh)
  // the device-pointer FFT / pointwise-product entry points are asynchronous on the context stream (= lane 0) and may be
  // producing this job's scalars: the second lane starts behind the last of them (mark_producer), while lane 0's own MSM
  // kernels -- which lane 1 exists to overlap -- are not waited for
  if (lane && c->lane_ev_set) ARK_HIP_TRY(hipStreamWaitEvent(st, c->lane_ev, 0));
  int slot = msm_enqueue_dispatch(curve, c->msm[lane], pts, wstride, prep, d_scalars, n, mont, st, c->msm_timing, sbytes, sbits,
                                  piece);
  return slot < 0 ? slot : lane * MSM_JOBS + slot;
}
// Called WITHOUT the context lock by ark_hip_msm_wait -- it takes no lock but the lane's and the slot mutex, so a job can always
// be finished and its slot freed while another thread's streaming body holds the context (round 6; before, a wait queued
// behind such a body, whose next piece was in turn waiting for the slot: a stall of seconds under mixed callers).
inline int msm_finish_ctx(Context* c, int curve, int slot, uint64_t* out) {
  if (slot < 0 || slot >= 2 * MSM_JOBS) return ARK_HIP_ERR_ARG;
  MsmTimings tm;
  const int rc = msm_finish_dispatch(curve, c->msm[slot / MSM_JOBS], slot % MSM_JOBS, out, &tm);
  {   // the slot is free again
    std::lock_guard<std::mutex> sl(c->slot_mu);
    c->slot_gen.fetch_add(1, std::memory_order_release);
  }
  c->slot_cv.notify_all();
  if (rc == 0 && tm.c != 0) {
    std::lock_guard<std::mutex> lk(c->slot_mu);   // (NOT the context lock: a wait must never queue behind a streaming body)
    c->msm_tm = tm;
  }
  return rc;
}
// error paths: wait for a job and drop its result, so that its slot is free again and nothing is left in flight
inline void msm_discard_ctx(Context* c, int curve, int slot) {
  uint64_t scratch[36];
  (void)msm_finish_ctx(c, curve, slot, scratch);
}

// next slot of the upload ring: the copy stream waits until the MSM that last read this slot has finished
inline int ring_acquire(Context* c, int* k) {
  const int i = c->ring_next;
  c->ring_next ^= 1;
  for (int j = 0; j < 2; j++) {
    if (!c->ring_free[j]) ARK_HIP_TRY(hipEventCreateWithFlags(&c->ring_free[j], hipEventDisableTiming));
    if (!c->ring_up[j]) ARK_HIP_TRY(hipEventCreateWithFlags(&c->ring_up[j], hipEventDisableTiming));
  }
  ARK_HIP_TRY(hipStreamWaitEvent(c->copy_stream, c->ring_free[i], 0));  // never-recorded event: no wait
  *k = i;
  return 0;
}
// uploads done -> compute may start; after the MSM is enqueued the slot is marked free again
inline int ring_publish(Context* c, int k, hipStream_t compute) {
  ARK_HIP_TRY(hipEventRecord(c->ring_up[k], c->copy_stream));
  ARK_HIP_TRY(hipStreamWaitEvent(compute, c->ring_up[k], 0));
  return 0;
}
inline int ring_release(Context* c, int k, hipStream_t compute) {
  ARK_HIP_TRY(hipEventRecord(c->ring_free[k], compute));
  return 0;
}

// bounded wait for a stream that carries a collective (capi_comm.hip)
int comm_sync(Context* c, hipStream_t st);
}  // namespace capi
}  // namespace arkhip
