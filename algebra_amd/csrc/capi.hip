// C ABI of libark_hip.so (see include/ark_hip.h for the contract and the reference items replaced).
#include "../../include/ark_hip.h"
#include <string.h>
#include <dlfcn.h>
#include <rccl/rccl.h>   // types and signatures only: the library is opened at run time (RcclApi)
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

using namespace arkhip;

namespace {

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
};
constexpr int MAX_DEV = 64;
Context* g_ctxs[MAX_DEV] = {};
std::mutex g_mu;
int g_default = -1;              // device of threads that never chose one
thread_local int t_dev = -1;

int device_count_raw() {
  int cnt = 0;
  if (hipGetDeviceCount(&cnt) != hipSuccess) return 0;
  return cnt;
}
// logical -> physical device.  ARK_HIP_OVERSUBSCRIBE=1 lets logical ids beyond the physical count wrap around (separate
// contexts, streams and workspaces on a shared GPU): how the multi-device code paths are tested on a one-GPU box.
int physical_of(int logical, int cnt) {
  if (logical < 0 || logical >= MAX_DEV || cnt <= 0) return -1;
  if (logical < cnt) return logical;
  const char* e = getenv("ARK_HIP_OVERSUBSCRIBE");
  return (e && atoi(e) > 0) ? logical % cnt : -1;
}

int get_ctx(int logical, Context** out) {
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

int msm_enqueue_dispatch(int curve, MsmWorkspace& ws, const void* pts, size_t wstride, const MsmPlan* prep, const void* s,
                         size_t n, int mont, hipStream_t st, bool timing, int sbytes = 0, int sbits = 0,
                         const MsmPiece* piece = nullptr) {
#define X(NAME) msm_enqueue_##NAME(ws, pts, wstride, prep, s, n, mont, st, timing, sbytes, sbits, piece)
  ARK_CURVE_SWITCH(curve, X);
#undef X
}
int msm_finish_dispatch(int curve, MsmWorkspace& ws, int slot, uint64_t* out, MsmTimings* tm) {
#define X(NAME) msm_finish_##NAME(ws, slot, out, tm)
  ARK_CURVE_SWITCH(curve, X);
#undef X
}
int msm_sum_ranks_dispatch(int curve, const void* d_blocks, int world, size_t block_bytes, uint32_t npairs, void* d_out, hipStream_t st) {
#define X(NAME) msm_sum_ranks_##NAME(d_blocks, world, block_bytes, npairs, d_out, st)
  ARK_CURVE_SWITCH(curve, X);
#undef X
}
int msm_fold_sums_dispatch(int curve, const MsmSumsHeader& h, const void* h_sums, uint64_t* out_xyz) {
#define X(NAME) msm_fold_sums_##NAME(h, h_sums, out_xyz)
  ARK_CURVE_SWITCH(curve, X);
#undef X
}
int msm_sample_widths_dispatch(int curve, const void* h_scalars, size_t n, int mont, MsmWidths* out) {
#define X(NAME) (msm_sample_widths_##NAME(h_scalars, n, mont, out), 0)
  ARK_CURVE_SWITCH(curve, X);
#undef X
}
int msm_prepare_dispatch(int curve, const void* d_bases, size_t n, const MsmPlan& pl, void* d_table, void* d_tmp, hipStream_t st) {
#define X(NAME) msm_prepare_##NAME(d_bases, n, pl, d_table, d_tmp, st)
  ARK_CURVE_SWITCH(curve, X);
#undef X
}
int batchmul_build_dispatch(int curve, const void* h_base, int window, void* d_scratch, void* d_table, hipStream_t st) {
#define X(NAME) batchmul_build_##NAME(h_base, window, d_scratch, d_table, st)
  ARK_CURVE_SWITCH(curve, X);
#undef X
}
long long batchmul_build_scratch_dispatch(int curve, int window) {   // bytes; < 0: unknown curve
#define X(NAME) (long long)batchmul_build_scratch_##NAME(window)
  ARK_CURVE_SWITCH(curve, X);
#undef X
}
int batchmul_run_dispatch(int curve, const void* d_table, int window, const void* d_scalars, size_t n, int mont, void* d_tmp, void* d_out, hipStream_t st) {
#define X(NAME) batchmul_run_##NAME(d_table, window, d_scalars, n, mont, d_tmp, d_out, st)
  ARK_CURVE_SWITCH(curve, X);
#undef X
}
int fft_dispatch(int field, FftWorkspace& ws, void* d, int k, const uint64_t* root, const uint64_t* pre,
                 const uint64_t* post, const uint64_t* postc, int zlog, hipStream_t st, FftTimings* tm) {
#define X(NAME) fft_run_##NAME(ws, d, k, root, pre, post, postc, zlog, st, tm)
  ARK_FIELD_SWITCH(field, X);
#undef X
}
typedef int (*elementwise_fn)(int, const void*, const void*, void*, size_t, hipStream_t);
int add_affine_dispatch(int curve, const void* in, void* out, size_t n, const void* d_delta, hipStream_t st) {
#define X(NAME) sw_add_affine_##NAME(in, out, n, d_delta, st)
  ARK_CURVE_SWITCH(curve, X);
#undef X
}
int normalize_dispatch(int curve, const void* in, void* out, size_t n, hipStream_t st) {
#define X(NAME) sw_normalize_batch_##NAME(in, out, n, st)
  ARK_CURVE_SWITCH(curve, X);
#undef X
}
int fr_mul_dispatch(int field, const void* a, const void* b, void* r, size_t n, hipStream_t st) {
#define X(NAME) test_field_op_##NAME(2, a, b, r, n, st)
  ARK_FIELD_SWITCH(field, X);
#undef X
}
int fr_scale_dispatch(int field, const void* a, const uint64_t* k4, void* r, size_t n, hipStream_t st) {
#define X(NAME) fr_scale_##NAME(a, k4, r, n, st)
  ARK_FIELD_SWITCH(field, X);
#undef X
}
bool field_is_one(int field, const uint64_t* x);
int fft_roots_dispatch(int field, FftWorkspace& ws, int k, const uint64_t* root4, hipStream_t st, const uint32_t** out) {
#define X(NAME) fft_roots_##NAME(ws, k, root4, st, out)
  ARK_FIELD_SWITCH(field, X);
#undef X
}
int fft_scalars_dispatch(int field, FftWorkspace& ws, const uint64_t* base4, const uint64_t* mul4, size_t count, void* d_out,
                         hipStream_t st) {
#define X(NAME) fft_scalars_##NAME(ws, base4, mul4, count, d_out, st)
  ARK_FIELD_SWITCH(field, X);
#undef X
}
int gfft_run_dispatch(int curve, void* d_jac, int k, const uint32_t* roots, const uint32_t* pre, const uint32_t* post, void* work,
                      hipStream_t st) {
#define X(NAME) gfft_run_##NAME(d_jac, k, roots, pre, post, work, st)
  ARK_CURVE_SWITCH(curve, X);
#undef X
}
size_t gfft_work_bytes_any(int curve, int k) {
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
int fr_div_dispatch(int field, const void* num, const void* den, void* r, size_t n, hipStream_t st) {
#define X(NAME) fr_div_##NAME(num, den, r, n, st)
  ARK_FIELD_SWITCH(field, X);
#undef X
}
int fr_op_dispatch(int field, int op, const void* a, const void* b, void* r, size_t n, hipStream_t st) {
#define X(NAME) test_field_op_##NAME(op, a, b, r, n, st)
  ARK_FIELD_SWITCH(field, X);
#undef X
}
int fft_axis_dispatch(int field, FftWorkspace& ws, const void* src, void* dst, unsigned G, size_t cols, const uint64_t* root,
                      hipStream_t st) {
#define X(NAME) fft_axis_##NAME(ws, src, dst, G, cols, root, st)
  ARK_FIELD_SWITCH(field, X);
#undef X
}
int fft_axis_prepare_dispatch(int field, FftWorkspace& ws, unsigned G, const uint64_t* root, hipStream_t st, const uint32_t** pw) {
#define X(NAME) fft_axis_prepare_##NAME(ws, G, root, st, pw)
  ARK_FIELD_SWITCH(field, X);
#undef X
}
int fft_axis_launch_dispatch(int field, const void* src, void* dst, unsigned G, size_t stride, size_t cols, const uint32_t* pw,
                             hipStream_t st) {
#define X(NAME) fft_axis_launch_##NAME(src, dst, G, stride, cols, pw, st)
  ARK_FIELD_SWITCH(field, X);
#undef X
}
elementwise_fn field_op_fn(int field) {
  switch (field) {
#ifndef ARK_HIP_DEV
    case ARK_HIP_BN254_FR: return test_field_op_BN254_FR;
    case ARK_HIP_BLS12_377_FR: return test_field_op_BLS12_377_FR;
    case ARK_HIP_BN254_FQ: return test_basefield_op_BN254_G1;  // base fields: through the G1 curve over them
    case ARK_HIP_BLS12_377_FQ: return test_basefield_op_BLS12_377_G1;
#endif
    case ARK_HIP_BLS12_381_FR: return test_field_op_BLS12_381_FR;
    case ARK_HIP_BLS12_381_FQ: return test_basefield_op_BLS12_381_G1;
  }
  return nullptr;
}
elementwise_fn basefield_op_fn(int curve) {
  switch (curve) {
#ifndef ARK_HIP_DEV
    case 0: return test_basefield_op_BN254_G1;
    case 2: return test_basefield_op_BLS12_377_G1;
    case 3: return test_basefield_op_BLS12_377_G2;
    case 4: return test_basefield_op_BLS12_381_G2;
#endif
    case 1: return test_basefield_op_BLS12_381_G1;
  }
  return nullptr;
}
elementwise_fn point_op_fn(int curve) {
  switch (curve) {
#ifndef ARK_HIP_DEV
    case 0: return test_point_op_BN254_G1;
    case 2: return test_point_op_BLS12_377_G1;
    case 3: return test_point_op_BLS12_377_G2;
    case 4: return test_point_op_BLS12_381_G2;
#endif
    case 1: return test_point_op_BLS12_381_G1;
  }
  return nullptr;
}

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
static void host_field_ops(int op, const uint64_t* a, const uint64_t* b, uint64_t* r, size_t n) {
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

bool field_is_one(int field, const uint64_t* x) {
  switch (field) {
#ifndef ARK_HIP_DEV
    case ARK_HIP_BN254_FR: return host_is_one<BN254_FR>(x);
    case ARK_HIP_BLS12_377_FR: return host_is_one<BLS12_377_FR>(x);
#endif
    case ARK_HIP_BLS12_381_FR: return host_is_one<BLS12_381_FR>(x);
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

int fft_any(Context* c, int field, const ark_hip_radix2_domain* dom, void* d_data, int inverse, int zlog,
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
int degree_aware_zlog(const ark_hip_radix2_domain* dom, size_t num_coeffs) {
  if (num_coeffs == 0 || num_coeffs * 4 > dom->size) return 0;
  size_t d = 2;  // at least two input elements are read (the last stage is always executed)
  while (d < num_coeffs) d <<= 1;
  int z = 0;
  while ((d << z) < dom->size) z++;
  return z;
}

size_t field_bytes(int field) { return (field == ARK_HIP_BLS12_381_FQ || field == ARK_HIP_BLS12_377_FQ) ? 48 : 32; }

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
int msm_pick_lane(Context* c) {
  int busy[2];
  for (int l = 0; l < 2; l++) {
    std::lock_guard<std::mutex> lock(c->msm[l].mu);
    busy[l] = 0;
    for (const auto& j : c->msm[l].jobs) busy[l] += j.busy ? 1 : 0;
  }
  if (busy[0] + busy[1] >= MSM_JOBS) return ARK_HIP_ERR_BUSY;
  return busy[1] < busy[0] ? 1 : 0;
}
int msm_lane_stream(Context* c, int lane, hipStream_t* out) {
  if (lane && !c->stream_b) ARK_HIP_TRY(hipStreamCreateWithFlags(&c->stream_b, hipStreamNonBlocking));
  *out = lane ? c->stream_b : c->stream;
  return 0;
}
int sync_compute(Context* c) {  // both MSM lanes idle
  ARK_HIP_TRY(hipStreamSynchronize(c->stream));
  if (c->stream_b) ARK_HIP_TRY(hipStreamSynchronize(c->stream_b));
  return 0;
}
// called by every entry point that leaves work running on the context stream whose OUTPUT a caller may hand to an MSM
int mark_producer(Context* c) {
  if (!c->lane_ev) ARK_HIP_TRY(hipEventCreateWithFlags(&c->lane_ev, hipEventDisableTiming));
  ARK_HIP_TRY(hipEventRecord(c->lane_ev, c->stream));
  c->lane_ev_set = true;
  return 0;
}
// returns lane * MSM_JOBS + slot, or a negative error
int msm_enqueue_ctx(Context* c, int curve, const void* pts, size_t wstride, const MsmPlan* prep, const void* d_scalars,
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
// May be called WITHOUT the context lock held (ark_hip_msm_wait): the timings go through a local and are published
// under the (recursive) lock.
int msm_finish_ctx(Context* c, int curve, int slot, uint64_t* out) {
  if (slot < 0 || slot >= 2 * MSM_JOBS) return ARK_HIP_ERR_ARG;
  MsmTimings tm;
  const int rc = msm_finish_dispatch(curve, c->msm[slot / MSM_JOBS], slot % MSM_JOBS, out, &tm);
  if (rc == 0 && tm.c != 0) {
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    c->msm_tm = tm;
  }
  return rc;
}
// error paths: wait for a job and drop its result, so that its slot is free again and nothing is left in flight
void msm_discard_ctx(Context* c, int curve, int slot) {
  uint64_t scratch[36];
  (void)msm_finish_ctx(c, curve, slot, scratch);
}

// next slot of the upload ring: the copy stream waits until the MSM that last read this slot has finished
int ring_acquire(Context* c, int* k) {
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
int ring_publish(Context* c, int k, hipStream_t compute) {
  ARK_HIP_TRY(hipEventRecord(c->ring_up[k], c->copy_stream));
  ARK_HIP_TRY(hipStreamWaitEvent(compute, c->ring_up[k], 0));
  return 0;
}
int ring_release(Context* c, int k, hipStream_t compute) {
  ARK_HIP_TRY(hipEventRecord(c->ring_free[k], compute));
  return 0;
}

// ---- base-set cache of the host-pointer entry points ----------------------------------------------------
void free_prepared(PreparedBases* pb) {  // the caller has made sure no job in flight reads the table
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
const HashKey& hash_key() {
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
int hash_threads() {
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
Hash128 base_hash(const uint64_t* p, size_t words) {
  HashPass h;
  h.begin(p, words);
  return h.end(nullptr);
}

void cache_configure(Context* c) {
  if (c->cache_budget < 0) {
    // default: a quarter of the device memory (ARK_HIP_BASE_CACHE_MB=0 or ark_hip_msm_cache_config(0, ..) turn it off).
    // The cache never changes a result: every hit is validated against a hash of the slice's full content, which host
    // threads compute while the device works (2^24 BLS12-381 G1: 40.2 ms per call cached, 40.2 pinned, 47.8 streamed,
    // 36.4 resident -- profiles/r4_trait_modes_sessionA.txt)
    long long budget = -1;
    if (const char* e = getenv("ARK_HIP_BASE_CACHE_MB")) budget = atoll(e) * (1ll << 20);
    if (budget < 0) {
      size_t fr = 0, tot = 0;
      budget = hipMemGetInfo(&fr, &tot) == hipSuccess ? (long long)(tot / 4) : (8ll << 30);
    }
    c->cache_budget = budget;
  }
  if (c->auto_prepare < 0) {
    const char* e = getenv("ARK_HIP_AUTO_PREPARE");
    c->auto_prepare = e ? atoi(e) : 0;
    if (c->auto_prepare < 0) c->auto_prepare = 0;
  }
}
long long cache_entry_bytes(const BaseCacheEntry& e) {   // device bytes an entry holds: the copy + its prepared table
  return (long long)e.dev.cap + (e.prepared ? (long long)e.prepared->table.cap : 0);
}
void cache_drop(Context* c, size_t idx) {
  BaseCacheEntry& e = c->base_cache[idx];
  if (e.prepared) free_prepared(e.prepared);
  e.dev.release();
  c->base_cache.erase(c->base_cache.begin() + (long)idx);
}
// drops the transparent entries (pinned sets stay until their unpin)
int cache_clear(Context* c) {
  bool any = false;
  for (auto& e : c->base_cache) any |= e.pins == 0;
  if (!any) return 0;
  if (int rc = sync_compute(c)) return rc;  // a job in flight may still read a cached copy
  for (size_t i = c->base_cache.size(); i-- > 0;)
    if (c->base_cache[i].pins == 0) cache_drop(c, i);
  return 0;
}
// a pinned set that CONTAINS [bases, bases + n points): index, and the offset in points; -1 if none
long cache_find_pinned(Context* c, int curve, const uint64_t* bases, size_t n, size_t* off_points) {
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
long cache_find_exact(Context* c, int curve, const void* bases, size_t n, bool pinned) {
  for (size_t i = 0; i < c->base_cache.size(); i++) {
    const BaseCacheEntry& e = c->base_cache[i];
    if (e.curve == curve && e.host == bases && e.n == n && (e.pins > 0) == pinned) return (long)i;
  }
  return -1;
}
// room for `bytes` more under the transparent budget: least recently used transparent entries go first
// returns false if the set cannot be cached
bool cache_make_room(Context* c, long long bytes, long keep = -1) {
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
void cache_forget(Context* c, int curve, const void* host, size_t n, bool pinned) {
  const long i = cache_find_exact(c, curve, host, n, pinned);
  if (i < 0) return;
  (void)sync_compute(c);
  (void)hipStreamSynchronize(c->copy_stream);
  cache_drop(c, (size_t)i);
}
// after `auto_prepare` hits on the WHOLE set: build its per-window table (a failed build leaves the plain path in place)
void cache_maybe_prepare(Context* c, BaseCacheEntry& e) {
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

// One MSM whose scalars (and, with `host_bases`, bases) come from host memory, against `d_bases` (resident; with
// `host_bases` as well: a resident copy being FILLED by this very call, piece by piece) or bases streamed through the ring: the pairs are cut into pieces; piece k+1 uploads on the copy stream -- and digit-recodes / sorts on the other MSM
// lane -- while piece k's accumulate kernel runs.
//   shared (2..8 pieces): the pieces are pieces of ONE MSM -- one plan, one bucket array, one reduction (MsmPiece);
//   otherwise (msm_chunks with its fixed 2^20 steps): independent MSMs whose results are added on the host (the
//   reference's own chunk sum, variable_base/mod.rs:542-557).
int msm_stream(Context* c, int curve, const void* d_bases, const uint64_t* host_bases, const uint64_t* scalars, size_t n,
               int mont, size_t step, uint64_t* out_xyz, bool allow_shared = true, bool growing = false,
               bool taper = false) {
  const size_t ab = (size_t)CURVES[curve].fe_words * 16;
  const size_t pw = (size_t)CURVES[curve].fe_words * 3;
  if (step == 0 || step > n) step = n;
  // piece sizes: equal steps, or -- `growing`, scalars-only uploads against resident bases -- each piece twice the one
  // before it: a piece's upload hides under the previous piece's kernels as long as it is less than ~3x as large (the MSM
  // spends 2.1 ns per pair, the PCIe copy 0.64 ns per 32-byte scalar), so the one upload nothing hides shrinks to
  // n / (2^P - 1) pairs and the per-piece costs (launches, the read-modify-write of every bucket) are paid P <= 6 times
  std::vector<size_t> sizes;
  if (growing && n >= ((size_t)3 << 18)) {
    int P = 1;
    while (P < 6 && (n / (((size_t)1 << (P + 1)) - 1)) >= ((size_t)1 << 18)) P++;
    size_t first = (n / (((size_t)1 << P) - 1)) & ~(size_t)255;
    size_t left = n, cur = first;
    for (int k = 0; k < P; k++) {
      const size_t take = k + 1 == P ? left : cur;
      sizes.push_back(take);
      left -= take;
      cur *= 2;
    }
    step = sizes.back();   // the largest piece sizes the ring buffers
  } else if (taper && host_bases && n >= ((size_t)1 << 21) && !getenv("ARK_HIP_STREAM_PIECES") &&
             !(getenv("ARK_HIP_STREAM_TAPER") && getenv("ARK_HIP_STREAM_TAPER")[0] == '0')) {
    // bases AND scalars cross PCIe (2^24 BLS12-381 G1: 2 GiB, ~40 ms of copy against 36 ms of kernels).  Equal eighths are
    // the best schedule measured (47.6 ms): every upload waits for the kernels two pieces back (two ring slots), so large
    // middle pieces stall the copy (2,4,10,8,4,2,1,1 / 32: 54.4 ms), and a halving tail (.., 4, 2, 1, 1 / 64: 48.7 ms) or
    // sixteenths (49.4 ms) pay more per piece than the shorter last piece saves (profiles/r4_trait_modes_and_schedules_sessionB.txt).
    // ARK_HIP_STREAM_SCHEDULE="a,b,c,.." (weights) overrides for experiments.
    std::vector<size_t> wts;
    if (const char* e = getenv("ARK_HIP_STREAM_SCHEDULE")) {
      for (const char* q = e; *q;) {
        char* end = nullptr;
        const size_t v = (size_t)strtoul(q, &end, 10);
        if (end == q) {   // not a number: strtoul consumed nothing (it used to loop here for ever) -- ignore the variable
          wts.clear();
          break;
        }
        if (v) wts.push_back(v);
        q = end;
        if (*q == ',') q++;
      }
    }
    if (wts.empty()) wts.assign(8, 1);
    size_t wsum = 0;
    for (size_t w : wts) wsum += w;
    size_t left = n;
    step = 0;
    for (size_t k = 0; k < wts.size() && left; k++) {
      size_t take = k + 1 == wts.size() ? left : ((n / wsum) * wts[k]) & ~(size_t)255;
      if (take == 0 || take > left) take = left;
      sizes.push_back(take);
      left -= take;
      step = take > step ? take : step;
    }
    if (left) {
      sizes.back() += left;
      step = sizes.back() > step ? sizes.back() : step;
    }
  } else {
    for (size_t off = 0; off < n; off += step) sizes.push_back(n - off < step ? n - off : step);
  }
  const size_t npieces = sizes.size();
  const bool shared = allow_shared && npieces >= 2 && npieces <= 16;
  MsmPlan plan{};
  if (shared) {
    // the window size from the width classes of a spread sample (~1024) of the host scalars (msm.cuh K0: what the device entry
    // measures exactly); the layout stays the full-width one -- a sample cannot bound the widest scalar
    MsmWidths widths{};
    bool skewed = false;
    static const bool probe_on = [] {
      const char* e = getenv("ARK_HIP_MSM_PROBE");
      return !(e && atoi(e) == 0);
    }();
    if (probe_on && scalars && n >= ((size_t)1 << 19) && msm_sample_widths_dispatch(curve, scalars, n, mont, &widths) == 0)
      skewed = msm_widths_skewed(widths);
    plan = msm_make_plan(n, msm_scalar_bits(curve), msm_mul_cost(curve), false, msm_lazy28(curve), skewed ? &widths : nullptr);
    if ((size_t)step * (size_t)plan.W >= (1ull << 32)) return ARK_HIP_ERR_SIZE;
    const size_t need = plan.nbuckets() * (size_t)CURVES[curve].fe_words * 32;  // XYZZ: four field elements
    if (c->piece_buckets.cap < need) {
      if (int rc = sync_compute(c)) return rc;
      if (c->piece_buckets.ensure(need)) return ARK_HIP_ERR_NOMEM;
    }
    for (int j = 0; j < 2; j++)
      if (!c->piece_ev[j]) ARK_HIP_TRY(hipEventCreateWithFlags(&c->piece_ev[j], hipEventDisableTiming));
  }
  std::vector<uint64_t> partials;
  int pending[2] = {-1, -1};
  bool pending_last[2] = {false, false};
  int npend = 0;
  auto drain_one = [&]() -> int {
    uint64_t part[36];
    const bool has_result = !shared || pending_last[0];
    const int rc = msm_finish_ctx(c, curve, pending[0], part);
    pending[0] = pending[1];
    pending_last[0] = pending_last[1];
    npend--;
    if (rc) return rc;
    if (has_result) partials.insert(partials.end(), part, part + pw);
    return 0;
  };
  auto fail = [&](int rc) -> int {  // nothing of this call stays in flight, no job slot stays taken
    while (npend) {
      msm_discard_ctx(c, curve, pending[0]);
      pending[0] = pending[1];
      npend--;
    }
    (void)hipStreamSynchronize(c->copy_stream);
    return rc;
  };
  if (n == 0) {
    int slot = msm_enqueue_ctx(c, curve, nullptr, 0, nullptr, nullptr, 0, mont);
    if (slot < 0) return slot;
    return msm_finish_ctx(c, curve, slot, out_xyz);
  }
  size_t off = 0;
  for (size_t piece_no = 0; piece_no < npieces; off += sizes[piece_no], piece_no++) {
    const size_t cnt = sizes[piece_no];
    if (npend == 2) {
      if (int rc = drain_one()) return fail(rc);
    }
    const int lane = msm_pick_lane(c);  // before anything is put in flight: BUSY must leave nothing behind
    if (lane < 0) return fail(lane);
    hipStream_t compute;
    if (int rc = msm_lane_stream(c, lane, &compute)) return fail(rc);
    int k = 0;
    if (int rc = ring_acquire(c, &k)) return fail(rc);
    const bool ring_bases = host_bases && !d_bases;  // host_bases with d_bases: fill the resident copy piece by piece
    if (c->ring_s[k].cap < cnt * 32 || (ring_bases && c->ring_b[k].cap < cnt * ab)) {
      if (int rc = sync_compute(c)) return fail(rc);  // growing frees memory an enqueued MSM may still read
      if (c->ring_s[k].ensure(step * 32) || (ring_bases && c->ring_b[k].ensure(step * ab))) return fail(ARK_HIP_ERR_NOMEM);
    }
    const void* pts = ring_bases ? (const void*)c->ring_b[k].p : (const void*)((const char*)d_bases + off * ab);
    if (host_bases)
      if (int rc = c->stager.upload((void*)pts, host_bases + off * (ab / 8), cnt * ab, c->copy_stream)) return fail(rc);
    if (int rc = c->stager.upload(c->ring_s[k].p, scalars + off * 4, cnt * 32, c->copy_stream)) return fail(rc);
    if (int rc = ring_publish(c, k, compute)) return fail(rc);
    MsmPiece piece{&plan, c->piece_buckets.p, piece_no == 0, piece_no + 1 == npieces,
                   piece_no == 0 ? nullptr : c->piece_ev[(piece_no - 1) & 1], c->piece_ev[piece_no & 1]};
    int slot = msm_enqueue_ctx(c, curve, pts, 0, nullptr, c->ring_s[k].p, cnt, mont, lane, 0, 0, shared ? &piece : nullptr);
    (void)ring_release(c, k, compute);
    if (slot < 0) return fail(slot);
    pending_last[npend] = piece_no + 1 == npieces;
    pending[npend++] = slot;
  }
  while (npend) {
    if (int rc = drain_one()) return fail(rc);
  }
  if (partials.size() == pw) {
    memcpy(out_xyz, partials.data(), pw * 8);
    return 0;
  }
  return ark_hip_sw_sum(curve, partials.data(), partials.size() / pw, out_xyz);
}
// pieces of one streamed MSM: the first piece's upload is the only one nothing hides, so many pieces -- as long as a
// piece keeps ~2^18 pairs (its ~20 launches and the read-modify-write of every bucket are per piece).  Measured on
// MI355X, BLS12-381 G1, repeat call with resident bases (profiles/r3_trait_surface.txt): 2^24: 1 / 2 / 4 / 8 pieces
// 52.6 / 47.1 / 45.3 / 44.0 ms (resident inputs: 41.3); 2^22: 17.0 / 15.2 / 14.0 / 13.5 (13.3); 2^20: 5.9 / 5.4 / 5.3 (4.8).
// Repeat calls against resident bases use GROWING pieces instead (msm_stream, `growing`): 2^24 43.95 -> 42.59 ms, 2^20
// 4.50 -> 4.01 ms (resident 3.97), 2^26 155.1 -> 147.9 ms (profiles/r3_trait_growing_pieces.txt); equal pieces remain for
// calls that upload their bases too (there the copy, not the kernels, sets the pace).
size_t msm_stream_step(size_t n) {
  size_t pieces = n >> 18;
  if (pieces < 1) pieces = 1;
  if (pieces > 8) pieces = 8;
  if (const char* e = getenv("ARK_HIP_STREAM_PIECES")) {
    const int v = atoi(e);
    if (v >= 1 && v <= 64) pieces = (size_t)v;
  }
  return (n + pieces - 1) / pieces;
}


// ---- sharded MSM: the ranks' PART SUMS are exchanged and added on the device, one host tail for the whole job ----------------
// (msm.cuh, "one process per GPU: the part sums of all ranks").  A rank's block = 64-byte header + its part sums, padded to
// a size that depends on the curve only, so that every rank posts the same byte count whatever its plan.  Ranks whose plans
// differ (unequal shard sizes), an empty shard or more than SUMS_MAX_PARTS parts are seen by EVERY rank in the gathered
// headers: all of them then take the fallback together -- finished partial results through msm_sharded_combine.
constexpr uint32_t SUMS_MAX_PARTS = 1024;
size_t sums_block_bytes(int curve) { return sizeof(MsmSumsHeader) + (size_t)SUMS_MAX_PARTS * CURVES[curve].fe_words * 32; }
int sums_buffers(Context* c, int curve, int world) {
  const size_t bb = sums_block_bytes(curve);
  if (c->comm_sums.cap < bb * (size_t)(world + 2)) {
    if (int rc = sync_compute(c)) return rc;
    if (c->comm_sums.ensure(bb * (size_t)(world + 2))) return ARK_HIP_ERR_NOMEM;
  }
  // pinned: [MAX_DEV staging headers | MAX_DEV gathered headers | summed parts]
  if (!c->comm_sums_pinned) ARK_HIP_TRY(hipHostMalloc(&c->comm_sums_pinned, sizeof(MsmSumsHeader) * 2 * MAX_DEV + (size_t)SUMS_MAX_PARTS * 12 * 32));
  return 0;
}
// the enqueued job `slot_full`'s block -> d_block (device), stream-ordered behind the job on lane 0's stream
int sums_write_block(Context* c, int curve, int slot_full, char* d_block, MsmSumsHeader* h_out, int stage_slot = 0) {
  MsmWorkspace& ws = c->msm[slot_full / MSM_JOBS];
  MsmSumsInfo info;
  const int rc = msm_job_sums(ws, slot_full % MSM_JOBS, &info);
  if (rc < 0) return ARK_HIP_ERR_ARG;
  const size_t pb = (size_t)CURVES[curve].fe_words * 32;
  MsmSumsHeader* hp = (MsmSumsHeader*)c->comm_sums_pinned + stage_slot;   // staging of this block's header (one slot per block
                                                                          // written in one call: the copies are asynchronous)
  const bool ran = rc == 0;   // a non-empty job: its kernels ran, ws.hctr holds ITS flags (an empty job touches none)
  if (rc == 1 || info.h.npairs == 0 || info.h.npairs > SUMS_MAX_PARTS) {
    memset(&info.h, 0, sizeof(info.h));   // no usable sums: every rank sees npairs == 0 and falls back
    info.d_sums = nullptr;
  }
  *hp = info.h;
  *h_out = info.h;
  ARK_HIP_TRY(hipMemcpyAsync(d_block, hp, sizeof(MsmSumsHeader), hipMemcpyHostToDevice, c->stream));
  if (info.d_sums)
    ARK_HIP_TRY(hipMemcpyAsync(d_block + sizeof(MsmSumsHeader), info.d_sums, (size_t)info.h.npairs * pb, hipMemcpyDeviceToDevice, c->stream));
  // the scalar-range flag travels whether or not sums do (a zeroed header -- empty shard, too many parts -- used to drop
  // it: the peers then took the fallback while this rank alone returned the error from its own finish)
  if (ran && ws.hctr.p)
    ARK_HIP_TRY(hipMemcpyAsync(d_block + offsetof(MsmSumsHeader, err), (const u32*)ws.hctr.p + 3, 4, hipMemcpyDeviceToDevice, c->stream));
  return 0;
}
// A rank whose local part failed BEFORE the collective (bad argument, no memory, busy lanes) must still take part in it, or
// its peers wait for ever: it posts a header that says so -- err = SUMS_ERR_LOCAL + (-code) -- and every rank returns that code.
constexpr uint32_t SUMS_ERR_LOCAL = 0x100;
int sums_write_failure(Context* c, char* d_block, int code, MsmSumsHeader* h_out) {
  MsmSumsHeader* hp = (MsmSumsHeader*)c->comm_sums_pinned;
  memset(hp, 0, sizeof(*hp));
  hp->err = SUMS_ERR_LOCAL + (uint32_t)(-code);
  *h_out = *hp;
  ARK_HIP_TRY(hipMemcpyAsync(d_block, hp, sizeof(MsmSumsHeader), hipMemcpyHostToDevice, c->stream));
  return 0;
}
// all `world` blocks sit in d_blocks: add them, bring the sums and the headers to the host, decide.  *agree = the ranks
// share one plan and out_xyz holds the whole job's result; otherwise the caller takes the fallback.  A scalar-range error on
// ANY rank is every rank's error.
int sums_reduce(Context* c, int curve, const MsmSumsHeader& mine, const char* d_blocks, int world, uint64_t* out_xyz, bool* agree) {
  const size_t bb = sums_block_bytes(curve), pb = (size_t)CURVES[curve].fe_words * 32;
  char* d_out = (char*)c->comm_sums.p + bb * (size_t)(world + 1);
  MsmSumsHeader* hh = (MsmSumsHeader*)c->comm_sums_pinned + MAX_DEV;   // the gathered headers
  char* h_sums = (char*)((MsmSumsHeader*)c->comm_sums_pinned + 2 * MAX_DEV);
  if (mine.npairs)
    if (int rc = msm_sum_ranks_dispatch(curve, d_blocks, world, bb, mine.npairs, d_out, c->stream)) return rc;
  ARK_HIP_TRY(hipMemcpy2DAsync(hh, sizeof(MsmSumsHeader), d_blocks, bb, sizeof(MsmSumsHeader), (size_t)world, hipMemcpyDeviceToHost, c->stream));
  if (mine.npairs) ARK_HIP_TRY(hipMemcpyAsync(h_sums, d_out, (size_t)mine.npairs * pb, hipMemcpyDeviceToHost, c->stream));
  ARK_HIP_TRY(hipStreamSynchronize(c->stream));
  bool same = mine.npairs != 0, err = false;
  for (int r = 0; r < world; r++)   // a rank that failed before the collective: its code is every rank's (lowest rank's first)
    if (hh[r].err >= SUMS_ERR_LOCAL) {
      *agree = false;
      return -(int)(hh[r].err - SUMS_ERR_LOCAL);
    }
  for (int r = 0; r < world; r++) {
    const MsmSumsHeader& o = hh[r];
    err |= o.err != 0;
    same &= o.c == mine.c && o.W == mine.W && o.narrow == mine.narrow && o.shared == mine.shared && o.nbits == mine.nbits &&
            o.log2L0 == mine.log2L0 && o.Q == mine.Q && o.npairs == mine.npairs;
  }
  *agree = same;
  if (err) return ARK_HIP_ERR_SCALAR_RANGE;
  if (!same) return 0;
  return msm_fold_sums_dispatch(curve, mine, h_sums, out_xyz);
}

// ---- RCCL, opened at run time ------------------------------------------------------------------------------
// librccl.so.1 by SONAME: a process that already carries a copy (PyTorch ships its own) gets that one, so two RCCL
// instances never meet in one process; otherwise the ROCm installation's.  ARK_HIP_RCCL_LIB overrides.
struct RcclApi {
  void* handle = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  bool tried = false;
};
RcclApi g_rccl;
std::mutex g_rccl_mu;
const RcclApi* rccl_api() {
  std::lock_guard<std::mutex> lk(g_rccl_mu);
  if (g_rccl.tried) return g_rccl.handle ? &g_rccl : nullptr;
  g_rccl.tried = true;
  const char* names[4] = {getenv("ARK_HIP_RCCL_LIB"), "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
  void* h = nullptr;
  for (const char* nm : names) {
    if (!nm || !*nm) continue;
    h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
    if (h) break;
  }
  if (!h) {
    fprintf(stderr, "ark_hip: RCCL not found (librccl.so.1): %s\n", dlerror());
    return nullptr;
  }
  RcclApi a;
  a.handle = h;
#define ARK_RCCL_SYM(field, name)                                  \
  a.field = (decltype(a.field))dlsym(h, name);                     \
  if (!a.field) {                                                  \
    fprintf(stderr, "ark_hip: %s missing from RCCL\n", name);      \
    return nullptr;                                                \
  }
  ARK_RCCL_SYM(GetUniqueId, "ncclGetUniqueId")
  ARK_RCCL_SYM(CommInitRank, "ncclCommInitRank")
  ARK_RCCL_SYM(CommDestroy, "ncclCommDestroy")
  ARK_RCCL_SYM(AllGather, "ncclAllGather")
  ARK_RCCL_SYM(Send, "ncclSend")
  ARK_RCCL_SYM(Recv, "ncclRecv")
  ARK_RCCL_SYM(GroupStart, "ncclGroupStart")
  ARK_RCCL_SYM(GroupEnd, "ncclGroupEnd")
  ARK_RCCL_SYM(GetErrorString, "ncclGetErrorString")
#undef ARK_RCCL_SYM
  a.tried = true;
  g_rccl = a;
  return &g_rccl;
}
#define ARK_RCCL_TRY(api, expr)                                                                         \
  do {                                                                                                  \
    ncclResult_t _r = (expr);                                                                           \
    if (_r != ncclSuccess) {                                                                            \
      fprintf(stderr, "ark_hip: %s failed: %s (%s:%d)\n", #expr, (api)->GetErrorString(_r), __FILE__, __LINE__); \
      return ARK_HIP_ERR_COMM;                                                                          \
    }                                                                                                   \
  } while (0)

// all-gather of one Projective per rank, summed in rank order: the same group element on every rank
int msm_sharded_combine(Context* c, int curve, const uint64_t* part, uint64_t* out_xyz) {
  const size_t pw = (size_t)CURVES[curve].fe_words * 3;
  if (!c->comm || c->comm_world == 1) {
    memcpy(out_xyz, part, pw * 8);
    return 0;
  }
  const RcclApi* api = rccl_api();
  if (!api) return ARK_HIP_ERR_COMM;
  const size_t G = (size_t)c->comm_world;
  if (c->comm_small.ensure((G + 1) * 36 * 8)) return ARK_HIP_ERR_NOMEM;
  if (!c->comm_pinned) ARK_HIP_TRY(hipHostMalloc(&c->comm_pinned, (size_t)(64 + 1) * 36 * 8));
  if (G > 64) return ARK_HIP_ERR_ARG;
  uint64_t* hp = (uint64_t*)c->comm_pinned;
  uint64_t* dsend = (uint64_t*)c->comm_small.p;
  uint64_t* dall = dsend + 36;
  memcpy(hp, part, pw * 8);
  ARK_HIP_TRY(hipMemcpyAsync(dsend, hp, pw * 8, hipMemcpyHostToDevice, c->stream));
  ARK_RCCL_TRY(api, api->AllGather(dsend, dall, pw, ncclUint64, c->comm, c->stream));
  ARK_HIP_TRY(hipMemcpyAsync(hp + 36, dall, G * pw * 8, hipMemcpyDeviceToHost, c->stream));
  ARK_HIP_TRY(hipStreamSynchronize(c->stream));
  return ark_hip_sw_sum(curve, hp + 36, G, out_xyz);
}

// ---- sharded FFT: constants of one rank's local transform and of the cross transform -------------------------
struct ShardConsts {
  int km = 0;                 // log2 of the local size m
  unsigned G = 1;
  size_t m = 0, sub = 0;
  uint64_t root_m[4], root_G[4], pre[4], post[4], postc[4];
  bool has_pre = false, has_post = false, has_postc = false;
};
template <class FP>
int shard_consts(const ark_hip_radix2_domain* dom, int rank, int world, int inverse, ShardConsts* o) {
  typedef Fp<FP> F;
  const int k = (int)dom->log_size_of_group;
  if (dom->size != ((uint64_t)1 << k) || k > FP::TWO_ADICITY) return ARK_HIP_ERR_ARG;
  if (world < 1 || world > 16 || (world & (world - 1)) || rank < 0 || rank >= world) return ARK_HIP_ERR_ARG;
  int lg = 0;
  while ((1 << lg) < world) lg++;
  if (2 * lg > k) return ARK_HIP_ERR_SIZE;   // G^2 must divide the size
  o->G = (unsigned)world;
  o->km = k - lg;
  o->m = (size_t)1 << o->km;
  o->sub = o->m >> lg;
  const bool coset = !host_is_one<FP>(dom->offset);
  const F w = F::load(inverse ? dom->group_gen_inv : dom->group_gen);
  const uint64_t eG[1] = {(uint64_t)world}, em[1] = {(uint64_t)o->m}, er[1] = {(uint64_t)rank};
  host_pow<FP>(w, eG, 1).store(o->root_m);    // generator of the size-m subgroup (or its inverse)
  host_pow<FP>(w, em, 1).store(o->root_G);    // primitive G-th root (or its inverse)
  const F tw = host_pow<FP>(w, er, 1);        // the rank's twiddle base: w_n^(+-rank)
  if (!inverse) {
    // local transform over i2 of x[rank + G i2] * g^(rank + G i2), then out[j2] *= w_n^(rank j2)
    if (coset) {
      const F g = F::load(dom->offset);
      host_pow<FP>(g, eG, 1).store(o->pre);
      o->has_pre = true;
      host_pow<FP>(g, er, 1).store(o->postc);
      o->has_postc = rank != 0;
    }
    tw.store(o->post);
    o->has_post = rank != 0;
    if (o->has_postc && !o->has_post) {  // (unreachable: both hinge on rank != 0) keep the pair consistent for fft_run_device
      F::one().store(o->post);
      o->has_post = true;
    }
  } else {
    // in[j2] *= w_n^(-rank j2), inverse transform over j2, then out[i2] *= n^-1 * g^-(rank + G i2)
    tw.store(o->pre);
    o->has_pre = rank != 0;
    F sc = F::load(dom->size_inv);
    if (coset) {
      const F gi = F::load(dom->offset_inv);
      host_pow<FP>(gi, eG, 1).store(o->post);
      o->has_post = true;
      sc = F::mul(sc, host_pow<FP>(gi, er, 1));
    }
    sc.store(o->postc);
    o->has_postc = true;
  }
  return 0;
}
int shard_consts_any(int field, const ark_hip_radix2_domain* dom, int rank, int world, int inverse, ShardConsts* o) {
  switch (field) {
#ifndef ARK_HIP_DEV
    case ARK_HIP_BN254_FR: return shard_consts<BN254_FR>(dom, rank, world, inverse, o);
    case ARK_HIP_BLS12_377_FR: return shard_consts<BLS12_377_FR>(dom, rank, world, inverse, o);
#endif
    case ARK_HIP_BLS12_381_FR: return shard_consts<BLS12_381_FR>(dom, rank, world, inverse, o);
  }
  return ARK_HIP_ERR_ARG;
}
// the size-m transform of one rank, twiddle and scalings fused into its first / last pass
int shard_local(Context* c, int field, const ShardConsts& sc, void* d_local, hipStream_t st) {
  return fft_dispatch(field, c->fft, d_local, sc.km, sc.root_m, sc.has_pre ? sc.pre : nullptr, sc.has_post ? sc.post : nullptr,
                      sc.has_postc ? sc.postc : nullptr, 0, st, nullptr);
}
int comm_streams(Context* c) {
  if (!c->comm_stream) ARK_HIP_TRY(hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
  for (int a = 0; a < 2; a++)
    for (int i = 0; i < COMM_MAX_SLICES; i++)
      if (!c->comm_ev[a][i]) ARK_HIP_TRY(hipEventCreateWithFlags(&c->comm_ev[a][i], hipEventDisableTiming));
  return 0;
}
// one slice of the all-to-all: columns [c0, c0 + cs) of block q go to rank q / come from rank q
int exchange_slice(Context* c, const RcclApi* api, const char* send, char* recv, size_t sub, size_t c0, size_t cs, hipStream_t st) {
  const int G = c->comm_world, me = c->comm_rank;
  ARK_RCCL_TRY(api, api->GroupStart());
  for (int q = 0; q < G; q++) {
    const size_t off = ((size_t)q * sub + c0) * 32;
    if (q == me) continue;
    ARK_RCCL_TRY(api, api->Send(send + off, cs * 32, ncclUint8, q, c->comm, st));
    ARK_RCCL_TRY(api, api->Recv(recv + off, cs * 32, ncclUint8, q, c->comm, st));
  }
  ARK_RCCL_TRY(api, api->GroupEnd());
  const size_t off = ((size_t)me * sub + c0) * 32;   // own block: a device copy
  ARK_HIP_TRY(hipMemcpyAsync(recv + off, send + off, cs * 32, hipMemcpyDeviceToDevice, st));
  return 0;
}
int comm_slices(size_t sub) {
  int s = sub >= ((size_t)1 << 15) ? 4 : 1;   // slices of >= 256 KiB per peer
  if (const char* e = getenv("ARK_HIP_COMM_SLICES")) {
    const int v = atoi(e);
    if (v >= 1 && v <= COMM_MAX_SLICES) s = v;
  }
  while (s > 1 && (sub % (size_t)s)) s >>= 1;
  return s;
}

}  // namespace

extern "C" {

int ark_hip_device_count(void) { return device_count_raw(); }

int ark_hip_init(int device) {
  Context* c = nullptr;
  if (device < 0) return ARK_HIP_ERR_ARG;
  int rc = get_ctx(device, &c);
  if (rc) return rc;
  t_dev = device;
  return 0;
}
int ark_hip_set_device(int device) { return ark_hip_init(device); }
int ark_hip_get_device(void) { return t_dev >= 0 ? t_dev : (g_default >= 0 ? g_default : 0); }

void ark_hip_shutdown(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  for (int i = 0; i < MAX_DEV; i++) {
    Context* c = g_ctxs[i];
    if (!c) continue;
    {
      std::lock_guard<std::recursive_mutex> cl(c->mu);  // waits for calls in flight on this device
      (void)hipSetDevice(c->physical);
      (void)hipStreamSynchronize(c->stream);
      if (c->stream_b) (void)hipStreamSynchronize(c->stream_b);
      (void)hipStreamSynchronize(c->copy_stream);
      for (int j = 0; j < 2; j++)
        if (c->fft_side[j]) (void)hipStreamSynchronize(c->fft_side[j]);
      while (!c->base_cache.empty()) cache_drop(c, c->base_cache.size() - 1);
      c->stager.release();
      c->piece_buckets.release();
      for (int j = 0; j < 2; j++)
        if (c->piece_ev[j]) (void)hipEventDestroy(c->piece_ev[j]);
      if (c->lane_ev) (void)hipEventDestroy(c->lane_ev);
      c->msm[0].release();
      c->msm[1].release();
      c->fft.release();
      c->stage_a.release();
      c->stage_b.release();
      c->stage_c.release();
      for (int j = 0; j < 2; j++) {
        c->ring_s[j].release();
        c->ring_b[j].release();
        if (c->ring_free[j]) (void)hipEventDestroy(c->ring_free[j]);
        if (c->ring_up[j]) (void)hipEventDestroy(c->ring_up[j]);
      }
      (void)hipStreamDestroy(c->stream);
      if (c->stream_b) (void)hipStreamDestroy(c->stream_b);
      (void)hipStreamDestroy(c->copy_stream);
      for (int j = 0; j < 2; j++)
        if (c->fft_side[j]) (void)hipStreamDestroy(c->fft_side[j]);
      for (int j = 0; j < 3; j++)
        if (c->fft_ev[j]) (void)hipEventDestroy(c->fft_ev[j]);
    }
    delete c;
    g_ctxs[i] = nullptr;
  }
  g_default = -1;
}

int ark_hip_synchronize(void) {
  ARK_SCOPE(sc);
  ARK_HIP_TRY(hipStreamSynchronize(sc.c->copy_stream));
  if (int rc = sync_compute(sc.c)) return rc;
  return 0;
}

const char* ark_hip_version(void) { return "ark_hip 0.4 (gfx950)"; }
int ark_hip_host_threads(int out[2]) {
  if (!out) return ARK_HIP_ERR_ARG;
  out[0] = HostPool::instance().helpers();
  out[1] = HostPool::instance().threads_created();
  return 0;
}

int ark_hip_curve_info(int curve, int* fe_words, int* scalar_field, int* base_field, int* ext_degree) {
  if (curve < 0 || curve > 4) return ARK_HIP_ERR_ARG;
  if (fe_words) *fe_words = CURVES[curve].fe_words;
  if (scalar_field) *scalar_field = CURVES[curve].scalar_field;
  if (base_field) *base_field = CURVES[curve].base_field;
  if (ext_degree) *ext_degree = CURVES[curve].ext;
  return 0;
}

// ---- device / pinned memory for hosts without their own HIP binding ----
int ark_hip_malloc(size_t bytes, void** out_dptr) {
  if (!out_dptr) return ARK_HIP_ERR_ARG;
  ARK_SCOPE(sc);
  *out_dptr = nullptr;
  if (bytes == 0) return 0;
  if (hipMalloc(out_dptr, bytes) != hipSuccess) return ARK_HIP_ERR_NOMEM;
  return 0;
}
int ark_hip_free(void* dptr) {
  if (!dptr) return 0;
  ARK_SCOPE(sc);
  if (int rc = sync_compute(sc.c)) return rc;
  ARK_HIP_TRY(hipFree(dptr));
  return 0;
}
int ark_hip_memcpy_h2d(void* dst_dptr, const void* src_host, size_t bytes) {
  if (bytes && (!dst_dptr || !src_host)) return ARK_HIP_ERR_ARG;
  ARK_SCOPE(sc);
  ARK_HIP_TRY(hipMemcpyAsync(dst_dptr, src_host, bytes, hipMemcpyHostToDevice, sc.c->stream));
  ARK_HIP_TRY(hipStreamSynchronize(sc.c->stream));
  return 0;
}
int ark_hip_memcpy_d2h(void* dst_host, const void* src_dptr, size_t bytes) {
  if (bytes && (!dst_host || !src_dptr)) return ARK_HIP_ERR_ARG;
  ARK_SCOPE(sc);
  ARK_HIP_TRY(hipMemcpyAsync(dst_host, src_dptr, bytes, hipMemcpyDeviceToHost, sc.c->stream));
  ARK_HIP_TRY(hipStreamSynchronize(sc.c->stream));
  return 0;
}
int ark_hip_host_alloc(size_t bytes, void** out_ptr) {
  if (!out_ptr) return ARK_HIP_ERR_ARG;
  ARK_SCOPE(sc);
  *out_ptr = nullptr;
  if (bytes == 0) return 0;
  if (hipHostMalloc(out_ptr, bytes) != hipSuccess) return ARK_HIP_ERR_NOMEM;
  return 0;
}
int ark_hip_host_free(void* ptr) {
  if (!ptr) return 0;
  ARK_SCOPE(sc);
  ARK_HIP_TRY(hipHostFree(ptr));
  return 0;
}

int ark_hip_curve_generator(int curve, uint64_t* out_xy) {
  if (!out_xy) return ARK_HIP_ERR_ARG;
  const uint64_t* g = nullptr;
  switch (curve) {
    case 0: g = GEN_BN254_G1; break;
    case 1: g = GEN_BLS12_381_G1; break;
    case 2: g = GEN_BLS12_377_G1; break;
    case 3: g = GEN_BLS12_377_G2; break;
    case 4: g = GEN_BLS12_381_G2; break;
    default: return ARK_HIP_ERR_ARG;
  }
  memcpy(out_xy, g, (size_t)CURVES[curve].fe_words * 16);
  return 0;
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
  int rc;
  {
    Scope sc;
    rc = sc.enter(h->logical);
    if (rc == 0) {
      // the event wait and the host tail run WITHOUT the context lock: other threads may enqueue meanwhile
      Context* c = sc.c;
      sc.lk.unlock();
      uint64_t scratch[36];
      rc = msm_finish_ctx(c, h->curve, h->slot, out_xyz ? out_xyz : scratch);
    }
  }
  delete h;
  return rc;
}

int ark_hip_msm_sw_device(int curve, const void* d_bases, const void* d_scalars, size_t n, int mont, uint64_t* out_xyz) {
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
int ark_hip_msm_sw(int curve, const uint64_t* bases, const uint64_t* scalars, size_t n, int mont, uint64_t* out_xyz) {
  if (curve < 0 || curve > 4 || !out_xyz || (n && (!bases || !scalars))) return ARK_HIP_ERR_ARG;
  ARK_SCOPE(sc);
  Context* c = sc.c;
  if (n == 0) return msm_stream(c, curve, nullptr, nullptr, nullptr, 0, mont, 0, out_xyz);
  return msm_with_bases(c, curve, bases, n, [&](const void* d_bases, bool fill, BaseCacheEntry* ce) -> int {
    if (!d_bases)   // no resident copy: bases and scalars both stream through the ring, the tail in shrinking pieces
      return msm_stream(c, curve, nullptr, bases, scalars, n, mont, msm_stream_step(n), out_xyz, true, false, true);
    if (fill)       // first call with this set: its bases cross PCIe with the scalars, piece k+1 under piece k's kernels
      return msm_stream(c, curve, d_bases, bases, scalars, n, mont, msm_stream_step(n), out_xyz, true, false, true);
    if (ce && ce->prepared) return ark_hip_msm_prepared((const ark_hip_msm_bases*)ce->prepared, scalars, n, mont, out_xyz);
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
int ark_hip_msm_sw_small_device(int curve, const void* d_bases, const void* d_scalars, size_t n, int scalar_bytes, int max_bits,
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
int ark_hip_msm_sw_small(int curve, const uint64_t* bases, const void* scalars, size_t n, int scalar_bytes, int max_bits,
                         uint64_t* out_xyz) {
  if (curve < 0 || curve > 4 || !out_xyz || (n && (!bases || !scalars))) return ARK_HIP_ERR_ARG;
  if (scalar_bytes != 1 && scalar_bytes != 2 && scalar_bytes != 4 && scalar_bytes != 8) return ARK_HIP_ERR_ARG;
  ARK_SCOPE(sc);
  Context* c = sc.c;
  if (n == 0) return ark_hip_msm_sw_small_device(curve, nullptr, nullptr, 0, scalar_bytes, max_bits, out_xyz);
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
    return ark_hip_msm_sw_small_device(curve, d_bases, c->stage_b.p, n, scalar_bytes, max_bits, out_xyz);
  });
}

// The narrow entries against a PREPARED base set: the per-window table was laid out for 255-bit scalars (its wide windows
// would leave a u32 vector with one dense and one sparse window); row 0 of the table IS the base set, so narrow scalars
// run as a plain MSM over it with a plan of their own.
int ark_hip_msm_prepared_small_device(const ark_hip_msm_bases* bases, const void* d_scalars, size_t n, int scalar_bytes,
                                      int max_bits, uint64_t* out_xyz) {
  if (!bases || !out_xyz) return ARK_HIP_ERR_ARG;
  const PreparedBases* pb = (const PreparedBases*)bases;
  if (n > pb->n) return ARK_HIP_ERR_ARG;
  Scope sc;
  if (int rc = sc.enter(pb->logical)) return rc;
  return ark_hip_msm_sw_small_device(pb->curve, pb->table.p, d_scalars, n, scalar_bytes, max_bits, out_xyz);
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
// Test hook (host only, no device): the verified cache's tag of `words` u64 words -- tests/test_capi_host.py checks that
// edits the round-4 hash could not see (a two-word edit built from its published constants) change it.
int ark_hip_test_base_hash(const uint64_t* p, size_t words, uint64_t out[2]) {
  if (!out || (words && !p)) return ARK_HIP_ERR_ARG;
  const Hash128 h = base_hash(p, words);
  out[0] = h.lo;
  out[1] = h.hi;
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
  const MsmTimings& t = sc.c->msm_tm;
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
int ark_hip_msm_prepared_device(const ark_hip_msm_bases* bases, const void* d_scalars, size_t n, int mont,
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
int ark_hip_msm_prepared(const ark_hip_msm_bases* bases, const uint64_t* scalars, size_t n, int mont, uint64_t* out_xyz) {
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
int ark_hip_msm_sw_chunks(int curve, const uint64_t* bases, size_t n_bases, const uint64_t* scalars, size_t n_scalars,
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

// ---- radix-2 domain / FFT --------------------------------------------------------------------------------
int ark_hip_radix2_domain_new(int field, size_t num_coeffs, ark_hip_radix2_domain* out) {
  if (!out) return ARK_HIP_ERR_ARG;
  switch (field) {
    case ARK_HIP_BN254_FR: return domain_new<BN254_FR>(num_coeffs, out);
    case ARK_HIP_BLS12_381_FR: return domain_new<BLS12_381_FR>(num_coeffs, out);
    case ARK_HIP_BLS12_377_FR: return domain_new<BLS12_377_FR>(num_coeffs, out);
  }
  return ARK_HIP_ERR_ARG;
}
int ark_hip_radix2_domain_get_coset(int field, const ark_hip_radix2_domain* dom, const uint64_t* offset,
                                    ark_hip_radix2_domain* out) {
  if (!dom || !offset || !out) return ARK_HIP_ERR_ARG;
  switch (field) {
    case ARK_HIP_BN254_FR: return domain_coset<BN254_FR>(dom, offset, out);
    case ARK_HIP_BLS12_381_FR: return domain_coset<BLS12_381_FR>(dom, offset, out);
    case ARK_HIP_BLS12_377_FR: return domain_coset<BLS12_377_FR>(dom, offset, out);
  }
  return ARK_HIP_ERR_ARG;
}

static int fft_device_entry(int field, const ark_hip_radix2_domain* dom, void* d, int inverse, size_t num_coeffs) {
  if (!dom || !d) return ARK_HIP_ERR_ARG;
  ARK_SCOPE(sc);
  int zlog = 0;
  if (!inverse && num_coeffs < dom->size) {
    // coefficients beyond num_coeffs are zero by contract (the reference resizes with zeros): make them so up to the
    // power of two the transform reads
    zlog = degree_aware_zlog(dom, num_coeffs);
    const size_t upto = (size_t)dom->size >> zlog;
    if (upto > num_coeffs)
      ARK_HIP_TRY(hipMemsetAsync((char*)d + num_coeffs * 32, 0, (upto - num_coeffs) * 32, sc.c->stream));
  }
  if (int rc = fft_any(sc.c, field, dom, d, inverse, zlog)) return rc;
  return mark_producer(sc.c);
}
static int fft_host_entry(int field, const ark_hip_radix2_domain* dom, uint64_t* data, int inverse, size_t num_coeffs) {
  if (!dom || !data) return ARK_HIP_ERR_ARG;
  ARK_SCOPE(sc);
  Context* c = sc.c;
  const size_t bytes = (size_t)dom->size * 32;
  if (num_coeffs > dom->size) return ARK_HIP_ERR_ARG;
  if (c->stage_a.ensure(bytes)) return ARK_HIP_ERR_NOMEM;
  if (int rc = c->stager.upload(c->stage_a.p, data, num_coeffs * 32, c->stream)) return rc;  // only what is there
  int rc = fft_device_entry(field, dom, c->stage_a.p, inverse, num_coeffs);
  if (rc) return rc;
  ARK_HIP_TRY(hipMemcpyAsync(data, c->stage_a.p, bytes, hipMemcpyDeviceToHost, c->stream));
  ARK_HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}
int ark_hip_fft_in_place(int field, const ark_hip_radix2_domain* dom, uint64_t* data) {
  return fft_host_entry(field, dom, data, 0, dom ? (size_t)dom->size : 0);
}
int ark_hip_ifft_in_place(int field, const ark_hip_radix2_domain* dom, uint64_t* data) {
  return fft_host_entry(field, dom, data, 1, dom ? (size_t)dom->size : 0);
}
int ark_hip_fft_in_place_device(int field, const ark_hip_radix2_domain* dom, void* d) {
  return fft_device_entry(field, dom, d, 0, dom ? (size_t)dom->size : 0);
}
int ark_hip_ifft_in_place_device(int field, const ark_hip_radix2_domain* dom, void* d) {
  return fft_device_entry(field, dom, d, 1, dom ? (size_t)dom->size : 0);
}
int ark_hip_fft_in_place_degree_aware(int field, const ark_hip_radix2_domain* dom, uint64_t* data, size_t num_coeffs) {
  return fft_host_entry(field, dom, data, 0, num_coeffs);
}
int ark_hip_fft_in_place_degree_aware_device(int field, const ark_hip_radix2_domain* dom, void* d, size_t num_coeffs) {
  if (dom && num_coeffs > dom->size) return ARK_HIP_ERR_ARG;
  return fft_device_entry(field, dom, d, 0, num_coeffs);
}

// `count` independent transforms over the same domain, each in place on its own device buffer of dom->size elements.
// Consecutive transforms go to three streams: one transform alone leaves ~20 % of the vector ALU idle around its
// pass boundaries (tail of one kernel, ramp of the next), another one in flight fills it (2^22: 0.53 -> 0.48 ms per
// transform, 2^20: 0.157 -> 0.108, 2^16: 44 -> 18 us; profiles/r2_fft_bench_shapes.txt).  Asynchronous like the single
// transform: later work on the context stream (and ark_hip_synchronize) waits for all of them.
int ark_hip_fft_batch_in_place_device(int field, const ark_hip_radix2_domain* dom, void* const* d_data, size_t count,
                                      int inverse) {
  if (!dom || (count && !d_data)) return ARK_HIP_ERR_ARG;
  for (size_t i = 0; i < count; i++)
    if (!d_data[i]) return ARK_HIP_ERR_ARG;
  ARK_SCOPE(sc);
  Context* c = sc.c;
  if (count == 0) return 0;
  const int lanes = count < 3 ? (int)count : 3;
  for (int j = 0; j < 3; j++)
    if (!c->fft_ev[j]) ARK_HIP_TRY(hipEventCreateWithFlags(&c->fft_ev[j], hipEventDisableTiming));
  for (int j = 0; j + 1 < lanes; j++)
    if (!c->fft_side[j]) ARK_HIP_TRY(hipStreamCreateWithFlags(&c->fft_side[j], hipStreamNonBlocking));
  // the side streams start after whatever is already queued on the context stream (it may produce their inputs)
  ARK_HIP_TRY(hipEventRecord(c->fft_ev[0], c->stream));
  for (int j = 0; j + 1 < lanes; j++) ARK_HIP_TRY(hipStreamWaitEvent(c->fft_side[j], c->fft_ev[0], 0));
  int rc = 0;
  for (size_t i = 0; i < count && rc == 0; i++) {
    const int lane = (int)(i % (size_t)lanes);
    rc = fft_any(c, field, dom, d_data[i], inverse, 0, lane == 0 ? c->stream : c->fft_side[lane - 1]);
  }
  for (int j = 0; j + 1 < lanes; j++) {  // join, also on error: nothing may outlive the call unordered
    (void)hipEventRecord(c->fft_ev[j + 1], c->fft_side[j]);
    (void)hipStreamWaitEvent(c->stream, c->fft_ev[j + 1], 0);
  }
  if (rc == 0) rc = mark_producer(c);
  return rc;
}

// r[i] = a[i] * b[i] over n Fr elements in device memory (Evaluations *= Evaluations,
// poly/src/evaluations/univariate/mod.rs MulAssign; the middle step of DensePolynomial multiplication,
// poly/src/polynomial/univariate/dense.rs:641-656).  Asynchronous on the context stream; r may alias a or b.
int ark_hip_fr_mul_device(int field, const void* d_a, const void* d_b, void* d_r, size_t n) {
  if (n && (!d_a || !d_b || !d_r)) return ARK_HIP_ERR_ARG;
  ARK_SCOPE(sc);
  if (int rc = fr_mul_dispatch(field, d_a, d_b, d_r, n, sc.c->stream)) return rc;
  return mark_producer(sc.c);
}

// `&DensePolynomial * &DensePolynomial` (poly/src/polynomial/univariate/dense.rs:641-656): zero if either factor is zero;
// otherwise evaluate both over the radix-2 domain of size >= na + nb - 1 (evaluate_over_domain_by_ref, univariate/mod.rs:
// 305-360), multiply the evaluations pointwise (Evaluations *=) and interpolate (evaluations/univariate/mod.rs:40-50).
// HOST pointers in and out; in between everything stays on the device: ONE upload of the two coefficient vectors (the zero
// padding is written on the device -- or never read: degree-aware transforms), the two forward transforms in flight
// together, the pointwise product, the inverse transform, ONE download of the na + nb - 1 coefficients.
// out: room for na + nb - 1 elements; *out_len: the product's coefficient count with leading zeros dropped, as
// DensePolynomial::from_coefficients_vec leaves it (0: the zero polynomial).  ARK_HIP_ERR_ARG when the field's 2-adicity
// cannot hold the domain (the reference panics: "field is not smooth enough to construct domain").
int ark_hip_poly_mul(int field, const uint64_t* a, size_t na, const uint64_t* b, size_t nb, uint64_t* out, size_t* out_len) {
  if (!out_len || (na && !a) || (nb && !b)) return ARK_HIP_ERR_ARG;
  *out_len = 0;
  auto all_zero = [](const uint64_t* p, size_t n) {
    for (size_t i = 0; i < 4 * n; i++)
      if (p[i]) return false;
    return true;
  };
  if (na == 0 || nb == 0 || all_zero(a, na) || all_zero(b, nb)) return 0;   // DensePolynomial::is_zero
  if (!out) return ARK_HIP_ERR_ARG;
  const size_t len = na + nb - 1;
  ark_hip_radix2_domain dom;
  if (int rc = ark_hip_radix2_domain_new(field, len, &dom)) return rc;
  const size_t n = (size_t)dom.size;
  ARK_SCOPE(sc);
  Context* c = sc.c;
  if (c->stage_a.cap < n * 32 || c->stage_b.cap < n * 32) {
    if (int rc = sync_compute(c)) return rc;
    if (c->stage_a.ensure(n * 32) || c->stage_b.ensure(n * 32)) return ARK_HIP_ERR_NOMEM;
  }
  if (int rc = c->stager.upload(c->stage_a.p, a, na * 32, c->stream)) return rc;
  if (int rc = c->stager.upload(c->stage_b.p, b, nb * 32, c->stream)) return rc;
  // forward transforms: short inputs take the degree-aware path (their padding is never read beyond the next power of
  // two); the two run on two streams (ark_hip_fft_batch_in_place_device's arrangement)
  auto pad = [&](void* d, size_t have) -> int {
    const int zlog = degree_aware_zlog(&dom, have);
    const size_t upto = n >> zlog;
    if (upto > have) ARK_HIP_TRY(hipMemsetAsync((char*)d + have * 32, 0, (upto - have) * 32, c->stream));
    return zlog;
  };
  const int za = pad(c->stage_a.p, na), zb = pad(c->stage_b.p, nb);
  if (za < 0 || zb < 0) return za < 0 ? za : zb;
  if (!c->fft_ev[0])
    for (int j = 0; j < 3; j++) ARK_HIP_TRY(hipEventCreateWithFlags(&c->fft_ev[j], hipEventDisableTiming));
  if (!c->fft_side[0]) ARK_HIP_TRY(hipStreamCreateWithFlags(&c->fft_side[0], hipStreamNonBlocking));
  ARK_HIP_TRY(hipEventRecord(c->fft_ev[0], c->stream));
  ARK_HIP_TRY(hipStreamWaitEvent(c->fft_side[0], c->fft_ev[0], 0));
  int rc = fft_any(c, field, &dom, c->stage_a.p, 0, za, c->stream);
  if (rc == 0) rc = fft_any(c, field, &dom, c->stage_b.p, 0, zb, c->fft_side[0]);
  (void)hipEventRecord(c->fft_ev[1], c->fft_side[0]);
  (void)hipStreamWaitEvent(c->stream, c->fft_ev[1], 0);
  if (rc == 0) rc = fr_mul_dispatch(field, c->stage_a.p, c->stage_b.p, c->stage_a.p, n, c->stream);
  if (rc == 0) rc = fft_any(c, field, &dom, c->stage_a.p, 1, 0, c->stream);
  if (rc) {
    (void)hipStreamSynchronize(c->stream);
    return rc;
  }
  ARK_HIP_TRY(hipMemcpyAsync(out, c->stage_a.p, len * 32, hipMemcpyDeviceToHost, c->stream));
  ARK_HIP_TRY(hipStreamSynchronize(c->stream));
  size_t top = len;   // truncate_leading_zeros (dense.rs)
  while (top > 0 && !(out[4 * top - 1] | out[4 * top - 2] | out[4 * top - 3] | out[4 * top - 4])) top--;
  *out_len = top;
  return 0;
}

// out = base^exp in Fr (host arithmetic): domain elements / twiddles for hosts without field code of their own
int ark_hip_fr_pow(int field, const uint64_t* base, uint64_t exp, uint64_t* out) {
  if (!base || !out) return ARK_HIP_ERR_ARG;
  uint64_t e[1] = {exp};
  switch (field) {
    case ARK_HIP_BN254_FR: host_pow<BN254_FR>(Fp<BN254_FR>::load(base), e, 1).store(out); return 0;
    case ARK_HIP_BLS12_381_FR: host_pow<BLS12_381_FR>(Fp<BLS12_381_FR>::load(base), e, 1).store(out); return 0;
    case ARK_HIP_BLS12_377_FR: host_pow<BLS12_377_FR>(Fp<BLS12_377_FR>::load(base), e, 1).store(out); return 0;
  }
  return ARK_HIP_ERR_ARG;
}

// G-point transform along the slow axis of a [G][cols] array in device memory: the cross-GPU stage of a
// sharded FFT (algebra_amd/dist.py).  root = primitive G-th root of unity to use (w_n^(n/G) or its inverse).
int ark_hip_fft_axis_device(int field, void* d_data, unsigned G, size_t cols, const uint64_t* root) {
  if (!d_data || !root) return ARK_HIP_ERR_ARG;
  ARK_SCOPE(sc);
  if (int rc = fft_axis_dispatch(field, sc.c->fft, d_data, d_data, G, cols, root, sc.c->stream)) return rc;
  return mark_producer(sc.c);
}

// ---- transform whose coefficients are GROUP elements --------------------------------------------------------------------
// EvaluationDomain::fft_in_place / ifft_in_place for T = Projective<P> (poly/src/domain/mod.rs:332-362 with
// radix2/fft.rs:74-119; the reference's own use: poly/src/test.rs:57): n = dom->size Jacobian points of `curve`, whose scalar
// field must be the domain's field, transformed in place -- forward: X_j = sum_i [(h g^j)^i] P_i; inverse:
// P_i = [n^-1 h^-i] sum_j [g^-ij] X_j.  The caller pads with identities (z = 0) to the domain size as the reference's
// resize does.  gfft.cuh.
static int gfft_entry(Context* c, int curve, const ark_hip_radix2_domain* dom, void* d_jac, int inverse) {
  const int field = CURVES[curve].scalar_field;
  const int k = (int)dom->log_size_of_group;
  if (k < 0 || k > 26 || dom->size != ((uint64_t)1 << k)) return ARK_HIP_ERR_ARG;
  const size_t n = (size_t)1 << k;
  const bool coset = !field_is_one(field, dom->offset);
  const uint32_t* roots = nullptr;
  if (k >= 1)
    if (int rc = fft_roots_dispatch(field, c->fft, k, inverse ? dom->group_gen_inv : dom->group_gen, c->stream, &roots)) return rc;
  const size_t wb = gfft_work_bytes_any(curve, k);
  if (wb == 0) return ARK_HIP_ERR_ARG;
  if (c->gfft_work.cap < wb || c->gfft_scal.cap < n * 32) {
    if (int rc = sync_compute(c)) return rc;
    if (c->gfft_work.ensure(wb) || c->gfft_scal.ensure(n * 32)) return ARK_HIP_ERR_NOMEM;
  }
  const uint32_t *pre = nullptr, *post = nullptr;
  if (!inverse && coset) {        // distribute_powers(coeffs, offset)                                   fft.rs:74-79
    if (int rc = fft_scalars_dispatch(field, c->fft, dom->offset, nullptr, n, c->gfft_scal.p, c->stream)) return rc;
    pre = (const uint32_t*)c->gfft_scal.p;
  }
  if (inverse) {                  // x[i] *= size_inv * offset_inv^i  (offset_inv = 1 off a coset)         fft.rs:81-88
    if (int rc = fft_scalars_dispatch(field, c->fft, dom->offset_inv, dom->size_inv, n, c->gfft_scal.p, c->stream)) return rc;
    post = (const uint32_t*)c->gfft_scal.p;
  }
  if (int rc = gfft_run_dispatch(curve, d_jac, k, roots, pre, post, c->gfft_work.p, c->stream)) return rc;
  return mark_producer(c);
}
int ark_hip_fft_group_in_place_device(int curve, const ark_hip_radix2_domain* dom, void* d_jac_points, int inverse) {
  if (curve < 0 || curve > 4 || !dom || !d_jac_points) return ARK_HIP_ERR_ARG;
  ARK_SCOPE(sc);
  return gfft_entry(sc.c, curve, dom, d_jac_points, inverse);
}
int ark_hip_fft_group_in_place(int curve, const ark_hip_radix2_domain* dom, uint64_t* jac_points, int inverse) {
  if (curve < 0 || curve > 4 || !dom || !jac_points) return ARK_HIP_ERR_ARG;
  ARK_SCOPE(sc);
  Context* c = sc.c;
  const size_t bytes = (size_t)dom->size * CURVES[curve].fe_words * 3 * 8;
  if (c->stage_a.cap < bytes) {
    if (int rc = sync_compute(c)) return rc;
    if (c->stage_a.ensure(bytes)) return ARK_HIP_ERR_NOMEM;
  }
  if (int rc = c->stager.upload(c->stage_a.p, jac_points, bytes, c->stream)) return rc;
  if (int rc = gfft_entry(c, curve, dom, c->stage_a.p, inverse)) {
    (void)hipStreamSynchronize(c->stream);
    return rc;   // the device worked on its own copy: the caller's points are intact
  }
  ARK_HIP_TRY(hipMemcpyAsync(jac_points, c->stage_a.p, bytes, hipMemcpyDeviceToHost, c->stream));
  ARK_HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

// The rest of the pointwise algebra on device-resident vectors of Fr (Evaluations +=, -=, negation; a polynomial or an
// evaluation vector times a field element -- poly/src/evaluations/univariate/mod.rs:104-180, polynomial/univariate/
// dense.rs:343-371, :604-622): what a chain evaluate_over_domain -> pointwise -> interpolate needs besides the transforms
// and ark_hip_fr_mul_device to stay on the device between ONE upload and ONE download.  Asynchronous on the context
// stream; r may alias a or b.
int ark_hip_fr_add_device(int field, const void* d_a, const void* d_b, void* d_r, size_t n) {
  if (n && (!d_a || !d_b || !d_r)) return ARK_HIP_ERR_ARG;
  ARK_SCOPE(sc);
  if (int rc = fr_op_dispatch(field, 0, d_a, d_b, d_r, n, sc.c->stream)) return rc;
  return mark_producer(sc.c);
}
int ark_hip_fr_sub_device(int field, const void* d_a, const void* d_b, void* d_r, size_t n) {
  if (n && (!d_a || !d_b || !d_r)) return ARK_HIP_ERR_ARG;
  ARK_SCOPE(sc);
  if (int rc = fr_op_dispatch(field, 1, d_a, d_b, d_r, n, sc.c->stream)) return rc;
  return mark_producer(sc.c);
}
int ark_hip_fr_neg_device(int field, const void* d_a, void* d_r, size_t n) {
  if (n && (!d_a || !d_r)) return ARK_HIP_ERR_ARG;
  ARK_SCOPE(sc);
  if (int rc = fr_op_dispatch(field, 4, d_a, nullptr, d_r, n, sc.c->stream)) return rc;
  return mark_producer(sc.c);
}
// r[i] = a[i] * k; k: one Montgomery element in HOST memory, read before the call returns
int ark_hip_fr_scale_device(int field, const void* d_a, const uint64_t* k, void* d_r, size_t n) {
  if (!k || (n && (!d_a || !d_r))) return ARK_HIP_ERR_ARG;
  ARK_SCOPE(sc);
  if (int rc = fr_scale_dispatch(field, d_a, k, d_r, n, sc.c->stream)) return rc;
  return mark_producer(sc.c);
}
// r[i] = a[i] / b[i] (Evaluations /= Evaluations, evaluations/univariate/mod.rs:142-163) and r[i] = 1 / a[i]
// (ark_ff::batch_inversion, ff/src/fields/mod.rs:358-385): a zero divisor gives zero, as the reference's batch inversion
// leaves zeros in place.  r may alias an operand.
int ark_hip_fr_div_device(int field, const void* d_a, const void* d_b, void* d_r, size_t n) {
  if (n && (!d_a || !d_b || !d_r)) return ARK_HIP_ERR_ARG;
  ARK_SCOPE(sc);
  if (int rc = fr_div_dispatch(field, d_a, d_b, d_r, n, sc.c->stream)) return rc;
  return mark_producer(sc.c);
}
int ark_hip_fr_inverse_device(int field, const void* d_a, void* d_r, size_t n) {
  if (n && (!d_a || !d_r)) return ARK_HIP_ERR_ARG;
  ARK_SCOPE(sc);
  if (int rc = fr_div_dispatch(field, nullptr, d_a, d_r, n, sc.c->stream)) return rc;
  return mark_producer(sc.c);
}
// device-to-device copy / byte fill on the context stream (a device vector's clone() and its zero-extension), ordered with
// the transforms and the pointwise kernels like every other *_device entry
int ark_hip_memcpy_d2d(void* dst_dptr, const void* src_dptr, size_t bytes) {
  if (bytes && (!dst_dptr || !src_dptr)) return ARK_HIP_ERR_ARG;
  ARK_SCOPE(sc);
  if (bytes) ARK_HIP_TRY(hipMemcpyAsync(dst_dptr, src_dptr, bytes, hipMemcpyDeviceToDevice, sc.c->stream));
  return mark_producer(sc.c);
}
int ark_hip_memset_device(void* dptr, int value, size_t bytes) {
  if (bytes && !dptr) return ARK_HIP_ERR_ARG;
  ARK_SCOPE(sc);
  if (bytes) ARK_HIP_TRY(hipMemsetAsync(dptr, value, bytes, sc.c->stream));
  return mark_producer(sc.c);
}

// 0: the saturated pass kernel (default), 1: the carry-free 9 x 29-bit pass kernel, -1: back to the environment's choice
int ark_hip_fft_set_kernel(int variant) {
  if (variant < -1 || variant > 1) return ARK_HIP_ERR_ARG;
  ARK_SCOPE(sc);
  std::lock_guard<std::mutex> lock(sc.c->fft.mu);
  sc.c->fft.kernel_variant = variant;
  return 0;
}
int ark_hip_fft_set_timing(int enable) {
  ARK_SCOPE(sc);
  sc.c->fft_timing = enable != 0;
  return 0;
}
int ark_hip_fft_last_timing(double out[10]) {
  if (!out) return ARK_HIP_ERR_ARG;
  ARK_SCOPE(sc);
  out[0] = sc.c->fft_tm.total;
  out[1] = sc.c->fft_tm.npass;
  for (int i = 0; i < 8; i++) out[2 + i] = sc.c->fft_tm.pass[i];
  return 0;
}

// ---- one process per GPU: RCCL inside the library ----------------------------------------------------------
int ark_hip_comm_unique_id(void* out_id) {
  if (!out_id) return ARK_HIP_ERR_ARG;
  const RcclApi* api = rccl_api();
  if (!api) return ARK_HIP_ERR_COMM;
  static_assert(sizeof(ncclUniqueId) == ARK_HIP_COMM_ID_BYTES, "ARK_HIP_COMM_ID_BYTES");
  ncclUniqueId id;
  ARK_RCCL_TRY(api, api->GetUniqueId(&id));
  memcpy(out_id, &id, sizeof(id));
  return 0;
}
int ark_hip_comm_init(const void* id, int rank, int world) {
  if (!id || world < 1 || rank < 0 || rank >= world) return ARK_HIP_ERR_ARG;
  if (world > MAX_DEV) return ARK_HIP_ERR_ARG;   // the gathered headers of the sharded MSM have MAX_DEV pinned slots (sums_buffers)
  ARK_SCOPE(sc);
  Context* c = sc.c;
  if (c->comm) return ARK_HIP_ERR_BUSY;   // one communicator per device: destroy first
  const RcclApi* api = rccl_api();
  if (!api) return ARK_HIP_ERR_COMM;
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof(uid));
  ncclComm_t comm = nullptr;
  ARK_RCCL_TRY(api, api->CommInitRank(&comm, world, uid, rank));
  c->comm = comm;
  c->comm_rank = rank;
  c->comm_world = world;
  return comm_streams(c);
}
int ark_hip_comm_info(int* rank, int* world) {
  ARK_SCOPE(sc);
  if (rank) *rank = sc.c->comm ? sc.c->comm_rank : 0;
  if (world) *world = sc.c->comm ? sc.c->comm_world : 1;
  return 0;
}
int ark_hip_comm_destroy(void) {
  ARK_SCOPE(sc);
  Context* c = sc.c;
  if (!c->comm) return 0;
  if (int rc = sync_compute(c)) return rc;
  if (c->comm_stream) ARK_HIP_TRY(hipStreamSynchronize(c->comm_stream));
  const RcclApi* api = rccl_api();
  if (!api) return ARK_HIP_ERR_COMM;
  ARK_RCCL_TRY(api, api->CommDestroy(c->comm));
  c->comm = nullptr;
  c->comm_rank = 0;
  c->comm_world = 1;
  return 0;
}

// local MSM -> exchange of the part sums -> one host tail (all ranks equal plans), or the fallback on finished partials
static int msm_sharded_finish(Context* c, int curve, int slot_full, uint64_t* out_xyz) {
  // slot_full < 0: this rank's enqueue (or its argument check) failed with that code.  With a communicator it still goes
  // through the exchange -- every rank reaches the collective or none does -- and all ranks return the code together.
  const bool have_job = slot_full >= 0;
  if (!c->comm || c->comm_world == 1) return have_job ? msm_finish_ctx(c, curve, slot_full, out_xyz) : slot_full;
  auto drop = [&]() { if (have_job) msm_discard_ctx(c, curve, slot_full); };   // frees the lane's job slot: on EVERY error exit
  const RcclApi* api = rccl_api();
  if (!api) {
    drop();
    return ARK_HIP_ERR_COMM;
  }
  const int world = c->comm_world;
  int rc = sums_buffers(c, curve, world);
  if (rc) {   // not even the exchange buffers: nothing can be posted (the peers' collective fails or times out in RCCL)
    drop();
    return rc;
  }
  const size_t bb = sums_block_bytes(curve);
  char* d_send = (char*)c->comm_sums.p;
  char* d_recv = d_send + bb;
  MsmSumsHeader mine{};
  rc = have_job ? sums_write_block(c, curve, slot_full, d_send, &mine) : slot_full;
  if (rc) {
    const int local = rc;
    if (int rc2 = sums_write_failure(c, d_send, local, &mine)) {
      drop();
      return rc2;
    }
  }
  // only the header and the parts a plan can have travel: SUMS_MAX_PARTS bounds the block, the count is what every rank posts
  {
    ncclResult_t r = api->AllGather(d_send, d_recv, bb, ncclUint8, c->comm, c->stream);
    if (r != ncclSuccess) {
      fprintf(stderr, "ark_hip: ncclAllGather of the part sums failed: %s\n", api->GetErrorString(r));
      drop();
      return ARK_HIP_ERR_COMM;
    }
  }
  bool agree = false;
  rc = sums_reduce(c, curve, mine, d_recv, world, out_xyz, &agree);
  if (rc || agree || !have_job) {
    drop();   // the job's own host tail is not needed
    return rc ? rc : (have_job ? 0 : slot_full);
  }
  uint64_t part[36];
  if (int rc2 = msm_finish_ctx(c, curve, slot_full, part)) return rc2;   // (finish frees the slot itself)
  return msm_sharded_combine(c, curve, part, out_xyz);
}
int ark_hip_msm_sw_device_sharded(int curve, const void* d_bases, const void* d_scalars, size_t n_local, int mont,
                                  uint64_t* out_xyz) {
  if (curve < 0 || curve > 4 || !out_xyz || (n_local && (!d_bases || !d_scalars))) return ARK_HIP_ERR_ARG;
  ARK_SCOPE(sc);
  int slot = msm_enqueue_ctx(sc.c, curve, d_bases, 0, nullptr, d_scalars, n_local, mont);
  return msm_sharded_finish(sc.c, curve, slot, out_xyz);   // (a failed enqueue included: every rank reaches the collective)
}
int ark_hip_msm_prepared_device_sharded(const ark_hip_msm_bases* bases, const void* d_scalars, size_t n_local, int mont,
                                        uint64_t* out_xyz) {
  if (!bases || !out_xyz) return ARK_HIP_ERR_ARG;
  const PreparedBases* pb = (const PreparedBases*)bases;
  Scope sc;
  if (int rc = sc.enter(pb->logical)) return rc;
  // more scalars than the set holds is THIS rank's error only: it still joins the exchange and every rank returns it
  int slot = n_local > pb->n ? ARK_HIP_ERR_ARG
                             : msm_enqueue_ctx(sc.c, pb->curve, pb->table.p, pb->n, &pb->plan, d_scalars, n_local, mont);
  return msm_sharded_finish(sc.c, pb->curve, slot, out_xyz);
}
// Test hook: the exchange of msm_sharded_finish with the ranks EMULATED in one process (one GPU): `world` local MSMs run one
// after the other, each block lands where the all-gather would put it, then the same sum kernel / header check / host tail --
// or, when the plans differ, the same fallback (partials summed on the host).  *path: 1 = part sums added on the device,
// 2 = fallback.  Everything but the RCCL call itself.
int ark_hip_test_msm_sharded_emulated(int curve, int world, const void* const* d_bases, const void* const* d_scalars,
                                      const size_t* n_local, int mont, uint64_t* out_xyz, int* path) {
  if (curve < 0 || curve > 4 || world < 1 || world > MAX_DEV || !d_bases || !d_scalars || !n_local || !out_xyz) return ARK_HIP_ERR_ARG;
  ARK_SCOPE(sc);
  Context* c = sc.c;
  if (int rc = sums_buffers(c, curve, world)) return rc;
  const size_t bb = sums_block_bytes(curve), pw = (size_t)CURVES[curve].fe_words * 3;
  char* d_recv = (char*)c->comm_sums.p + bb;
  std::vector<uint64_t> partials((size_t)world * pw);
  MsmSumsHeader h0{};
  for (int r = 0; r < world; r++) {
    int slot = msm_enqueue_ctx(c, curve, d_bases[r], 0, nullptr, d_scalars[r], n_local[r], mont);
    if (slot < 0) return slot;
    MsmSumsHeader h{};
    int rc = sums_write_block(c, curve, slot, d_recv + (size_t)r * bb, &h, r);
    if (r == 0) h0 = h;
    const int rc2 = msm_finish_ctx(c, curve, slot, &partials[(size_t)r * pw]);   // (the fallback's input; also frees the slot)
    if (rc || rc2) return rc ? rc : rc2;
  }
  bool agree = false;
  if (int rc = sums_reduce(c, curve, h0, d_recv, world, out_xyz, &agree)) return rc;
  if (path) *path = agree ? 1 : 2;
  if (agree) return 0;
  return ark_hip_sw_sum(curve, partials.data(), (size_t)world, out_xyz);
}

int ark_hip_fft_shard_local_device(int field, const ark_hip_radix2_domain* dom, int rank, int world, void* d_local,
                                   int inverse) {
  if (!dom || !d_local) return ARK_HIP_ERR_ARG;
  ShardConsts k;
  if (int rc = shard_consts_any(field, dom, rank, world, inverse, &k)) return rc;
  ARK_SCOPE(sc);
  if (int rc = shard_local(sc.c, field, k, d_local, sc.c->stream)) return rc;
  return mark_producer(sc.c);
}
int ark_hip_fft_shard_cross_device(int field, const ark_hip_radix2_domain* dom, int world, const void* d_src, void* d_dst,
                                   int inverse) {
  if (!dom || !d_src || !d_dst) return ARK_HIP_ERR_ARG;
  ShardConsts k;
  if (int rc = shard_consts_any(field, dom, 0, world, inverse, &k)) return rc;
  ARK_SCOPE(sc);
  if (int rc = fft_axis_dispatch(field, sc.c->fft, d_src, d_dst, k.G, k.sub, k.root_G, sc.c->stream)) return rc;
  return mark_producer(sc.c);
}
int ark_hip_fft_sharded_device(int field, const ark_hip_radix2_domain* dom, void* d_local, int inverse) {
  if (!dom || !d_local) return ARK_HIP_ERR_ARG;
  ARK_SCOPE(sc);
  Context* c = sc.c;
  if (!c->comm || c->comm_world == 1) {   // no communicator: the single-GPU transform
    if (int rc = fft_any(c, field, dom, d_local, inverse, 0)) return rc;
    return mark_producer(c);
  }
  const RcclApi* api = rccl_api();
  if (!api) return ARK_HIP_ERR_COMM;
  ShardConsts k;
  if (int rc = shard_consts_any(field, dom, c->comm_rank, c->comm_world, inverse, &k)) return rc;
  if (int rc = comm_streams(c)) return rc;
  if (c->comm_tmp.cap < k.m * 32) {
    if (int rc = sync_compute(c)) return rc;
    ARK_HIP_TRY(hipStreamSynchronize(c->comm_stream));
    if (c->comm_tmp.ensure(k.m * 32)) return ARK_HIP_ERR_NOMEM;
  }
  char* loc = (char*)d_local;
  char* tmp = (char*)c->comm_tmp.p;
  const int S = comm_slices(k.sub);
  const size_t cs = k.sub / (size_t)S;
  const uint32_t* pw = nullptr;
  if (int rc = fft_axis_prepare_dispatch(field, c->fft, k.G, k.root_G, c->stream, &pw)) return rc;
  if (!inverse) {
    if (int rc = shard_local(c, field, k, d_local, c->stream)) return rc;
    ARK_HIP_TRY(hipEventRecord(c->comm_ev[0][0], c->stream));
    ARK_HIP_TRY(hipStreamWaitEvent(c->comm_stream, c->comm_ev[0][0], 0));
    for (int i = 0; i < S; i++) {   // slice i+1 travels while the G-point kernel works on slice i
      if (int rc = exchange_slice(c, api, loc, tmp, k.sub, (size_t)i * cs, cs, c->comm_stream)) return rc;
      ARK_HIP_TRY(hipEventRecord(c->comm_ev[1][i], c->comm_stream));
      ARK_HIP_TRY(hipStreamWaitEvent(c->stream, c->comm_ev[1][i], 0));
      if (int rc = fft_axis_launch_dispatch(field, tmp + (size_t)i * cs * 32, loc + (size_t)i * cs * 32, k.G, k.sub, cs, pw, c->stream))
        return rc;
    }
  } else {
    for (int i = 0; i < S; i++) {   // the exchange of slice i travels while the G-point kernel works on slice i+1
      if (int rc = fft_axis_launch_dispatch(field, loc + (size_t)i * cs * 32, tmp + (size_t)i * cs * 32, k.G, k.sub, cs, pw, c->stream))
        return rc;
      ARK_HIP_TRY(hipEventRecord(c->comm_ev[0][i], c->stream));
      ARK_HIP_TRY(hipStreamWaitEvent(c->comm_stream, c->comm_ev[0][i], 0));
      if (int rc = exchange_slice(c, api, tmp, loc, k.sub, (size_t)i * cs, cs, c->comm_stream)) return rc;
    }
    ARK_HIP_TRY(hipEventRecord(c->comm_ev[1][0], c->comm_stream));
    ARK_HIP_TRY(hipStreamWaitEvent(c->stream, c->comm_ev[1][0], 0));
    if (int rc = shard_local(c, field, k, d_local, c->stream)) return rc;
  }
  ARK_HIP_TRY(hipGetLastError());
  return mark_producer(c);
}

// ---- host-side group helpers (no device involved) ----
int ark_hip_sw_sum(int curve, const uint64_t* jac_points, size_t n, uint64_t* out_xyz) {
  if (!out_xyz || (n && !jac_points)) return ARK_HIP_ERR_ARG;
  switch (curve) {
    case 0: return host_sum<BN254_G1>(jac_points, n, out_xyz);
    case 1: return host_sum<BLS12_381_G1>(jac_points, n, out_xyz);
    case 2: return host_sum<BLS12_377_G1>(jac_points, n, out_xyz);
    case 3: return host_sum<BLS12_377_G2>(jac_points, n, out_xyz);
    case 4: return host_sum<BLS12_381_G2>(jac_points, n, out_xyz);
  }
  return ARK_HIP_ERR_ARG;
}
int ark_hip_test_msm_host_fold(int curve, const uint64_t* parts, int windows, int nbits, int log2_l0, const int* widths,
                               uint64_t* out_xyz) {
  if (!parts || !widths || !out_xyz || windows < 1 || windows > 256 || nbits < 0 || nbits > 31 || log2_l0 < 0 || log2_l0 > 16)
    return ARK_HIP_ERR_ARG;
  switch (curve) {
    case 0: return host_fold<BN254_G1>(parts, windows, nbits, log2_l0, widths, out_xyz);
    case 1: return host_fold<BLS12_381_G1>(parts, windows, nbits, log2_l0, widths, out_xyz);
    case 2: return host_fold<BLS12_377_G1>(parts, windows, nbits, log2_l0, widths, out_xyz);
    case 3: return host_fold<BLS12_377_G2>(parts, windows, nbits, log2_l0, widths, out_xyz);
    case 4: return host_fold<BLS12_381_G2>(parts, windows, nbits, log2_l0, widths, out_xyz);
  }
  return ARK_HIP_ERR_ARG;
}
int ark_hip_sw_into_affine(int curve, const uint64_t* jac_points, size_t n, uint64_t* out_xy) {
  if (n && (!jac_points || !out_xy)) return ARK_HIP_ERR_ARG;
  switch (curve) {
    case 0: return host_into_affine<BN254_G1>(jac_points, n, out_xy);
    case 1: return host_into_affine<BLS12_381_G1>(jac_points, n, out_xy);
    case 2: return host_into_affine<BLS12_377_G1>(jac_points, n, out_xy);
    case 3: return host_into_affine<BLS12_377_G2>(jac_points, n, out_xy);
    case 4: return host_into_affine<BLS12_381_G2>(jac_points, n, out_xy);
  }
  return ARK_HIP_ERR_ARG;
}

// out[i] = in[i] + delta on the device (affine in/out); d_in may equal d_out
int ark_hip_sw_add_affine_device(int curve, const void* d_in, void* d_out, size_t n, const uint64_t* delta_xy) {
  if (curve < 0 || curve > 4 || !delta_xy || (n && (!d_in || !d_out))) return ARK_HIP_ERR_ARG;
  ARK_SCOPE(sc);
  Context* c = sc.c;
  size_t ab = (size_t)CURVES[curve].fe_words * 16;
  if (c->stage_c.ensure(ab)) return ARK_HIP_ERR_NOMEM;
  ARK_HIP_TRY(hipMemcpyAsync(c->stage_c.p, delta_xy, ab, hipMemcpyHostToDevice, c->stream));
  int rc = add_affine_dispatch(curve, d_in, d_out, n, c->stage_c.p, c->stream);
  if (rc) return rc;
  ARK_HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

// CurveGroup::normalize_batch for n Projective points in device memory -> n Affine points (device memory)
int ark_hip_sw_normalize_batch_device(int curve, const void* d_jac, void* d_out_xy, size_t n) {
  if (curve < 0 || curve > 4 || (n && (!d_jac || !d_out_xy))) return ARK_HIP_ERR_ARG;
  ARK_SCOPE(sc);
  int rc = normalize_dispatch(curve, d_jac, d_out_xy, n, sc.c->stream);
  if (rc) return rc;
  ARK_HIP_TRY(hipStreamSynchronize(sc.c->stream));
  return 0;
}

// The same from HOST memory (what the Rust hook behind CurveGroup::normalize_batch hands over, group.rs:302-319): one
// upload of the n Projective points, the lane-batched inversion kernel, one download of the n Affine points.
int ark_hip_sw_normalize_batch(int curve, const uint64_t* jac_points, size_t n, uint64_t* out_xy) {
  if (curve < 0 || curve > 4 || (n && (!jac_points || !out_xy))) return ARK_HIP_ERR_ARG;
  if (n == 0) return 0;
  ARK_SCOPE(sc);
  Context* c = sc.c;
  const size_t fb = (size_t)CURVES[curve].fe_words * 8;
  if (c->stage_a.cap < n * 3 * fb || c->stage_b.cap < n * 2 * fb) {
    if (int rc = sync_compute(c)) return rc;
    if (c->stage_a.ensure(n * 3 * fb) || c->stage_b.ensure(n * 2 * fb)) return ARK_HIP_ERR_NOMEM;
  }
  if (int rc = c->stager.upload(c->stage_a.p, jac_points, n * 3 * fb, c->stream)) return rc;
  if (int rc = normalize_dispatch(curve, c->stage_a.p, c->stage_b.p, n, c->stream)) return rc;
  ARK_HIP_TRY(hipMemcpyAsync(out_xy, c->stage_b.p, n * 2 * fb, hipMemcpyDeviceToHost, c->stream));
  ARK_HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

// ---- test hooks ----
static int run_elementwise(size_t abytes, size_t bbytes, size_t rbytes, const void* a, const void* b, void* r,
                           elementwise_fn fn, int op, size_t n) {
  if (!fn) return ARK_HIP_ERR_ARG;  // curve/field not in this (development) build
  ARK_SCOPE(sc);
  Context* c = sc.c;
  if (n == 0) return 0;
  if (c->stage_a.ensure(abytes) || c->stage_b.ensure(bbytes ? bbytes : 16) || c->stage_c.ensure(rbytes)) return ARK_HIP_ERR_NOMEM;
  ARK_HIP_TRY(hipMemcpyAsync(c->stage_a.p, a, abytes, hipMemcpyHostToDevice, c->stream));
  if (b) ARK_HIP_TRY(hipMemcpyAsync(c->stage_b.p, b, bbytes, hipMemcpyHostToDevice, c->stream));
  int rc = fn(op, c->stage_a.p, b ? c->stage_b.p : nullptr, c->stage_c.p, n, c->stream);
  if (rc) return rc;
  ARK_HIP_TRY(hipMemcpyAsync(r, c->stage_c.p, rbytes, hipMemcpyDeviceToHost, c->stream));
  ARK_HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

int ark_hip_test_field_op(int field, int op, const uint64_t* a, const uint64_t* b, uint64_t* r, size_t n) {
  const bool lazy_op = op >= 20 && op <= 22;   // the 28-bit-limb device forms (testops.cuh)
  if (!a || !r || op < 0 || (op > 8 && !lazy_op) || op == 6) return ARK_HIP_ERR_ARG;
  size_t fb = field_bytes(field);
  if (field < 0 || field > 5) return ARK_HIP_ERR_ARG;
  if ((field == ARK_HIP_BN254_FQ || field == ARK_HIP_BLS12_381_FQ || field == ARK_HIP_BLS12_377_FQ) && op > 5 && !lazy_op)
    return ARK_HIP_ERR_ARG;
  return run_elementwise(n * fb, b ? n * fb : 0, n * fb, a, b, r, field_op_fn(field), op, n);
}

int ark_hip_test_basefield_op(int curve, int op, const uint64_t* a, const uint64_t* b, uint64_t* r, size_t n) {
  if (curve < 0 || curve > 4 || !a || !r || op < 0 || op > 5) return ARK_HIP_ERR_ARG;
  size_t fb = (size_t)CURVES[curve].fe_words * 8;
  return run_elementwise(n * fb, b ? n * fb : 0, n * fb, a, b, r, basefield_op_fn(curve), op, n);
}

int ark_hip_test_host_basefield_op(int curve, int op, const uint64_t* a, const uint64_t* b, uint64_t* r, size_t n) {
  if (curve < 0 || curve > 4 || !a || !r || op < 0 || op > 5 || (op <= 2 && !b)) return ARK_HIP_ERR_ARG;
  switch (curve) {
#ifndef ARK_HIP_DEV
    case ARK_HIP_BN254_G1: host_field_ops<BN254_G1::F>(op, a, b, r, n); return 0;
    case ARK_HIP_BLS12_377_G1: host_field_ops<BLS12_377_G1::F>(op, a, b, r, n); return 0;
    case ARK_HIP_BLS12_377_G2: host_field_ops<BLS12_377_G2::F>(op, a, b, r, n); return 0;
    case ARK_HIP_BLS12_381_G2: host_field_ops<BLS12_381_G2::F>(op, a, b, r, n); return 0;
#endif
    case ARK_HIP_BLS12_381_G1: host_field_ops<BLS12_381_G1::F>(op, a, b, r, n); return 0;
  }
  return ARK_HIP_ERR_ARG;
}

int ark_hip_test_point_op(int curve, int kind, const uint64_t* acc, const uint64_t* other, uint64_t* out, size_t n) {
  const bool lazy_kind = kind >= 12 && kind <= 15;   // the carry-free forms of kinds 2 / 3 / 4 (testops.cuh)
  if (curve < 0 || curve > 4 || !acc || !out || kind < 2 || (kind > 7 && !lazy_kind)) return ARK_HIP_ERR_ARG;
  size_t fb = (size_t)CURVES[curve].fe_words * 8;
  size_t abytes = n * fb * (kind == 7 ? 2 : 4);
  size_t bbytes = (kind == 2 || kind == 3 || kind == 12 || kind == 13) ? n * fb * 2 : ((kind == 4 || kind >= 14) ? n * fb * 4 : 0);
  size_t rbytes = n * fb * (kind == 6 ? 3 : 4);
  if (bbytes && !other) return ARK_HIP_ERR_ARG;
  return run_elementwise(abytes, bbytes, rbytes, acc, bbytes ? other : nullptr, out, point_op_fn(curve), kind, n);
}

}  // extern "C"
