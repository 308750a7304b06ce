// A persistent, BOUNDED pool of host helper threads for the short host-side tails of the library (the windows' own sums of
// an MSM's tail, msm.cuh: msm_finish / msm_host_fold).
//
// Rounds 4-5 created std::threads per call (7 per short MSM, up to 7 more per G2 call) and let them spin on an atomic until
// the GPU had finished -- under the reference's callers, which invoke `msm` from rayon pools (variable_base/mod.rs:546-550
// nests them), that is 7 N spinning threads for N concurrent callers.  Here:
//   * ONE pool per process, created on first use (never on a per-call path afterwards), at most ARK_HIP_HOST_TAIL_THREADS
//     helpers (default 7, capped by the cores the process may use minus one; 0: no pool, every tail on its caller's thread);
//   * helpers are PARKED on a condition variable while no batch is open;
//   * a batch is a set of independent tasks behind a gate.  The caller opens the batch BEFORE it waits for the GPU (the
//     helpers' wake-up hides under the kernels), opens the gate when the inputs have landed, and then claims tasks itself
//     from the same counter: it never waits for a helper to START anything -- only for tasks a helper has already claimed,
//     i.e. is running.  Busy helpers (other callers' batches) simply do not show up, and the caller does all of it;
//   * a helper waiting at a closed gate spins (pause + sched_yield, so that a saturated host's own threads get the core) for
//     at most ARK_HIP_HOST_TAIL_SPIN_US (default 1000: the length of the short jobs the early start exists for), then sleeps in
//     SLEEP_US steps: a short job queued behind 100 ms of other device work costs its helpers 1 ms of a core each, not
//     100 ms.
// The thread count of the process is therefore bounded by the pool size whatever the number of concurrent callers.  The pool
// is never destroyed (its threads are detached and end with the process: no join in a static destructor, which would hang in
// a forked child that inherited the bookkeeping but not the threads -- such a child simply gets no help and runs its tails
// on the calling thread).
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>
#if defined(__linux__)
#include <sched.h>
#endif

namespace arkhip {

class HostPool {
 public:
  typedef void (*TaskFn)(void* ctx, int task);
  struct Batch {
    TaskFn fn = nullptr;
    void* ctx = nullptr;
    int ntasks = 0;
    int want = 0;                    // helpers this batch can use
    int helpers = 0;                 // helpers attached (guarded by the pool's mutex)
    std::atomic<int> next{0};        // next unclaimed task
    std::atomic<int> done{0};        // finished tasks
    std::atomic<int> gate{0};        // 0: closed, 1: open, -1: cancelled (no task will run)
  };
  typedef std::shared_ptr<Batch> Handle;

  static constexpr int SLEEP_US = 50;

  // the process-wide pool (one instance per shared object: inline function-local static; leaked on purpose, see above)
  static HostPool& instance() {
    static HostPool* p = new HostPool;
    return *p;
  }
  // helpers the pool has (0: none -- callers run their tails alone)
  int helpers() {
    start();
    return nhelpers_;
  }
  // threads this pool has EVER created (tests: stays at helpers() whatever the number of calls)
  int threads_created() const { return created_.load(); }

  // Open a batch of `ntasks` independent tasks fn(ctx, 0 .. ntasks-1) behind a closed gate.  Returns an empty handle when
  // there is nobody to help (no pool, one task): the caller then simply runs its tasks itself (run_all does).
  Handle open(TaskFn fn, void* ctx, int ntasks) {
    start();
    if (nhelpers_ == 0 || ntasks < 2) return Handle();
    Handle b;
    try {
      b = std::make_shared<Batch>();
    } catch (...) {
      return Handle();
    }
    b->fn = fn;
    b->ctx = ctx;
    b->ntasks = ntasks;
    b->want = ntasks - 1 < nhelpers_ ? ntasks - 1 : nhelpers_;
    {
      std::lock_guard<std::mutex> lk(mu_);
      try {
        open_.push_back(b);
      } catch (...) {
        return Handle();
      }
    }
    cv_.notify_all();
    return b;
  }
  // Open the gate and run the batch to completion with the caller taking part.  `ctx` / `fn` must stay valid until this
  // returns; afterwards no helper touches them (helpers hold the Batch itself by shared_ptr, never the caller's frame).
  void run(const Handle& b) {
    b->gate.store(1, std::memory_order_release);
    drain(*b);
    // tasks a helper claimed before the counter ran out are RUNNING: wait for them (tens of microseconds)
    while (b->done.load(std::memory_order_acquire) < b->ntasks) relax();
    close(b);
  }
  // Open the gate WITHOUT taking part: the helpers work on the batch while the caller does something else (the verified
  // cache's hashing pass under the MSM it validates); the caller joins later with run(), which finishes what is left.
  void start_async(const Handle& b) { b->gate.store(1, std::memory_order_release); }
  // Give the batch up without running anything (error paths).  Tasks are claimed only behind an open gate, so none runs.
  void cancel(const Handle& b) {
    b->gate.store(-1, std::memory_order_release);
    close(b);
  }
  // convenience: tasks with nothing to wait for (the gate opens at once)
  void run_all(TaskFn fn, void* ctx, int ntasks) {
    Handle b = open(fn, ctx, ntasks);
    if (!b) {
      for (int i = 0; i < ntasks; i++) fn(ctx, i);
      return;
    }
    run(b);
  }

 private:
  HostPool() {}
  HostPool(const HostPool&) = delete;
  static void relax() {
#if defined(__x86_64__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    asm volatile("yield");
#endif
  }
  static int usable_cores() {
#if defined(__linux__)
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof(set), &set) == 0) {
      const int n = CPU_COUNT(&set);
      if (n > 0) return n;
    }
#endif
    const int hw = (int)std::thread::hardware_concurrency();
    return hw > 0 ? hw : 1;
  }
  void start() {
    std::call_once(once_, [this]() {
      int n = 7;
      if (const char* e = getenv("ARK_HIP_HOST_TAIL_THREADS")) n = atoi(e);
      if (n > 16) n = 16;
      const int cores = usable_cores();
      if (n > cores - 1) n = cores - 1;   // the caller is a worker too
      if (const char* e = getenv("ARK_HIP_HOST_TAIL_SPIN_US")) spin_us_ = atoi(e) < 0 ? 0 : atoi(e);
      for (int i = 0; i < n; i++) {
        try {
          std::thread([this]() { worker(); }).detach();
          created_.fetch_add(1);
          nhelpers_++;
        } catch (...) {   // no more threads to be had (EAGAIN under a thread / cgroup limit): a smaller pool
          break;
        }
      }
    });
  }
  static void drain(Batch& b) {
    for (;;) {
      const int i = b.next.fetch_add(1, std::memory_order_acq_rel);
      if (i >= b.ntasks) return;
      b.fn(b.ctx, i);
      b.done.fetch_add(1, std::memory_order_acq_rel);
    }
  }
  void close(const Handle& b) {
    std::lock_guard<std::mutex> lk(mu_);
    for (auto it = open_.begin(); it != open_.end(); ++it)
      if (it->get() == b.get()) {
        open_.erase(it);
        break;
      }
  }
  void worker() {
    for (;;) {
      Handle b;
      {
        std::unique_lock<std::mutex> lk(mu_);
        for (;;) {
          for (auto& o : open_)
            if (o->helpers < o->want && o->gate.load(std::memory_order_relaxed) >= 0 &&
                o->next.load(std::memory_order_relaxed) < o->ntasks) {
              b = o;
              break;
            }
          if (b) break;
          cv_.wait(lk);
        }
        b->helpers++;
      }
      // at the gate: a bounded spin, then short sleeps
      int g = b->gate.load(std::memory_order_acquire);
      if (g == 0) {
        const auto t0 = std::chrono::steady_clock::now();
        unsigned it = 0;
        while ((g = b->gate.load(std::memory_order_acquire)) == 0) {
          relax();
          if ((++it & 31u) == 0) {
            if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(spin_us_)) break;
            std::this_thread::yield();
          }
        }
        while ((g = b->gate.load(std::memory_order_acquire)) == 0)
          std::this_thread::sleep_for(std::chrono::microseconds(SLEEP_US));
      }
      if (g > 0) drain(*b);
      // detached: `b` (the Batch, not the caller's frame) dies with the last reference
    }
  }

  std::once_flag once_;
  int nhelpers_ = 0;
  int spin_us_ = 1000;
  std::atomic<int> created_{0};
  std::deque<Handle> open_;
  std::mutex mu_;
  std::condition_variable cv_;
};


// One parked host thread per logical device for the one-process-for-all-GPUs entries (ark_hip_msm_sw_multi & co.): every
// device's share is a BLOCKING call on that device's context, so the shares need a thread each -- persistent ones, created
// when an entry first asks for that many devices (rounds 2-5 created and joined n_gpus threads per call).  Calls of these
// entries serialise on the pool (they would contend for the same devices anyway); share 0 runs on the calling thread.
class DeviceThreads {
 public:
  typedef void (*ShareFn)(void* ctx, int device);
  static DeviceThreads& instance() {
    static DeviceThreads* p = new DeviceThreads;   // leaked on purpose, as HostPool
    return *p;
  }
  // fn(ctx, g) for g = 0 .. n - 1, concurrently; returns when all have returned.  false: the threads could not be had
  // (nothing has run for g >= 1; the caller falls back to running the shares one after another)
  bool run(int n, ShareFn fn, void* ctx) {
    std::lock_guard<std::mutex> call(call_mu_);
    if (n > 1 && !grow(n - 1)) return false;
    {
      std::lock_guard<std::mutex> lk(mu_);
      fn_ = fn;
      ctx_ = ctx;
      pending_ = n - 1;
      for (int g = 1; g < n; g++) slot_[(size_t)(g - 1)] = g;
      epoch_++;
    }
    if (n > 1) cv_.notify_all();
    fn(ctx, 0);
    std::unique_lock<std::mutex> lk(mu_);
    done_cv_.wait(lk, [this]() { return pending_ == 0; });
    return true;
  }

 private:
  DeviceThreads() {}
  bool grow(int want) {
    std::lock_guard<std::mutex> lk(mu_);
    while ((int)slot_.size() < want) {
      const int idx = (int)slot_.size();
      try {
        slot_.push_back(-1);
        std::thread([this, idx]() { worker(idx); }).detach();
      } catch (...) {
        if ((int)slot_.size() > idx) slot_.pop_back();
        return false;
      }
    }
    return true;
  }
  void worker(int idx) {
    unsigned long seen = 0;
    for (;;) {
      int g;
      ShareFn fn;
      void* ctx;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&]() { return epoch_ != seen && slot_[(size_t)idx] >= 0; });
        seen = epoch_;
        g = slot_[(size_t)idx];
        slot_[(size_t)idx] = -1;
        fn = fn_;
        ctx = ctx_;
      }
      fn(ctx, g);
      std::lock_guard<std::mutex> lk(mu_);
      if (--pending_ == 0) done_cv_.notify_all();
    }
  }
  std::mutex call_mu_, mu_;
  std::condition_variable cv_, done_cv_;
  std::vector<int> slot_;   // slot_[i]: the device thread i is to serve in this epoch, -1: none
  unsigned long epoch_ = 0;
  int pending_ = 0;
  ShareFn fn_ = nullptr;
  void* ctx_ = nullptr;
};

}  // namespace arkhip
