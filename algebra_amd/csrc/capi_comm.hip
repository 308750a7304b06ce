// C ABI of libark_hip.so, unit 4 of 5: one process per GPU -- RCCL opened at run time, the sharded MSM (part sums
// all-gathered and added on the device) and the one-exchange sharded FFT (see include/ark_hip.h).
#include "capi_core.hpp"
#include "capi_hostmath.hpp"
using namespace arkhip;
using namespace arkhip::capi;

namespace {

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
// Bounded wait for a stream that carries a collective (the exchange has only ever run with a world of one or
// emulated ranks -- a peer that never arrives must cost an error code, not a hang): the stream is polled until
// ARK_HIP_COMM_TIMEOUT_MS (default 120 000; 0 = wait for ever) has passed; then the communicator is aborted (ncclCommAbort,
// which also ends the stuck kernel), dropped from the context, and the call returns ARK_HIP_ERR_COMM.  Every rank that waits
// for the same collective runs the same clock, so all of them leave.
// (comm_sync: declared in capi_core.hpp, defined below)
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
  if (int rc = comm_sync(c, c->stream)) return rc;   // behind the all-gather: bounded
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
  decltype(&ncclCommAbort) CommAbort = nullptr;   // optional: frees a collective that never completes (comm_sync)
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
  a.CommAbort = (decltype(a.CommAbort))dlsym(h, "ncclCommAbort");
  a.tried = true;
  g_rccl = a;
  return &g_rccl;
}
}  // namespace
namespace arkhip {
namespace capi {
int comm_sync(Context* c, hipStream_t st) {
  static const long limit_ms = [] {
    const char* e = getenv("ARK_HIP_COMM_TIMEOUT_MS");
    return e ? atol(e) : 120000L;
  }();
  if (limit_ms <= 0 || !c->comm || c->comm_world <= 1) {
    ARK_HIP_TRY(hipStreamSynchronize(st));
    return 0;
  }
  const auto t0 = std::chrono::steady_clock::now();
  unsigned spins = 0;
  for (;;) {
    const hipError_t e = hipStreamQuery(st);
    if (e == hipSuccess) return 0;
    if (e != hipErrorNotReady) {
      (void)hipGetLastError();
      return -1000 - (int)e;
    }
    (void)hipGetLastError();   // "not ready" is not an error to leave behind
    if ((++spins & 1023u) == 0) {
      const long ms = (long)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
      if (ms > limit_ms) break;
      if (ms > 5) std::this_thread::sleep_for(std::chrono::microseconds(200));   // a long wait: stop burning the core
    }
  }
  fprintf(stderr, "ark_hip: a collective did not complete within %ld ms (ARK_HIP_COMM_TIMEOUT_MS): communicator of rank %d aborted\n",
          limit_ms, c->comm_rank);
  const RcclApi* api = rccl_api();
  if (api && api->CommAbort) (void)api->CommAbort(c->comm);   // ends the stuck kernel; the communicator is gone afterwards
  c->comm = nullptr;
  c->comm_rank = 0;
  c->comm_world = 1;
  return ARK_HIP_ERR_COMM;
}
}  // namespace capi
}  // namespace arkhip
namespace {
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
  if (int rc = comm_sync(c, c->stream)) return rc;   // behind the all-gather: bounded
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

}  // extern "C"

// test support (reached through ark_hip_test_msm_sharded_emulated of capi_test.hip, i.e. libark_hip_test.so only): not exported
namespace arkhip {
namespace capi {
// Test hook: the exchange of msm_sharded_finish with the ranks EMULATED in one process (one GPU): `world` local MSMs run one
// after the other, each block lands where the all-gather would put it, then the same sum kernel / header check / host tail --
// or, when the plans differ, the same fallback (partials summed on the host).  *path: 1 = part sums added on the device,
// 2 = fallback.  Everything but the RCCL call itself.
int msm_sharded_emulated(int curve, int world, const void* const* d_bases, const void* const* d_scalars,
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
}  // namespace capi
}  // namespace arkhip
