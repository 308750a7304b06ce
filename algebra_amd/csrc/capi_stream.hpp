// Streamed MSMs: host scalars (and bases) cut into pieces that upload through a two-slot ring under the previous piece's
// kernels (msm_stream), and the piece-size rule.
#pragma once
#include "capi_core.hpp"
namespace arkhip {
namespace capi {

// One MSM whose scalars (and, with `host_bases`, bases) come from host memory, against `d_bases` (resident; with
// `host_bases` as well: a resident copy being FILLED by this very call, piece by piece) or bases streamed through the ring: the pairs are cut into pieces; piece k+1 uploads on the copy stream -- and digit-recodes / sorts on the other MSM
// lane -- while piece k's accumulate kernel runs.
//   shared (2..8 pieces): the pieces are pieces of ONE MSM -- one plan, one bucket array, one reduction (MsmPiece);
//   otherwise (msm_chunks with its fixed 2^20 steps): independent MSMs whose results are added on the host (the
//   reference's own chunk sum, variable_base/mod.rs:542-557).
inline int msm_stream(Context* c, int curve, const void* d_bases, const uint64_t* host_bases, const uint64_t* scalars, size_t n,
               int mont, size_t step, uint64_t* out_xyz, bool allow_shared = true, bool growing = false,
               bool taper = false) {
  const size_t ab = (size_t)CURVES[curve].fe_words * 16;
  const size_t pw = (size_t)CURVES[curve].fe_words * 3;
  if (step == 0 || step > n) step = n;
  // piece sizes: equal steps, or -- `growing`, scalars-only uploads against resident bases -- each piece twice (from 2^23
  // pairs: three times, below) the one before it: a piece's upload hides under the previous piece's kernels as long as it is less than ~3x as large (the MSM
  // spends 2.1 ns per pair, the PCIe copy 0.64 ns per 32-byte scalar), so the one upload nothing hides shrinks to
  // n / (2^P - 1) pairs and the per-piece costs (launches, the read-modify-write of every bucket) are paid P <= 6 times
  std::vector<size_t> sizes;
  auto parse_weights = [](const char* e) {   // "a,b,c,..": positive weights; anything else: empty (the variable is ignored)
    std::vector<size_t> wts;
    for (const char* q = e; q && *q;) {
      char* end = nullptr;
      const size_t v = (size_t)strtoul(q, &end, 10);
      if (end == q) {   // not a number: strtoul consumed nothing (it used to loop here for ever)
        wts.clear();
        break;
      }
      if (v) wts.push_back(v);
      q = end;
      if (*q == ',') q++;
    }
    return wts;
  };
  auto by_weights = [&](const std::vector<size_t>& wts) {
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
  };
  const std::vector<size_t> grow_wts = growing ? parse_weights(getenv("ARK_HIP_STREAM_GROW_SCHEDULE")) : std::vector<size_t>();
  if (growing && n >= ((size_t)3 << 18) && grow_wts.size() >= 2 && grow_wts.size() <= 16) {
    by_weights(grow_wts);   // experiments: the growing pieces' weights, e.g. "1,3,9,27"
  } else if (growing && n >= ((size_t)1 << 23)) {
    // from 2^23 pairs: each piece THREE times the one before it, at most four (round 6, profiles/r6_growing_pieces.txt): a
    // piece's upload still hides under the previous piece's kernels (0.54 ns of PCIe per scalar against 2.1 ns of MSM per pair
    // allow a factor of ~3.9), and four pieces pay the per-piece costs -- ~30 launches, sparse runs in the early pieces, the
    // read-modify-write of every bucket -- four times instead of six: 2^24 40.4 -> 39.25 ms, 2^23 21.4 -> 21.05, 2^25 72.8 -> 71.5,
    // 2^26 140.0 -> 137.8 (2^22 and below: the doubling pieces remain the faster ones)
    int P = 1;
    size_t geo = 1, pw3 = 1;   // geo = 1 + 3 + .. + 3^(P-1)
    while (P < 4 && n / (geo + pw3 * 3) >= ((size_t)1 << 18)) {
      pw3 *= 3;
      geo += pw3;
      P++;
    }
    std::vector<size_t> wts;
    for (size_t k = 0, w = 1; k < (size_t)P; k++, w *= 3) wts.push_back(w);
    by_weights(wts);
  } else if (growing && n >= ((size_t)3 << 18)) {
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
    std::vector<size_t> wts = parse_weights(getenv("ARK_HIP_STREAM_SCHEDULE"));
    if (wts.empty()) wts.assign(8, 1);
    by_weights(wts);
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
    // a lane with a free job slot -- before anything of this piece is put in flight.  All four taken: first this call's own
    // oldest piece is finished (its slot is ours to free); then other threads' jobs are given a bounded time to be waited for
    // (ark_hip_msm_wait takes no context lock, so they can be, although this body holds it); then BUSY, which leaves nothing
    // behind (the synchronous entries retry the whole call: capi_msm.hip retry_while_busy)
    int lane = msm_pick_lane(c);
    if (lane == ARK_HIP_ERR_BUSY && npend) {
      if (int rc = drain_one()) return fail(rc);
      lane = msm_pick_lane(c);
    }
    for (int tries = 0; lane == ARK_HIP_ERR_BUSY && tries < 25; tries++) {   // <= 25 x 20 ms
      {
        std::unique_lock<std::mutex> lk(c->slot_mu);
        const uint64_t gen = c->slot_gen.load(std::memory_order_acquire);
        c->slot_cv.wait_for(lk, std::chrono::milliseconds(20), [&]() { return c->slot_gen.load(std::memory_order_acquire) != gen; });
      }
      lane = msm_pick_lane(c);
    }
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
inline size_t msm_stream_step(size_t n) {
  size_t pieces = n >> 18;
  if (pieces < 1) pieces = 1;
  if (pieces > 8) pieces = 8;
  if (const char* e = getenv("ARK_HIP_STREAM_PIECES")) {
    const int v = atoi(e);
    if (v >= 1 && v <= 64) pieces = (size_t)v;
  }
  return (n + pieces - 1) / pieces;
}


}  // namespace capi
}  // namespace arkhip
