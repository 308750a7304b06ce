/*
 * ark_hip.h -- C ABI of libark_hip.so: MI355X (gfx950) implementation of the one data-parallel hot
 * path of arkworks-rs/algebra:
 *     ark_ec::VariableBaseMSM::msm over short-Weierstrass G1/G2   (Pippenger bucket method)
 *     ark_poly::Radix2EvaluationDomain::{fft,ifft}_in_place over Fr (Cooley-Tukey radix-2)
 * Every entry point below is what the reference's Rust side would bind through FFI; the
 * reference item each one replaces is cited (paths relative to the arkworks-rs/algebra tree).
 * INTEGRATION.md shows the Rust `extern "C"` block and the trait impls that call them.
 *
 * Data layout == the reference's in-memory layout, no conversion on either side:
 *   Fp           N little-endian u64 limbs, Montgomery form, value < p   (ff/src/biginteger/mod.rs:34,
 *                                                                          ff/src/fields/models/fp/mod.rs:109-115)
 *   Fp2          c0 | c1                                                  (quadratic_extension.rs:100-106)
 *   Affine       x | y, identity = all-zero (ZeroFlag = ())               (short_weierstrass/affine.rs:30-37,91-104)
 *   Projective   x | y | z Jacobian, identity = (R, R, 0)                 (short_weierstrass/group.rs:34-41,145-151)
 *   scalar       4 u64 limbs: BigInt<4> canonical, or Fr Montgomery
 * All functions return 0 on success and a negative code on error; none throws or aborts.
 * Devices and threads: the library keeps one context (streams, workspaces) per GPU.  A call runs on the
 * calling thread's current device (ark_hip_set_device; default: the first device initialised in the process)
 * and holds that context's lock for its whole body, so any number of host threads (rayon workers) may call
 * concurrently: calls on one GPU serialise, calls on different GPUs overlap.  One process per GPU
 * (torch.distributed / RCCL) and one process driving all GPUs (ark_hip_msm_sw_multi) are both supported.
 * Limits: MSM n < 2^31 and n * windows < 2^32 (n <= 2^28 for 255-bit scalars; 2^28 is tested); FFT log2(size) <= 30.
 * Environment: ARK_HIP_WAIT=block makes an MSM wait for the GPU with a blocking hipEventSynchronize; by default the
 * calling thread polls the completion event (the MSM is on its caller's critical path; a sleeping thread was measured
 * to add up to 1 ms per call on some hosts).  ARK_HIP_MSM_C / ARK_HIP_MSM_C_PREPARED force the window size (tuning).
 * ARK_HIP_HOST_TAIL_THREADS=k (default 7; 0: none): size of the ONE process-wide pool of parked helper threads that shares
 * the short host-side tails with the calling thread -- the windows' own sums of a short MSM's tail, the verified cache's hashing
 * pass.  Created on first use, never per call; see ark_hip_host_threads.
 * ARK_HIP_COPY_THREADS=n (default 0 = the HIP runtime's own pageable path): n worker threads stage uploads from ordinary
 * host memory through page-locked buffers.  ARK_HIP_STREAM_PIECES: pieces a host-scalar MSM is cut into (default
 * n / 2^18, at most 8).  ARK_HIP_BASE_CACHE_MB / ARK_HIP_AUTO_PREPARE: see ark_hip_msm_cache_config.
 */
#ifndef ARK_HIP_H
#define ARK_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden: exactly the functions declared between this push and the pop below are its
 * dynamic symbols. */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

/* field ids */
enum { ARK_HIP_BN254_FQ = 0, ARK_HIP_BN254_FR = 1, ARK_HIP_BLS12_381_FQ = 2, ARK_HIP_BLS12_381_FR = 3,
       ARK_HIP_BLS12_377_FQ = 4, ARK_HIP_BLS12_377_FR = 5 };
/* curve ids (SWCurveConfig instances) */
enum { ARK_HIP_BN254_G1 = 0, ARK_HIP_BLS12_381_G1 = 1, ARK_HIP_BLS12_377_G1 = 2, ARK_HIP_BLS12_377_G2 = 3,
       ARK_HIP_BLS12_381_G2 = 4 };
/* error codes */
enum { ARK_HIP_OK = 0, ARK_HIP_ERR_ARG = -1, ARK_HIP_ERR_SIZE = -2, ARK_HIP_ERR_NOMEM = -3,
       ARK_HIP_ERR_SCALAR_RANGE = -4, ARK_HIP_ERR_NO_DEVICE = -5, ARK_HIP_ERR_BUSY = -6,
       ARK_HIP_ERR_COMM = -7 /* RCCL missing or failed */ /* HIP runtime errors: <= -1000 */ };

/* ---- runtime ---- */
int ark_hip_device_count(void);
/* Make GPU `device` the calling thread's current device and create its context (streams, workspaces) if needed.
 * The first device initialised becomes the default of threads that never choose one.  Idempotent.
 * (ARK_HIP_OVERSUBSCRIBE=1 in the environment lets ids beyond the physical count wrap around -- separate contexts
 * on a shared GPU -- so that multi-device code can be exercised on a one-GPU box.) */
int ark_hip_init(int device);
int ark_hip_set_device(int device);   /* same as ark_hip_init */
int ark_hip_get_device(void);
/* Destroys every context (waits for calls in flight). */
void ark_hip_shutdown(void);
/* Waits for everything enqueued on the current device's streams. */
int ark_hip_synchronize(void);
const char* ark_hip_version(void);
/* Host threads the library keeps: out[0] = helper threads of the process-wide pool that shares the short host-side tails
 * (the windows' sums of an MSM's tail, the verified cache's hashing pass) with the calling thread -- created once, on first
 * use, parked while idle; ARK_HIP_HOST_TAIL_THREADS (default 7, 0: none) and the cores the process may use bound it;
 * out[1] = threads that pool has created since the process started (stays at out[0]: no thread is created per call).
 * Callable without a GPU. */
int ark_hip_host_threads(int out[2]);
/* u64 words per base-field element (4, 6 or 12), scalar field id, base field id, extension degree */
int ark_hip_curve_info(int curve, int* fe_words, int* scalar_field, int* base_field, int* ext_degree);

/* Device memory for hosts that have no HIP binding of their own: upload a fixed base set (an SRS) once with
 * ark_hip_malloc + ark_hip_memcpy_h2d and pass the device pointer to ark_hip_msm_sw_device for every proof. */
int ark_hip_malloc(size_t bytes, void** out_dptr);
int ark_hip_free(void* dptr);
int ark_hip_memcpy_h2d(void* dst_dptr, const void* src_host, size_t bytes);
int ark_hip_memcpy_d2h(void* dst_host, const void* src_dptr, size_t bytes);
/* device-to-device copy and byte fill, asynchronous on the context stream (ordered with every *_device entry) */
int ark_hip_memcpy_d2d(void* dst_dptr, const void* src_dptr, size_t bytes);
int ark_hip_memset_device(void* dptr, int value, size_t bytes);
/* Page-locked host memory: scalar vectors produced into it upload at full PCIe rate and truly asynchronously
 * (ark_hip_msm_prepared_async). */
int ark_hip_host_alloc(size_t bytes, void** out_ptr);
int ark_hip_host_free(void* ptr);

/* SWCurveConfig::GENERATOR (e.g. curves/bls12_381/src/curves/g1.rs:199-205) as Affine limbs */
int ark_hip_curve_generator(int curve, uint64_t* out_xy);

/* ---- MSM ----
 * Replaces SWCurveConfig::msm (ec/src/models/short_weierstrass/mod.rs:112-119) -> 
 * VariableBaseMSM::msm_unchecked / msm_bigint (ec/src/scalar_mul/variable_base/mod.rs:59-85).
 * bases: n Affine points; scalars: n x 4 limbs; scalars_are_montgomery != 0 for the `msm(&[Fr])`
 * entry (the into_bigint pass of mod.rs:60-62 then runs on the device), 0 for `msm_bigint`.
 * The length check (Err(min_len), mod.rs:73-77) stays on the caller's side: one n here.
 * out_xyz: Projective.  Bases must lie in the prime-order subgroup (as Affine deserialisation guarantees; the
 * reference's own msm_signed relies on r*P = O as well, mod.rs:251-285).  Scalars: any BigInt<4> below
 * 2^MODULUS_BIT_SIZE is accepted and gives the reference's result (values in [r, 2^bits) are reduced once);
 * a scalar with higher bits set returns ARK_HIP_ERR_SCALAR_RANGE -- the reference's make_digits (mod.rs:754-794)
 * silently drops bits above ceil(bits/c)*c there, i.e. its result depends on the window size it happened to pick. */
int ark_hip_msm_sw(int curve, const uint64_t* bases, const uint64_t* scalars, size_t n, int scalars_are_montgomery,
                   uint64_t* out_xyz);
/* ark_hip_msm_sw is a FUNCTION OF ITS TWO SLICES, like the reference (`bases: &[Affine]` is borrowed for the call,
 * variable_base/mod.rs:59-85): whatever the library keeps between calls never changes a result.  Provers hand the SAME
 * base slice -- an SRS -- to call after call (bench-templates/src/macros/ec.rs:223-240 does exactly that), so device
 * copies of base slices are kept in two ways:
 *
 * Verified cache (on by default).  Device copies keyed by (curve, host address, length), validated on EVERY call by a
 * keyed 128-bit hash of the slice's FULL content: host threads (ARK_HIP_HASH_THREADS, default 8) hash the slice while the device already
 * works from the cached copy, and the result is returned only if the hash equals that of the content the copy was filled
 * from -- otherwise the copy is refreshed and the MSM rerun.  An in-place edit of the slice, of a single limb, is therefore
 * honoured on the next call; a hit costs one pass over the host slice instead of its PCIe transfer (2^24 BLS12-381 G1:
 * 40 ms per call against 48 ms streamed).  A slice that does not fit the budget streams over PCIe with its scalars, piece
 * k+1 under piece k's kernels.  Least-recently-used sets (with their tables) are dropped beyond the budget.
 *   ark_hip_msm_cache_config(budget_bytes, auto_prepare_after): budget -1 keeps the current value, -2 restores the
 *     default (16 GiB, at most a quarter of the device memory; ARK_HIP_BASE_CACHE_MB raises or lowers it), 0 disables and empties the cache (every call
 *     then streams bases and scalars and nothing is retained); auto_prepare_after = K > 0 builds the per-window table of a
 *     resident set -- pinned or cached -- once the WHOLE set has been the operand K times (default 0 = never, or
 *     ARK_HIP_AUTO_PREPARE; a cached set's table counts against the budget), < 0 keeps the current value.
 *   ark_hip_msm_cache_clear: drops the cached sets (pinned sets stay).
 *   ark_hip_msm_cache_stats: [cached sets, their device bytes, hits, misses, refreshed (content changed), evicted,
 *     pinned sets, pinned hits].
 *
 * Pinned base sets (no host pass at all).  ark_hip_msm_bases_pin(curve, bases, n) uploads the set now and declares
 * bases[0 .. n) IMMUTABLE until the matching ark_hip_msm_bases_unpin(curve, bases, n).  Every ark_hip_msm_sw /
 * ark_hip_msm_sw_small call on this device whose base slice lies inside a pinned range (at a point boundary: the whole set,
 * msm_unchecked's truncation, the steps of msm_chunks / ChunkedPippenger) runs against the resident copy and uploads only
 * its scalars -- no validation, the declaration is the contract (the Rust guard ark_hip::msm::ResidentBases holds the
 * shared borrow for the pin's lifetime, so the compiler enforces it).  Pins of the same (curve, address, n) nest; pinned
 * sets are outside the cache's budget.  ARK_HIP_ERR_NOMEM if the copy does not fit, ARK_HIP_ERR_ARG for an unpin without
 * a pin. */
int ark_hip_msm_bases_pin(int curve, const uint64_t* bases, size_t n);
int ark_hip_msm_bases_unpin(int curve, const uint64_t* bases, size_t n);
int ark_hip_msm_cache_config(long long budget_bytes, int auto_prepare_after);
int ark_hip_msm_cache_clear(void);
int ark_hip_msm_cache_stats(uint64_t out[8]);
/* The validation pass of the verified cache (round 5): a keyed 128-bit universal hash (NH with 64-bit words under a key
 * drawn from the OS once per process -- two different slices of one length collide with probability <= 2^-63 over the key,
 * and nothing in the source predicts it).  The pass is timed on every call; while its predicted duration exceeds what
 * streaming the slice over PCIe took when the copy was filled (a host whose cores are saturated by the caller's own thread
 * pool), calls stream instead and every eighth call hashes again.  out: [0] calls streamed for that reason, [1] the
 * latest pass in microseconds, [2] its smoothed rate in MB/s, [3] host threads per pass (ARK_HIP_HASH_THREADS). */
int ark_hip_msm_cache_hash_stats(uint64_t out[4]);
/* VariableBaseMSM::msm_u1 / msm_u8 / msm_u16 / msm_u32 / msm_u64 (variable_base/mod.rs:87-117; CPU bodies msm_binary /
 * msm_u8.. :373-434): scalars are n unsigned integers of scalar_bytes (1, 2, 4 or 8) bytes -- exactly the reference's
 * &[bool] (one byte each, max_bits = 1), &[u8], &[u16], &[u32], &[u64] -- whose low max_bits bits may be set (0 = all
 * 8 * scalar_bytes).  Only the ceil((max_bits + 1) / c) windows such scalars reach are built: nothing is widened to 32
 * bytes (a u8 vector uploads 1/32 of what msm_bigint would) and no empty window is sorted.  The host-pointer form shares
 * ark_hip_msm_sw's
 * pinned sets and verified cache (on by default). */
int ark_hip_msm_sw_small(int curve, const uint64_t* bases, const void* scalars, size_t n, int scalar_bytes, int max_bits,
                         uint64_t* out_xyz);
int ark_hip_msm_sw_small_device(int curve, const void* d_bases, const void* d_scalars, size_t n, int scalar_bytes,
                                int max_bits, uint64_t* out_xyz);
/* Same with bases/scalars already in this GPU's memory (device pointers); out_xyz is a host pointer.
 * Scalar widths: msm_signed (variable_base/mod.rs:242-347) sorts the scalars into width classes (+-u1, +-u8, +-u16,
 * +-u32, +-u64, the rest) and runs one MSM per class.  This entry runs ONE pipeline and plans it from the same classes:
 * from 2^19 pairs, with nothing else of the context in flight, a probe measures the bit length of min(s, r - s) over a
 * sample of 4096 scalars and -- unless they look uniform -- over all of them (one 40-byte read-back), and the window size and
 * the number of windows follow the digits the scalars really have.  The result is the same group element either way;
 * ARK_HIP_MSM_PROBE=0 plans every call for n uniform full-width scalars.  A prepared base set whose scalars turn out no wider
 * than 48 bits runs this plain pipeline on row 0 of its table (the bases themselves).  The host-pointer entry above estimates the
 * classes from about 1024 of the host scalars and takes only the window size from them. */
int ark_hip_msm_sw_device(int curve, const void* d_bases, const void* d_scalars, size_t n, int scalars_are_montgomery,
                          uint64_t* out_xyz);
/* Asynchronous form: the device work is enqueued and the call returns; ark_hip_msm_wait blocks until the result is
 * there, finishes it (window combine, a few hundred point operations on the host) and frees the job.  Up to 4 jobs
 * may be in flight per device: an *_async entry reports ARK_HIP_ERR_BUSY beyond that, while the SYNCHRONOUS entries
 * (ark_hip_msm_sw, _sw_device, _sw_small(_device), _prepared(_device), _prepared_small_device, _sw_chunks) wait for a slot
 * -- they may be called from any number of host threads at once (the reference's callers are rayon pools) and never
 * return BUSY.  Inputs must stay valid until the wait returns.
 * With a job already in flight the next one runs on the device's second MSM lane (own stream and workspace): its
 * digits / sort / reduction overlap the first job's accumulate kernel (two jobs in flight: +20 % MSMs/s at 2^20, +5 % at
 * 2^24).  Synchronous calls from one thread never use (or allocate) the second lane.
 * The *_async entries never wait for device work already queued (a transform producing the scalars, an earlier job): the
 * width probe of the synchronous entries (one 40-byte read-back that plans narrow / skewed scalar vectors) is skipped
 * here, and the plan is the one for n uniform full-width scalars -- correct for any input, optimal for those. */
typedef struct ark_hip_msm_job ark_hip_msm_job;
int ark_hip_msm_sw_device_async(int curve, const void* d_bases, const void* d_scalars, size_t n,
                                int scalars_are_montgomery, ark_hip_msm_job** out_job);
int ark_hip_msm_wait(ark_hip_msm_job* job, uint64_t* out_xyz);

/* ---- prepared base sets ----
 * Provers run many MSMs against ONE fixed base set (an SRS).  ark_hip_msm_bases_prepare uploads it once and
 * precomputes, for every window w of the digit decomposition, the multiples 2^(offset_w) * P_i (288 GB of HBM pay for
 * the W-fold table: 18 GiB for 2^24 BLS12-381 G1 points).  All windows then share one bucket set, which allows wider
 * windows -- fewer mixed additions per scalar -- and shrinks the bucket reduction W-fold.  Role in the reference: the
 * `bases: &[Affine]` argument of VariableBaseMSM::msm (variable_base/mod.rs:59-85) held by the caller across calls;
 * the precomputation itself is the fixed-base idea of BatchMulPreprocessing (scalar_mul/mod.rs:156-245) applied per base.
 * ark_hip_msm_prepared* compute sum_i s_i * P_i over the first n <= n_bases pairs (msm_unchecked's truncation). */
typedef struct ark_hip_msm_bases ark_hip_msm_bases;
int ark_hip_msm_bases_prepare(int curve, const uint64_t* bases, size_t n, ark_hip_msm_bases** out);
int ark_hip_msm_bases_prepare_device(int curve, const void* d_bases, size_t n, ark_hip_msm_bases** out);
int ark_hip_msm_bases_free(ark_hip_msm_bases* bases);
int ark_hip_msm_bases_info(const ark_hip_msm_bases* bases, size_t* n, int* window_bits, int* windows, size_t* table_bytes);
int ark_hip_msm_prepared(const ark_hip_msm_bases* bases, const uint64_t* scalars, size_t n, int scalars_are_montgomery,
                         uint64_t* out_xyz);
int ark_hip_msm_prepared_device(const ark_hip_msm_bases* bases, const void* d_scalars, size_t n,
                                int scalars_are_montgomery, uint64_t* out_xyz);
/* Narrow scalars (ark_hip_msm_sw_small's scalar_bytes / max_bits) against a prepared base set: runs as a plain MSM over
 * the base set itself (row 0 of the table) with a window plan for the narrow scalars. */
int ark_hip_msm_prepared_small_device(const ark_hip_msm_bases* bases, const void* d_scalars, size_t n, int scalar_bytes,
                                      int max_bits, uint64_t* out_xyz);
/* Asynchronous forms.  The host-scalar one uploads through a two-slot ring on a copy stream: the upload of MSM k+1's
 * scalars overlaps MSM k's kernels. */
int ark_hip_msm_prepared_async(const ark_hip_msm_bases* bases, const uint64_t* scalars, size_t n,
                               int scalars_are_montgomery, ark_hip_msm_job** out_job);
int ark_hip_msm_prepared_device_async(const ark_hip_msm_bases* bases, const void* d_scalars, size_t n,
                                      int scalars_are_montgomery, ark_hip_msm_job** out_job);

/* VariableBaseMSM::msm_chunks (variable_base/mod.rs:119-150): Fr (Montgomery) scalars; the streams are aligned at
 * their end (the first n_bases - n_scalars bases are skipped); steps of `step` pairs (0 = the reference's 2^20), each
 * one msm_bigint, results added.  Step k+1 uploads while step k computes. */
int ark_hip_msm_sw_chunks(int curve, const uint64_t* bases, size_t n_bases, const uint64_t* scalars, size_t n_scalars,
                          size_t step, uint64_t* out_xyz);

/* One MSM across n_gpus GPUs driven from this process: base-range shards (the reference's own split,
 * variable_base/mod.rs:521-557), one host thread + context per device, partial results summed on the host.
 * _multi: host inputs, split evenly; _multi_device: shard g already resident on GPU g (n_per_gpu[g] pairs). */
int ark_hip_msm_sw_multi(int curve, int n_gpus, const uint64_t* bases, const uint64_t* scalars, size_t n,
                         int scalars_are_montgomery, uint64_t* out_xyz);
int ark_hip_msm_sw_multi_device(int curve, int n_gpus, const void* const* d_bases, const void* const* d_scalars,
                                const size_t* n_per_gpu, int scalars_are_montgomery, uint64_t* out_xyz);
/* The same over PREPARED shards: shards[g] was prepared on GPU g (ark_hip_set_device(g); ark_hip_msm_bases_prepare over
 * the g-th base range); the scalars are host memory, cut at the shards' sizes in order, uploaded and run on all devices
 * concurrently (the asynchronous prepared entry per device), partial results summed on the host. */
int ark_hip_msm_prepared_multi(int n_gpus, const ark_hip_msm_bases* const* shards, const uint64_t* scalars, size_t n,
                               int scalars_are_montgomery, uint64_t* out_xyz);

/* The window plan (widest window in bits, number of windows) the library picks for an MSM of n pairs on `curve`, plain
 * (prepared = 0) or over a prepared base set; pure host arithmetic. */
int ark_hip_msm_plan(int curve, size_t n, int prepared, int* window_bits, int* windows);
/* The plan of a plain msm_bigint whose n scalars fall into the given width classes of b = bits(min(s, r - s)) -- counts[k]
 * scalars with TOP[k-1] < b <= TOP[k], TOP = {0, 1, 8, 16, 32, 64, 128, 192, 256} (msm_signed's partition, mod.rs:251-285,
 * with two classes above u64) -- and whose widest has max_bits: what ark_hip_msm_sw_device does after its probe.  Host
 * arithmetic only. */
int ark_hip_msm_plan_widths(int curve, size_t n, uint32_t max_bits, const uint32_t counts[9], int* window_bits, int* windows);
/* Per-phase device times of the last MSM finished on this device with timing enabled (ms):
 * [digits, partition histogram + scan, partition scatter + finish + bucket order, accumulate (incl. heavy
 *  buckets), reduce, total, window_bits, windows] */
int ark_hip_msm_set_timing(int enable);
int ark_hip_msm_last_timing(double out[8]);

/* ---- fixed-base batch multiplication ----
 * ScalarMul::batch_mul / BatchMulPreprocessing (ec/src/scalar_mul/mod.rs:104-251): out[i] = v[i] * g for ONE group element
 * g, results affine (the reference's Vec<MulBase>).  table_new = BatchMulPreprocessing::new(base, num_scalars): builds the
 * table of multiples on the device (base_xyz: Projective; num_scalars sizes the table as in the reference, :222-228, by
 * the device's own cost rule: 12-bit rows, 16-bit rows from 2^24 scalars -- the window never changes a result);
 * batch_mul = BatchMulPreprocessing::batch_mul.  Scalars: n x 4 limbs, Fr (Montgomery) when
 * scalars_are_montgomery != 0 as in the reference, or canonical BigInt<4> (any 256-bit value is multiplied exactly). */
typedef struct ark_hip_batch_mul_table ark_hip_batch_mul_table;
int ark_hip_batch_mul_table_new(int curve, const uint64_t* base_xyz, size_t num_scalars, ark_hip_batch_mul_table** out);
int ark_hip_batch_mul_table_free(ark_hip_batch_mul_table* table);
int ark_hip_batch_mul(const ark_hip_batch_mul_table* table, const uint64_t* scalars, size_t n, int scalars_are_montgomery,
                      uint64_t* out_xy);
int ark_hip_batch_mul_device(const ark_hip_batch_mul_table* table, const void* d_scalars, size_t n,
                             int scalars_are_montgomery, void* d_out_xy);

/* ---- host-side group helpers (run on the CPU; tiny) ----
 * Sum of n Projective points: the multi-GPU combine of per-rank partial MSMs (the reference's
 * chunk sum, ec/src/scalar_mul/variable_base/mod.rs:542-557).  The reduction operator is
 * elliptic-curve addition, so it cannot be an RCCL sum: ranks all-gather 3 field elements each. */
int ark_hip_sw_sum(int curve, const uint64_t* jac_points, size_t n, uint64_t* out_xyz);
/* From<Projective> for Affine (ec/src/models/short_weierstrass/affine.rs:374-396): the unique
 * representative the reference's assert_eq! compares; identity -> (0, 0). */
int ark_hip_sw_into_affine(int curve, const uint64_t* jac_points, size_t n, uint64_t* out_xy);

/* out[i] = in[i] + delta for n affine points in device memory (affine result, one inversion per
 * point; d_in may equal d_out).  Used to grow synthetic base sets P_i = (a + i*b)G on the device
 * (bench.py / tests); mathematically Affine + Affine -> into_affine (group.rs:332-413, affine.rs:374-396). */
int ark_hip_sw_add_affine_device(int curve, const void* d_in, void* d_out, size_t n, const uint64_t* delta_xy);

/* CurveGroup::normalize_batch (ec/src/models/short_weierstrass/group.rs:302-319) for n Projective points in
 * device memory -> n Affine points in device memory; identity -> (0, 0). */
int ark_hip_sw_normalize_batch_device(int curve, const void* d_jac, void* d_out_xy, size_t n);
/* the same from host memory: n Projective in, n Affine out (the Rust hook behind CurveGroup::normalize_batch) */
int ark_hip_sw_normalize_batch(int curve, const uint64_t* jac_points, size_t n, uint64_t* out_xy);

/* ---- Radix-2 evaluation domain ----
 * Mirror of Radix2EvaluationDomain<F>'s public fields (poly/src/domain/radix2/mod.rs:22-42). */
typedef struct {
  uint64_t size;
  uint32_t log_size_of_group;
  uint32_t _pad;
  uint64_t size_as_field_element[4];
  uint64_t size_inv[4];
  uint64_t group_gen[4];
  uint64_t group_gen_inv[4];
  uint64_t offset[4];
  uint64_t offset_inv[4];
  uint64_t offset_pow_size[4];
} ark_hip_radix2_domain;

/* Radix2EvaluationDomain::new (radix2/mod.rs:55-83): size = num_coeffs.next_power_of_two();
 * returns ARK_HIP_ERR_SIZE where the reference returns None (log size > TWO_ADICITY). */
int ark_hip_radix2_domain_new(int field, size_t num_coeffs, ark_hip_radix2_domain* out);
/* get_coset (radix2/mod.rs:85-92); ARK_HIP_ERR_ARG if offset == 0 (reference: None) */
int ark_hip_radix2_domain_get_coset(int field, const ark_hip_radix2_domain* dom, const uint64_t* offset,
                                    ark_hip_radix2_domain* out);
/* fft_in_place / ifft_in_place (radix2/mod.rs:140-153) on exactly dom->size elements (the caller
 * has already done the reference's resize(size, zero)); natural order in and out. */
int ark_hip_fft_in_place(int field, const ark_hip_radix2_domain* dom, uint64_t* data);
int ark_hip_ifft_in_place(int field, const ark_hip_radix2_domain* dom, uint64_t* data);
/* Same on device memory; asynchronous on the context stream (ark_hip_synchronize() to wait). */
int ark_hip_fft_in_place_device(int field, const ark_hip_radix2_domain* dom, void* d_data);
int ark_hip_ifft_in_place_device(int field, const ark_hip_radix2_domain* dom, void* d_data);
/* fft_in_place on a coefficient vector of num_coeffs <= size elements (radix2/mod.rs:140-147): the buffer holds
 * dom->size elements, only the first num_coeffs are read (the rest is treated as zero, as the reference's resize
 * does).  With num_coeffs * 4 <= size this is the degree-aware path (fft.rs:29-71): the first
 * log2(size / next_pow2(num_coeffs)) butterfly stages are skipped and only the coefficients are uploaded. */
int ark_hip_fft_in_place_degree_aware(int field, const ark_hip_radix2_domain* dom, uint64_t* data, size_t num_coeffs);
int ark_hip_fft_in_place_degree_aware_device(int field, const ark_hip_radix2_domain* dom, void* d_data, size_t num_coeffs);
/* `count` independent transforms over one domain (forward, or inverse != 0), each in place on its own device buffer of
 * dom->size elements -- the several polynomials a prover transforms at once.  Up to three run concurrently on the GPU,
 * which fills the vector-ALU slots a single transform leaves idle around its pass boundaries: per transform 2^22
 * 0.53 -> 0.48 ms, 2^20 0.157 -> 0.108 ms, 2^16 44 -> 18 us.  Asynchronous on the context stream like the single call.
 * No counterpart in the reference (its callers loop over fft_in_place, poly/src/domain/mod.rs:92-112). */
int ark_hip_fft_batch_in_place_device(int field, const ark_hip_radix2_domain* dom, void* const* d_data, size_t count,
                                      int inverse);
/* r[i] = a[i] * b[i] over n Fr elements in device memory: `Evaluations *= &Evaluations`
 * (poly/src/evaluations/univariate/mod.rs), the pointwise step between the two FFTs and the IFFT of
 * DensePolynomial multiplication (poly/src/polynomial/univariate/dense.rs:641-656).  Asynchronous. */
int ark_hip_fr_mul_device(int field, const void* d_a, const void* d_b, void* d_r, size_t n);
/* The transform over GROUP elements (round 5): EvaluationDomain::fft_in_place / ifft_in_place for T = Projective<P>
 * (poly/src/domain/mod.rs:332-362, radix2/fft.rs:74-119; the reference's own use: poly/src/test.rs:57 with G1Projective).
 * jac_points: dom->size Jacobian points (x | y | z, the reference's Projective) of `curve`, whose scalar field must be the
 * field the domain was made for; the caller pads with identities (z = 0) to the domain size, as the reference's resize does.
 * Forward: X_j = sum_i [(h g^j)^i] P_i; inverse: P_i = [n^-1 h^-i] sum_j [g^-ij] X_j.  Results are group elements: the
 * Projective representatives differ from the reference's, into_affine() agrees.  The device form is asynchronous. */
int ark_hip_fft_group_in_place(int curve, const ark_hip_radix2_domain* dom, uint64_t* jac_points, int inverse);
int ark_hip_fft_group_in_place_device(int curve, const ark_hip_radix2_domain* dom, void* d_jac_points, int inverse);
/* The rest of the pointwise algebra of device-resident Fr vectors (round 5): Evaluations += / -= / negation
 * (poly/src/evaluations/univariate/mod.rs:104-180) and a vector times ONE field element (dense.rs:604-622; k is a host
 * pointer to one Montgomery element, read before the call returns).  With the transforms above, a chain
 * evaluate_over_domain -> pointwise -> interpolate (polynomial/univariate/mod.rs:305-360, evaluations/univariate/
 * mod.rs:40-50) runs between ONE upload and ONE download.  Asynchronous on the context stream; r may alias a or b. */
int ark_hip_fr_add_device(int field, const void* d_a, const void* d_b, void* d_r, size_t n);
int ark_hip_fr_sub_device(int field, const void* d_a, const void* d_b, void* d_r, size_t n);
int ark_hip_fr_neg_device(int field, const void* d_a, void* d_r, size_t n);
int ark_hip_fr_scale_device(int field, const void* d_a, const uint64_t* k, void* d_r, size_t n);
/* r[i] = a[i] / b[i] (Evaluations /= Evaluations, mod.rs:142-163) and r[i] = 1 / a[i] (ark_ff::batch_inversion,
 * ff/src/fields/mod.rs:358-385): a zero divisor gives zero, as the reference's batch inversion leaves zeros in place. */
int ark_hip_fr_div_device(int field, const void* d_a, const void* d_b, void* d_r, size_t n);
int ark_hip_fr_inverse_device(int field, const void* d_a, void* d_r, size_t n);
/* `&DensePolynomial * &DensePolynomial` (poly/src/polynomial/univariate/dense.rs:641-656: evaluate both factors over the
 * radix-2 domain of size >= na + nb - 1, multiply pointwise, interpolate) from HOST coefficient vectors: ONE upload of a and
 * b, both forward transforms in flight together (short factors on the degree-aware path), the pointwise product and the
 * inverse transform on the device, ONE download.  out: room for na + nb - 1 elements; *out_len: coefficients of the product
 * with leading zeros dropped as DensePolynomial::from_coefficients_vec leaves them (0: a factor was the zero polynomial).
 * ARK_HIP_ERR_ARG when the field's two-adicity cannot hold the domain (the reference panics there). */
int ark_hip_poly_mul(int field, const uint64_t* a, size_t na, const uint64_t* b, size_t nb, uint64_t* out, size_t* out_len);
/* base^exp in Fr on the host (domain elements / twiddles for hosts without field code of their own) */
int ark_hip_fr_pow(int field, const uint64_t* base, uint64_t exp, uint64_t* out);
/* the G-point transform along the slow axis of a [G][cols] device array (G = 2, 4, 8 or 16), in place:
 * out[j][c] = sum_i root^(i j) in[i][c] */
int ark_hip_fft_axis_device(int field, void* d_data, unsigned G, size_t cols, const uint64_t* root);

/* ---- one process per GPU: RCCL inside the library ---------------------------------------------------------------
 * The reference chunks an MSM by base range and sums the chunk results (variable_base/mod.rs:521-557); an FFT shards by
 * coefficient range with ONE transpose between two rounds of local butterflies (the four-step form of the radix-2
 * recursion, radix2/fft.rs:190-349).  Across the GPUs of a node both need exactly one exchange, and the library runs it
 * itself on RCCL (loaded at run time: librccl.so.1, the copy the process already has if any -- e.g. PyTorch's):
 *   id:    rank 0 calls ark_hip_comm_unique_id and ships the ARK_HIP_COMM_ID_BYTES to the other ranks by any means the
 *          host has (MPI, a TCP socket, torch.distributed's store);
 *   init:  every rank calls ark_hip_comm_init(id, rank, world) on its device (collective: ncclCommInitRank);
 *   then the *_sharded entries below are collective calls: every rank makes the same sequence of them.
 * Without a communicator (or with world = 1) they run the single-GPU path. */
#define ARK_HIP_COMM_ID_BYTES 128
int ark_hip_comm_unique_id(void* out_id);
int ark_hip_comm_init(const void* id, int rank, int world);
int ark_hip_comm_info(int* rank, int* world);   /* (0, 1) when this device has no communicator */
int ark_hip_comm_destroy(void);
/* One MSM over the union of all ranks' (base, scalar) shards, device-resident; every rank receives the same sum.  The
 * exchange is an all-gather of one Projective per rank (3 field elements: elliptic-curve addition is not an RCCL reduce
 * op) followed by world - 1 point additions in rank order.  _prepared_: this rank's shard as a prepared base set. */
int ark_hip_msm_sw_device_sharded(int curve, const void* d_bases, const void* d_scalars, size_t n_local,
                                  int scalars_are_montgomery, uint64_t* out_xyz);
int ark_hip_msm_prepared_device_sharded(const ark_hip_msm_bases* bases, const void* d_scalars, size_t n_local,
                                        int scalars_are_montgomery, uint64_t* out_xyz);
/* Radix-2 FFT / IFFT of dom->size = G m elements over G = world ranks (G a power of two <= 16, G^2 | size), m per rank,
 * in place on d_local, ONE all-to-all.  With i = i1 + G i2 and j = j1 m + j2 (i1, j1 < G; i2, j2 < m; sub = m / G):
 *     X[j1 m + j2] = sum_i1 w_G^(i1 j1) * w_n^(i1 j2) * ( sum_i2 w_m^(i2 j2) x[i1 + G i2] )
 *   forward  in:  d_local[i2] = x[rank + G i2]                       (cyclic by coefficient index)
 *            out: d_local[j1 sub + t] = X[j1 m + rank sub + t]       (G rows of the rank's j2 range)
 *            = size-m transform with the twiddle w_n^rank fused into its last pass, all-to-all, G-point transform
 *   inverse  takes the forward's output layout back to the forward's input layout (G-point transform, all-to-all,
 *            twiddle fused into the first pass of the size-m inverse transform, 1/n into its last).
 * A coset domain (dom->offset != 1) is honoured as in the single-GPU transform.  The exchange is cut into slices that
 * travel on a second stream while the G-point kernel works on the previous slice.  Asynchronous on the context stream. */
int ark_hip_fft_sharded_device(int field, const ark_hip_radix2_domain* dom, void* d_local, int inverse);
/* The two local halves of that transform for a host that brings its own exchange (MPI, gloo, ...), and for testing the
 * decomposition on one GPU:  _local: the size-m transform with the per-rank twiddle (forward: first step; inverse: last
 * step);  _cross: the G-point transform of a [G][sub] array, d_src -> d_dst (may alias).  Forward = local, all-to-all
 * (block q of every rank to rank q), cross;  inverse = cross, all-to-all, local. */
int ark_hip_fft_shard_local_device(int field, const ark_hip_radix2_domain* dom, int rank, int world, void* d_local,
                                   int inverse);
int ark_hip_fft_shard_cross_device(int field, const ark_hip_radix2_domain* dom, int world, const void* d_src, void* d_dst,
                                   int inverse);
/* Which pass kernel the transforms of the current device use: 0 = saturated 32-bit limbs (default), 1 = carry-free 9 x 29-bit
 * limbs (same results bit for bit; measured no faster: DESIGN.md section 5), -1 = the environment's choice (ARK_HIP_FFT_LAZY). */
int ark_hip_fft_set_kernel(int variant);
int ark_hip_fft_set_timing(int enable);
/* [total_ms, npass, pass0_ms, pass1_ms, ...] of the last timed device transform */
int ark_hip_fft_last_timing(double out[10]);

#ifdef ARK_HIP_TEST_HOOKS
/* ---- device-arithmetic test hooks (used by tests/ to check the kernels' field and point
 * arithmetic against the oracle; host pointers) ----
 * Exported by libark_hip_test.so ONLY (the same objects as libark_hip.so + csrc/capi_test.hip; `make` builds both): the shipped
 * library has no ark_hip_test_* symbol.  Declared when ARK_HIP_TEST_HOOKS is defined.
 * op: 0 add, 1 sub, 2 mul, 3 sqr, 4 neg, 5 dbl, 7 into_bigint, 8 from_bigint; 20 / 21 / 22: x y, x^2, 2 x y computed through
 * the carry-free 28-bit-limb form of the accumulate kernels (device product, square, sum of two products) */
int ark_hip_test_field_op(int field, int op, const uint64_t* a, const uint64_t* b, uint64_t* r, size_t n);
int ark_hip_test_basefield_op(int curve, int op, const uint64_t* a, const uint64_t* b, uint64_t* r, size_t n);
/* the same ops (0 add, 1 sub, 2 mul, 3 sqr, 4 neg, 5 dbl) through the HOST builds of the field arithmetic -- what the MSM's serial
 * tail runs on (64-bit limbs) -- on the calling thread: no GPU involved */
int ark_hip_test_host_basefield_op(int curve, int op, const uint64_t* a, const uint64_t* b, uint64_t* r, size_t n);
/* kind: 2 bucket += affine, 3 bucket -= affine, 4 bucket += bucket, 5 bucket double, 6 bucket -> jacobian,
 * 7 affine double_to_bucket.  acc/other/out are arrays of n elements. */
int ark_hip_test_point_op(int curve, int kind, const uint64_t* acc, const uint64_t* other, uint64_t* out, size_t n);
/* test hook: msm_sharded's exchange with the ranks emulated in one process on one GPU (everything but the RCCL call): world
 * local MSMs, their part sums added on the device, one host tail -- or the fallback when the shards' plans differ.
 * *path: 1 = device-side sum of the part sums, 2 = fallback (finished partials summed on the host). */
int ark_hip_test_msm_sharded_emulated(int curve, int world, const void* const* d_bases, const void* const* d_scalars,
                                      const size_t* n_local, int scalars_are_montgomery, uint64_t* out_xyz, int* path);
/* HOST-ONLY test hook (no device needed): the serial tail of an MSM (ec/src/scalar_mul/variable_base/mod.rs:489-502, the
 * window combine) as msm_finish runs it.  parts: windows x (nbits + 1) bucket-form points (x | y | zz | zzz), row w =
 * U_(w,0) .. U_(w,nbits-1), A_w; widths: the windows' bit widths.  out = sum_w 2^(off_w) (A_w + 2^log2_l0 sum_b 2^b U_(w,b))
 * as a Projective (x | y | z). */
/* Host only: the verified cache's 128-bit tag of a buffer (keyed per process: equal only within one process). */
int ark_hip_test_base_hash(const uint64_t* p, size_t words, uint64_t out[2]);
int ark_hip_test_msm_host_fold(int curve, const uint64_t* parts, int windows, int nbits, int log2_l0, const int* widths,
                               uint64_t* out_xyz);

#endif /* ARK_HIP_TEST_HOOKS */

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif
