#!/usr/bin/env python3
"""bench.py -- the hot path on N MI355X (one process per GPU).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): G1 scalar-muls/sec of a BLS12-381 G1 MSM, inputs resident in HBM.
  value   ONE MSM per step through the PLAIN entry -- ark_hip_msm_sw_device on raw bases, nothing precomputed: what
          VariableBaseMSM::msm / msm_bigint compute (variable_base/mod.rs:59-85).
          N = 1: 2^24 pairs (the configuration the metric is quoted on), `"scaling": "weak"` by the contract's convention
          for a one-GPU line.
          N > 1: BASELINE config 4 -- ONE 2^26 MSM split over the N ranks by base range (STRONG scaling: total work fixed,
          2^26 / N pairs per GPU; shard [r*2^26/N, (r+1)*2^26/N) lives on rank r; the ranks' part sums are all-gathered
          over RCCL and added on the device -- elliptic-curve addition, so not an RCCL reduction op; no other collective
          exists on the path).  `"scaling": "strong"`; its one-GPU reference is the N = 1 line's
          `config4_strong_2_26.value` (the same 2^26 job on one GPU), NOT that line's `value` (a 2^24 job).  Two more
          legs ride along at N > 1, each labelled: `weak_2_24_per_gpu` (2^24 pairs on every GPU: one 2^(24+log2 N) job)
          and `strong_2_24` (the 2^24 job itself split N ways: 2^24 / N pairs per GPU).
  extras  (never `value`)  `prepared`: the same job against a PREPARED base set (ark_hip_msm_bases_prepare: per-window
          multiples built once, outside the step, for a fixed SRS); `pipelined`: two asynchronous jobs in flight;
          `trait_surface`: the same job through ark_hip_msm_sw from HOST pointers -- what SWCurveConfig::msm and the
          msm_bigint hook call -- first call (bases + scalars over PCIe) and repeat calls (resident-base cache: scalars
          only), PCIe-inclusive; `config4_strong_2_26`: BASELINE config 4 -- ONE 2^26 MSM split over the N ranks (N = 1:
          the whole job on one GPU, the strong-scaling reference); the other BASELINE configs; the Fr FFT at 2^22; the
          CPU baseline (the oracle's restatement of ark-ec's msm_bigint_wnaf on this box's host cores).
The roofline figures of the dominant kernel are measured with HIP events on the library's stream inside this run.

Synthetic inputs (tools/synth.py): bases P_i = (a + i*b)*G grown on the GPU from the generator; uniform
scalars in [0, r).  After timing, every result is checked bit-exactly against k*G with k = sum_i s_i (a + i b) mod r
(exact big-integer identity) -- and k*G itself comes from the ORACLE's scalar multiplication (oracle/, CPU: independent of the
device arithmetic; `kG_source` in the line says so), as the at-size tests take it.  oracle/ is test infrastructure: this file
uses it as the checker and, in the two `cpu_baseline` legs, as the thing timed on the host -- never on the measured path.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import synth as S  # noqa: E402

CURVE = "BLS12_381_G1"
FIELD = "BLS12_381_FR"
R_MOD = S.R[FIELD]  # bls12_381 fr.rs:4-5
A0, B0 = S.A0, S.B0
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
# hardware-anchored multiply bound: the BEST issue rate of v_mad_u64_u32 measured on this chip -- 34.96e12 lane-ops/s, four
# independent chains per lane at eight waves per SIMD (profiles/r3_issue_rates.txt); the same loop at the accumulate
# kernel's two waves per SIMD issues 26.6e12/s (same file), reported beside it as a second, labelled figure.  No figure is
# quoted against a loop the kernel beats.
MAD_U64_U32_PER_S = 34.96e12
MAD_U64_U32_PER_S_TWO_WAVES = 26.6e12
SIMPLE_VALU_PER_S = 55.04e12   # v_add_u32, the same file: the plain vector instruction's issue rate
# Fr products: the carry-free 9 x 29-bit product back to back in a loop, best measured rate (eight waves per SIMD) and the
# rate at two waves (profiles/r4_ubench_product_rate_29bit.txt, BLS12-381 Fr)
FR_PRODUCTS_PER_S = {"carry-free": 177.2e9, "saturated": 135.7e9}
FR_PRODUCTS_PER_S_TWO_WAVES = {"carry-free": 154.5e9, "saturated": 121.5e9}
# multiply-adds per mixed addition.  Saturated 32-bit limbs (ec.cuh): 8 products of 2 * 12^2 + one sum of two products
# under a single reduction (3 * 12^2), each followed by a carry instruction.  Carry-free 28-bit limbs (ec28.cuh, the
# default for the Fp384 G1 curves): 6 products of 2 * 14^2, 2 squares of 105 + 14^2, one two-product sum of 3 * 14^2.
MADS_PER_MIXED_ADD = {"saturated": 8 * 288 + 432, "lazy28": 6 * 392 + 2 * 301 + 588}
LAZY = not os.environ.get("ARK_HIP_MSM_LAZY", "1").startswith("0")   # the library's own switch (msm.cuh)
ACC_KERNEL = "msm_accumulate_lazy_kernel" if LAZY else "msm_accumulate_kernel"
LOG_PER_GPU = 24                    # pairs per GPU of the headline job (BASELINE config 2)
LOG_CONFIG4 = 26                    # BASELINE config 4: one 2^26 MSM over all ranks
limbs4 = S.limbs4


def gen_scalars(n, seed):
    return S.gen_scalars(n, seed, R_MOD)


_ORACLE = [None, "unset"]


def kg_affine(cid, k, r):
    """k*G as affine limbs, the expected value of every checked leg: from the oracle's scalar multiplication (CPU), or --
    only when the oracle library is not there -- from the product's own 1-point MSM (the line's `kG_source` says which)."""
    if _ORACLE[1] == "unset":
        try:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_lib as O  # test infrastructure: here only as the checker of the timed results
            _ORACLE[0], _ORACLE[1] = O, "oracle/ scalar_mul on the host (independent of the device arithmetic)"
        except Exception as e:  # noqa: BLE001
            _ORACLE[0], _ORACLE[1] = None, "the product's own 1-point MSM (oracle library unavailable: %s)" % repr(e)[:80]
    O = _ORACLE[0]
    if O is None:
        return S.mul_gen(cid, k, r)
    return O.to_affine(cid, O.scalar_mul(cid, O.generator(cid), S.limbs4(k % r)))


def fft_issue_bound(kf, nf, dev_ms):
    """The vector-issue bound of the saturated pass kernel (VERDICT r4 next #1d): instructions per 4-point group of a
    two-stage round counted in the shipped ISA (hipcc -S of fft_bls12_381_fr.hip: 512 v_mad_u64_u32 + 512 v_addc_co_u32 of
    the four Comba products, 380 others -- additions / subtractions with their conditional corrections, LDS traffic, index
    arithmetic), priced at the issue rates measured on this chip (profiles/r3_issue_rates.txt, eight waves per SIMD:
    v_mad_u64_u32 34.96e12 lane-ops/s, v_add_u32 55.04e12)."""
    P = (kf + 7) // 8
    base, rem = kf // P, kf % P
    kps = [base + (1 if i < rem else 0) for i in range(P)]
    for i in range(P):
        for j in range(P - 1, i, -1):
            if kps[i] % 2 and kps[j] % 2 and kps[i] < 8 and kps[j] > 2:
                kps[i] += 1
                kps[j] -= 1
                break
    if P <= 3:
        kps.sort()                                # round 6: the shortest pass first (fft.cuh fft_run_device)
    rounds2 = sum(kp // 2 for kp in kps)          # two-stage rounds: nf / 4 groups of four points each
    rounds1 = sum(kp % 2 for kp in kps)           # single-stage rounds: nf / 2 butterflies, half a group's work each
    groups = nf / 4.0 * (rounds2 + 0.5 * rounds1)
    mads, others = 512.0, 512.0 + 380.0
    t = groups * (mads / MAD_U64_U32_PER_S + others / SIMPLE_VALU_PER_S)
    return {"what": "time the transform's vector instructions need at the chip's measured issue rates: %d two-stage rounds "
                    "(+ %d single) x 2^%d 4-point groups x (512 v_mad_u64_u32 at 34.96e12 lane-ops/s + 892 plain vector "
                    "instructions at 55.04e12)" % (rounds2, rounds1, kf - 2),
            "bound_ms": t * 1e3, "achieved_ms": dev_ms, "frac": t * 1e3 / dev_ms, "pass_plan": kps,
            "source": "profiles/r3_issue_rates.txt; instruction counts: the kernel's ISA, see DESIGN.md section 5"}


def usable_cores():
    """Host cores this process may actually use: the smaller of the affinity mask and the cgroup CPU quota (the GPU
    boxes show 256 logical CPUs behind a 16-core quota; 256 threads on 16 cores' worth of time run 4x slower than 16:
    profiles/r3_cpu_baseline_host_scaling.txt)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, (q + per // 2) // per))
        except (OSError, ValueError):
            pass
    return n


def pmc_traffic(kernel, log_n):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes (profiles/r*_pmc_traffic.json,
    produced by tools/pmc_traffic.py on this same command; FETCH_SIZE / WRITE_SIZE collected in separate passes,
    corrected as MI355X_MICROARCH.md prescribes).  None when no matching measurement is committed."""
    for name in ("r6_pmc_traffic.json", "r5_pmc_traffic.json", "r4_pmc_traffic.json", "r3_pmc_traffic.json", "r2_pmc_traffic.json", "r1_pmc_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                d = json.load(f)
            e = d.get(kernel)
            if e and e.get("log_n") == log_n:
                return e.get("hbm_bytes_per_launch"), "profiles/" + name
        except Exception:
            pass
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--log-n", type=int, default=None,
                    help="N = 1: log2 of the pairs of the headline MSM (default 24: BASELINE config 2).  N > 1: log2 of the "
                         "pairs PER GPU of the weak-scaling leg, and of the job the strong_2_24 leg splits")
    ap.add_argument("--log-total", type=int, default=None,
                    help="N > 1: log2 of the pairs of the ONE job split over all ranks -- the headline value (default 26: "
                         "BASELINE config 4)")
    ap.add_argument("--fft-log-n", type=int, default=22)
    ap.add_argument("--fft-steps", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-log-n", type=int, default=24)
    ap.add_argument("--no-extras", action="store_true", help="headline only: skip every side measurement")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import algebra_amd as A
    from algebra_amd import curves as cv
    from algebra_amd import dist as D
    from algebra_amd._lib import check, lib

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d"
                         % (args.gpus, world, args.gpus))
    # one process per GPU.  (ARK_BENCH_BACKEND=gloo lets the N > 1 code path be exercised on a box with fewer GPUs
    # than ranks -- ranks then share devices; never used for reported numbers.)
    backend = os.environ.get("ARK_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    L = lib()
    check(L.ark_hip_init(dev_index), "ark_hip_init")
    cid = cv.curve_id(CURVE)
    start_guard = None
    if world > 1:
        # nothing below is allowed to hang the driver: if the communicators, the pre-flight or the timed headline steps have
        # not finished after ARK_BENCH_HEADLINE_WATCHDOG_S (default 900 s) the rank says why on stderr and leaves with code 3
        import threading

        def give_up():
            sys.stderr.write("bench.py: rank %d: no headline after the watchdog's limit (a collective did not return?); "
                             "ARK_BENCH_EXCHANGE=torch keeps the exchange in torch.distributed\n" % rank)
            sys.stderr.flush()
            os._exit(3)
        start_guard = threading.Timer(float(os.environ.get("ARK_BENCH_HEADLINE_WATCHDOG_S", "900")), give_up)
        start_guard.daemon = True
        start_guard.start()
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
        # bring the communicators up outside the timed region (RCCL initialises lazily on the first collective): the
        # library's own (the data path: ark_hip_msm_sw_device_sharded / ark_hip_fft_sharded_device) and, as the fallback
        # exchange should that fail on this box, torch.distributed's
        D.combine_partials(cid, np.zeros(cv.projective_words(cid), dtype=np.uint64))
        def all_ranks(ok):
            f = torch.tensor([1.0 if ok else 0.0], device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(f, op=dist.ReduceOp.MIN)
            return f.item() != 0.0
        why = ""
        try:
            if os.environ.get("ARK_BENCH_EXCHANGE", "") == "torch":
                ok, why = False, "ARK_BENCH_EXCHANGE=torch"
            else:
                ok = D.comm_init()
        except Exception as e:  # noqa: BLE001 -- the headline must survive a communicator problem
            ok, why = False, repr(e)[:120]
        if backend == "nccl" and all_ranks(ok):
            # pre-flight: one tiny sharded MSM (4 pairs per rank, k * G checked below by the real job's own check) so that
            # a collective that cannot run on this box is found here, on every rank at once, not inside the timed region
            try:
                tb = S.grow_bases(cid, 4, (A0 + rank * 4 * B0) % R_MOD, B0, R_MOD)
                ts = torch.from_numpy(gen_scalars(4, 0x7E57 + rank).view(np.int64)).cuda()
                torch.cuda.synchronize()
                D.msm_bigint_sharded(cid, tb, ts)
                ok2 = True
            except Exception as e:  # noqa: BLE001
                ok2, why = False, repr(e)[:120]
            if not all_ranks(ok2):
                ok = False
        else:
            ok = False
        if not ok and D.library_comm_active():
            try:
                D.comm_destroy()
            except Exception:  # noqa: BLE001
                D._LIB_COMM["world"] = 0
        exchange = "RCCL inside libark_hip.so (ark_hip_msm_sw_device_sharded: the ranks' part sums all-gathered, added on the " \
                   "device, one host tail for the whole job)" if ok else \
            "torch.distributed all_gather (%s backend%s)" % (backend, "; library communicator unavailable: " + why if why else "")

    if world == 1:
        exchange = "none (one GPU)"
    else:
        # what RCCL itself saw: the library communicator's rank / world and the RCCL build
        cr, cw = C.c_int(-1), C.c_int(-1)
        L.ark_hip_comm_info(C.byref(cr), C.byref(cw))
        try:
            nv = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:  # noqa: BLE001
            nv = "unknown"
        exchange += "; library communicator rank %d of %d; RCCL %s" % (cr.value, cw.value, nv)
    log_weak = args.log_n if args.log_n is not None else LOG_PER_GPU
    if world == 1:
        n = 1 << log_weak                   # pairs on this GPU
        headline_scaling = "weak"
    else:
        log_total = args.log_total if args.log_total is not None else LOG_CONFIG4
        n = (1 << log_total) // world       # BASELINE config 4: ONE 2^26 job, strong scaling
        headline_scaling = "strong"
    log_local = int(np.log2(n)) if n & (n - 1) == 0 else float(np.log2(n))
    n_total = n * world                     # pairs of the one MSM
    ab = cv.affine_bytes(cid)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        check(L.ark_hip_synchronize(), "sync")

    def all_max(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def make_inputs(n_local, first, seed):
        """this rank's shard [first, first + n_local) of the global arithmetic progression of bases + its scalars"""
        bases = S.grow_bases(cid, n_local, (A0 + first * B0) % R_MOD, B0, R_MOD)
        sc_h = gen_scalars(n_local, seed)
        sc = torch.from_numpy(sc_h.view(np.int64)).cuda()
        torch.cuda.synchronize()
        return bases, sc_h, sc

    def expected_affine(sc_h, first):
        """k*G for the whole job: every rank contributes its share of k"""
        k = S.dlog_of_msm(sc_h, A0, B0, R_MOD, first_index=first)
        if world > 1:
            ks = [None] * world
            dist.all_gather_object(ks, k)
            k = sum(ks) % R_MOD
        return kg_affine(cid, k, R_MOD) if rank == 0 else None

    def run_timed(local_msm, sc, warmup, steps, sharded=None):
        """K steps of: local MSM on this rank's shard, then (N > 1) all-gather of the partials + EC sum"""
        def step():
            if sharded is not None and D.library_comm_active():
                return sharded(sc)             # local MSM + all-gather + sum inside the library
            return D.combine_partials(cid, local_msm(sc))
        for _ in range(warmup):
            res = step()
        check(L.ark_hip_msm_set_timing(1), "set_timing")
        phases = np.zeros(8)
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            res = step()
            tm = (C.c_double * 8)()
            L.ark_hip_msm_last_timing(tm)
            phases += np.array(list(tm))
        barrier()
        elapsed = all_max(time.perf_counter() - t0)
        check(L.ark_hip_msm_set_timing(0), "set_timing")
        return res, elapsed, phases / max(steps, 1)

    # ---- the headline job: ONE MSM of n_total pairs through the plain entry, base-range shards ----------------
    first = rank * n
    bases, scalars_h, scalars = make_inputs(n, first, 0xA11CE + rank)
    result, elapsed, phases = run_timed(lambda sc: A.msm_bigint(cid, bases, sc), scalars, args.warmup, args.steps,
                                        sharded=lambda sc: D.msm_bigint_sharded(cid, bases, sc))
    want = expected_affine(scalars_h, first)
    exact = bool(np.array_equal(A.into_affine(cid, result), want)) if rank == 0 else None
    extras = not args.no_extras
    if start_guard is not None:
        start_guard.cancel()

    # ---- the line is assembled by a closure over the legs' results, so that a watchdog can print what is there ------
    prepared = pipelined = trait = config4 = others = fft = fft_sharded = cpu = sizes = None
    side_legs = {}

    def accumulate_fracs(n_, ph):
        """(HBM GB/s on algorithmic bytes, mixed additions, v_mad_u64_u32 lane-ops/s) of one accumulate launch"""
        acc_ms_, W_, cbits_ = float(ph[3]), int(ph[7]), int(ph[6])
        ach = 128.0 * n_ / (acc_ms_ * 1e-3) / 1e9  # algorithmic bytes: 96 B base + 32 B scalar per pair
        # mixed additions actually executed by the accumulate kernel: one per (scalar, window) with a non-zero digit,
        # minus the first point of every non-empty bucket (a copy, not an addition)
        entries = n_ * W_ * (1.0 - 2.0 ** -cbits_)
        nbuckets = W_ * (1 << (cbits_ - 1))
        madds_ = entries - nbuckets * (1.0 - np.exp(-entries / nbuckets))
        return ach, madds_, madds_ * MADS_PER_MIXED_ADD["lazy28" if LAZY else "saturated"] / (acc_ms_ * 1e-3)

    def build_line():
        acc_ms = float(phases[3])
        W = int(phases[7])
        cbits = int(phases[6])
        achieved, madds, mads_per_s = accumulate_fracs(n, phases)
        mads_per_add = MADS_PER_MIXED_ADD["lazy28" if LAZY else "saturated"]
        traffic, traffic_src = pmc_traffic(ACC_KERNEL, log_local)
        out = {
            "metric": "G1 scalar-muls/sec (MSM, 2^%d%s)" % (int(round(np.log2(n_total))), "" if world == 1 else
                                                             ", ONE job split over %d GPUs" % world),
            "value": n_total * args.steps / elapsed,
            "unit": "scalar-muls/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / args.steps,
            "higher_is_better": True,
            "scaling": headline_scaling,
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": "BLS12-381 G1 MSM, one job of 2^%d random bases/scalars, device resident, plain entry "
                                   "(raw bases, nothing precomputed: VariableBaseMSM::msm_bigint)%s"
                                   % (int(round(np.log2(n_total))), "" if world == 1 else
                                      " -- BASELINE config 4, split by base range over %d GPUs (strong scaling: the total is "
                                      "fixed; one-GPU reference = `config4_strong_2_26.value` of the --gpus 1 line)" % world),
                       "arithmetic": "exact integers on v_mad_u64_u32: Montgomery Fp384, bucket accumulation on %s, "
                                     "everything else on saturated 32-bit limbs" %
                                     ("carry-free 28-bit limbs" if LAZY else "saturated 32-bit limbs"),
                       "curve": CURVE, "window_bits": cbits, "windows": W,
                       "pairs_per_gpu": n, "sharding": "base-range, %d rank(s), partials all-gathered" % world,
                       "exchange": exchange},
            "bit_exact_vs_kG": exact,
            "kG_source": _ORACLE[1],
            "phases_ms": {"digits": phases[0], "partition_hist_scan": phases[1], "partition_sort_order": phases[2],
                          "accumulate": phases[3], "reduce": phases[4], "device_total": phases[5]},
            "roofline": {"bound": "hbm", "kernel": ACC_KERNEL,
                         "achieved": achieved, "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "alu": {"what": "v_mad_u64_u32 lane-ops/s issued by the mixed additions the accumulate kernel "
                                         "executes (%d per addition, %s) vs the instruction's measured issue rate on "
                                         "this chip" % (mads_per_add, "carry-free 28-bit limbs: 6 products, 2 squares, "
                                                        "one two-product sum" if LAZY else
                                                        "saturated 32-bit limbs: 8 products + one two-product sum"),
                                 "mixed_additions": madds, "achieved": mads_per_s, "peak": MAD_U64_U32_PER_S,
                                 "frac": mads_per_s / MAD_U64_U32_PER_S,
                                 "two_waves": {"what": "the same instruction's issue rate at the kernel's occupancy (two "
                                                       "waves per SIMD, 211 VGPRs)",
                                               "peak": MAD_U64_U32_PER_S_TWO_WAVES,
                                               "frac": mads_per_s / MAD_U64_U32_PER_S_TWO_WAVES},
                                 "peak_source": "profiles/r3_issue_rates.txt: v_mad_u64_u32, four independent chains per "
                                                "lane -- 34.96e12 lane-ops/s at eight waves per SIMD (the best rate "
                                                "measured on this chip: `peak`), 26.6e12/s at two waves"},
                         "note": "MSM is integer-ALU bound (SURVEY 8d): the HBM fraction is tiny by construction"},
            "cpu_baseline": cpu,
            "trait_surface": trait,
            "prepared": prepared,
            "pipelined": pipelined,
            "config4_strong_2_26": config4 if world == 1 else "this line's `value` (N > 1: the headline IS config 4)",
            "north_star_sizes": sizes,
            "other_scalings": side_legs if world > 1 else None,
            "other_configs": others,
            "fft": fft,
            "fft_sharded": fft_sharded,
        }
        return out

    # N > 1: the exchange legs (RCCL inside the library: sharded MSM, sharded FFT) have only ever run on one-GPU boxes with
    # emulated ranks -- a collective that hangs on real hardware must not cost the headline line.  Once `value` exists a
    # watchdog prints the line with whatever legs have finished and ends the process (every rank runs its own).
    watchdog = None
    if world > 1:
        import threading
        limit = float(os.environ.get("ARK_BENCH_WATCHDOG_S", "1200"))

        def show():   # rank 0, three seconds before every rank leaves: the line with whatever legs have finished
            line = build_line()
            line["watchdog"] = "side legs did not finish within %.0f s of the headline: line printed without them" % limit
            print(json.dumps(line), flush=True)

        def leave():  # every rank at the same moment, exit code 0 (a rank that outlives its peers dies in its collective)
            os._exit(0)
        if rank == 0:
            early = threading.Timer(max(limit - 3.0, 0.0), show)
            early.daemon = True
            early.start()
        watchdog = threading.Timer(limit, leave)
        watchdog.daemon = True
        watchdog.start()


    # ---- N > 1: the two other scalings of the same path, each labelled (never `value`) ---------------------------
    if world > 1 and extras:
        for key, n_leg, scal, what in (
                ("weak_2_%d_per_gpu" % log_weak, 1 << log_weak, "weak",
                 "2^%d pairs on every GPU: one 2^%d * %d job" % (log_weak, log_weak, world)),
                ("strong_2_%d" % log_weak, (1 << log_weak) // world, "strong",
                 "the 2^%d job (BASELINE config 2) split %d ways; its one-GPU reference is the N = 1 line's `value`"
                 % (log_weak, world))):
            try:
                if n_leg == n:
                    lb, lsh, ls = bases, scalars_h, scalars
                else:
                    lb, lsh, ls = make_inputs(n_leg, rank * n_leg, 0x1E6 + rank)
                st = min(args.steps, 5)
                res_l, el_l, ph_l = run_timed(lambda sc: A.msm_bigint(cid, lb, sc), ls, 1, st,
                                              sharded=lambda sc: D.msm_bigint_sharded(cid, lb, sc))
                kl = expected_affine(lsh, rank * n_leg)
                if rank == 0:
                    side_legs[key] = {"what": what, "scaling": scal, "pairs_per_gpu": n_leg,
                                      "value": n_leg * world * st / el_l, "unit": "scalar-muls/s",
                                      "ms_per_step": el_l * 1e3 / st, "steps": st,
                                      "window_bits": int(ph_l[6]), "windows": int(ph_l[7]), "accumulate_ms": ph_l[3],
                                      "bit_exact_vs_kG": bool(np.array_equal(A.into_affine(cid, res_l), kl))}
                if n_leg != n:
                    del lb, ls
                    torch.cuda.empty_cache()
            except Exception as e:  # noqa: BLE001 -- a side measurement must never cost the headline line
                if rank == 0:
                    side_legs[key] = {"error": repr(e)[:200]}

    # ---- extension: the same job against a PREPARED base set (fixed SRS; the table is built outside the step) ----
    prepared = None
    pipelined = None
    if extras:
        try:
            t0 = time.perf_counter()
            pb = A.PreparedBases(cid, bases)
            prep_s = time.perf_counter() - t0
            st = min(args.steps, 5)
            r2, e2, ph2 = run_timed(pb.msm_bigint, scalars, 1, st)
            if rank == 0:
                prepared = {"what": "same job, base set prepared ONCE outside the step (ark_hip_msm_bases_prepare: per-window "
                                    "multiples of a fixed SRS in HBM); an extension, not the VariableBaseMSM::msm operation",
                            "value": n_total * st / e2, "ms_per_step": e2 * 1e3 / st, "prepare_s": prep_s,
                            "table_gib": pb.info()["table_bytes"] / 2**30, "window_bits": int(ph2[6]),
                            "windows": int(ph2[7]), "accumulate_ms": ph2[3],
                            "bit_exact_vs_kG": bool(np.array_equal(A.into_affine(cid, r2), want))}
            # two asynchronous jobs in flight: job k+1's digits / sort / reduction run under job k's accumulate kernel
            # on the library's second MSM lane -- steady state of a prover committing to one polynomial after another
            if world == 1:
                st = max(args.steps, 4)
                pb.msm_bigint_async(scalars).wait()
                barrier()
                t0 = time.perf_counter()
                pend, last = [], None
                for _ in range(st):
                    pend.append(pb.msm_bigint_async(scalars))
                    if len(pend) == 2:
                        last = pend.pop(0).wait()
                while pend:
                    last = pend.pop(0).wait()
                barrier()
                ep = time.perf_counter() - t0
                pipelined = {"what": "prepared base set, two asynchronous jobs in flight (ark_hip_msm_prepared_device_async)",
                             "value": n_total * st / ep, "ms_per_step": ep * 1e3 / st, "steps": st,
                             "bit_exact_vs_kG": bool(np.array_equal(A.into_affine(cid, last), want))}
            pb.free()
            del pb
        except Exception as e:  # noqa: BLE001 -- a side measurement must never cost the headline line
            prepared = {"error": repr(e)[:200]}
        torch.cuda.empty_cache()

    # ---- the trait surface: the same job through ark_hip_msm_sw from HOST pointers (N = 1) ---------------------
    trait = None
    if extras and world == 1:
        try:
            hb = bases.cpu().numpy().view(np.uint64).reshape(n, -1)   # ordinary (pageable) host memory, like a Rust Vec

            def timed_calls(reps):
                t0 = time.perf_counter()
                for _ in range(reps):
                    res = A.msm_bigint(cid, hb, scalars_h)
                return (time.perf_counter() - t0) * 1e3 / reps, res

            ok_all = True
            A.base_cache_config(0, 0)                                  # nothing retained: bases + scalars streamed
            A.msm_bigint(cid, hb, scalars_h)                           # (ring buffers allocated)
            off_ms, res = timed_calls(3)
            ok_all &= bool(np.array_equal(A.into_affine(cid, res), want))
            A.base_cache_config(-2, 0)                                 # the default: verified cache
            A.base_cache_clear()
            s0 = A.base_cache_stats()
            first_ms, res = timed_calls(1)
            ok_all &= bool(np.array_equal(A.into_affine(cid, res), want))
            rep_ms, res = timed_calls(4)
            ok_all &= bool(np.array_equal(A.into_affine(cid, res), want))
            s1 = A.base_cache_stats()
            A.base_cache_clear()
            with A.pin_bases(cid, hb):
                A.msm_bigint(cid, hb, scalars_h)
                pin_ms, res = timed_calls(4)
            ok_all &= bool(np.array_equal(A.into_affine(cid, res), want))
            trait = {"what": "ark_hip_msm_sw(host bases, host scalars): the call behind SWCurveConfig::msm / the msm_bigint "
                             "hook; PCIe-inclusive, pageable host memory.  DEFAULT settings = the verified resident-base "
                             "cache: a repeat call works from the device copy while 8 host threads hash the full 2^%d x 96 B "
                             "slice, and returns only if the hash matches (an edited slice is re-uploaded and the MSM rerun); "
                             "`cache_off`: bases + scalars streamed on every call, nothing retained; `pinned`: "
                             "ark_hip_msm_bases_pin (the Rust ResidentBases guard), no host pass at all" % log_local,
                     "first_call_ms": first_ms, "repeat_call_ms": rep_ms, "repeat_value": n / (rep_ms * 1e-3),
                     "cache_off_call_ms": off_ms, "pinned_repeat_call_ms": pin_ms,
                     "host_bytes_first_call": int(hb.nbytes + scalars_h.nbytes), "host_bytes_repeat_call": int(scalars_h.nbytes),
                     "cache": {k: s1[k] - s0[k] for k in ("hits", "misses", "refreshed", "evicted")},
                     "bit_exact_vs_kG": ok_all}
            del hb
        except Exception as e:  # noqa: BLE001
            trait = {"error": repr(e)[:200]}
            try:
                A.base_cache_config(-2, 0)
            except Exception:  # noqa: BLE001
                pass

    # ---- BASELINE config 4: ONE 2^26 MSM split over the ranks (strong scaling; N = 1: the whole job on one GPU) ----
    config4 = None
    if extras and world == 1 and log_local == LOG_PER_GPU:
        try:
            nb_ = (1 << LOG_CONFIG4) // world
            if nb_ == n:
                bb, bsh, bs = bases, scalars_h, scalars
            else:
                del bases, scalars
                bases = scalars = None
                torch.cuda.empty_cache()
                bb, bsh, bs = make_inputs(nb_, rank * nb_, 0xB16 + rank)
            st = 2 if world == 1 else min(args.steps, 5)
            res_b, el_b, ph_b = run_timed(lambda sc: A.msm_bigint(cid, bb, sc), bs, 1, st,
                                          sharded=lambda sc: D.msm_bigint_sharded(cid, bb, sc))
            kb = expected_affine(bsh, rank * nb_)
            if rank == 0:
                config4 = {"what": "one 2^%d MSM, plain entry, %d pairs per GPU over %d GPU(s)" % (LOG_CONFIG4, nb_, world),
                           "value": (1 << LOG_CONFIG4) * st / el_b, "ms_per_step": el_b * 1e3 / st, "scaling": "strong",
                           "window_bits": int(ph_b[6]), "windows": int(ph_b[7]), "accumulate_ms": ph_b[3],
                           "bit_exact_vs_kG": bool(np.array_equal(A.into_affine(cid, res_b), kb))}
            if world == 1:
                try:
                    t0 = time.perf_counter()
                    bpb = A.PreparedBases(cid, bb)
                    bprep = time.perf_counter() - t0
                    res_p, el_p, ph_p = run_timed(bpb.msm_bigint, bs, 1, 2)
                    config4["prepared"] = {"value": nb_ * 2 / el_p, "ms_per_step": el_p * 1e3 / 2, "window_bits": int(ph_p[6]),
                                           "windows": int(ph_p[7]), "table_gib": bpb.info()["table_bytes"] / 2**30,
                                           "prepare_s": bprep,
                                           "bit_exact_vs_kG": bool(np.array_equal(A.into_affine(cid, res_p), kb))}
                    bpb.free()
                except Exception as e:  # noqa: BLE001
                    config4["prepared"] = {"error": repr(e)[:200]}
            # ---- the north star's size axis on one GPU: 2^20 .. 2^26, plain entry, each with its phase times, the accumulate
            # kernel's HBM fraction (algorithmic bytes) and multiply-issue fraction, checked against k*G.  The inputs are
            # PREFIXES of the 2^26 job's (P_i = (a + i b)G, i < 2^k; the first 2^k scalars), so nothing is generated twice.
            if world == 1 and config4 is not None:
                sizes = {}
                for lg in range(20, LOG_CONFIG4 + 1):
                    try:
                        nk = 1 << lg
                        bk, sk = bb[: nk * ab], bs[:nk]
                        stk = 10 if lg <= 22 else (5 if lg <= 24 else 2)
                        if lg == LOG_CONFIG4:
                            res_k, el_k, ph_k, stk = res_b, el_b, ph_b, st
                            wk = kb
                        else:
                            res_k, el_k, ph_k = run_timed(lambda sc: A.msm_bigint(cid, bk, sc), sk, 1, stk)
                            wk = kg_affine(cid, S.dlog_of_msm(bsh[:nk], A0, B0, R_MOD), R_MOD)
                        ach_k, madds_k, mads_k = accumulate_fracs(nk, ph_k)
                        sizes["2^%d" % lg] = {
                            "ms_per_step": el_k * 1e3 / stk, "value": nk * stk / el_k, "steps": stk,
                            "window_bits": int(ph_k[6]), "windows": int(ph_k[7]),
                            "phases_ms": {"digits": ph_k[0], "sort": ph_k[1] + ph_k[2], "accumulate": ph_k[3], "reduce": ph_k[4],
                                          "device_total": ph_k[5]},
                            "hbm_frac": ach_k / HBM_PEAK_GBPS, "alu_frac": mads_k / MAD_U64_U32_PER_S,
                            "bit_exact_vs_kG": bool(np.array_equal(A.into_affine(cid, res_k), wk))}
                    except Exception as e:  # noqa: BLE001
                        sizes["2^%d" % lg] = {"error": repr(e)[:200]}
                sizes["what"] = ("BLS12-381 G1 MSM, plain entry, device resident, one GPU: the north star's 2^20 .. 2^26 axis; "
                                 "hbm_frac = 128 B x n / accumulate time / 8 TB/s, alu_frac = the accumulate kernel's "
                                 "v_mad_u64_u32 lane-ops/s over the chip's best measured issue rate (as `roofline`); 2 / 4 / 8 "
                                 "GPUs: UNMEASURED ON HARDWARE (no multi-GPU node in the build loop; the driver's SCALE run is "
                                 "the measurement)")
            if nb_ != n:
                del bb, bs
                torch.cuda.empty_cache()
        except Exception as e:  # noqa: BLE001
            config4 = {"error": repr(e)[:200]}

    # ---- the other BASELINE configurations, one line each (rank 0, single GPU) --------------------------------
    others = None
    if world == 1 and extras:
        others = {}
        for cname, lg, st in (("BN254_G1", 16, 20), ("BLS12_381_G1", 20, 10), ("BLS12_377_G2", 22, 3)):
            try:
                c2 = cv.curve_id(cname)
                r2 = S.R[cv.scalar_field(c2)]
                n2 = 1 << lg
                b2 = S.grow_bases(c2, n2, A0, B0, r2)
                sh2 = S.gen_scalars(n2, 0xC0DE + lg, r2)
                s2 = torch.from_numpy(sh2.view(np.int64)).cuda()
                torch.cuda.synchronize()
                kg2 = kg_affine(c2, S.dlog_of_msm(sh2, A0, B0, r2), r2)
                p2 = A.PreparedBases(c2, b2)
                entry = {}
                for label, fn in (("plain", lambda: A.msm_bigint(c2, b2, s2)), ("prepared", p2.msm_bigint)):
                    res2 = fn(s2) if label == "prepared" else fn()
                    t1 = time.perf_counter()
                    for _ in range(st):
                        res2 = fn(s2) if label == "prepared" else fn()
                    dt2 = (time.perf_counter() - t1) / st
                    entry[label] = {"ms_per_step": dt2 * 1e3, "value": n2 / dt2,
                                    "bit_exact_vs_kG": bool(np.array_equal(A.into_affine(c2, res2), kg2))}
                others["%s MSM 2^%d" % (cname, lg)] = entry
                p2.free()
                del b2, s2
                torch.cuda.empty_cache()
            except Exception as e:  # noqa: BLE001 -- side measurements never cost the headline line
                others["%s MSM 2^%d" % (cname, lg)] = {"error": repr(e)[:200]}

    # ---- the reference bench's non-uniform scalars (bench-templates/src/macros/ec.rs:244-372) at the headline size, plus a
    # witness-like vector: msm_bigint on device-resident inputs, each against k*G.  Not the headline (uniform scalars are).
    if others is not None:
        try:
            rng = np.random.default_rng(0x5CA1A)
            nsk = n
            bsk = bases if bases is not None else S.grow_bases(cid, nsk, A0, B0, R_MOD)   # (config 4 may have released them)
            skew = {}

            def limb_rows(v):
                a = np.zeros((nsk, 4), dtype=np.uint64)
                a[:, 0] = v
                return a

            def witness():   # 60 % zeros, 30 % ones, 5 % minus one, 5 % full width
                a = S.gen_scalars(nsk, 0x717, R_MOD)
                u = rng.random(nsk)
                a[u < 0.60] = 0
                one = np.zeros(4, dtype=np.uint64)
                one[0] = 1
                a[(u >= 0.60) & (u < 0.90)] = one
                a[(u >= 0.90) & (u < 0.95)] = np.array(S.limbs4(R_MOD - 1), dtype=np.uint64)
                return a

            for label, make in (("bool", lambda: limb_rows(rng.integers(0, 2, size=nsk, dtype=np.uint64))),
                                ("u8", lambda: limb_rows(rng.integers(0, 1 << 8, size=nsk, dtype=np.uint64))),
                                ("u16", lambda: limb_rows(rng.integers(0, 1 << 16, size=nsk, dtype=np.uint64))),
                                ("u32", lambda: limb_rows(rng.integers(0, 1 << 32, size=nsk, dtype=np.uint64))),
                                ("u64", lambda: limb_rows(rng.integers(0, 1 << 64, size=nsk, dtype=np.uint64))),
                                ("witness_60_30_5_5", witness)):
                sk_h = make()
                sk = torch.from_numpy(sk_h.view(np.int64)).cuda()
                torch.cuda.synchronize()
                want = kg_affine(cid, S.dlog_of_msm(sk_h, A0, B0, R_MOD), R_MOD)
                res_sk = A.msm_bigint(cid, bsk, sk)
                t1 = time.perf_counter()
                for _ in range(3):
                    res_sk = A.msm_bigint(cid, bsk, sk)
                dt_sk = (time.perf_counter() - t1) / 3
                skew[label] = {"ms_per_step": dt_sk * 1e3, "value": nsk / dt_sk,
                               "bit_exact_vs_kG": bool(np.array_equal(A.into_affine(cid, res_sk), want))}
                del sk
            others["BLS12_381_G1 MSM 2^%d, non-uniform scalars through msm_bigint" % log_local] = skew
            if bsk is not bases:
                del bsk
                torch.cuda.empty_cache()
        except Exception as e:  # noqa: BLE001
            others["non-uniform scalars"] = {"error": repr(e)[:200]}

    # ---- FFT leg (rank 0's GPU; the FFT config is single-GPU) ---------------------------------------
    # the reference's bench shapes (poly/benches/fft.rs:71-152): in-place FFT, IFFT and their coset variants on Vec<Fr>
    fft = None
    if rank == 0 and args.fft_steps > 0:
        kf = args.fft_log_n
        nf = 1 << kf
        dom = A.Radix2EvaluationDomain.new(FIELD, nf)
        seven = np.array(S.limbs4(7 * (1 << 256) % R_MOD), dtype=np.uint64)   # Fr::GENERATOR, Montgomery form (fft.rs:107)
        cos = dom.get_coset(seven)
        x_h = gen_scalars(nf, 7)                                        # canonical < r is a valid Montgomery residue too
        x = torch.from_numpy(x_h.view(np.int64)).cuda()
        y = x.clone()
        torch.cuda.synchronize()
        fwd = L.ark_hip_fft_in_place_device
        inv = L.ark_hip_ifft_in_place_device
        sref, cref = C.byref(dom._s), C.byref(cos._s)
        for _ in range(2):
            check(fwd(dom.field, sref, y.data_ptr()), "fft")
            check(inv(dom.field, sref, y.data_ptr()), "ifft")
        check(L.ark_hip_synchronize(), "sync")
        roundtrip_ok = bool(torch.equal(x, y))
        for _ in range(2):
            check(fwd(cos.field, cref, y.data_ptr()), "coset fft")
            check(inv(cos.field, cref, y.data_ptr()), "coset ifft")
        check(L.ark_hip_synchronize(), "sync")
        roundtrip_ok = roundtrip_ok and bool(torch.equal(x, y))

        def timed_dev(fn, ref, warm_ms=50.0):
            """args.fft_steps transforms back to back.  warm_ms > 0: the chip is first kept busy with the same transform for
            that long, so that the timed region runs at its SUSTAINED clocks -- after ~20 ms of idle time (host-side work
            between bench legs) the first milliseconds of device work run 17 % slower (power state ramp, measured:
            profiles/r5_fft_warm_cold.txt: 0.527 ms per 2^22 transform cold, 0.438 warm, same box, same binary)."""
            if warm_ms > 0:
                w0 = time.perf_counter()
                while (time.perf_counter() - w0) * 1e3 < warm_ms:
                    for _ in range(8):
                        check(fn(dom.field, ref, y.data_ptr()), "fft")
                    check(L.ark_hip_synchronize(), "sync")
            else:
                time.sleep(0.03)
            e0 = time.perf_counter()
            for _ in range(args.fft_steps):
                check(fn(dom.field, ref, y.data_ptr()), "fft")
            check(L.ark_hip_synchronize(), "sync")
            return (time.perf_counter() - e0) * 1e3 / args.fft_steps

        fft_cold_ms = timed_dev(fwd, sref, warm_ms=0.0)   # what round 4's line reported: the leg starts after host-side work
        fft_ms = timed_dev(fwd, sref)
        ifft_ms = timed_dev(inv, sref)
        cfft_ms = timed_dev(fwd, cref)
        cifft_ms = timed_dev(inv, cref)
        # per-pass device times (HIP events on the library stream), averaged over as many transforms as `ms_per_step`: the
        # chip's clocks drift by +-8 % over milliseconds under this kernel (profiles/r5_fft_schedule_ab.txt), so ONE
        # event-timed transform and the mean of twenty are different statistics -- round 4's "76 us per call that is not
        # kernel time" was that, not launch overhead (the trace shows 0 us between passes, 6 us between transforms)
        timed_dev(fwd, sref)                         # (leaves the chip at its sustained clocks for the event-timed transforms)
        check(L.ark_hip_fft_set_timing(1), "fft timing")
        ft = (C.c_double * 10)()
        dev_all, pass_sum = [], None
        for _ in range(max(1, args.fft_steps)):
            check(fwd(dom.field, sref, y.data_ptr()), "fft")
            L.ark_hip_fft_last_timing(ft)
            dev_all.append(ft[0])
            npass = int(ft[1])
            cur = np.array([ft[2 + i] for i in range(npass)])
            pass_sum = cur if pass_sum is None else pass_sum + cur
        check(L.ark_hip_fft_set_timing(0), "fft timing")
        dev_ms = float(np.mean(dev_all))
        dev_min = float(np.min(dev_all))
        pass_ms = [float(v) for v in pass_sum / len(dev_all)]
        # the same transform as a batch of 8 polynomials (three in flight: ark_hip_fft_batch_in_place_device)
        ys = [x.clone() for _ in range(8)]
        torch.cuda.synchronize()       # (as above: torch's stream is not the library's)
        ptrs = (C.c_void_p * 8)(*[t.data_ptr() for t in ys])
        check(L.ark_hip_fft_batch_in_place_device(dom.field, sref, ptrs, 8, 0), "fft batch")
        check(L.ark_hip_synchronize(), "sync")
        timed_dev(fwd, sref)                         # sustained clocks for the batch loop as well
        e0 = time.perf_counter()
        reps_b = max(1, args.fft_steps // 8)
        for _ in range(reps_b):
            check(L.ark_hip_fft_batch_in_place_device(dom.field, sref, ptrs, 8, 0), "fft batch")
        check(L.ark_hip_synchronize(), "sync")
        batch_ms = (time.perf_counter() - e0) * 1e3 / (8 * reps_b)
        batch_same = bool(torch.equal(ys[0], ys[7]))
        del ys
        # the trait surface: Vec<Fr> in host memory, in place (radix2/mod.rs:140-153) -- both PCIe crossings inside the call
        hx = x_h.copy()
        dom.fft_in_place(hx)
        t1 = time.perf_counter()
        host_reps = 3
        for _ in range(host_reps):
            dom.fft_in_place(hx)
        host_ms = (time.perf_counter() - t1) * 1e3 / host_reps
        del hx
        # multiply work: every two-stage round costs one Fr product per element, a single-stage round half of one; the
        # transform's last two stages multiply only by w^(n/4) (a quarter), its very last stage not at all
        lazy_fft = os.environ.get("ARK_HIP_FFT_LAZY", "0").startswith("1")   # the library's own switch (fft.cuh): default saturated
        kps = None
        products = None
        if not lazy_fft:
            # n/2 per stage minus the trivial twiddles the saturated kernel skips (one per block of every stage: n - 1 in all)
            products = float(nf // 2 * kf - (nf - 1))
        if lazy_fft and kf > 10:
            P_ = (kf + 7) // 8
            base_, rem_ = kf // P_, kf % P_
            kps = [base_ + (1 if i < rem_ else 0) for i in range(P_)]
            for i in range(P_):                       # the plan's pairing of odd passes (fft.cuh fft_run_device)
                for j in range(P_ - 1, i, -1):
                    if kps[i] % 2 and kps[j] % 2 and kps[i] < 8 and kps[j] > 2:
                        kps[i] += 1
                        kps[j] -= 1
                        break
            products = 0.0
            for i, kp_ in enumerate(kps):
                products += nf * (kp_ // 2) + (nf / 2) * (kp_ % 2)
                if i == len(kps) - 1:
                    products -= (nf / 2) if kp_ % 2 else (3 * nf / 4)
        fft_cpu = None
        if not args.no_cpu_baseline:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_lib as O  # test infrastructure: used here only as the timed CPU baseline
            cores = usable_cores()
            fid = O.FID[FIELD]
            O.fft(fid, x_h, kf, None, False, cores)                     # warm-up (page faults, thread start)
            t1 = time.perf_counter()
            creps = 3
            for _ in range(creps):
                ref_out = O.fft(fid, x_h, kf, None, False, cores)
            cpu_s = (time.perf_counter() - t1) / creps
            z = x.clone()
            torch.cuda.synchronize()   # the clone runs on torch's stream, the transform on the library's
            check(fwd(dom.field, sref, z.data_ptr()), "fft")
            check(L.ark_hip_synchronize(), "sync")
            same = bool(np.array_equal(z.cpu().numpy().view(np.uint64).reshape(-1), ref_out))
            fft_cpu = {"value": nf / cpu_s, "unit": "elements/s", "cores": cores, "kind": "port",
                       "sample": "the whole 2^%d transform, the oracle's threaded in-order radix-2 FFT (oracle/: restatement of "
                                 "radix2/fft.rs), %.2f s per transform on %d threads; GPU result limb for limb equal: %s"
                                 % (kf, cpu_s, cores, same)}
        kname = "fft_pass29_kernel" if lazy_fft else "fft_pass_kernel"
        fft = {
            "metric": "BLS12-381 Fr radix-2 FFT elements/sec (2^%d, in place, device resident)" % kf,
            "value": nf / (fft_ms * 1e-3), "unit": "elements/s", "ms_per_step": fft_ms,
            "ms_per_step_after_30ms_idle": fft_cold_ms, "value_after_30ms_idle": nf / (fft_cold_ms * 1e-3),
            "timing": "%d transforms back to back on the library stream after 50 ms of the same transform (sustained clocks); "
                      "`ms_per_step_after_30ms_idle`: the same loop entered from an idle chip (round 4's figure: the power "
                      "state ramps for the first milliseconds)" % args.fft_steps,
            "device_ms": dev_ms, "device_ms_fastest_of_%d" % len(dev_all): dev_min, "passes": pass_ms,
            "shapes_ms": {"fft": fft_ms, "ifft": ifft_ms, "coset_fft": cfft_ms, "coset_ifft": cifft_ms},
            "ifft_fft_roundtrip_exact": roundtrip_ok,
            "arithmetic": "exact integers on v_mad_u64_u32: %s" % ("carry-free 9 x 29-bit limbs" if lazy_fft else "saturated 32-bit limbs"),
            "batch_of_8": {"what": "8 polynomials over the same domain per call, three transforms in flight",
                           "ms_per_transform": batch_ms, "value": nf / (batch_ms * 1e-3), "results_agree": batch_same},
            "trait_surface": {"what": "ark_hip_fft_in_place on a host Vec<Fr> (the call behind Radix2EvaluationDomain::"
                                      "fft_in_place): 2 x %d MiB over PCIe inside the call, pageable memory" % (nf * 32 >> 20),
                              "ms_per_call": host_ms, "value": nf / (host_ms * 1e-3)},
            "cpu_baseline": fft_cpu,
            "roofline": {"bound": "hbm", "kernel": "%s (x%d passes)" % (kname, npass),
                         "achieved": 64.0 * nf / (dev_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": 64.0 * nf / (dev_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                         "traffic": pmc_traffic(kname, kf)[0],
                         "traffic_source": pmc_traffic(kname, kf)[1],
                         "alu": None if products is None else {
                             "what": "Fr products/s executed by the transform (%.3g products%s) vs the best rate of the same "
                                     "product back to back in a loop on this chip"
                                     % (products, "" if kps is None else ", pass plan %s" % kps),
                             "achieved": products / (dev_ms * 1e-3),
                             "peak": FR_PRODUCTS_PER_S["carry-free" if lazy_fft else "saturated"],
                             "frac": products / (dev_ms * 1e-3) / FR_PRODUCTS_PER_S["carry-free" if lazy_fft else "saturated"],
                             "two_waves": {"peak": FR_PRODUCTS_PER_S_TWO_WAVES["carry-free" if lazy_fft else "saturated"],
                                           "frac": products / (dev_ms * 1e-3)
                                           / FR_PRODUCTS_PER_S_TWO_WAVES["carry-free" if lazy_fft else "saturated"]},
                             "peak_source": "profiles/r4_ubench_product_rate_29bit.txt (BLS12-381 Fr): saturated product 135.7 G/s "
                                            "at eight waves per SIMD / 121.5 at two; carry-free 177.2 / 154.5"},
                         "valu_issue": None if lazy_fft or kf <= 10 else fft_issue_bound(kf, nf, dev_ms)},
        }

    # ---- sharded FFT leg (N > 1): 2^fft_log_n coefficients per GPU, all-to-all exchanges over RCCL ----------
    fft_sharded = None
    if world > 1 and args.fft_steps > 0 and extras:
        try:
            nloc = 1 << args.fft_log_n
            ntot = nloc * world
            xs = torch.from_numpy(gen_scalars(nloc, 11 + rank).view(np.int64)).cuda()   # this rank's cyclic slice
            ys = D.fft_sharded(FIELD, ntot, xs)                       # warm-up (tables, communicator)
            back = D.fft_sharded(FIELD, ntot, ys, inverse=True)
            rt_ok = bool(torch.equal(back, xs))
            if D.library_comm_active():   # timed in place through the C entry: no clone, no host synchronisation per step
                domt = A.Radix2EvaluationDomain.new(FIELD, ntot)
                sref_t = C.byref(domt._s)
                buf = xs.clone()
                torch.cuda.synchronize()
                barrier()
                t1 = time.perf_counter()
                for _ in range(args.fft_steps):
                    check(L.ark_hip_fft_sharded_device(domt.field, sref_t, buf.data_ptr(), 0), "fft sharded")
                barrier()
                dt = time.perf_counter() - t1
            else:
                barrier()
                t1 = time.perf_counter()
                for _ in range(args.fft_steps):
                    ys = D.fft_sharded(FIELD, ntot, xs)
                barrier()
                dt = time.perf_counter() - t1
            tt = torch.tensor([dt, 0.0 if rt_ok else 1.0], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            fft_sharded = {"metric": "BLS12-381 Fr radix-2 FFT elements/sec (2^%d per GPU, sharded: ONE all-to-all, cyclic in / row-block out)"
                                     % args.fft_log_n,
                           "value": ntot * args.fft_steps / float(tt[0].item()), "unit": "elements/s",
                           "ms_per_step": float(tt[0].item()) * 1e3 / args.fft_steps, "log_n_total": int(np.log2(ntot)),
                           "ifft_fft_roundtrip_exact": float(tt[1].item()) == 0.0,
                           "exchange": "RCCL inside libark_hip.so (ark_hip_fft_sharded_device)" if D.library_comm_active()
                                       else "torch.distributed all_to_all_single"}
        except Exception as e:  # never lose the MSM line to the secondary leg
            fft_sharded = {"error": repr(e)[:300]}

    # ---- CPU baseline: the oracle's msm_bigint_wnaf restatement on the host cores, bounded sample ---
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as O  # test infrastructure: used here only as the timed CPU baseline
        ns = 1 << min(args.cpu_sample_log_n, log_local)
        if bases is None or bases.numel() < ns * ab:
            bases, scalars_h, scalars = make_inputs(ns, 0, 0xA11CE)
        hb = bases[: ns * ab].cpu().numpy().view(np.uint64).reshape(ns, -1)
        cores = usable_cores()
        # warm-up, then the median of repeated runs.  One small warm-up (2^20: page faults of the bucket arenas,
        # thread start), then ARK_BENCH_CPU_REPS (default 3) runs of the sample; value = sample size / median
        nw = min(ns, 1 << 20)
        O.msm(O.CID[CURVE], hb[:nw], scalars_h[:nw], O.WNAF, cores)
        reps = max(1, int(os.environ.get("ARK_BENCH_CPU_REPS", "3")))
        runs = []
        for _ in range(reps):
            t1 = time.perf_counter()
            ref = O.msm(O.CID[CURVE], hb, scalars_h[:ns], O.WNAF, cores)
            runs.append(time.perf_counter() - t1)
        cpu_s = float(np.median(runs))
        same = bool(np.array_equal(O.to_affine(O.CID[CURVE], ref),
                                   A.into_affine(cid, A.msm_bigint(cid, bases[: ns * ab], scalars[:ns]))))
        cpu = {"value": ns / cpu_s, "unit": "scalar-muls/s", "cores": cores, "kind": "port",
               "runs_s": runs,
               "sample": "%s 2^%d of the same bases/scalars, msm_bigint_wnaf restatement (oracle/: a plain-C __int128 port, no "
                         "assembly -- NOT ark-ec's own speed), one 2^%d warm-up then the median of %d runs: %.1f s on %d threads "
                         "(%d logical CPUs visible, affinity / cgroup quota allow %d); GPU result on the sample "
                         "bit-exact: %s"
                         % ("all" if ns == n else "first", int(np.log2(ns)), int(np.log2(nw)), reps, cpu_s, cores,
                            os.cpu_count() or 1, cores, same)}

    if watchdog is not None:
        watchdog.cancel()
        if rank == 0:
            early.cancel()
    if rank == 0:
        print(json.dumps(build_line()), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
