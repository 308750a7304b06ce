#!/usr/bin/env python3
"""bench.py -- the hot path on N MI355X (one process per GPU).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): G1 scalar-muls/sec of a BLS12-381 G1 MSM at 2^24 points per GPU, inputs
resident in HBM; the JSON line also carries the Fr radix-2 FFT rate at 2^22 (`fft`), the roofline
figures of the dominant kernel measured with HIP events inside this run, and the CPU baseline
(the oracle's restatement of ark-ec's msm_bigint_wnaf, timed on this box's host cores on a bounded
sample).  One "step" = one complete MSM over the rank's 2^24 (base, scalar) pairs: digit recoding,
bucket sort, bucket accumulation, bucket reduction, window combine, result on the host.

Multi-GPU: the MSM shards by base range -- rank r owns pairs [r*n, (r+1)*n) of a
global N*n-point MSM (weak scaling), computes its partial on its GPU, the 144-byte partials are
all-gathered over RCCL and summed (elliptic-curve addition, so not an RCCL reduction op).

Synthetic inputs: bases P_i = (a + i*b)*G grown on the GPU from the generator;
uniform scalars in [0, r) by top-limb-masked rejection.  After timing, the result is checked
bit-exactly against k*G with k = sum_i s_i (a + i b) mod r (exact big-integer identity).
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

CURVE = "BLS12_381_G1"
FIELD = "BLS12_381_FR"
R_MOD = 52435875175126190479447740508185965837690552500527637822603658699938581184513  # bls12_381 fr.rs:4-5
A0 = 0xA11CE + (1 << 64) + (2 << 128)
B0 = 0xB0B + (3 << 64)
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
FP384_MUL_PEAK = 59.9e9  # measured ceiling of the Montgomery product sequence (ubench/mulbench.hip), products/s


def pmc_traffic(kernel, log_n):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes (profiles/r1_pmc_traffic.json,
    produced by tools/pmc_traffic.py; FETCH_SIZE / WRITE_SIZE collected in separate passes, corrected as
    MI355X_MICROARCH.md prescribes).  None when no matching measurement is committed."""
    try:
        with open(os.path.join(ROOT, "profiles", "r1_pmc_traffic.json")) as f:
            d = json.load(f)
        e = d.get(kernel)
        if e and e.get("log_n") == log_n:
            return e.get("hbm_bytes_per_launch")
    except Exception:
        pass
    return None


def limbs4(v):
    return np.array([(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)


def gen_scalars(n, seed):
    """uniform in [0, r): 255-bit draws, rejection (ff/src/fields/models/fp/mod.rs:525-547 style)."""
    rng = np.random.default_rng(seed)
    out = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64)
    out[:, 3] &= np.uint64((1 << 63) - 1)
    rl = [int(x) for x in limbs4(R_MOD)]

    def ge_r(a):
        ge = np.ones(a.shape[0], dtype=bool)   # equal so far -> counts as >=
        decided = np.zeros(a.shape[0], dtype=bool)
        for k in (3, 2, 1, 0):
            gt = a[:, k] > np.uint64(rl[k])
            lt = a[:, k] < np.uint64(rl[k])
            ge = np.where(~decided & lt, False, ge)
            decided |= gt | lt
        return ge

    bad = np.nonzero(ge_r(out))[0]
    while bad.size:
        new = rng.integers(0, 1 << 64, size=(bad.size, 4), dtype=np.uint64)
        new[:, 3] &= np.uint64((1 << 63) - 1)
        out[bad] = new
        bad = bad[ge_r(new)]
    return out


def _exact_sum(x):
    """exact integer sum of a uint64 array whose entries are < 2^58 (chunks of 32 stay < 2^63)."""
    pad = (-x.size) % 32
    if pad:
        x = np.concatenate([x, np.zeros(pad, dtype=np.uint64)])
    return int(x.reshape(-1, 32).sum(axis=1, dtype=np.uint64).astype(object).sum())


def dlog_of_msm(scalars, a, b):
    """k = sum_i s_i (a + i b) mod r, exact: the discrete log of the MSM of P_i = (a + i b)G."""
    n = scalars.shape[0]
    idx = np.arange(n, dtype=np.uint64)  # n <= 2^26
    s_sum = 0
    is_sum = 0
    for k in range(4):
        for half, sh in ((scalars[:, k] & np.uint64(0xFFFFFFFF), 0), (scalars[:, k] >> np.uint64(32), 32)):
            w = 1 << (64 * k + sh)
            s_sum += w * _exact_sum(half)
            is_sum += w * _exact_sum(half * idx)
    return (s_sum * a + is_sum * b) % R_MOD


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--log-n", type=int, default=24, help="log2 of (base, scalar) pairs per GPU")
    ap.add_argument("--fft-log-n", type=int, default=22)
    ap.add_argument("--fft-steps", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-log-n", type=int, default=24)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import algebra_amd as A
    from algebra_amd import curves as cv
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
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)

    cid = cv.curve_id(CURVE)
    n = 1 << args.log_n
    ab = cv.affine_bytes(cid)

    # ---- synthetic inputs, generated on the device ------------------------------------------------
    gen = np.zeros(cv.affine_words(cid), dtype=np.uint64)
    check(L.ark_hip_curve_generator(cid, gen.ctypes.data_as(C.c_void_p)), "generator")

    def mul_gen(k):  # k*G as affine limbs, via a 1-point MSM on the GPU
        return A.into_affine(cid, A.msm_bigint(cid, gen.reshape(1, -1), limbs4(k % R_MOD).reshape(1, 4)))

    a_r = (A0 + rank * n * B0) % R_MOD  # this rank's shard starts at global index rank*n
    bases = torch.zeros(n * ab, dtype=torch.uint8, device="cuda")
    p0 = mul_gen(a_r)
    bases[:ab] = torch.from_numpy(p0.view(np.uint8)).cuda()
    torch.cuda.synchronize()
    m = 1
    while m < n:
        cnt = min(m, n - m)
        d = np.ascontiguousarray(mul_gen(m * B0))
        check(L.ark_hip_sw_add_affine_device(cid, bases.data_ptr(), bases.data_ptr() + m * ab, cnt,
                                             d.ctypes.data_as(C.c_void_p)), "sw_add_affine_device")
        m += cnt
    scalars_h = gen_scalars(n, 0xA11CE + rank)
    scalars = torch.from_numpy(scalars_h.view(np.int64)).cuda()
    torch.cuda.synchronize()

    from algebra_amd import dist as D
    if world > 1:
        # bring the communicator up outside the timed region (RCCL initialises lazily on the first collective)
        D.combine_partials(cid, np.zeros(cv.projective_words(cid), dtype=np.uint64))

    def step():
        # local MSM on this rank's base range, then (N > 1) RCCL all-gather of the 144-byte partials + EC sum
        return D.msm_bigint_sharded(cid, bases, scalars)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        check(L.ark_hip_synchronize(), "sync")

    for _ in range(args.warmup):
        result = step()
    # ---- timed region: exactly K steps -------------------------------------------------------------
    check(L.ark_hip_msm_set_timing(1), "set_timing")
    acc_ms = []
    phases = np.zeros(8)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        result = step()
        tm = (C.c_double * 8)()
        L.ark_hip_msm_last_timing(tm)
        acc_ms.append(tm[3])
        phases += np.array(list(tm))
    barrier()
    elapsed = time.perf_counter() - t0
    check(L.ark_hip_msm_set_timing(0), "set_timing")
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    phases /= max(args.steps, 1)

    # ---- exactness of the timed result: MSM == k*G, k = sum s_i (a + i b) ---------------------------
    k_r = dlog_of_msm(scalars_h, a_r, B0)
    if world > 1:
        ks = [None] * world
        dist.all_gather_object(ks, k_r)
        k_tot = sum(ks) % R_MOD
    else:
        k_tot = k_r
    exact = None
    if rank == 0:
        exact = bool(np.array_equal(A.into_affine(cid, result), mul_gen(k_tot)))

    # ---- FFT leg (rank 0's GPU; the FFT config is single-GPU) ---------------------------------------
    fft = None
    if rank == 0 and args.fft_steps > 0:
        kf = args.fft_log_n
        nf = 1 << kf
        dom = A.Radix2EvaluationDomain.new(FIELD, nf)
        x = torch.from_numpy(gen_scalars(nf, 7).view(np.int64)).cuda()  # canonical < r is a valid Montgomery residue too
        y = x.clone()
        torch.cuda.synchronize()
        fwd = L.ark_hip_fft_in_place_device
        inv = L.ark_hip_ifft_in_place_device
        sref = C.byref(dom._s)
        for _ in range(2):
            check(fwd(dom.field, sref, y.data_ptr()), "fft")
            check(inv(dom.field, sref, y.data_ptr()), "ifft")
        check(L.ark_hip_synchronize(), "sync")
        roundtrip_ok = bool(torch.equal(x, y))
        e0, e1 = time.perf_counter(), None
        for _ in range(args.fft_steps):
            check(fwd(dom.field, sref, y.data_ptr()), "fft")
        check(L.ark_hip_synchronize(), "sync")
        e1 = time.perf_counter()
        fft_ms = (e1 - e0) * 1e3 / args.fft_steps
        # per-pass device times of one transform (HIP events on the library stream)
        check(L.ark_hip_fft_set_timing(1), "fft timing")
        check(fwd(dom.field, sref, y.data_ptr()), "fft")
        ft = (C.c_double * 10)()
        L.ark_hip_fft_last_timing(ft)
        check(L.ark_hip_fft_set_timing(0), "fft timing")
        dev_ms = ft[0]
        fft = {
            "metric": "BLS12-381 Fr radix-2 FFT elements/sec (2^%d, in place, device resident)" % kf,
            "value": nf / (fft_ms * 1e-3), "unit": "elements/s", "ms_per_step": fft_ms,
            "device_ms": dev_ms, "passes": [ft[2 + i] for i in range(int(ft[1]))],
            "ifft_fft_roundtrip_exact": roundtrip_ok,
            "roofline": {"bound": "hbm", "kernel": "fft_pass_kernel (x%d passes)" % int(ft[1]),
                         "achieved": 64.0 * nf / (dev_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": 64.0 * nf / (dev_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                         "traffic": pmc_traffic("fft_pass_kernel", kf)},
        }

    # ---- sharded FFT leg (N > 1): 2^fft_log_n coefficients per GPU, all-to-all exchanges over RCCL ----------
    fft_sharded = None
    if world > 1 and args.fft_steps > 0:
        try:
            nloc = 1 << args.fft_log_n
            ntot = nloc * world
            xs = torch.from_numpy(gen_scalars(nloc, 11 + rank).view(np.int64)).cuda()
            ys = D.fft_sharded(FIELD, ntot, xs)                       # warm-up (tables, communicator)
            back = D.fft_sharded(FIELD, ntot, ys, inverse=True)
            rt_ok = bool(torch.equal(back, xs))
            barrier()
            t1 = time.perf_counter()
            for _ in range(args.fft_steps):
                ys = D.fft_sharded(FIELD, ntot, xs)
            barrier()
            dt = time.perf_counter() - t1
            tt = torch.tensor([dt, 0.0 if rt_ok else 1.0], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            fft_sharded = {"metric": "BLS12-381 Fr radix-2 FFT elements/sec (2^%d per GPU, sharded, block layout in/out)"
                                     % args.fft_log_n,
                           "value": ntot * args.fft_steps / float(tt[0].item()), "unit": "elements/s",
                           "ms_per_step": float(tt[0].item()) * 1e3 / args.fft_steps, "log_n_total": int(np.log2(ntot)),
                           "ifft_fft_roundtrip_exact": float(tt[1].item()) == 0.0, "exchanges": "3 all-to-all"}
        except Exception as e:  # never lose the MSM line to the secondary leg
            fft_sharded = {"error": repr(e)[:300]}

    # ---- CPU baseline: the oracle's msm_bigint_wnaf restatement on the host cores, bounded sample ---
    cpu = None
    if rank == 0 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as O  # test infrastructure: used here only as the timed CPU baseline
        ns = 1 << min(args.cpu_sample_log_n, args.log_n)
        hb = bases[: ns * ab].cpu().numpy().view(np.uint64).reshape(ns, -1)
        cores = os.cpu_count() or 1
        t1 = time.perf_counter()
        ref = O.msm(O.CID[CURVE], hb, scalars_h[:ns], O.WNAF, cores)
        cpu_s = time.perf_counter() - t1
        same = bool(np.array_equal(O.to_affine(O.CID[CURVE], ref),
                                   A.into_affine(cid, A.msm_bigint(cid, bases[: ns * ab], scalars[:ns]))))
        cpu = {"value": ns / cpu_s, "unit": "scalar-muls/s", "cores": cores, "kind": "port",
               "sample": "%s 2^%d of the same bases/scalars, msm_bigint_wnaf restatement (oracle/), %.1f s; "
                         "GPU result on the sample bit-exact: %s"
                         % ("all" if ns == n else "first", int(np.log2(ns)), cpu_s, same)}

    if rank == 0:
        total_pairs = n * world * args.steps
        acc_avg_ms = float(np.mean(acc_ms))
        achieved = 128.0 * n / (acc_avg_ms * 1e-3) / 1e9  # algorithmic bytes: 96 B base + 32 B scalar per pair
        out = {
            "metric": "G1 scalar-muls/sec (MSM, 2^%d per GPU)" % args.log_n,
            "value": total_pairs / elapsed,
            "unit": "scalar-muls/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": "BLS12-381 G1 MSM, 2^%d random bases/scalars per GPU, device resident" % args.log_n,
                       "arithmetic": "Montgomery Fp384 on 32-bit limbs (v_mad_u64_u32), exact integers",
                       "curve": CURVE, "window_bits": int(phases[6]), "windows": int(phases[7]),
                       "sharding": "base-range, %d rank(s)" % world},
            "bit_exact_vs_kG": exact,
            "phases_ms": {"digits": phases[0], "partition_hist_scan": phases[1], "partition_sort_order": phases[2],
                          "accumulate": phases[3], "reduce": phases[4], "device_total": phases[5]},
            "roofline": {"bound": "hbm", "kernel": "msm_accumulate_kernel", "achieved": achieved, "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                         "traffic": pmc_traffic("msm_accumulate_kernel", args.log_n),
                         "alu": {"what": "Fp384 products/s in the accumulate kernel (10 per mixed addition) vs the "
                                         "measured ceiling of the multiply sequence",
                                 "achieved": 10.0 * n * phases[7] / (acc_avg_ms * 1e-3), "peak": FP384_MUL_PEAK,
                                 "frac": 10.0 * n * phases[7] / (acc_avg_ms * 1e-3) / FP384_MUL_PEAK},
                         "note": "MSM is integer-ALU bound (SURVEY 8d): the HBM fraction is tiny by construction"},
            "cpu_baseline": cpu,
            "fft": fft,
            "fft_sharded": fft_sharded,
        }
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
