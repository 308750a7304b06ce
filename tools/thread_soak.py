#!/usr/bin/env python3
"""Concurrency soak (round 6): T host threads call MIXED entry points of the library at once -- device-pointer MSMs, host-pointer
MSMs (verified cache on: hashing passes on the shared helper pool), prepared sets (synchronous and asynchronous jobs), narrow
scalars, msm_chunks, transforms from host and device memory, batched transforms, normalize_batch -- on three curves and three
fields, every result compared with an answer the ORACLE computed beforehand.  What it looks for: lost wake-ups and races in the
job-slot wait (capi_msm.hip: retry_while_busy), the helper pool (hostpool.hpp), the cache, the staging buffers and the two MSM
lanes under callers that outnumber the four job slots eight to one (the reference's callers are rayon pools).
    python tools/thread_soak.py [threads=32] [seconds=60] [seed=1] [only: substring of the work items' names]"""
import ctypes as C
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import algebra_amd as A
import oracle_lib as O
from algebra_amd._lib import ArkHipError, lib

T = int(sys.argv[1]) if len(sys.argv) > 1 else 32
SECONDS = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
SEED = int(sys.argv[3]) if len(sys.argv) > 3 else 1
ONLY = sys.argv[4] if len(sys.argv) > 4 else ""
rng = np.random.default_rng(SEED)
A4 = np.array([0xA11CE, 1, 2, 0], dtype=np.uint64)
B4 = np.array([0xB0B, 3, 0, 0], dtype=np.uint64)


def threads_of_process():
    with open("/proc/self/status") as f:
        for line in f:
            if line.startswith("Threads:"):
                return int(line.split()[1])
    return -1


# ---- the work items: inputs + the oracle's answers, computed once, single-threaded ---------------------------------------
cases = []   # (name, callable returning True / False)
busy = [0]   # asynchronous enqueues answered ARK_HIP_ERR_BUSY (expected under 32 callers)
CURVES = (("BLS12_381_G1", 1 << 13), ("BN254_G1", 1 << 13), ("BLS12_377_G2", 1 << 10))
keep = []    # device tensors and prepared sets stay alive for the whole run
for cname, cap in CURVES:
    cid = O.CID[cname]
    sfid = O.curve_info(cid)[1]
    bases = O.gen_bases(cid, A4, B4, cap)
    d_bases = torch.from_numpy(bases.view(np.int64)).cuda()
    pb = A.PreparedBases(cid, bases)
    keep += [d_bases, pb]
    for k in range(4):
        n = int(rng.integers(cap // 8, cap + 1))
        sc = O.gen_scalars(sfid, 100 * cid + k, n)
        if k == 3:                                   # a skewed vector: half zeros and ones, heavy runs
            sc[::2, 1:] = 0
            sc[::2, 0] &= np.uint64(1)
        exp = O.to_affine(cid, O.msm(cid, bases[:n], sc, O.SIGNED, 8))
        d_sc = torch.from_numpy(sc.view(np.int64)).cuda()
        hb = np.ascontiguousarray(bases[:n])
        keep += [d_sc, hb]
        cases.append(("msm_device %s %d" % (cname, n),
                      lambda cid=cid, b=d_bases, s=d_sc, n=n, e=exp: np.array_equal(A.into_affine(cid, A.msm_bigint(cid, b[:n], s)), e)))
        cases.append(("msm_host %s %d" % (cname, n),
                      lambda cid=cid, b=hb, s=sc, e=exp: np.array_equal(A.into_affine(cid, A.msm_bigint(cid, b, s)), e)))
        cases.append(("msm_prepared %s %d" % (cname, n),
                      lambda cid=cid, p=pb, s=sc, e=exp: np.array_equal(A.into_affine(cid, p.msm_bigint(s)), e)))
        def async_case(cid=cid, p=pb, s=sc, e=exp):
            try:
                job = p.msm_bigint_async(s)
            except ArkHipError as ex:    # the *_async entries report BUSY with four jobs in flight (the synchronous ones wait)
                if ex.code == -6:
                    busy[0] += 1
                    return True
                raise
            return np.array_equal(A.into_affine(cid, job.wait()), e)
        cases.append(("msm_prepared_async %s %d" % (cname, n), async_case))
    small = rng.integers(0, 1 << 16, size=cap, dtype=np.uint64).astype(np.uint16)
    sc4 = np.zeros((cap, 4), dtype=np.uint64)
    sc4[:, 0] = small
    exp = O.to_affine(cid, O.msm(cid, bases, sc4, O.SIGNED, 8))
    cases.append(("msm_u16 %s" % cname, lambda cid=cid, b=bases, s=small, e=exp: np.array_equal(A.into_affine(cid, A.msm_u16(cid, b, s)), e)))
    scm = O.gen_scalars(sfid, 777 + cid, cap, montgomery=True)
    expc = O.to_affine(cid, O.msm(cid, bases, scm, O.SIGNED, 8, montgomery_scalars=True))
    cases.append(("msm_chunks %s" % cname, lambda cid=cid, b=bases, s=scm, e=expc: np.array_equal(A.into_affine(cid, A.msm_chunks(cid, b, s, step=1 << 11)), e)))
for fname in ("BLS12_381_FR", "BN254_FR", "BLS12_377_FR"):
    fid = O.FID[fname]
    for log_n in (6, 11, 14, 17):
        x = O.gen_scalars(fid, 50 + log_n, 1 << log_n, montgomery=True)
        dom = A.Radix2EvaluationDomain.new(fname, 1 << log_n)
        cos = dom.get_coset(O.field_const(fid, 3))
        ef = O.fft(fid, x, log_n, None, False, 8)
        ei = O.fft(fid, x, log_n, O.field_const(fid, 3), True, 8)
        d_x = torch.from_numpy(x.view(np.int64)).cuda()
        keep += [d_x]
        cases.append(("fft_host %s 2^%d" % (fname, log_n), lambda d=dom, x=x, e=ef: np.array_equal(d.fft(x).reshape(-1), e)))
        cases.append(("coset_ifft_host %s 2^%d" % (fname, log_n), lambda d=cos, x=x, e=ei: np.array_equal(d.ifft(x).reshape(-1), e)))
        cases.append(("fft_device %s 2^%d" % (fname, log_n),
                      lambda d=dom, x=d_x, e=ef: np.array_equal(d.fft(x.clone()).cpu().numpy().view(np.uint64).reshape(-1), e)))
if ONLY:
    cases = [c for c in cases if any(o in c[0] for o in ONLY.split(","))]
print("thread soak: %d work items prepared, %d threads for %.0f s (seed %d)" % (len(cases), T, SECONDS, SEED), flush=True)

out = (C.c_int * 2)()
lib().ark_hip_host_threads(out)
pool0 = (out[0], out[1])
threads0 = threads_of_process()
bad, done, errs = [], [0] * T, []
stop = time.time() + SECONDS


def worker(t):
    r = np.random.default_rng(SEED * 1000 + t)
    try:
        while time.time() < stop:
            name, fn = cases[int(r.integers(0, len(cases)))]
            if not fn():
                bad.append((t, name))
            done[t] += 1
    except Exception as e:  # noqa: BLE001
        errs.append((t, repr(e)))


th = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
t0 = time.time()
for x in th:
    x.start()
peak = threads0
while any(x.is_alive() for x in th):
    time.sleep(0.5)
    peak = max(peak, threads_of_process())
for x in th:
    x.join()
lib().ark_hip_host_threads(out)
print("thread soak: %d calls in %.1f s on %d threads, %d mismatches, %d exceptions (%d asynchronous enqueues answered BUSY, as documented); "
      "helper pool %s -> %s; process threads %d -> peak %d"
      % (sum(done), time.time() - t0, T, len(bad), len(errs), busy[0], pool0, (out[0], out[1]), threads0, peak))
for b in bad[:10]:
    print("  MISMATCH", b)
for e in errs[:10]:
    print("  EXCEPTION", e)
ok = not bad and not errs and (out[0], out[1]) == pool0 and peak <= threads0 + T + 2
sys.exit(0 if ok else 1)
