#!/usr/bin/env python3
"""Host-core scaling of the CPU baseline (the oracle's msm_bigint_wnaf restatement): what the box really offers
(affinity mask, cgroup quota) and scalar-muls/s against the thread count.  Test infrastructure only."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
print("os.cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try:
        print(f, open(f).read().strip())
    except OSError:
        pass
cid = O.CID["BLS12_381_G1"]
a4 = np.array([0xA11CE, 1, 2, 0], dtype=np.uint64)
b4 = np.array([0xB0B, 3, 0, 0], dtype=np.uint64)
n = 1 << log_n
bases = O.gen_bases(cid, a4, b4, n)
sc = O.gen_scalars(O.curve_info(cid)[1], 1, n)
for th in [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else "8,16,32,64,128,256".split(","))]:
    t = time.perf_counter()
    O.msm(cid, bases, sc, O.WNAF, th)
    dt = time.perf_counter() - t
    print("threads %4d  2^%d  %.2f s  %.3e scalar-muls/s" % (th, log_n, dt, n / dt), flush=True)
