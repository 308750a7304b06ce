#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc result databases (one pass with FETCH_SIZE, one with WRITE_SIZE -- the TCC block cannot
hold both, MI355X_MICROARCH.md "rocprofv3 PMC slots") into profiles/r<N>_pmc_traffic.json: HBM bytes per launch of the
dominant kernels, as bench.py's roofline.traffic reads them.

Units / corrections (MI355X_MICROARCH.md, section HBM): FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
reports half of the bytes of a wide coalesced streaming read, so the read side is doubled for the streaming kernels
(FFT passes).  Calibration inside the same pass: msm_digits_kernel streams exactly 512 MiB of scalars and reports
FETCH_SIZE = 256.05 MiB (factor 0.500), WRITE_SIZE = 832.0 MiB for 832 MiB written (factor 1.000).
For the MSM accumulate kernel the reads are 96-byte random gathers (6 x 16 B per lane), an access width the guide
calls uncalibrated: the raw counter equals 2 x 64 B per gathered point almost exactly (a 96-byte row always straddles
two 64-byte lines), i.e. 64-byte requests counted at face value, so the raw figure is used there and the doubled one
is recorded as an upper bound.

usage: tools/pmc_traffic.py FETCH_DB WRITE_DB LOG_N_MSM LOG_N_FFT > profiles/r2_pmc_traffic.json
"""
import json
import sqlite3
import sys


def per_kernel(db, counter, min_ns=100000):
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name, count(*), avg(value), max(value) from counters_collection where counter_name = ? "
                     "and duration >= ? group by kernel_name", (counter, min_ns)).fetchall()
    return {r[0]: (r[1], r[2], r[3]) for r in rows}


def pick(d, frag):
    for k, v in d.items():
        if frag in k:
            return v
    return None


def main():
    fdb, wdb, log_msm, log_fft = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    fetch = per_kernel(fdb, "FETCH_SIZE")
    write = per_kernel(wdb, "WRITE_SIZE")
    out = {}
    for frag, logn in (("msm_accumulate_shared_kernel", log_msm), ("msm_accumulate_kernel", log_msm),
                       ("fft_pass_kernel", log_fft)):
        f, w = pick(fetch, frag), pick(write, frag)
        if not f or not w:
            continue
        # use the largest launches (the timed full-size ones); avg over them
        fetch_kib, write_kib = f[2] if frag.startswith("msm") else f[1], w[2] if frag.startswith("msm") else w[1]
        gather = frag.startswith("msm")
        rd = (1.0 if gather else 2.0) * fetch_kib * 1024
        out[frag] = {
            "log_n": logn,
            "fetch_size_kib_raw": fetch_kib, "write_size_kib_raw": write_kib,
            "read_bytes": rd, "read_bytes_if_doubled": 2.0 * fetch_kib * 1024, "write_bytes": write_kib * 1024,
            "hbm_bytes_per_launch": rd + write_kib * 1024,
            "launches_seen": f[0],
            "note": ("random 96-B gathers: raw FETCH_SIZE (= 2 x 64 B lines per point); WRITE_SIZE as reported" if gather else
                     "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of a wide read); WRITE_SIZE as reported"),
        }
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
