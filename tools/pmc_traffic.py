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

usage: tools/pmc_traffic.py FETCH_DB WRITE_DB LOG_N_MSM LOG_N_FFT [GATHER_CAL_FETCH_DB [GATHERS]] > profiles/r3_pmc_traffic.json
"""
import json
import sqlite3
import sys


def per_kernel(db, counter, min_ns=100000):
    """kernel -> (launches used, avg value, max value) over the FULL-SIZE launches of each kernel: those lasting at least
    half as long as its longest one (bench.py also launches the kernels on tiny set-up jobs)"""
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name, value, duration from counters_collection where counter_name = ? "
                     "and duration >= ?", (counter, min_ns)).fetchall()
    by = {}
    for name, value, dur in rows:
        by.setdefault(name, []).append((value, dur))
    out = {}
    for name, lst in by.items():
        dmax = max(d for _, d in lst)
        big = [v for v, d in lst if d >= 0.5 * dmax]
        out[name] = (len(big), sum(big) / len(big), max(big))
    return out


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
    for frag, logn in (("msm_accumulate_shared_lazy_kernel", log_msm), ("msm_accumulate_lazy_kernel", log_msm),
                       ("msm_accumulate_shared_kernel", log_msm), ("msm_accumulate_kernel", log_msm),
                       ("fft_pass_kernel", log_fft)):
        f, w = pick(fetch, frag), pick(write, frag)
        if not f or not w:
            continue
        fetch_kib, write_kib = f[1], w[1]   # averages over the full-size launches
        gather = frag.startswith("msm")
        rd = (1.0 if gather else 2.0) * fetch_kib * 1024
        out[frag] = {
            "log_n": logn,
            "fetch_size_kib_raw": fetch_kib, "write_size_kib_raw": write_kib,
            "read_bytes": rd, "read_bytes_if_doubled": 2.0 * fetch_kib * 1024, "write_bytes": write_kib * 1024,
            "hbm_bytes_per_launch": rd + write_kib * 1024,
            "launches_averaged": f[0],
            "note": ("random 96-B gathers: raw FETCH_SIZE (= 2 x 64 B lines per point); WRITE_SIZE as reported" if gather else
                     "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of a wide read); WRITE_SIZE as reported"),
        }
    if len(sys.argv) > 5:
        # calibration of the gather pattern: FETCH_SIZE pass over csrc/ubench/ubench.bin, whose k_gather96 issues a known
        # number of random 96-byte gathers (6 x 16 B per lane, as the accumulate kernel's Affine load does)
        cal = per_kernel(sys.argv[5], "FETCH_SIZE", 0)
        g = pick(cal, "k_gather96")
        if g:
            gathers = float(sys.argv[6]) if len(sys.argv) > 6 else 256.0 * 1024 * 256 * 128   # blocks x threads x per_thread of the timed launch
            out["gather96_calibration"] = {
                "fetch_size_kib_raw_max_launch": g[2], "gathers": gathers, "bytes_requested": gathers * 96,
                "raw_bytes_per_gather": g[2] * 1024 / gathers,
                "note": "a 96-byte row straddles two 64-byte lines: 128 B per gather at face value means FETCH_SIZE counts "
                        "the lines of a gather un-halved (unlike a wide streaming read)"}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
