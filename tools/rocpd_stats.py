#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.2) `--kernel-trace` result database (rocpd sqlite, *_results.db) as the
per-kernel statistics table `--stats` would print: calls, total/avg/min/max duration, share of GPU time.
Launches shorter than --min-us are also reported separately so that the tiny set-up launches of bench.py
(1-point MSMs that build the synthetic bases) do not dilute the averages of the timed launches.

usage: tools/rocpd_stats.py gpurun_out/prof/r1_results.db [--min-us 1000] > profiles/<name>.txt
"""
import argparse
import sqlite3


def table(c, where, title):
    rows = c.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e6, min(end-start)/1e6, "
                     "max(end-start)/1e6, max(vgpr_count), max(lds_size), max(scratch_size) from kernels %s "
                     "group by name order by 3 desc" % where).fetchall()
    tot = sum(r[2] for r in rows) or 1.0
    print("## " + title)
    print("%-72s %6s %11s %10s %10s %10s %6s %5s %7s %7s" % ("kernel", "calls", "total_ms", "avg_ms", "min_ms", "max_ms",
                                                          "pct", "vgpr", "lds", "scratch"))
    for r in rows:
        print("%-72s %6d %11.3f %10.4f %10.4f %10.4f %6.2f %5d %7d %7d" % (r[0][:72], r[1], r[2], r[3], r[4], r[5],
                                                                       100 * r[2] / tot, r[6], r[7], r[8]))
    print()


def pmc_table(c, min_ns):
    """per-kernel sums of every collected hardware counter (rocprofv3 --pmc), launches >= min duration"""
    rows = c.execute("select kernel_name, counter_name, count(*), sum(value), avg(value), avg(duration)/1e6 "
                     "from counters_collection where duration >= %d group by kernel_name, counter_name "
                     "order by 4 desc" % min_ns).fetchall()
    print("## hardware counters per kernel (launches >= %.0f us)" % (min_ns / 1000))
    print("%-64s %-14s %6s %16s %16s %10s" % ("kernel", "counter", "calls", "sum", "avg/launch", "avg_ms"))
    for r in rows:
        print("%-64s %-14s %6d %16.1f %16.1f %10.4f" % (r[0][:64], r[1], r[2], r[3], r[4], r[5]))
    print()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--min-us", type=float, default=1000.0)
    ap.add_argument("--pmc", action="store_true", help="database comes from a --pmc run: print counter sums")
    ap.add_argument("--timeline", type=int, default=0, help="also list the LAST n launches in order: start, duration, idle gap before")
    a = ap.parse_args()
    c = sqlite3.connect(a.db)
    if a.pmc:
        pmc_table(c, int(a.min_us * 1000))
        return
    table(c, "", "all kernel launches")
    table(c, "where (end-start) >= %d" % int(a.min_us * 1000), "launches >= %.0f us (the timed full-size steps)" % a.min_us)
    if a.timeline:
        rows = c.execute("select name, start, end from kernels order by start desc limit %d" % a.timeline).fetchall()[::-1]
        print("## the last %d launches in start order (us): start, duration, idle gap since the latest end so far" % len(rows))
        t0, last_end, busy = rows[0][1], rows[0][1], 0
        for name, st, en in rows:
            print("%10.1f %9.1f %8.1f  %s" % ((st - t0) / 1e3, (en - st) / 1e3, max(0, st - last_end) / 1e3, name[:90]))
            busy += en - max(st, last_end) if en > last_end else 0
            last_end = max(last_end, en)
        print("span %.1f us, busy %.1f us" % ((last_end - t0) / 1e3, busy / 1e3))


if __name__ == "__main__":
    main()
