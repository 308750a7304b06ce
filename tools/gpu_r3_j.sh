#!/bin/bash
# round 3, session J: parity + bench line of the restructured 28-bit accumulate kernels, host-core scaling of the CPU
# baseline, kernel trace, HBM traffic (FETCH_SIZE / WRITE_SIZE in separate passes, 3 full-size launches each) with the
# 96-byte-gather calibration on the microbenchmark
mkdir -p gpurun_out/r3j
O=$PWD/gpurun_out/r3j
R=$PWD
export TMPDIR=/tmp
(timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -15) > $O/tests.log
(timeout 300 python tools/cpu_scaling.py 20 8,16,32,64,128,256 > $O/cpu_scaling.txt) 2>&1
(timeout 600 python bench.py > $O/bench.json) 2> $O/bench.err
cd /tmp
BENCH3="python $R/bench.py --steps 3 --warmup 0 --no-cpu-baseline --no-extras --fft-steps 3"
timeout 600 rocprofv3 --kernel-trace -d $O/p_kt -o kt -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras > $O/kt.out 2> $O/kt.err
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $O/p_fetch -o f -- $BENCH3 > $O/fetch.out 2> $O/fetch.err
timeout 600 rocprofv3 --pmc WRITE_SIZE -d $O/p_write -o w -- $BENCH3 > $O/write.out 2> $O/write.err
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $O/p_cal -o c -- $R/algebra_amd/csrc/ubench/ubench.bin > $O/cal.out 2> $O/cal.err
cd $R
db() { find $O/$1 -name "*results.db" | head -1; }
python tools/rocpd_stats.py $(db p_kt) --min-us 200 > $O/kernel_stats.txt 2>> $O/post.err
python tools/rocpd_stats.py $(db p_fetch) --pmc --min-us 100 > $O/pmc_fetch.txt 2>> $O/post.err
python tools/rocpd_stats.py $(db p_write) --pmc --min-us 100 > $O/pmc_write.txt 2>> $O/post.err
python tools/rocpd_stats.py $(db p_cal) --pmc --min-us 100 > $O/pmc_cal.txt 2>> $O/post.err
python tools/pmc_traffic.py $(db p_fetch) $(db p_write) 24 22 $(db p_cal) 33554432 > $O/pmc_traffic.json 2>> $O/post.err
rm -rf $O/p_kt $O/p_fetch $O/p_write $O/p_cal
echo done > $O/done
