#!/bin/bash
# round 3, final session: the committed build (one-Horner host tail, batch_mul by num_scalars):
# parity, smoke, bench line, kernel trace, FETCH / WRITE_SIZE (3 full-size launches), N = 2 over gloo
mkdir -p gpurun_out/r3fin3
O=$PWD/gpurun_out/r3fin3
R=$PWD
export TMPDIR=/tmp
(timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -6) > $O/tests.log
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log) 2>&1
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
(ARK_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 2 --steps 3 --warmup 1 --log-n 21 --fft-log-n 20 --fft-steps 4 > $O/bench_n2_gloo.json) 2> $O/bench_n2_gloo.err
(timeout 900 python tools/soak.py 300 2>&1 | tail -4) > $O/soak.log
echo done > $O/done
