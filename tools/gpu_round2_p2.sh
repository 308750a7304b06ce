#!/bin/bash
# GPU session P2 of round 2 (after the FFT radix-4 rounds, MSM lanes, LDS-parked reduction): final parity run, full bench line, and the profiles of the final kernels
# (kernel trace, FETCH/WRITE_SIZE, SQ counters of the bench command and of the G2 MSM; N = 2 bench path over gloo)
mkdir -p gpurun_out/r2p2
O=$PWD/gpurun_out/r2p2
R=$PWD
export TMPDIR=/tmp
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15) > $O/tests.log
(timeout 900 python bench.py --steps 10 --warmup 2 > $O/bench.json) 2> $O/bench.err
cd /tmp
BENCH="python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras"
BENCH1="python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras --fft-steps 1"
G2="python $R/tools/msm_bench.py BLS12_377_G2 22 2 both"
SQ="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"
run() {  # name, rocprof args..., -- command
  local name=$1; shift
  timeout 900 rocprofv3 "$@" > $O/$name.out 2> $O/$name.err
}
run kt --kernel-trace -d $O/p_kt -o kt -- $BENCH
run kt_plain --kernel-trace -d $O/p_ktp -o kt -- $BENCH --no-prepare
run fetch --pmc FETCH_SIZE -d $O/p_fetch -o f -- $BENCH1
run write --pmc WRITE_SIZE -d $O/p_write -o w -- $BENCH1
run fetch_plain --pmc FETCH_SIZE -d $O/p_fetchp -o f -- $BENCH1 --no-prepare
run write_plain --pmc WRITE_SIZE -d $O/p_writep -o w -- $BENCH1 --no-prepare
run sq --pmc $SQ -d $O/p_sq -o s -- $BENCH1
run g2_kt --kernel-trace -d $O/p_g2kt -o kt -- $G2
run g2_sq --pmc $SQ -d $O/p_g2sq -o s -- $G2
cd $R
db() { find $O/$1 -name "*results.db" | head -1; }
python tools/rocpd_stats.py $(db p_kt) --min-us 1000 > $O/kernel_stats.txt 2>> $O/post.err
python tools/rocpd_stats.py $(db p_ktp) --min-us 1000 > $O/kernel_stats_plain.txt 2>> $O/post.err
python tools/rocpd_stats.py $(db p_g2kt) --min-us 1000 > $O/kernel_stats_g2.txt 2>> $O/post.err
python tools/rocpd_stats.py $(db p_fetch) --pmc --min-us 100 > $O/pmc_fetch.txt 2>> $O/post.err
python tools/rocpd_stats.py $(db p_write) --pmc --min-us 100 > $O/pmc_write.txt 2>> $O/post.err
python tools/rocpd_stats.py $(db p_sq) --pmc --min-us 100 > $O/pmc_sq.txt 2>> $O/post.err
python tools/rocpd_stats.py $(db p_g2sq) --pmc --min-us 100 > $O/pmc_sq_g2.txt 2>> $O/post.err
python tools/pmc_traffic.py $(db p_fetch) $(db p_write) 24 22 > $O/pmc_traffic.json 2>> $O/post.err
python tools/pmc_traffic.py $(db p_fetchp) $(db p_writep) 24 22 > $O/pmc_traffic_plain.json 2>> $O/post.err
rm -rf $O/p_*
# N = 2 code path of bench.py on this one GPU (gloo carries the combine; never a reported number)
ARK_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 \
  bench.py --gpus 2 --steps 2 --warmup 1 --log-n 21 --fft-steps 2 --fft-log-n 16 > $O/bench_n2_gloo.json 2> $O/bench_n2_gloo.err
echo done > $O/done
