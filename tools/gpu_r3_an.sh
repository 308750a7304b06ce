#!/bin/bash
# session AN: SQ counters of the shipped kernels (plain 2^24 step + FFT): VALU instruction count and busy fraction of the
# 28-bit-limb accumulate / reduce kernels, wait states of the sort kernels
mkdir -p gpurun_out/r3an
O=$PWD/gpurun_out/r3an
R=$PWD
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_LDS -d $O/p_sq -o sq -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras --fft-steps 1 > $O/sq.out 2> $O/sq.err
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_WAIT_INST_LDS -d $O/p_sq2 -o sq2 -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras --fft-steps 1 > $O/sq2.out 2> $O/sq2.err
cd $R
db() { find $O/$1 -name "*results.db" | head -1; }
python tools/rocpd_stats.py $(db p_sq) --pmc --min-us 100 > $O/pmc_sq.txt 2>> $O/post.err
python tools/rocpd_stats.py $(db p_sq2) --pmc --min-us 100 > $O/pmc_sq2.txt 2>> $O/post.err
rm -rf $O/p_sq $O/p_sq2
echo done > $O/done
