#!/bin/bash
# GPU session L of round 2: soak of the new paths, full bench line (all extras), BN254 window check, refreshed G2 profile
mkdir -p gpurun_out/r2l
O=$PWD/gpurun_out/r2l
R=$PWD
export TMPDIR=/tmp
(timeout 400 python tools/soak.py 240 > $O/soak.txt) 2> $O/soak.err
(timeout 900 python bench.py --steps 10 --warmup 2 > $O/bench.json) 2> $O/bench.err
for c in 20 21 22; do
  ARK_HIP_MSM_C_PREPARED=$c timeout 300 python tools/msm_bench.py BN254_G1 24 3 prepared >> $O/bn254_c.txt 2>> $O/err.txt
done
cd /tmp
G2="python $R/tools/msm_bench.py BLS12_377_G2 22 2 both"
SQ="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"
timeout 600 rocprofv3 --kernel-trace -d $O/p_g2kt -o kt -- $G2 > $O/g2_kt.out 2> $O/g2_kt.err
timeout 600 rocprofv3 --pmc $SQ -d $O/p_g2sq -o s -- $G2 > $O/g2_sq.out 2> $O/g2_sq.err
cd $R
db() { find $O/$1 -name "*results.db" | head -1; }
python tools/rocpd_stats.py $(db p_g2kt) --min-us 1000 > $O/kernel_stats_g2.txt 2>> $O/err.txt
python tools/rocpd_stats.py $(db p_g2sq) --pmc --min-us 100 > $O/pmc_sq_g2.txt 2>> $O/err.txt
rm -rf $O/p_*
echo done > $O/done
