#!/bin/bash
# GPU session B of round 2: full parity suite, sweeps after the lookahead / sort-split / chunk-tree changes, kernel trace.
mkdir -p gpurun_out/r2b
O=gpurun_out/r2b
export TMPDIR=/tmp
(timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -60) > $O/tests.log
for s in "24 5" "20 10" "16 20" "23 5" "26 2"; do
  timeout 400 python tools/msm_bench.py BLS12_381_G1 $s both >> $O/sweep.txt 2>> $O/sweep.err
done
for c in 20 22 24; do
  ARK_HIP_MSM_C_PREPARED=$c timeout 300 python tools/msm_bench.py BLS12_381_G1 24 3 prepared >> $O/sweep_c.txt 2>> $O/sweep.err
done
for l in 8 16 32; do
  ARK_HIP_MSM_L0=$l timeout 300 python tools/msm_bench.py BLS12_381_G1 24 3 prepared >> $O/sweep_l0.txt 2>> $O/sweep.err
done
timeout 300 python tools/msm_bench.py BLS12_377_G2 22 3 both >> $O/sweep_g2.txt 2>> $O/sweep.err
timeout 300 python tools/msm_bench.py BN254_G1 24 3 both >> $O/sweep_g2.txt 2>> $O/sweep.err
R=$PWD
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d $R/$O/prof -o r2b -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras > $R/$O/prof_bench.json 2> $R/$O/prof.err
cd $R
DB=$(find $O/prof -name "*results.db" | head -1)
python tools/rocpd_stats.py $DB --min-us 1000 > $O/kernel_stats.txt 2>> $O/prof.err
rm -rf $O/prof
echo done > $O/done
