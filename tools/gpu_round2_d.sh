#!/bin/bash
# GPU session D of round 2: relaxed / lane-pair reduce + heavy kernels
mkdir -p gpurun_out/r2d
O=gpurun_out/r2d
export TMPDIR=/tmp
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -40) > $O/tests.log
timeout 300 python tools/msm_bench.py BLS12_377_G2 22 3 both >> $O/sweep_g2.txt 2>> $O/sweep.err
timeout 300 python tools/msm_bench.py BLS12_381_G2 20 3 both >> $O/sweep_g2.txt 2>> $O/sweep.err
timeout 300 python tools/msm_bench.py BLS12_377_G2 18 5 both >> $O/sweep_g2.txt 2>> $O/sweep.err
for s in "24 5" "20 10" "16 20"; do
  timeout 400 python tools/msm_bench.py BLS12_381_G1 $s both >> $O/sweep.txt 2>> $O/sweep.err
done
timeout 300 python tools/msm_bench.py BN254_G1 24 3 both >> $O/sweep.txt 2>> $O/sweep.err
echo done > $O/done
