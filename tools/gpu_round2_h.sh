#!/bin/bash
mkdir -p gpurun_out/r2h
O=gpurun_out/r2h
export TMPDIR=/tmp
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -30) > $O/tests.log
(timeout 300 python tools/diag_u32.py > $O/diag_u32.txt) 2>> $O/err.txt
for s in "16 20" "18 20" "20 10" "22 5" "24 5"; do
  timeout 400 python tools/msm_bench.py BLS12_381_G1 $s both >> $O/sweep.txt 2>> $O/err.txt
done
timeout 300 python tools/msm_bench.py BLS12_377_G2 22 3 both >> $O/sweep.txt 2>> $O/err.txt
timeout 300 python tools/msm_bench.py BLS12_377_G2 18 5 both >> $O/sweep.txt 2>> $O/err.txt
echo done > $O/done
