#!/bin/bash
mkdir -p gpurun_out/r2k
O=gpurun_out/r2k
export TMPDIR=/tmp
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -30) > $O/tests.log
timeout 400 python tools/msm_bench.py BLS12_377_G2 22 3 both >> $O/sweep.txt 2>> $O/err.txt
timeout 400 python tools/msm_bench.py BLS12_381_G2 20 3 both >> $O/sweep.txt 2>> $O/err.txt
timeout 400 python tools/msm_bench.py BLS12_381_G1 24 3 both >> $O/sweep.txt 2>> $O/err.txt
echo done > $O/done
