#!/bin/bash
mkdir -p gpurun_out/r2j
O=gpurun_out/r2j
export TMPDIR=/tmp
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -30) > $O/tests.log
(timeout 300 python tools/diag_u32.py > $O/diag_u32.txt) 2>> $O/err.txt
(timeout 600 python tools/small_scalar_bench.py 20 > $O/distributions.txt) 2>> $O/err.txt
timeout 400 python tools/msm_bench.py BLS12_381_G1 24 5 both >> $O/sweep.txt 2>> $O/err.txt
timeout 400 python tools/msm_bench.py BLS12_377_G2 22 3 both >> $O/sweep.txt 2>> $O/err.txt
echo done > $O/done
