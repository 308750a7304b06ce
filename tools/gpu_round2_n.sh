#!/bin/bash
# GPU session N of round 2: product with one-statement columns / carry-initialised top word: ubench, parity, end to end
mkdir -p gpurun_out/r2n
O=$PWD/gpurun_out/r2n
export TMPDIR=/tmp
echo "## old product (HEAD)" > $O/mulbench.txt
timeout 120 algebra_amd/csrc/ubench/mulbench_old.bin | grep sat32 >> $O/mulbench.txt 2>> $O/err.txt
echo "## new product" >> $O/mulbench.txt
timeout 120 algebra_amd/csrc/ubench/mulbench.bin | grep sat32 >> $O/mulbench.txt 2>> $O/err.txt
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15) > $O/tests.log
timeout 400 python tools/msm_bench.py BLS12_381_G1 24 3 both >> $O/sweep.txt 2>> $O/err.txt
timeout 400 python tools/msm_bench.py BLS12_381_G1 20 5 both >> $O/sweep.txt 2>> $O/err.txt
timeout 400 python tools/msm_bench.py BN254_G1 16 5 both >> $O/sweep.txt 2>> $O/err.txt
timeout 400 python tools/msm_bench.py BLS12_377_G2 22 3 both >> $O/sweep.txt 2>> $O/err.txt
timeout 200 python tools/fft_bench.py 22 >> $O/sweep.txt 2>> $O/err.txt
echo done > $O/done
