#!/bin/bash
# round 3, session R: asm-chained 28-bit products in the accumulate kernels: device-form parity, A/B, product rates
mkdir -p gpurun_out/r3r
O=$PWD/gpurun_out/r3r
export TMPDIR=/tmp
(timeout 600 python -m pytest tests/test_gpu_field_points.py tests/test_gpu_msm.py -m gpu -q -x 2>&1 | tail -6) > $O/tests_quick.log
(cd algebra_amd/csrc/ubench && timeout 300 ./mulbench.bin > $O/mulbench.txt 2>&1)
for rep in 1 2; do
for lz in 1 0; do
  (echo "== LAZY=$lz 2^24"; ARK_HIP_MSM_LAZY=$lz timeout 300 python tools/msm_bench.py BLS12_381_G1 24 3 both) >> $O/ab.txt 2>> $O/ab.err
done
done
(echo "== LAZY=1 2^20"; timeout 300 python tools/msm_bench.py BLS12_381_G1 20 5 both) >> $O/ab.txt 2>> $O/ab.err
(echo "== LAZY=1 377 2^22"; timeout 300 python tools/msm_bench.py BLS12_377_G1 22 3 both) >> $O/ab.txt 2>> $O/ab.err
(timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -6) > $O/tests.log
echo done > $O/done
