#!/bin/bash
# session AH: instruction issue rates (ubench/issuebench.hip); host tail as one Horner over bit positions, A/B at small n;
# parity of the new tail (MSM test files)
mkdir -p gpurun_out/r3ah
O=$PWD/gpurun_out/r3ah
export TMPDIR=/tmp
timeout 120 algebra_amd/csrc/ubench/issuebench.bin > $O/issue.txt 2> $O/issue.err
for rep in 1 2; do
for v in algebra_amd/variants/a_before.so algebra_amd/variants/b_tail.so; do
  for cfg in "BN254_G1 16" "BLS12_381_G1 16" "BLS12_381_G1 20" "BLS12_377_G2 16"; do
    (echo "== $v $cfg"; ARK_HIP_LIB=$PWD/$v timeout 300 python tools/msm_bench.py $cfg 20 plain | grep -v amdgpu.ids) >> $O/ab.txt 2>> $O/ab.err
  done
done
done
(timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_msm_prepared.py tests/test_gpu_trait_surface.py -m gpu -q -x 2>&1 | tail -5) > $O/tests.log
echo done > $O/done
