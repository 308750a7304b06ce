#!/bin/bash
# session AJ: batch_mul -- table by num_scalars (12 / 16-bit rows), 28-bit-limb batch kernel, host-side doubling chain,
# lane-batched table normalisation; before/after on one box + the new tests
mkdir -p gpurun_out/r3aj
O=$PWD/gpurun_out/r3aj
export TMPDIR=/tmp
for v in algebra_amd/variants/a_before.so algebra_amd/libark_hip.so; do
  (echo "== $v"; ARK_HIP_LIB=$PWD/$v timeout 600 python tools/batchmul_bench.py BLS12_381_G1 4 10 16 20 22 24 | grep -v amdgpu.ids
   ARK_HIP_LIB=$PWD/$v timeout 600 python tools/batchmul_bench.py BLS12_377_G2 10 16 20 | grep -v amdgpu.ids
   ARK_HIP_LIB=$PWD/$v timeout 600 python tools/batchmul_bench.py BN254_G1 16 20 | grep -v amdgpu.ids) >> $O/ab.txt 2>> $O/ab.err
done
for w in 8 12 16; do
  (echo "== new, ARK_HIP_BATCHMUL_WINDOW=$w"; ARK_HIP_BATCHMUL_WINDOW=$w timeout 600 python tools/batchmul_bench.py BLS12_381_G1 4 16 20 22 24 | grep -v amdgpu.ids) >> $O/ab.txt 2>> $O/ab.err
done
(echo "== new, ARK_HIP_MSM_LAZY=0"; ARK_HIP_MSM_LAZY=0 timeout 600 python tools/batchmul_bench.py BLS12_381_G1 20 24 | grep -v amdgpu.ids) >> $O/ab.txt 2>> $O/ab.err
(timeout 900 python -m pytest tests/test_gpu_msm_prepared.py -m gpu -q -x -k batch_mul 2>&1 | tail -8) > $O/tests.log
echo done > $O/done
