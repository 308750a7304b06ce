#!/bin/bash
# session AS: wave-shuffle block scans in the scatter / finish kernels, alone (the first attempt was bundled with an LDS
# preload of the run starts and came out slower)
mkdir -p gpurun_out/r3as
O=$PWD/gpurun_out/r3as
export TMPDIR=/tmp
for rep in 1 2 3; do
for v in a_xcd b_shfl; do
  for cfg in "BLS12_381_G1 24" "BLS12_381_G1 20" "BLS12_381_G1 26"; do
    (echo "== $v $cfg"; ARK_HIP_LIB=$PWD/algebra_amd/variants/$v.so timeout 300 python tools/msm_bench.py $cfg 5 plain | grep -v amdgpu.ids) >> $O/ab.txt 2>> $O/ab.err
  done
done
done
(ARK_HIP_LIB=$PWD/algebra_amd/variants/b_shfl.so timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_msm_prepared.py tests/test_gpu_configs.py -m gpu -q -x 2>&1 | tail -3) > $O/tests.log
echo done > $O/done
