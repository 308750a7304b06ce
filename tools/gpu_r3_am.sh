#!/bin/bash
# session AM: partition sort traffic -- pass A's output as u32 index|sign + u16 bucket bits (pass B's counting sweep reads
# 2 B per entry), histogram pass writing 32-byte pieces (8 tiles per workgroup)
mkdir -p gpurun_out/r3am
O=$PWD/gpurun_out/r3am
export TMPDIR=/tmp
for rep in 1 2; do
for v in a_before b_split c_split_hist; do
  for cfg in "BLS12_381_G1 24" "BLS12_381_G1 20" "BLS12_381_G1 26"; do
    (echo "== $v $cfg"; ARK_HIP_LIB=$PWD/algebra_amd/variants/$v.so timeout 300 python tools/msm_bench.py $cfg 5 plain | grep -v amdgpu.ids) >> $O/ab.txt 2>> $O/ab.err
  done
done
done
(ARK_HIP_LIB=$PWD/algebra_amd/variants/c_split_hist.so timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_msm_prepared.py tests/test_gpu_configs.py tests/test_gpu_trait_surface.py -m gpu -q -x 2>&1 | tail -5) > $O/tests.log
echo done > $O/done
