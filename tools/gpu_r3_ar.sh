#!/bin/bash
# session AR: pass-A tile size again on the batched-load, XCD-aware kernels (16384-key tiles below 2^26?)
mkdir -p gpurun_out/r3ar
O=$PWD/gpurun_out/r3ar
export TMPDIR=/tmp
for rep in 1 2; do
for t in 8192 16384; do
  for cfg in "BLS12_381_G1 24" "BLS12_381_G1 22" "BLS12_381_G1 20" "BLS12_381_G1 25"; do
    (echo "== tile $t $cfg"; ARK_HIP_MSM_TILE=$t ARK_HIP_LIB=$PWD/algebra_amd/variants/b_xcd.so timeout 300 python tools/msm_bench.py $cfg 5 plain | grep -v amdgpu.ids) >> $O/ab.txt 2>> $O/ab.err
  done
done
done
echo done > $O/done
