#!/bin/bash
# quick A/B on ONE box (boxes differ by several percent): every variant library x ARK_HIP_MSM_LAZY in {1, 0}
# usage: tools/gpu_ab.sh TAG "CURVE LOGN" [mode]
tag=$1; cfg=${2:-"BLS12_381_G1 24"}; mode=${3:-plain}
mkdir -p gpurun_out/$tag
O=$PWD/gpurun_out/$tag
export TMPDIR=/tmp
for rep in 1 2; do
for v in $(ls algebra_amd/variants/*.so); do
  for lz in 1 0; do
    (echo "== $v LAZY=$lz $cfg"; ARK_HIP_MSM_LAZY=$lz ARK_HIP_LIB=$PWD/$v timeout 300 python tools/msm_bench.py $cfg 3 $mode) >> $O/ab.txt 2>> $O/ab.err
  done
done
done
echo done > $O/done
