#!/bin/bash
# round 3, session M: lazy (28-bit) against saturated accumulate kernels of the shipped build across sizes and on BLS12-377 G1
mkdir -p gpurun_out/r3m
O=$PWD/gpurun_out/r3m
export TMPDIR=/tmp
for cfg in "BLS12_377_G1 22" "BLS12_381_G1 16" "BLS12_381_G1 20" "BLS12_381_G1 23" "BLS12_381_G1 26"; do
  for lz in 1 0; do
    (echo "== LAZY=$lz $cfg"; ARK_HIP_MSM_LAZY=$lz timeout 300 python tools/msm_bench.py $cfg 3 both) >> $O/ab.txt 2>> $O/ab.err
  done
done
echo done > $O/done
