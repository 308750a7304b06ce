#!/bin/bash
# level-0 chunk 1 / 2 (no running sums at all) and shorter bit-stage chunks for small bucket counts
mkdir -p gpurun_out/r2l0
O=$PWD/gpurun_out/r2l0
for cfg in "BLS12_377_G2 12" "BLS12_377_G2 16" "BLS12_381_G1 12" "BLS12_381_G1 16" "BN254_G1 16" "BLS12_381_G1 20"; do
  set -- $cfg
  for l0 in 1 2 4 8; do
    for ch in 256 1024; do
      echo -n "L0=$l0 chunk=$ch  " >> $O/l0.txt
      ARK_HIP_MSM_L0=$l0 ARK_HIP_MSM_CHUNK=$ch timeout 200 python tools/msm_bench.py $1 $2 20 prepared >> $O/l0.txt 2>> $O/err.txt
    done
  done
done
echo done > $O/done
