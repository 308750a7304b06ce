#!/bin/bash
mkdir -p gpurun_out/r2g1c
O=$PWD/gpurun_out/r2g1c
for cv in BLS12_381_G1 BN254_G1; do
for k in 10 12 14; do
  for c in 11 12 13 14 15 16 17; do
    ARK_HIP_MSM_C_PREPARED=$c timeout 200 python tools/msm_bench.py $cv $k 20 prepared >> $O/g1c.txt 2>> $O/err.txt
  done
done
done
echo done > $O/done
