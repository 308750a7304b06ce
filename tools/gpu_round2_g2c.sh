#!/bin/bash
mkdir -p gpurun_out/r2g2c
O=$PWD/gpurun_out/r2g2c
for k in 12 16 18; do
  for c in 13 15 16 17 18 19; do
    ARK_HIP_MSM_C_PREPARED=$c timeout 200 python tools/msm_bench.py BLS12_377_G2 $k 10 prepared >> $O/g2c.txt 2>> $O/err.txt
  done
done
echo done > $O/done
