#!/bin/bash
# round 3, session N: super-bucket bits of the partition sort at 2^25 / 2^26 (pass A histogram size against direct placement in pass B)
mkdir -p gpurun_out/r3n
O=$PWD/gpurun_out/r3n
export TMPDIR=/tmp
V=$PWD/algebra_amd/variants/libark_hip_hb.so
for cfg in "BLS12_381_G1 26" "BLS12_381_G1 25"; do
  for hb in default 9 10; do
    (echo "== HB=$hb $cfg"; if [ $hb = default ]; then unset ARK_HIP_MSM_HB; else export ARK_HIP_MSM_HB=$hb; fi; ARK_HIP_LIB=$V timeout 300 python tools/msm_bench.py $cfg 3 plain) >> $O/hb.txt 2>> $O/hb.err
  done
done
echo done > $O/done
