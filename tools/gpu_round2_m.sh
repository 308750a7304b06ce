#!/bin/bash
# GPU session M of round 2: planner check at 2^23/2^25 + accumulator-parking experiment (3 waves per SIMD)
mkdir -p gpurun_out/r2m
O=$PWD/gpurun_out/r2m
R=$PWD
export TMPDIR=/tmp
for lib in libark_hip_w3.so libark_hip_w3y.so libark_hip.so; do
  echo "## $lib" >> $O/park.txt
  ARK_HIP_LIB=$R/algebra_amd/$lib timeout 400 python tools/msm_bench.py BLS12_381_G1 24 3 both >> $O/park.txt 2>> $O/err.txt
  ARK_HIP_LIB=$R/algebra_amd/$lib timeout 400 python tools/msm_bench.py BLS12_381_G1 20 5 both >> $O/park.txt 2>> $O/err.txt
done
echo "## planner: default vs forced" >> $O/plan.txt
timeout 300 python tools/msm_bench.py BLS12_381_G1 23 3 prepared >> $O/plan.txt 2>> $O/err.txt
ARK_HIP_MSM_C_PREPARED=20 timeout 300 python tools/msm_bench.py BLS12_381_G1 23 3 prepared >> $O/plan.txt 2>> $O/err.txt
timeout 300 python tools/msm_bench.py BN254_G1 23 3 prepared >> $O/plan.txt 2>> $O/err.txt
ARK_HIP_MSM_C_PREPARED=20 timeout 300 python tools/msm_bench.py BN254_G1 23 3 prepared >> $O/plan.txt 2>> $O/err.txt
timeout 300 python tools/msm_bench.py BLS12_381_G1 25 2 prepared >> $O/plan.txt 2>> $O/err.txt
ARK_HIP_MSM_C_PREPARED=22 timeout 300 python tools/msm_bench.py BLS12_381_G1 25 2 prepared >> $O/plan.txt 2>> $O/err.txt
echo done > $O/done
