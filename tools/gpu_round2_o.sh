#!/bin/bash
# GPU session O of round 2: reduction geometry sweep; parked accumulator (3 waves) on top of the new product
mkdir -p gpurun_out/r2o
O=$PWD/gpurun_out/r2o
R=$PWD
export TMPDIR=/tmp
timeout 600 python tools/reduce_sweep.py BLS12_381_G1 24 prepared > $O/reduce_sweep.txt 2>> $O/err.txt
timeout 600 python tools/reduce_sweep.py BLS12_381_G1 24 plain >> $O/reduce_sweep.txt 2>> $O/err.txt
timeout 600 python tools/reduce_sweep.py BLS12_377_G2 22 prepared >> $O/reduce_sweep.txt 2>> $O/err.txt
for lib in libark_hip_w3y.so libark_hip.so; do
  echo "## $lib" >> $O/park.txt
  ARK_HIP_LIB=$R/algebra_amd/$lib timeout 400 python tools/msm_bench.py BLS12_381_G1 24 3 both >> $O/park.txt 2>> $O/err.txt
done
echo done > $O/done
