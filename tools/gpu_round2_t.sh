#!/bin/bash
# GPU session T of round 2: lane-batched inversion (table of a prepared set, batch_mul, normalize_batch): parity, then before / after
mkdir -p gpurun_out/r2t
O=$PWD/gpurun_out/r2t
R=$PWD
export TMPDIR=/tmp
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -8) > $O/tests.log
for lib in libark_hip_prev.so libark_hip.so; do
  echo "## $lib" >> $O/aux.txt
  ARK_HIP_LIB=$R/algebra_amd/$lib timeout 300 python tools/aux_bench.py BLS12_381_G1 20 >> $O/aux.txt 2>> $O/err.txt
  ARK_HIP_LIB=$R/algebra_amd/$lib timeout 300 python tools/aux_bench.py BLS12_381_G1 24 >> $O/aux.txt 2>> $O/err.txt
  ARK_HIP_LIB=$R/algebra_amd/$lib timeout 300 python tools/aux_bench.py BLS12_377_G2 20 >> $O/aux.txt 2>> $O/err.txt
done
timeout 300 python tools/msm_bench.py BLS12_381_G1 24 3 prepared >> $O/aux.txt 2>> $O/err.txt
echo done > $O/done
