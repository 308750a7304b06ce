#!/bin/bash
# round 3, session Z: reduction geometry (level-0 chunk x bit-stage chunk) on the 28-bit reduction kernels
mkdir -p gpurun_out/r3z
O=$PWD/gpurun_out/r3z
export TMPDIR=/tmp
for cfg in "BLS12_381_G1 24 plain" "BLS12_381_G1 24 prepared" "BLS12_381_G1 20 plain" "BLS12_381_G1 26 plain"; do
  (timeout 600 python tools/reduce_sweep.py $cfg >> $O/reduce_sweep.txt) 2>> $O/err.txt
done
echo done > $O/done
