#!/bin/bash
# round 3, session F: lazy accumulate kernel variants (accumulator kept out of scratch; 1 vs 2 waves per SIMD), product rates
mkdir -p gpurun_out/r3f
O=$PWD/gpurun_out/r3f
export TMPDIR=/tmp
(cd algebra_amd/csrc/ubench && timeout 120 ./mulbench.bin > $O/mulbench.txt 2>&1)
for v in $(ls algebra_amd/variants/*.so); do
  for cfg in "BLS12_381_G1 24" "BLS12_381_G1 20"; do
    (echo "== $v $cfg"; ARK_HIP_LIB=$PWD/$v timeout 300 python tools/msm_bench.py $cfg 3 both) >> $O/variants.txt 2>> $O/variants.err
  done
done
echo done > $O/done
