#!/bin/bash
# round 3, session D: accumulate kernels on 28-bit limbs (A/B against the saturated kernels), parity, narrow-scalar sweep
mkdir -p gpurun_out/r3d
O=$PWD/gpurun_out/r3d
export TMPDIR=/tmp
(timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -30) > $O/tests.log
for lz in 1 0; do
  for cfg in "BLS12_381_G1 24" "BLS12_381_G1 20" "BLS12_381_G1 16" "BLS12_377_G1 22" "BLS12_381_G1 26"; do
    (echo "== ARK_HIP_MSM_LAZY=$lz $cfg"; ARK_HIP_MSM_LAZY=$lz timeout 300 python tools/msm_bench.py $cfg 3 both) >> $O/lazy_ab.txt 2>> $O/lazy_ab.err
  done
done
(timeout 300 python tools/small_scalar_bench.py > $O/small_scalar.txt) 2> $O/small_scalar.err
for c in 12 14 16 17 18 20 22; do
  (echo "== u64-direct c=$c"; ARK_HIP_MSM_C=$c timeout 120 python tools/small_scalar_bench.py 20 2>/dev/null | grep -E "^u64|^u32") >> $O/small_c_sweep.txt
done
echo done > $O/done
