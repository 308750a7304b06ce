#!/bin/bash
# round 3, session AA: two window groups inside one MSM (second group's sort under the first group's accumulate kernel)
mkdir -p gpurun_out/r3aa
O=$PWD/gpurun_out/r3aa
export TMPDIR=/tmp
(timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -8) > $O/tests.log
for rep in 1 2; do
for cfg in "BLS12_381_G1 24" "BLS12_381_G1 20" "BLS12_381_G1 22" "BLS12_381_G1 26" "BN254_G1 24" "BLS12_377_G2 22"; do
  for g in 2 1; do
    (echo "== GROUPS=$g $cfg"; ARK_HIP_MSM_GROUPS=$g timeout 300 python tools/msm_bench.py $cfg 3 plain) >> $O/ab.txt 2>> $O/ab.err
  done
done
done
echo done > $O/done
