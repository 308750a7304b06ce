#!/bin/bash
# GPU session AD of round 2: fewer launches per MSM (single-kernel scans for small arrays, threshold inside find_heavy, one memset)
mkdir -p gpurun_out/r2ad
O=$PWD/gpurun_out/r2ad
export TMPDIR=/tmp
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6) > $O/tests.log
(timeout 200 python tools/soak.py 45 > $O/soak.txt) 2> $O/soak.err
for cfg in "BN254_G1 16 30" "BLS12_381_G1 14 30" "BLS12_381_G1 16 30" "BLS12_381_G1 18 20" "BLS12_381_G1 20 10" "BLS12_381_G1 24 3"; do
  set -- $cfg
  timeout 300 python tools/msm_bench.py $1 $2 $3 both >> $O/small.txt 2>> $O/err.txt
done
echo done > $O/done
