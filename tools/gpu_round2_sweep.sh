#!/bin/bash
# final size x curve sweep of the shipped build (every line checked against k*G)
mkdir -p gpurun_out/r2sweep
O=$PWD/gpurun_out/r2sweep
export TMPDIR=/tmp
for c in BN254_G1 BLS12_381_G1 BLS12_377_G1; do
  for k in 12 16 18 20 22 24; do
    steps=20; [ $k -ge 22 ] && steps=3
    timeout 400 python tools/msm_bench.py $c $k $steps both >> $O/sweep.txt 2>> $O/err.txt
  done
done
timeout 600 python tools/msm_bench.py BLS12_381_G1 26 2 both >> $O/sweep.txt 2>> $O/err.txt
for c in BLS12_381_G2 BLS12_377_G2; do
  for k in 12 16 18 20 22; do
    steps=10; [ $k -ge 22 ] && steps=3
    timeout 400 python tools/msm_bench.py $c $k $steps both >> $O/sweep.txt 2>> $O/err.txt
  done
done
timeout 300 python tools/small_scalar_bench.py 20 > $O/dists.txt 2>> $O/err.txt
echo done > $O/done
