#!/bin/bash
# round 3, session X: the re-fitted plain-path planner against its neighbours at small n and on BN254 / BLS12-377 G1
mkdir -p gpurun_out/r3x
O=$PWD/gpurun_out/r3x
export TMPDIR=/tmp
(timeout 600 python tools/c_sweep.py BN254_G1 12,14,16,18,20 plain > $O/c_sweep_bn254.txt) 2> $O/err.txt
(timeout 600 python tools/c_sweep.py BLS12_381_G1 10,12,14,15,17,19 plain > $O/c_sweep_381_small.txt) 2>> $O/err.txt
(timeout 600 python tools/c_sweep.py BLS12_377_G1 16,20,22 plain > $O/c_sweep_377.txt) 2>> $O/err.txt
echo done > $O/done
