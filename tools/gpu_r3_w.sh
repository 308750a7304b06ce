#!/bin/bash
# round 3, session W: window sizes around the planner's choice after the 28-bit kernels (accumulate and reduction rates changed)
mkdir -p gpurun_out/r3w
O=$PWD/gpurun_out/r3w
export TMPDIR=/tmp
(timeout 900 python tools/c_sweep.py BLS12_381_G1 16,18,20,22,23,24 both > $O/c_sweep_381.txt) 2> $O/err.txt
(timeout 600 python tools/c_sweep.py BLS12_377_G1 20,22 both > $O/c_sweep_377.txt) 2>> $O/err.txt
echo done > $O/done
