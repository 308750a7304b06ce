#!/bin/bash
# round 3, session P: batched-affine accumulation priced against the XYZZ mixed addition (csrc/ubench/affbench.hip)
mkdir -p gpurun_out/r3p
O=$PWD/gpurun_out/r3p
(cd algebra_amd/csrc/ubench && timeout 600 ./affbench.bin > $O/affbench.txt 2>&1)
echo done > $O/done
