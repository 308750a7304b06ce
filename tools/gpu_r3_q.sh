#!/bin/bash
# round 3, session Q: product rates incl. the asm-chained 28-bit product
mkdir -p gpurun_out/r3q
O=$PWD/gpurun_out/r3q
(cd algebra_amd/csrc/ubench && timeout 300 ./mulbench.bin > $O/mulbench.txt 2>&1)
echo done > $O/done
