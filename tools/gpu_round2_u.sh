#!/bin/bash
# GPU session U of round 2: does a second MSM's sort overlap the first one's accumulate? (two contexts on one GPU); MSM at 2^27 / 2^28
mkdir -p gpurun_out/r2u
O=$PWD/gpurun_out/r2u
export TMPDIR=/tmp
ARK_HIP_OVERSUBSCRIBE=1 timeout 600 python tools/overlap_probe.py 24 6 > $O/overlap.txt 2> $O/err.txt
ARK_HIP_OVERSUBSCRIBE=1 timeout 600 python tools/overlap_probe.py 20 40 >> $O/overlap.txt 2>> $O/err.txt
timeout 900 python tools/msm_bench.py BLS12_381_G1 27 1 plain > $O/big.txt 2>> $O/err.txt
timeout 900 python tools/msm_bench.py BLS12_381_G1 28 1 plain >> $O/big.txt 2>> $O/err.txt
echo done > $O/done
