#!/bin/bash
# A/B of the host tail's helper pool across sizes on one box, interleaved three times:
#   pool for every job with >= 8 windows (default)  |  pool for short jobs only (ARK_HIP_HOST_TAIL_ALL=0: rounds 5-6a)  |  no pool
out=gpurun_out/$1; mkdir -p $out
for rep in 1 2 3; do for mode in "ARK_HIP_HOST_TAIL_ALL=1" "ARK_HIP_HOST_TAIL_ALL=0" "ARK_HIP_HOST_TAIL_THREADS=0"; do for lg in 16 19 20 21 22; do
  echo "== $mode 2^$lg"
  env $mode python tools/msm_bench.py BLS12_381_G1 $lg 20 plain 2>&1 | grep -v amdgpu.ids
done; done; done > $out/tail_ab.txt 2>&1
cat $out/tail_ab.txt
