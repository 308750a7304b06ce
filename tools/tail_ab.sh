#!/bin/bash
# A/B of the host tail's helper pool across sizes on one box: ARK_HIP_HOST_TAIL_THREADS=7 (default) against 0, interleaved twice.
out=gpurun_out/$1; mkdir -p $out
for rep in 1 2; do for t in 7 0; do for lg in 16 18 19 20 21 22 24; do
  echo "== ARK_HIP_HOST_TAIL_THREADS=$t 2^$lg"
  ARK_HIP_HOST_TAIL_THREADS=$t python tools/msm_bench.py BLS12_381_G1 $lg 10 plain 2>&1 | grep -v amdgpu.ids
done; done; done > $out/tail_ab.txt 2>&1
cat $out/tail_ab.txt
