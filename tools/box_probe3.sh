#!/bin/bash
# Round 6: is the "slow box" mode (profiles/r5_box_states.txt: about one session in seven runs the latency-bound kernels 1.5-2 x
# slower) tied to the scratch stack frame of the out-of-line doubling?  On ONE box, back to back: the round-5 kernels (development
# variant built from commit 3d2e131: scratch 240-368 B in the accumulate / reduce / heavy kernels) and this build's (scratch 0).
#   gpurun -- 'bash tools/box_probe3.sh TAG'   ->  gpurun_out/TAG/probe.txt
out=gpurun_out/$1; mkdir -p $out
r5=$PWD/algebra_amd/variants/libark_hip_r5.so
for rep in 1 2; do
  for lib in $r5 $PWD/algebra_amd/libark_hip.so; do
    for lg in 21 16; do
      echo "== $(basename $lib) 2^$lg"
      ARK_HIP_LIB=$lib python tools/msm_bench.py BLS12_381_G1 $lg 6 plain 2>&1 | grep -v amdgpu.ids
    done
  done
done > $out/probe.txt 2>&1
if [ -n "$2" ]; then   # also the headline size, both builds, twice
  for rep in 1 2; do for lib in $r5 $PWD/algebra_amd/libark_hip.so; do
    echo "== $(basename $lib) 2^24"
    ARK_HIP_LIB=$lib python tools/msm_bench.py BLS12_381_G1 24 3 plain 2>&1 | grep -v amdgpu.ids
  done; done >> $out/probe.txt 2>&1
fi
red=$(grep -A1 "libark_hip_r5.so 2^21" $out/probe.txt | sed -n 's/.*reduce \([0-9.]*\)\].*/\1/p' | head -1)
if python3 -c "import sys; sys.exit(0 if float('${red:-0}') > 1.5 else 1)"; then
  echo "SLOW BOX for the round-5 kernels (reduce 2^21 = $red ms)" >> $out/probe.txt
  # a kernel trace of each build on this box
  for lib in $r5 $PWD/algebra_amd/libark_hip.so; do
    export ARK_HIP_LIB=$lib
    suffix=_$(basename $lib .so) KT_TIMELINE=40 bash tools/gpu_session.sh $1 ktpy:msm_bench.py:BLS12_381_G1:21:3:plain
  done
else
  echo "normal box for the round-5 kernels (reduce 2^21 = $red ms)" >> $out/probe.txt
fi
cat $out/probe.txt
