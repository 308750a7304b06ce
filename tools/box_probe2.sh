#!/bin/bash
# If this box is one of the slow-reduction ones (tools/box_probe.sh), look closer: kernel timeline of one 2^21 MSM.
out=gpurun_out/$1; mkdir -p $out
python tools/msm_bench.py BLS12_381_G1 21 6 plain 2>&1 | grep -v amdgpu.ids > $out/first.txt
cat $out/first.txt
red=$(sed -n 's/.*reduce \([0-9.]*\)\].*/\1/p' $out/first.txt | head -1)
if python3 -c "import sys; sys.exit(0 if float('$red') > 1.3 else 1)"; then
  echo "SLOW BOX (reduce $red)"
  KT_TIMELINE=40 bash tools/gpu_session.sh $1 ktpy:msm_bench.py:BLS12_381_G1:21:3:plain
  for v in "HSA_ENABLE_SCRATCH_ASYNC_RECLAIM=0" "HSA_SCRATCH_SINGLE_LIMIT=4294967295" "GPU_MAX_HW_QUEUES=1" "HIP_FORCE_DEV_KERNARG=1"; do
    echo "== $v"; env $v python tools/msm_bench.py BLS12_381_G1 21 6 plain 2>&1 | grep -v amdgpu.ids
  done > $out/env_ab.txt 2>&1
  cat $out/env_ab.txt
else
  echo "normal box (reduce $red)"
fi
