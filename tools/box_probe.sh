#!/bin/bash
# What kind of box is this?  Some gpurun boxes run the latency-bound kernels (bucket reduction, small MSMs) 1.5-2 x slower than
# others with the throughput kernels unchanged (profiles/r5_bench_line_after_soak_anomalous_box_state.json).  Prints the
# partition / clock / power state next to a reduction-heavy MSM so that the two kinds can be told apart.
#   gpurun -- 'bash tools/box_probe.sh > gpurun_out/box_probe.txt 2>&1'
echo "== rocm-smi"
rocm-smi --showcomputepartition --showmemorypartition --showperflevel --showclocks --showpower --showmaxpower --showtemp 2>&1 | grep -v "^$" | head -60
echo "== rocminfo (compute units, clocks)"
rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|Max Clock|Memory Properties|Size:" | head -20
echo "== uname / cpu"
uname -r; nproc; grep -m1 "model name" /proc/cpuinfo
echo "== MSM 2^21 / 2^16 (BLS12-381 G1): reduce ~1.15 / ~0.27 ms on a normal box"
python tools/msm_bench.py BLS12_381_G1 21 6 plain 2>&1 | grep -v amdgpu.ids
python tools/msm_bench.py BLS12_381_G1 16 10 plain 2>&1 | grep -v amdgpu.ids
echo "== clocks right after"
rocm-smi --showclocks 2>&1 | grep -E "sclk|mclk|fclk|socclk" | head -8
echo "== sclk / power sampled every 0.25 s during 400 MSMs of 2^21 (the first ~2 s are the interpreter starting)"
python tools/msm_bench.py BLS12_381_G1 21 400 plain 2>&1 | grep -v amdgpu.ids &
for i in $(seq 1 24); do sleep 0.25; echo "t=$i $(rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|Socket" | sed -e 's/GPU\[0\]//' -e 's/[[:space:]]\+/ /g' | tr '\n' ' ')"; done
wait
