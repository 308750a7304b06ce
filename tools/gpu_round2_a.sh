#!/bin/bash
# GPU session A of round 2: parity suite, bench line, plain/prepared sweeps, 3-wave A/B.  Outputs under gpurun_out/r2a/.
mkdir -p gpurun_out/r2a
O=gpurun_out/r2a
export TMPDIR=/tmp
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -40) > $O/tests.log
(timeout 600 python bench.py --steps 5 --warmup 2 > $O/bench.json) 2> $O/bench.err
for s in "24 5" "20 10" "16 20" "22 5" "23 5"; do
  timeout 300 python tools/msm_bench.py BLS12_381_G1 $s both >> $O/sweep.txt 2>> $O/sweep.err
done
for c in 20 21 22 23 24; do
  ARK_HIP_MSM_C_PREPARED=$c timeout 300 python tools/msm_bench.py BLS12_381_G1 24 3 prepared >> $O/sweep_c.txt 2>> $O/sweep.err
done
ARK_HIP_LIB=$PWD/algebra_amd/libark_hip_w3.so timeout 300 python tools/msm_bench.py BLS12_381_G1 24 5 both >> $O/sweep_w3.txt 2>> $O/sweep.err
timeout 300 python tools/msm_bench.py BLS12_377_G2 22 3 both >> $O/sweep_g2.txt 2>> $O/sweep.err
timeout 300 python tools/msm_bench.py BN254_G1 24 3 both >> $O/sweep_g2.txt 2>> $O/sweep.err
timeout 120 python tools/fft_bench.py 16 20 22 24 26 > $O/fft.txt 2>> $O/sweep.err
echo done > $O/done
