#!/bin/bash
# GPU session C of round 2: lane-pair Fp2 accumulate (G2), shorter bit-stage chunks
mkdir -p gpurun_out/r2c
O=gpurun_out/r2c
export TMPDIR=/tmp
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -40) > $O/tests.log
timeout 300 python tools/msm_bench.py BLS12_377_G2 22 3 both >> $O/sweep_g2.txt 2>> $O/sweep.err
timeout 300 python tools/msm_bench.py BLS12_381_G2 20 3 both >> $O/sweep_g2.txt 2>> $O/sweep.err
timeout 300 python tools/msm_bench.py BLS12_377_G2 18 5 both >> $O/sweep_g2.txt 2>> $O/sweep.err
for s in "24 5" "20 10" "16 20"; do
  timeout 400 python tools/msm_bench.py BLS12_381_G1 $s both >> $O/sweep.txt 2>> $O/sweep.err
done
for ch in 512 2048 4096; do
  ARK_HIP_MSM_CHUNK=$ch timeout 300 python tools/msm_bench.py BLS12_381_G1 24 3 prepared >> $O/sweep_chunk.txt 2>> $O/sweep.err
done
echo done > $O/done
