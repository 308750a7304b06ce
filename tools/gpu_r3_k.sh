#!/bin/bash
# round 3, session K: RCCL inside the library (world-of-one communicator on hardware), one-exchange sharded FFT
# (single-process emulation, gloo ranks sharing the GPU, coset), prepared multi-device MSM; then the whole GPU suite
mkdir -p gpurun_out/r3k
O=$PWD/gpurun_out/r3k
export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gpu_dist_fft.py tests/test_gpu_msm_prepared.py -m gpu -q -x 2>&1 | tail -30) > $O/tests_new.log
(timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -30) > $O/tests.log
echo done > $O/done
