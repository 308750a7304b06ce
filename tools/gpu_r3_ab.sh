#!/bin/bash
# round 3, session AB: FFT pass kernel with the next stage pair's twiddles requested a round ahead, against the shipped one
mkdir -p gpurun_out/r3ab
O=$PWD/gpurun_out/r3ab
export TMPDIR=/tmp
for rep in 1 2; do
for v in fftnew fftold; do
  (echo "== $v"; ARK_HIP_LIB=$PWD/algebra_amd/variants/libark_hip_$v.so timeout 300 python tools/fft_bench.py 16 20 22 24 26) >> $O/fft.txt 2>> $O/err.txt
done
done
(timeout 600 python -m pytest tests/test_gpu_fft.py tests/test_gpu_configs.py -m gpu -q -x 2>&1 | tail -4) > $O/tests.log
echo done > $O/done
