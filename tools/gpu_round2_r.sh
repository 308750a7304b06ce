#!/bin/bash
# GPU session R of round 2: the reference's FFT bench shapes (poly/benches/fft.rs) + FFT sizes up to 2^26
mkdir -p gpurun_out/r2r
O=$PWD/gpurun_out/r2r
export TMPDIR=/tmp
timeout 900 python tools/fft_shapes.py 4 22 > $O/fft_shapes.txt 2> $O/err.txt
timeout 600 python tools/fft_bench.py 16 18 20 22 24 26 > $O/fft_sizes.txt 2>> $O/err.txt
echo done > $O/done
