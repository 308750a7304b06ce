#!/bin/bash
# GPU session S of round 2: cached coset power tables -- parity, soak, the FFT bench shapes again
mkdir -p gpurun_out/r2s
O=$PWD/gpurun_out/r2s
export TMPDIR=/tmp
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -8) > $O/tests.log
(timeout 300 python tools/soak.py 90 > $O/soak.txt) 2> $O/soak.err
timeout 900 python tools/fft_shapes.py 4 22 > $O/fft_shapes.txt 2> $O/err.txt
echo done > $O/done
