#!/bin/bash
# GPU session E of round 2: batch_mul, streaming measurements, scalar distributions, bench line, FFT shapes
mkdir -p gpurun_out/r2e
O=gpurun_out/r2e
export TMPDIR=/tmp
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -40) > $O/tests.log
(timeout 600 python tools/stream_bench.py 24 6 > $O/stream.txt) 2>> $O/err.txt
(timeout 600 python tools/small_scalar_bench.py 20 > $O/distributions.txt) 2>> $O/err.txt
(timeout 900 python bench.py --steps 10 --warmup 2 > $O/bench.json) 2> $O/bench.err
(timeout 300 python tools/fft_bench.py 16 18 20 22 24 > $O/fft.txt) 2>> $O/err.txt
echo done > $O/done
