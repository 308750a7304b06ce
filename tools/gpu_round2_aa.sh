#!/bin/bash
# GPU session AA of round 2: batched FFT entry (three transforms in flight): parity, shapes, bench line
mkdir -p gpurun_out/r2aa
O=$PWD/gpurun_out/r2aa
export TMPDIR=/tmp
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -8) > $O/tests.log
timeout 600 python tools/fft_shapes.py 4 22 > $O/fft_shapes.txt 2> $O/err.txt
(timeout 900 python bench.py --steps 10 --warmup 2 > $O/bench.json) 2> $O/bench.err
echo done > $O/done
