#!/bin/bash
mkdir -p gpurun_out/r2g
O=gpurun_out/r2g
export TMPDIR=/tmp
(timeout 900 python tools/small_n_sweep.py BLS12_381_G1 > $O/small_n.txt) 2> $O/err.txt
(timeout 600 python tools/small_scalar_bench.py 20 > $O/distributions.txt) 2>> $O/err.txt
echo done > $O/done
