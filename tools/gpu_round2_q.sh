#!/bin/bash
# GPU session Q of round 2: completion wait by polling vs blocking (wall vs device time), then the bench line
mkdir -p gpurun_out/r2q
O=$PWD/gpurun_out/r2q
export TMPDIR=/tmp
for mode in block spin block spin; do
  echo "## ARK_HIP_WAIT=$mode" >> $O/wait.txt
  ARK_HIP_WAIT=$mode timeout 400 python tools/msm_bench.py BLS12_381_G1 24 5 prepared >> $O/wait.txt 2>> $O/err.txt
done
(timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -5) > $O/tests.log
(timeout 900 python bench.py --steps 10 --warmup 2 > $O/bench.json) 2> $O/bench.err
echo done > $O/done
