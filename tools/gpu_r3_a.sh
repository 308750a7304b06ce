#!/bin/bash
# round 3, session A: parity of the new trait-surface path, bench line, staging/piece sweep, lazy-limb product rate
mkdir -p gpurun_out/r3a
O=$PWD/gpurun_out/r3a
R=$PWD
export TMPDIR=/tmp
(timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -15) > $O/tests.log
(timeout 600 python bench.py > $O/bench.json) 2> $O/bench.err
for t in 0 2 4 8; do
  (ARK_HIP_COPY_THREADS=$t timeout 300 python tools/trait_probe.py --log-n 24 $( [ $t = 4 ] && echo --auto-prepare ) >> $O/trait_probe.txt) 2>> $O/trait_probe.err
done
(cd algebra_amd/csrc/ubench && timeout 120 ./mulbench.bin > $O/mulbench.txt 2>&1)
nproc > $O/host.txt; free -g >> $O/host.txt
echo done > $O/done
