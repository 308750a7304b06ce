#!/bin/bash
# round 3, session C: shared-bucket streamed MSM (trait surface), narrow-scalar entries, full parity
mkdir -p gpurun_out/r3c
O=$PWD/gpurun_out/r3c
export TMPDIR=/tmp
(timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -30) > $O/tests.log
for t in 4 0; do
  (ARK_HIP_COPY_THREADS=$t timeout 300 python tools/trait_probe.py --log-n 24 --pieces 1,2,4,6,8 >> $O/trait_probe.txt) 2>> $O/trait_probe.err
done
(ARK_HIP_COPY_THREADS=4 timeout 300 python tools/trait_probe.py --log-n 20 --pieces 1,2,4 >> $O/trait_probe.txt) 2>> $O/trait_probe.err
(ARK_HIP_COPY_THREADS=4 timeout 300 python tools/trait_probe.py --log-n 22 --pieces 1,2,4,8 >> $O/trait_probe.txt) 2>> $O/trait_probe.err
(timeout 300 python tools/small_scalar_bench.py > $O/small_scalar.txt) 2> $O/small_scalar.err
echo done > $O/done
