#!/bin/bash
# GPU session W of round 2: two MSM lanes per context (a second job's sort under the first one's accumulate):
# parity + soak, then streaming / pipelined throughput before and after at 2^20 and 2^24
mkdir -p gpurun_out/r2w
O=$PWD/gpurun_out/r2w
R=$PWD
export TMPDIR=/tmp
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -8) > $O/tests.log
(timeout 300 python tools/soak.py 60 > $O/soak.txt) 2> $O/soak.err
for lib in libark_hip_prev.so libark_hip.so; do
  echo "## $lib  2^20" >> $O/stream.txt
  ARK_HIP_LIB=$R/algebra_amd/$lib timeout 400 python tools/stream_bench.py 20 40 >> $O/stream.txt 2>> $O/err.txt
  echo "## $lib  2^24" >> $O/stream.txt
  ARK_HIP_LIB=$R/algebra_amd/$lib timeout 600 python tools/stream_bench.py 24 8 >> $O/stream.txt 2>> $O/err.txt
done
echo done > $O/done
