#!/bin/bash
# round 3, session O: pass-A tile of 16384 keys at 2^26 against 8192
mkdir -p gpurun_out/r3o
O=$PWD/gpurun_out/r3o
export TMPDIR=/tmp
V=$PWD/algebra_amd/variants/libark_hip_tile.so
for t in 16384 8192 16384 8192; do
  (echo "== TILE=$t 2^26"; ARK_HIP_MSM_TILE=$t ARK_HIP_LIB=$V timeout 300 python tools/msm_bench.py BLS12_381_G1 26 3 plain) >> $O/tile.txt 2>> $O/tile.err
done
(echo "== default 2^24"; ARK_HIP_LIB=$V timeout 300 python tools/msm_bench.py BLS12_381_G1 24 3 plain) >> $O/tile.txt 2>> $O/tile.err
(echo "== TILE=16384 2^24"; ARK_HIP_MSM_TILE=16384 ARK_HIP_LIB=$V timeout 300 python tools/msm_bench.py BLS12_381_G1 24 3 plain) >> $O/tile.txt 2>> $O/tile.err
echo done > $O/done
