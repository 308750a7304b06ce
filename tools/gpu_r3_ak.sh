#!/bin/bash
# session AK: accumulate kernel without the next-point prefetch and without the out-of-line doubling call, at 2 and at a
# real 3 waves per SIMD (the earlier "3 waves" build never left occupancy 2: the noinline callee's own register count)
mkdir -p gpurun_out/r3ak
O=$PWD/gpurun_out/r3ak
export TMPDIR=/tmp
for rep in 1 2; do
for v in a_cur b_nopf2 c_nopf3; do
  for cfg in "BLS12_381_G1 24" "BLS12_381_G1 20"; do
    (echo "== $v $cfg"; ARK_HIP_LIB=$PWD/algebra_amd/variants/libark_hip_$v.so timeout 300 python tools/msm_bench.py $cfg 5 plain | grep -v amdgpu.ids) >> $O/ab.txt 2>> $O/ab.err
  done
done
done
echo done > $O/done
