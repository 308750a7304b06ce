#!/bin/bash
# session AQ: XCD-aware tile mapping in the sort's pass A (neighbouring tiles' runs merge in ONE L2); WRITE_SIZE before / after
mkdir -p gpurun_out/r3aq
O=$PWD/gpurun_out/r3aq
R=$PWD
export TMPDIR=/tmp
for rep in 1 2; do
for v in a_mlp b_xcd; do
  for cfg in "BLS12_381_G1 24" "BLS12_381_G1 20" "BLS12_381_G1 26" "BLS12_381_G1 22"; do
    (echo "== $v $cfg"; ARK_HIP_LIB=$PWD/algebra_amd/variants/$v.so timeout 300 python tools/msm_bench.py $cfg 5 plain | grep -v amdgpu.ids) >> $O/ab.txt 2>> $O/ab.err
  done
done
done
cd /tmp
for v in a_mlp b_xcd; do
  ARK_HIP_LIB=$R/algebra_amd/variants/$v.so timeout 300 rocprofv3 --pmc WRITE_SIZE -d $O/p_$v -o w -- python $R/tools/msm_bench.py BLS12_381_G1 24 2 plain > $O/w_$v.out 2> $O/w_$v.err
  python $R/tools/rocpd_stats.py $(find $O/p_$v -name "*results.db" | head -1) --pmc --min-us 100 > $O/pmc_write_$v.txt 2>> $O/post.err
  rm -rf $O/p_$v
done
cd $R
(ARK_HIP_LIB=$PWD/algebra_amd/variants/b_xcd.so timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_msm_prepared.py tests/test_gpu_configs.py tests/test_gpu_trait_surface.py -m gpu -q -x 2>&1 | tail -5) > $O/tests.log
echo done > $O/done
