#!/bin/bash
# session AP: + finish kernel holding its entries in registers between the counting and the placement sweep (one HBM read)
# scatter kernel); b_mlp: 66 VGPRs (one scatter workgroup per CU), c_mlp64: 64 VGPRs (two)
mkdir -p gpurun_out/r3ap
O=$PWD/gpurun_out/r3ap
export TMPDIR=/tmp
for rep in 1 2; do
for v in a_before c_mlp64 d_regs; do
  for cfg in "BLS12_381_G1 24" "BLS12_381_G1 20" "BLS12_381_G1 26"; do
    (echo "== $v $cfg"; ARK_HIP_LIB=$PWD/algebra_amd/variants/$v.so timeout 300 python tools/msm_bench.py $cfg 5 plain | grep -v amdgpu.ids) >> $O/ab.txt 2>> $O/ab.err
  done
done
done
(ARK_HIP_LIB=$PWD/algebra_amd/variants/d_regs.so timeout 900 python -m pytest tests/test_gpu_msm.py tests/test_gpu_msm_prepared.py tests/test_gpu_configs.py tests/test_gpu_trait_surface.py -m gpu -q -x 2>&1 | tail -5) > $O/tests.log
echo done > $O/done
