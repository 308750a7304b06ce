#!/bin/bash
# round 3, session V: reduction / heavy-run kernels on 28-bit limbs against the saturated ones (same box), then the suite
mkdir -p gpurun_out/r3v
O=$PWD/gpurun_out/r3v
export TMPDIR=/tmp
for cfg in "BLS12_381_G1 24" "BLS12_381_G1 20" "BLS12_381_G1 16" "BLS12_381_G1 26"; do
  for v in redlazy redsat; do
    (echo "== $v $cfg"; ARK_HIP_LIB=$PWD/algebra_amd/variants/libark_hip_$v.so timeout 300 python tools/msm_bench.py $cfg 3 both) >> $O/ab.txt 2>> $O/ab.err
  done
done
(ARK_HIP_LIB=$PWD/algebra_amd/variants/libark_hip_redlazy.so timeout 300 python tools/small_scalar_bench.py > $O/small_scalar_lazy.txt) 2>> $O/ab.err
(ARK_HIP_LIB=$PWD/algebra_amd/variants/libark_hip_redsat.so timeout 300 python tools/small_scalar_bench.py > $O/small_scalar_sat.txt) 2>> $O/ab.err
(timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -8) > $O/tests.log
echo done > $O/done
