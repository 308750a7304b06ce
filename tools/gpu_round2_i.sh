#!/bin/bash
mkdir -p gpurun_out/r2i
O=$PWD/gpurun_out/r2i
R=$PWD
export TMPDIR=/tmp
for s in "16 20" "18 20" "20 10"; do
  timeout 400 python tools/msm_bench.py BLS12_381_G1 $s both >> $O/sweep.txt 2>> $O/err.txt
done
timeout 300 python tools/msm_bench.py BN254_G1 16 20 both >> $O/sweep.txt 2>> $O/err.txt
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d $O/p -o kt -- python $R/tools/diag_u32.py > $O/diag.txt 2> $O/diag.err
cd $R
python tools/rocpd_stats.py $(find $O/p -name "*results.db" | head -1) --min-us 50 > $O/diag_kernels.txt 2>> $O/err.txt
rm -rf $O/p
echo done > $O/done
