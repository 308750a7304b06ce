#!/bin/bash
# round 3, session E (after the container was replaced): re-establish the numbers of the committed build --
# parity, bench line, kernel trace, 28-bit-limb A/B, trait surface, narrow scalars, product-rate microbench
mkdir -p gpurun_out/r3e
O=$PWD/gpurun_out/r3e
R=$PWD
export TMPDIR=/tmp
(timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -15) > $O/tests.log
(timeout 600 python bench.py > $O/bench.json) 2> $O/bench.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d $O/p_kt -o kt -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras > $O/kt.out 2> $O/kt.err
cd $R
python tools/rocpd_stats.py $(find $O/p_kt -name "*results.db" | head -1) --min-us 200 > $O/kernel_stats.txt 2>> $O/post.err
rm -rf $O/p_kt
for lz in 1 0; do
  for cfg in "BLS12_381_G1 24" "BLS12_381_G1 20" "BLS12_381_G1 16" "BLS12_377_G1 22" "BLS12_381_G1 23"; do
    (echo "== ARK_HIP_MSM_LAZY=$lz $cfg"; ARK_HIP_MSM_LAZY=$lz timeout 300 python tools/msm_bench.py $cfg 3 both) >> $O/lazy_ab.txt 2>> $O/lazy_ab.err
  done
done
(ARK_HIP_COPY_THREADS=4 timeout 300 python tools/trait_probe.py --log-n 24 --pieces 1,2,4,8 >> $O/trait_probe.txt) 2>> $O/trait_probe.err
(timeout 300 python tools/small_scalar_bench.py > $O/small_scalar.txt) 2> $O/small_scalar.err
(cd algebra_amd/csrc/ubench && ls && timeout 120 ./mulbench.bin > $O/mulbench.txt 2>&1)
nproc > $O/host.txt; free -g >> $O/host.txt
echo done > $O/done
