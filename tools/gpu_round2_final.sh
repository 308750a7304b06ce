#!/bin/bash
# final GPU session of round 2: parity of the shipped build, smoke, the bench line, kernel trace of the bench command
mkdir -p gpurun_out/r2final
O=$PWD/gpurun_out/r2final
R=$PWD
export TMPDIR=/tmp
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6) > $O/tests.log
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log) 2>&1
(timeout 900 python bench.py > $O/bench.json) 2> $O/bench.err
cd /tmp
timeout 900 rocprofv3 --kernel-trace -d $O/p_kt -o kt -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras > $O/kt.out 2> $O/kt.err
cd $R
python tools/rocpd_stats.py $(find $O/p_kt -name "*results.db" | head -1) --min-us 1000 > $O/kernel_stats.txt 2>> $O/post.err
rm -rf $O/p_kt
echo done > $O/done
