#!/bin/bash
# round 3, session T: after the shr_mod work-around: device self-check of fp28.cuh, full GPU suite
mkdir -p gpurun_out/r3t
O=$PWD/gpurun_out/r3t
export TMPDIR=/tmp
(cd algebra_amd/csrc/ubench && timeout 120 ./lazycheck.bin | grep -v "limb\|word") > $O/lazycheck.txt 2>&1
(timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -8) > $O/tests.log
echo done > $O/done
