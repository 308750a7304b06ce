#!/bin/bash
# session AI: full GPU suite on the build with the one-Horner host tail (durations: the 2^25 / 2^27 at-size cases are new)
mkdir -p gpurun_out/r3ai
O=$PWD/gpurun_out/r3ai
export TMPDIR=/tmp
(timeout 1500 python -m pytest tests -m gpu -q --durations=10 2>&1 | tail -25) > $O/tests.log
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log) 2>&1
echo done > $O/done
