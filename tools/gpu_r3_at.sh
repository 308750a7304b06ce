#!/bin/bash
# session AT: repeat calls of the trait-surface entry stream their scalars in GROWING pieces (each twice the previous one)
mkdir -p gpurun_out/r3at
O=$PWD/gpurun_out/r3at
export TMPDIR=/tmp
for ln in 24 22 20 26; do
  (timeout 600 python tools/trait_probe.py --log-n $ln --pieces 4,8 | grep -v amdgpu.ids) >> $O/trait.txt 2>> $O/trait.err
done
(timeout 600 python tools/trait_probe.py --log-n 22 --curve BLS12_377_G2 --pieces 4,8 | grep -v amdgpu.ids) >> $O/trait.txt 2>> $O/trait.err
(timeout 600 python tools/trait_probe.py --log-n 24 --curve BN254_G1 --pieces 4,8 | grep -v amdgpu.ids) >> $O/trait.txt 2>> $O/trait.err
(timeout 900 python -m pytest tests/test_gpu_trait_surface.py -m gpu -q -x 2>&1 | tail -3) > $O/tests.log
echo done > $O/done
