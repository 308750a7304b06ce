#!/bin/bash
# round 3, session L: C++ mirror with the communicator; bench.py --gpus 8 / 2 end to end with the ranks sharing the one GPU
# (gloo carries the exchanges: RCCL wants one GPU per rank) at reduced n -- the N > 1 code path, never a reported number
mkdir -p gpurun_out/r3l
O=$PWD/gpurun_out/r3l
export TMPDIR=/tmp
(timeout 600 python -m pytest tests/test_gpu_cpp_mirror.py tests/test_gpu_configs.py -m gpu -q -x 2>&1 | tail -8) > $O/tests.log
(ARK_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 8 --steps 3 --warmup 1 --log-n 18 --fft-log-n 16 --fft-steps 4 > $O/bench_n8_gloo.json) 2> $O/bench_n8_gloo.err
(ARK_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 3 --warmup 1 --log-n 21 --fft-log-n 20 --fft-steps 4 > $O/bench_n2_gloo.json) 2> $O/bench_n2_gloo.err
echo done > $O/done
