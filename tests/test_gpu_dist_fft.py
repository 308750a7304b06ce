"""Sharded radix-2 FFT (algebra_amd.dist.fft_sharded) end to end: 2 and 4 ranks share the one GPU of the test box,
gloo carries the three all-to-alls (RCCL needs one GPU per rank); the local stages run the real HIP kernels.
Checked limb-for-limb against the oracle's single transform."""
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib as O

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, fname, log_n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import torch
    from algebra_amd import dist as D
    fid = O.FID[fname]
    n = 1 << log_n
    m = n // world
    x = O.gen_scalars(fid, 99, n, montgomery=True)
    xs = torch.from_numpy(x[rank * m:(rank + 1) * m].view(np.int64)).cuda()
    y = D.fft_sharded(fname, n, xs)
    exp = O.fft(fid, x, log_n, None, False, 4).reshape(n, 4)
    ok_f = np.array_equal(y.cpu().numpy().view(np.uint64), exp[rank * m:(rank + 1) * m])
    back = D.fft_sharded(fname, n, y, inverse=True)
    ok_i = np.array_equal(back.cpu().numpy().view(np.uint64), x[rank * m:(rank + 1) * m])
    q.put((rank, bool(ok_f), bool(ok_i)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,log_n", [(2, 12), (4, 16), (2, 20)])
def test_fft_sharded_matches_single_transform(world, log_n):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, "BLS12_381_FR", log_n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res), res
    assert all(r[2] for r in res), res
