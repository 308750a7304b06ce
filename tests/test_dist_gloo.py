"""The N > 1 path on CPU: world_size 2, gloo.  The per-rank MSM itself needs a GPU (the product has no CPU
path), so here each rank's partial comes from the oracle; what is under test is the product's
sharding rule, the all-gather of partial Projective points and the host-side curve-addition combine."""
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib as O

A4 = np.array([0xA11CE, 1, 2, 0], dtype=np.uint64)
B4 = np.array([0xB0B, 3, 0, 0], dtype=np.uint64)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, cname, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import algebra_amd as A
    from algebra_amd import dist as D
    cid = O.CID[cname]
    bases = O.gen_bases(cid, A4, B4, n)
    scalars = O.gen_scalars(O.curve_info(cid)[1], 5, n)
    lo, hi = D.shard_bounds(n, rank, world)
    oracle_local = lambda c, b, s: O.msm(c, b, s, O.SIGNED, 1)
    total = D.msm_bigint_sharded(cid, bases[lo:hi], scalars[lo:hi], local_msm=oracle_local)
    full = O.msm(cid, bases, scalars, O.SIGNED, 1)
    ok = np.array_equal(A.into_affine(cid, total), O.to_affine(cid, full))
    q.put((rank, lo, hi, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("cname,n", [("BLS12_381_G1", 301), ("BLS12_377_G2", 64)])
def test_sharded_msm_world2_gloo(cname, n):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, cname, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[0] for r in res] == [0, 1]
    assert res[0][1] == 0 and res[0][2] == res[1][1] and res[1][2] == n   # contiguous cover
    assert all(r[3] for r in res)


def test_shard_bounds_cover():
    from algebra_amd import dist as D
    for n in (0, 1, 7, 8, 1 << 20, (1 << 26) + 3):
        for world in (1, 2, 4, 8):
            b = [D.shard_bounds(n, r, world) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1
