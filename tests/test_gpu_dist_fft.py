"""Sharded radix-2 FFT end to end (ark_hip_fft_sharded_device and its two local halves, algebra_amd.dist.fft_sharded).
The test box has ONE GPU and RCCL wants one GPU per rank, so:
  * 2 and 4 ranks (processes) share the GPU and gloo carries the ONE all-to-all; both local halves run the real HIP kernels;
  * the same decomposition driven from one process with device copies as the exchange, forward / inverse / coset;
  * the library's own RCCL path with a genuine world-of-one communicator (dlopen, ncclCommInitRank, the entry points).
Checked limb-for-limb against the oracle's single transform, in the layouts include/ark_hip.h documents."""
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
    sub = m // world
    x = O.gen_scalars(fid, 99, n, montgomery=True).reshape(n, 4)
    mine = np.ascontiguousarray(D.shard_input(x, rank, world))                   # x[rank + G i2]
    xs = torch.from_numpy(mine.view(np.int64)).cuda()
    y = D.fft_sharded(fname, n, xs)
    exp = O.fft(fid, x, log_n, None, False, 4).reshape(n, 4)
    want = np.concatenate([exp[j1 * m + rank * sub: j1 * m + (rank + 1) * sub] for j1 in range(world)])
    ok_f = np.array_equal(y.cpu().numpy().view(np.uint64), want)                 # X[j1 m + rank sub + t]
    back = D.fft_sharded(fname, n, y, inverse=True)
    ok_i = np.array_equal(back.cpu().numpy().view(np.uint64), mine)
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


def _emulated(fname, log_n, world, offset, inverse_too=True):
    """all ranks' local halves from ONE process, the exchange as array slicing: the decomposition itself"""
    import ctypes as C
    import torch
    import algebra_amd as A
    from algebra_amd import dist as D
    from algebra_amd._lib import check, lib
    L = lib()
    fid = O.FID[fname]
    n = 1 << log_n
    m, sub = n // world, n // world // world
    dom = A.Radix2EvaluationDomain.new(fname, n)
    if offset is not None:
        dom = dom.get_coset(offset)
    sref = C.byref(dom._s)
    x = O.gen_scalars(fid, 1234, n, montgomery=True).reshape(n, 4)
    exp = O.fft(fid, x, log_n, offset, False, 4).reshape(n, 4)
    loc = [torch.from_numpy(np.ascontiguousarray(D.shard_input(x, r, world)).view(np.int64)).cuda() for r in range(world)]
    for r in range(world):
        check(L.ark_hip_fft_shard_local_device(fid, sref, r, world, loc[r].data_ptr(), 0), "local")
    check(L.ark_hip_synchronize(), "sync")
    blocks = [t.reshape(world, sub, 4) for t in loc]
    recv = [torch.stack([blocks[i1][q] for i1 in range(world)]).contiguous() for q in range(world)]   # the all-to-all
    outs = []
    for q in range(world):
        dst = torch.empty_like(recv[q])
        check(L.ark_hip_fft_shard_cross_device(fid, sref, world, recv[q].data_ptr(), dst.data_ptr(), 0), "cross")
        outs.append(dst)
    check(L.ark_hip_synchronize(), "sync")
    got = D.unshard_output([o.reshape(m, 4).cpu().numpy().view(np.uint64) for o in outs])
    assert np.array_equal(got, exp)
    if not inverse_too:
        return
    # inverse: cross, all-to-all, local -- back to the cyclic coefficient layout
    for q in range(world):
        check(L.ark_hip_fft_shard_cross_device(fid, sref, world, outs[q].data_ptr(), outs[q].data_ptr(), 1), "cross inv")
    check(L.ark_hip_synchronize(), "sync")
    back = [torch.stack([outs[q][i1] for q in range(world)]).reshape(m, 4).contiguous() for i1 in range(world)]
    for r in range(world):
        check(L.ark_hip_fft_shard_local_device(fid, sref, r, world, back[r].data_ptr(), 1), "local inv")
    check(L.ark_hip_synchronize(), "sync")
    for r in range(world):
        assert np.array_equal(back[r].cpu().numpy().view(np.uint64), D.shard_input(x, r, world))


@pytest.mark.parametrize("fname,log_n,world", [("BLS12_381_FR", 10, 2), ("BLS12_381_FR", 14, 8), ("BLS12_381_FR", 16, 16),
                                               ("BN254_FR", 12, 4), ("BLS12_377_FR", 13, 4), ("BLS12_381_FR", 20, 8)])
def test_sharded_decomposition_single_process(fname, log_n, world):
    _emulated(fname, log_n, world, None)


@pytest.mark.parametrize("fname,log_n,world", [("BLS12_381_FR", 12, 4), ("BLS12_377_FR", 10, 2)])
def test_sharded_decomposition_coset(fname, log_n, world):
    off = O.gen_scalars(O.FID[fname], 77, 1, montgomery=True).reshape(4)
    _emulated(fname, log_n, world, off)


def test_library_rccl_world_of_one():
    """the RCCL plumbing on hardware: unique id, ncclCommInitRank (world 1), the sharded MSM and FFT entry points through
    the communicator, destroy.  (World > 1 needs one GPU per rank: the driver's multi-GPU run.)"""
    import ctypes as C
    import torch
    import algebra_amd as A
    from algebra_amd._lib import check, lib
    L = lib()
    ident = (C.c_ubyte * 128)()
    check(L.ark_hip_comm_unique_id(ident), "unique id")
    assert any(ident)
    check(L.ark_hip_comm_init(ident, 0, 1), "comm init")
    try:
        rk, wd = C.c_int(-1), C.c_int(-1)
        check(L.ark_hip_comm_info(C.byref(rk), C.byref(wd)), "info")
        assert (rk.value, wd.value) == (0, 1)
        assert L.ark_hip_comm_init(ident, 0, 1) != 0          # one communicator per device
        cid = O.CID["BLS12_381_G1"]
        n = 3000
        a4 = np.array([0xA11CE, 1, 2, 0], dtype=np.uint64)
        b4 = np.array([0xB0B, 3, 0, 0], dtype=np.uint64)
        bases = O.gen_bases(cid, a4, b4, n)
        sc = O.gen_scalars(O.curve_info(cid)[1], 3, n)
        out = np.zeros(18, dtype=np.uint64)
        db = torch.from_numpy(bases.view(np.int64)).cuda()
        ds = torch.from_numpy(sc.view(np.int64)).cuda()
        check(L.ark_hip_msm_sw_device_sharded(cid, db.data_ptr(), ds.data_ptr(), n, 0, out.ctypes.data_as(C.c_void_p)), "sharded msm")
        assert np.array_equal(A.into_affine(cid, out), O.to_affine(cid, O.msm(cid, bases, sc, O.SIGNED, 2)))
        fid = O.FID["BLS12_381_FR"]
        x = O.gen_scalars(fid, 5, 1 << 12, montgomery=True)
        dom = A.Radix2EvaluationDomain.new("BLS12_381_FR", 1 << 12)
        dx = torch.from_numpy(x.view(np.int64)).cuda()
        check(L.ark_hip_fft_sharded_device(fid, C.byref(dom._s), dx.data_ptr(), 0), "sharded fft")
        check(L.ark_hip_synchronize(), "sync")
        assert np.array_equal(dx.cpu().numpy().view(np.uint64).reshape(-1), O.fft(fid, x, 12, None, False, 2))
    finally:
        check(L.ark_hip_comm_destroy(), "destroy")
