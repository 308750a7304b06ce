"""The exchange behind ark_hip_msm_sw_device_sharded / ark_hip_msm_prepared_device_sharded with more than one rank: every
rank's PART SUMS (not a finished partial result) are all-gathered behind a 64-byte header, added by one kernel, and folded
by ONE host tail; ranks whose plans differ fall back -- together -- to finished partials (variable_base/mod.rs:542-557:
the chunk sum).  The build loop has one-GPU boxes and RCCL wants one GPU per rank, so the ranks are emulated in one process
(ark_hip_test_msm_sharded_emulated: everything but the ncclAllGather call itself, which the world-of-one RCCL test and
the gloo tests cover) and checked against the oracle."""
import ctypes as C

import numpy as np
import pytest

import algebra_amd as A
import oracle_lib as O
from algebra_amd._lib import test_lib

pytestmark = pytest.mark.gpu

A4 = np.array([0xA11CE, 1, 2, 0], dtype=np.uint64)
B4 = np.array([0xB0B, 3, 0, 0], dtype=np.uint64)


def emulated(cid, bases, scalars, cuts, montgomery=False):
    """shards [cuts[r], cuts[r+1]) of (bases, scalars), one per emulated rank -> (rc, Projective, path)"""
    import torch
    world = len(cuts) - 1
    db, ds = [], []
    for r in range(world):
        lo, hi = cuts[r], cuts[r + 1]
        db.append(torch.from_numpy(np.ascontiguousarray(bases[lo:hi]).view(np.int64)).cuda() if hi > lo else None)
        ds.append(torch.from_numpy(np.ascontiguousarray(scalars[lo:hi]).view(np.int64)).cuda() if hi > lo else None)
    torch.cuda.synchronize()
    pb = (C.c_void_p * world)(*[t.data_ptr() if t is not None else None for t in db])
    ps = (C.c_void_p * world)(*[t.data_ptr() if t is not None else None for t in ds])
    pn = (C.c_size_t * world)(*[cuts[r + 1] - cuts[r] for r in range(world)])
    out = np.zeros(3 * O.fe_words(cid), dtype=np.uint64)
    path = C.c_int(0)
    rc = test_lib().ark_hip_test_msm_sharded_emulated(cid, world, pb, ps, pn, int(montgomery), out.ctypes.data_as(C.c_void_p), C.byref(path))
    return rc, out, path.value


@pytest.mark.parametrize("cname", O.CURVES)
@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_equal_shards_sum_their_part_sums_on_the_device(cname, world):
    cid = O.CID[cname]
    per = 700 if cname.endswith("G2") else 3000
    n = per * world
    bases = O.gen_bases(cid, A4, B4, n)
    scalars = O.gen_scalars(O.curve_info(cid)[1], 60 + world, n)
    rc, got, path = emulated(cid, bases, scalars, [r * per for r in range(world + 1)])
    assert rc == 0 and path == 1
    assert np.array_equal(A.into_affine(cid, got), O.to_affine(cid, O.msm(cid, bases, scalars, O.SIGNED, 4)))


def test_montgomery_scalars_and_a_size_with_many_parts():
    cid = O.CID["BLS12_381_G1"]
    world, per = 2, 1 << 16
    n = per * world
    bases = O.gen_bases(cid, A4, B4, n)
    fid = O.curve_info(cid)[1]
    mont = O.gen_scalars(fid, 5, n, montgomery=True)
    rc, got, path = emulated(cid, bases, mont, [0, per, n], montgomery=True)
    assert rc == 0 and path == 1
    assert np.array_equal(A.into_affine(cid, got), O.to_affine(cid, O.msm(cid, bases, mont, O.SIGNED, 8, montgomery_scalars=True)))


def test_unequal_or_empty_shards_fall_back_together():
    cid = O.CID["BLS12_381_G1"]
    n = 40000
    bases = O.gen_bases(cid, A4, B4, n)
    scalars = O.gen_scalars(O.curve_info(cid)[1], 9, n)
    want = O.to_affine(cid, O.msm(cid, bases, scalars, O.SIGNED, 4))
    for cuts, must_fall_back in (([0, 100, n], True), ([0, 0, n], True), ([0, 17000, 17000, n], True), ([0, 13333, 26666, n], False)):
        rc, got, path = emulated(cid, bases, scalars, cuts)
        assert rc == 0, cuts
        assert path == 2 or not must_fall_back, (cuts, path)   # (13333 / 13333 / 13334 may or may not share a plan)
        assert np.array_equal(A.into_affine(cid, got), want), cuts


def test_scalar_range_error_on_one_rank_is_everyones():
    cid = O.CID["BLS12_381_G1"]
    per = 2000
    bases = O.gen_bases(cid, A4, B4, 2 * per)
    scalars = O.gen_scalars(O.curve_info(cid)[1], 3, 2 * per).copy()
    scalars[per + 5, 3] = np.uint64(1) << np.uint64(63)          # >= 2^255 on the second rank only
    rc, _, _ = emulated(cid, bases, scalars, [0, per, 2 * per])
    assert rc == -4
