"""The entry the trait surface lands in: ark_hip_msm_sw from HOST pointers -- what SWCurveConfig::msm and the msm_bigint
hook (patches/0001, rust/ark-hip/src/msm.rs) call with Rust slices -- with its resident-base cache (hit / miss / content
replaced in place / eviction / disabled / auto-prepare), the streamed scalar pieces, and the ordering of the second MSM
lane behind producers on the context stream.  Parity against the oracle through the C ABI."""
import ctypes as C

import numpy as np
import pytest

import algebra_amd as A
import oracle_lib as O
from algebra_amd._lib import check, lib

pytestmark = pytest.mark.gpu

A4 = np.array([0xA11CE, 1, 2, 0], dtype=np.uint64)
B4 = np.array([0xB0B, 3, 0, 0], dtype=np.uint64)
C4 = np.array([0xC0FFEE, 7, 0, 0], dtype=np.uint64)


def sf(cid):
    return O.curve_info(cid)[1]


def aff(cid, p):
    return A.into_affine(cid, p)


def oracle_aff(cid, bases, scalars, **kw):
    return O.to_affine(cid, O.msm(cid, bases, scalars, O.SIGNED, 4, **kw))


@pytest.fixture(autouse=True)
def fresh_cache():
    A.base_cache_config(8 << 30, 0)
    A.base_cache_clear()
    yield
    A.base_cache_clear()
    A.base_cache_config(8 << 30, 0)


def delta(before, after):
    return {k: after[k] - before[k] for k in ("hits", "misses", "refreshed", "evicted")}


@pytest.mark.parametrize("cname", O.CURVES)
def test_cache_miss_then_hits(cname):
    cid = O.CID[cname]
    n = 1 << 11 if cname.endswith("G2") else 1 << 13
    bases = O.gen_bases(cid, A4, B4, n)
    s0 = A.base_cache_stats()
    for k in range(3):
        scalars = O.gen_scalars(sf(cid), 100 + k, n)           # new scalars every call, the same base array
        assert np.array_equal(aff(cid, A.msm_bigint(cid, bases, scalars)), oracle_aff(cid, bases, scalars)), k
    mont = O.gen_scalars(sf(cid), 9, n, montgomery=True)       # SWCurveConfig::msm: Fr elements
    assert np.array_equal(aff(cid, A.msm(cid, bases, mont)), oracle_aff(cid, bases, mont, montgomery_scalars=True))
    s1 = A.base_cache_stats()
    assert delta(s0, s1) == {"hits": 3, "misses": 1, "refreshed": 0, "evicted": 0}
    assert s1["entries"] == 1 and s1["bytes"] >= bases.nbytes
    # msm_unchecked truncation = a different length at the same address: its own entry
    k = n // 3
    sc = O.gen_scalars(sf(cid), 5, k)
    assert np.array_equal(aff(cid, A.msm_bigint(cid, bases[:k], sc)), oracle_aff(cid, bases[:k], sc))
    assert A.base_cache_stats()["entries"] == 2


def test_cache_notices_replaced_content():
    cid = O.CID["BLS12_381_G1"]
    n = 1 << 12
    bases = O.gen_bases(cid, A4, B4, n).copy()
    other = O.gen_bases(cid, C4, B4, n)
    scalars = O.gen_scalars(sf(cid), 1, n)
    assert np.array_equal(aff(cid, A.msm_bigint(cid, bases, scalars)), oracle_aff(cid, bases, scalars))
    s0 = A.base_cache_stats()
    bases[:] = other                                           # same address, same length, new SRS
    got = aff(cid, A.msm_bigint(cid, bases, scalars))
    assert np.array_equal(got, oracle_aff(cid, other, scalars))
    assert delta(s0, A.base_cache_stats()) == {"hits": 0, "misses": 0, "refreshed": 1, "evicted": 0}
    assert np.array_equal(aff(cid, A.msm_bigint(cid, bases, scalars)), got)   # and it is a plain hit afterwards
    # an edit of ONE sampled point (n <= 4096: every point is sampled) is noticed as well
    bases[7] = O.gen_bases(cid, B4, A4, 1)[0]
    assert np.array_equal(aff(cid, A.msm_bigint(cid, bases, scalars)), oracle_aff(cid, bases, scalars))
    # the documented escape hatch for edits the sample could miss
    A.base_cache_clear()
    assert A.base_cache_stats()["entries"] == 0
    assert np.array_equal(aff(cid, A.msm_bigint(cid, bases, scalars)), oracle_aff(cid, bases, scalars))


def test_cache_eviction_lru_and_disabled():
    cid = O.CID["BN254_G1"]
    n = 1 << 12
    sets = [O.gen_bases(cid, np.array([11 + i, i, 0, 0], dtype=np.uint64), B4, n) for i in range(3)]
    scalars = O.gen_scalars(sf(cid), 3, n)
    want = [oracle_aff(cid, b, scalars) for b in sets]
    A.base_cache_config(int(2.6 * sets[0].nbytes), -1)         # room for two sets (capacity rounding included)
    s0 = A.base_cache_stats()
    for i in (0, 1, 0, 2, 1):                                   # 2 evicts the least recently used (1), then 1 evicts 0
        assert np.array_equal(aff(cid, A.msm_bigint(cid, sets[i], scalars)), want[i]), i
    d = delta(s0, A.base_cache_stats())
    assert d["misses"] == 4 and d["hits"] == 1 and d["evicted"] == 2, d
    assert A.base_cache_stats()["entries"] == 2
    A.base_cache_config(0, -1)                                  # off: bases stream with the scalars on every call
    assert A.base_cache_stats()["entries"] == 0
    s0 = A.base_cache_stats()
    for i in range(3):
        assert np.array_equal(aff(cid, A.msm_bigint(cid, sets[i], scalars)), want[i])
    assert delta(s0, A.base_cache_stats()) == {"hits": 0, "misses": 0, "refreshed": 0, "evicted": 0}


@pytest.mark.parametrize("pieces", [1, 2, 3, 5])
def test_streamed_pieces_match_oracle(pieces, monkeypatch):
    monkeypatch.setenv("ARK_HIP_STREAM_PIECES", str(pieces))
    for cname, n in (("BLS12_381_G1", 12345), ("BLS12_377_G2", 1500), ("BN254_G1", 1)):
        cid = O.CID[cname]
        bases = O.gen_bases(cid, A4, B4, n)
        scalars = O.gen_scalars(sf(cid), 77 + pieces, n)
        for _ in range(2):                                      # miss (bases uploaded) and hit (scalars only)
            assert np.array_equal(aff(cid, A.msm_bigint(cid, bases, scalars)), oracle_aff(cid, bases, scalars)), (cname, pieces)
    # cache off: bases AND scalars stream piecewise (the msm_chunks machinery)
    A.base_cache_config(0, -1)
    cid = O.CID["BLS12_381_G1"]
    bases = O.gen_bases(cid, A4, B4, 5000)
    scalars = O.gen_scalars(sf(cid), 5, 5000)
    assert np.array_equal(aff(cid, A.msm_bigint(cid, bases, scalars)), oracle_aff(cid, bases, scalars))


def test_streamed_pieces_at_size(monkeypatch):
    # 2^21 pairs through the default piece rule (two pieces on the two lanes), miss then hit, against k*G
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import synth as S
    cid = O.CID["BLS12_381_G1"]
    r = S.R["BLS12_381_FR"]
    n = 1 << 21
    bases = S.grow_bases(cid, n, S.A0, S.B0, r).cpu().numpy().view(np.uint64).reshape(n, -1)
    sc = S.gen_scalars(n, 0x51, r)
    want = S.mul_gen(cid, S.dlog_of_msm(sc, S.A0, S.B0, r), r)
    for _ in range(2):
        assert np.array_equal(aff(cid, A.msm_bigint(cid, bases, sc)), want)
    st = A.base_cache_stats()          # (mul_gen's one-point MSMs own the other entries)
    assert st["bytes"] >= bases.nbytes and st["hits"] >= 1
    # a hit streams its scalars in GROWING pieces (each twice the one before: 3 pieces here); the equal-piece rule and a
    # ragged length (two growing pieces + remainder) give the same point
    monkeypatch.setenv("ARK_HIP_STREAM_GROWING", "0")
    assert np.array_equal(aff(cid, A.msm_bigint(cid, bases, sc)), want)
    monkeypatch.delenv("ARK_HIP_STREAM_GROWING")
    m = (3 << 18) + 1001
    want_m = S.mul_gen(cid, S.dlog_of_msm(sc[:m], S.A0, S.B0, r), r)
    for _ in range(2):
        assert np.array_equal(aff(cid, A.msm_bigint(cid, bases[:m], sc[:m])), want_m)


def test_auto_prepare_after_hits():
    cid = O.CID["BLS12_381_G1"]
    n = 3000
    bases = O.gen_bases(cid, A4, B4, n)
    A.base_cache_config(-1, 2)                                  # the third call builds the per-window table
    for k in range(5):
        scalars = O.gen_scalars(sf(cid), 40 + k, n)
        assert np.array_equal(aff(cid, A.msm_bigint(cid, bases, scalars)), oracle_aff(cid, bases, scalars)), k
    # content replaced under a prepared entry: table dropped, plain path again, still exact
    bases2 = bases.copy()
    scalars = O.gen_scalars(sf(cid), 1, n)
    for k in range(4):
        assert np.array_equal(aff(cid, A.msm_bigint(cid, bases2, scalars)), oracle_aff(cid, bases2, scalars))
    bases2[:] = O.gen_bases(cid, C4, A4, n)
    assert np.array_equal(aff(cid, A.msm_bigint(cid, bases2, scalars)), oracle_aff(cid, bases2, scalars))


def test_multi_device_entry_caches_per_shard(monkeypatch):
    monkeypatch.setenv("ARK_HIP_OVERSUBSCRIBE", "1")
    cid = O.CID["BLS12_381_G1"]
    n = 5001
    bases = O.gen_bases(cid, A4, B4, n)
    scalars = O.gen_scalars(sf(cid), 8, n)
    want = oracle_aff(cid, bases, scalars)
    for _ in range(2):
        assert np.array_equal(aff(cid, A.msm_bigint_multi(cid, 3, bases, scalars)), want)


def test_second_lane_waits_for_context_stream_producers():
    """a device iFFT queued on the context stream produces the scalars of an asynchronous MSM that lands on
    the second lane (a job is already in flight): the lane must start behind the transform."""
    import torch
    cid = O.CID["BLS12_381_G1"]
    fid = O.FID["BLS12_381_FR"]
    k = 16
    n = 1 << k
    bases = O.gen_bases(cid, A4, B4, n)
    pb = A.PreparedBases(cid, bases)
    dom = A.Radix2EvaluationDomain.new("BLS12_381_FR", n)
    evals = O.gen_scalars(fid, 21, n, montgomery=True)
    coeffs = O.fft(fid, evals, k, None, True, 4)               # the oracle's inverse transform
    want = oracle_aff(cid, bases, coeffs.reshape(n, 4), montgomery_scalars=True)
    other = torch.from_numpy(O.gen_scalars(sf(cid), 2, n).view(np.int64)).cuda()
    L = lib()
    for rep in range(4):
        x = torch.from_numpy(evals.view(np.int64)).cuda()
        torch.cuda.synchronize()
        j0 = pb.msm_bigint_async(other)                        # occupies lane 0
        check(L.ark_hip_ifft_in_place_device(dom.field, C.byref(dom._s), x.data_ptr()), "ifft")   # asynchronous
        j1 = pb.msm_bigint_async(x, montgomery=True)           # lane 1, reads the transform's output
        got = j1.wait()
        j0.wait()
        assert np.array_equal(aff(cid, got), want), rep
    pb.free()
