"""The headline configuration under `pytest -m gpu`: BLS12-381 G1 MSM at 2^24 -- BASELINE.json config 2,
the size `bench.py` quotes `value` on -- through every entry that serves it: the plain device entry (c = 20, 13 windows),
a prepared base set (c = 22, 12 windows), and the host-pointer entry the Rust hooks bind (`ark_hip_msm_sw`: default =
bases and scalars streamed in tapered pieces; pinned = scalars only in growing pieces; Fr scalars in Montgomery form as
`SWCurveConfig::msm` passes them), each against k*G with k = sum s_i (a + i b) computed exactly and k*G taken from the
ORACLE (reference: test-templates/src/msm.rs:17-32 compares msm against the naive sum; here the naive sum is the closed
form).  Plus BASELINE config 3 through the host-pointer FFT entries at 2^22 (forward, inverse, coset) limb for limb."""
import os
import sys

import numpy as np
import pytest

import algebra_amd as A
import oracle_lib as O

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import synth as S  # noqa: E402

pytestmark = pytest.mark.gpu

CNAME, LOGN = "BLS12_381_G1", 24


@pytest.fixture(scope="module")
def headline():
    import torch
    cid = O.CID[CNAME]
    r = S.R["BLS12_381_FR"]
    n = 1 << LOGN
    bases = S.grow_bases(cid, n, S.A0, S.B0, r)
    sc = S.gen_scalars(n, 0x2424, r)
    k = S.dlog_of_msm(sc, S.A0, S.B0, r)
    kg = O.to_affine(cid, O.scalar_mul(cid, O.generator(cid), S.limbs4(k)))   # independent of the device arithmetic
    yield {"cid": cid, "r": r, "n": n, "bases": bases, "sc": sc, "kg": kg}
    del bases
    torch.cuda.empty_cache()


def test_plan_is_the_one_the_bench_times():
    cid = O.CID[CNAME]
    assert A.msm_plan(cid, 1 << LOGN, prepared=False) == (20, 13)
    assert A.msm_plan(cid, 1 << LOGN, prepared=True) == (22, 12)


def test_headline_2_24_plain_device_entry(headline):
    import torch
    h = headline
    d_sc = torch.from_numpy(h["sc"].view(np.int64)).cuda()
    got = A.msm_bigint(h["cid"], h["bases"], d_sc)               # ark_hip_msm_sw_device: what bench.py's `value` times
    assert np.array_equal(A.into_affine(h["cid"], got), h["kg"])
    del d_sc


def test_headline_2_24_prepared_base_set(headline):
    import torch
    h = headline
    d_sc = torch.from_numpy(h["sc"].view(np.int64)).cuda()
    pb = A.PreparedBases(h["cid"], h["bases"])
    try:
        assert np.array_equal(A.into_affine(h["cid"], pb.msm_bigint(d_sc)), h["kg"])
    finally:
        pb.free()
        del d_sc
        torch.cuda.empty_cache()


def test_headline_2_24_host_pointer_entry(headline):
    """ark_hip_msm_sw from pageable host memory: cache off (bases + scalars streamed), default settings (miss, then a hit
    validated by the full-content hash of 1.5 GiB), pinned first call and repeat, Montgomery scalars."""
    h = headline
    cid, n = h["cid"], h["n"]
    host_bases = h["bases"].cpu().numpy().view(np.uint64).reshape(n, -1)
    A.base_cache_config(0, -1)
    s0 = A.base_cache_stats()
    assert np.array_equal(A.into_affine(cid, A.msm_bigint(cid, host_bases, h["sc"])), h["kg"])
    s1 = A.base_cache_stats()
    assert s1["entries"] == 0 and s1["pinned"] == 0 and s1["pinned_hits"] == s0["pinned_hits"]   # nothing retained
    A.base_cache_config(-2, -1)                                  # the default: verified cache
    for _ in range(2):
        assert np.array_equal(A.into_affine(cid, A.msm_bigint(cid, host_bases, h["sc"])), h["kg"])
    s2 = A.base_cache_stats()
    assert s2["misses"] - s1["misses"] == 1 and s2["hits"] - s1["hits"] == 1 and s2["bytes"] >= host_bases.nbytes
    A.base_cache_clear()
    with A.pin_bases(cid, host_bases):
        for _ in range(2):
            assert np.array_equal(A.into_affine(cid, A.msm_bigint(cid, host_bases, h["sc"])), h["kg"])
        # SWCurveConfig::msm hands over Fr elements: s R mod r, converted back on the device (mod.rs:60-62)
        fid = O.curve_info(cid)[1]
        mont = O.field_op(fid, "from_bigint", h["sc"]).reshape(n, 4)
        assert np.array_equal(A.into_affine(cid, A.msm(cid, host_bases, mont)), h["kg"])
        assert A.base_cache_stats()["pinned_hits"] - s0["pinned_hits"] >= 2


def test_fft_2_22_host_pointer_entries_vs_oracle():
    """BASELINE config 3 as the trait surface calls it: `Vec<Fr>` in host memory, in place (radix2/mod.rs:140-153)."""
    fname, log_n = "BLS12_381_FR", 22
    fid = O.FID[fname]
    n = 1 << log_n
    threads = min(os.cpu_count() or 8, 64)
    x = O.gen_scalars(fid, 0x2222, n, montgomery=True)
    d = A.Radix2EvaluationDomain.new(fname, n)
    y = d.fft(x)                                                  # numpy in -> ark_hip_fft_in_place on a host copy
    assert isinstance(y, np.ndarray)
    assert np.array_equal(y.reshape(-1), O.fft(fid, x, log_n, None, False, threads))
    assert np.array_equal(d.ifft(y).reshape(-1), x.reshape(-1))   # round trip through ark_hip_ifft_in_place
    assert np.array_equal(d.ifft(x).reshape(-1), O.fft(fid, x, log_n, None, True, threads))
    gen = O.field_const(fid, 3)                                   # Fr::GENERATOR, the offset of poly/benches/fft.rs:107
    dc = d.get_coset(gen)
    assert np.array_equal(dc.fft(x).reshape(-1), O.fft(fid, x, log_n, gen, False, threads))


def test_2_20_bases_from_the_oracle_through_every_entry():
    """the at-size tests above build their bases ON THE DEVICE (tools/synth.py grow_bases) and check
    k*G -- independent of the MSM kernels, but no oracle-generated base set was fed through the device entries above 2^18.
    Here 2^20 bases come from the ORACLE's generator (oracle/ gen_bases: host arithmetic only), the expected point from the
    oracle's own msm_bigint_wnaf restatement on the same arrays, and the product path runs them through the plain device
    entry, a prepared base set, and the host-pointer entry (streamed with the cache off; default settings: miss, then a hit
    validated by the keyed hash)."""
    import torch
    cid = O.CID[CNAME]
    n = 1 << 20
    a4 = np.array([0x5EED, 11, 0, 0], dtype=np.uint64)
    b4 = np.array([0xFACE, 0, 5, 0], dtype=np.uint64)
    bases = O.gen_bases(cid, a4, b4, n)
    sc = O.gen_scalars(O.curve_info(cid)[1], 0x2020, n)
    want = O.to_affine(cid, O.msm(cid, bases, sc, O.WNAF, 16))
    d_b = torch.from_numpy(bases.view(np.int64)).cuda()
    d_s = torch.from_numpy(sc.view(np.int64)).cuda()
    torch.cuda.synchronize()
    assert np.array_equal(A.into_affine(cid, A.msm_bigint(cid, d_b, d_s)), want)
    pb = A.PreparedBases(cid, d_b)
    assert np.array_equal(A.into_affine(cid, pb.msm_bigint(d_s)), want)
    pb.free()
    A.base_cache_config(0, 0)
    assert np.array_equal(A.into_affine(cid, A.msm_bigint(cid, bases, sc)), want)
    A.base_cache_config(-2, 0)
    A.base_cache_clear()
    s0 = A.base_cache_stats()
    for _ in range(2):
        assert np.array_equal(A.into_affine(cid, A.msm_bigint(cid, bases, sc)), want)
    s1 = A.base_cache_stats()
    assert (s1["misses"] - s0["misses"], s1["hits"] - s0["hits"]) == (1, 1)
    A.base_cache_clear()
