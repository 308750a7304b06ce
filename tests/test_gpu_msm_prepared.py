"""Prepared base sets (ark_hip_msm_bases_*: per-window multiples, one shared bucket set), the asynchronous / streaming
entry points (jobs, pinned uploads, msm_chunks), the one-process multi-GPU entry and HashMapPippenger -- parity against
the oracle through the C ABI."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

import algebra_amd as A
import oracle_lib as O
import hip_lib as H
import pyref as P
from algebra_amd._lib import check, lib

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import synth as S  # noqa: E402

pytestmark = pytest.mark.gpu

A4 = np.array([0xA11CE, 1, 2, 0], dtype=np.uint64)
B4 = np.array([0xB0B, 3, 0, 0], dtype=np.uint64)


def sf(cid):
    return O.curve_info(cid)[1]


def aff(cid, p):
    return A.into_affine(cid, p)


def oracle_aff(cid, bases, scalars, **kw):
    return O.to_affine(cid, O.msm(cid, bases, scalars, O.SIGNED, 4, **kw))


@pytest.mark.parametrize("cname", O.CURVES)
@pytest.mark.parametrize("n", [0, 1, 2, 33, 1000, 1 << 12])
def test_prepared_matches_oracle(cname, n):
    cid = O.CID[cname]
    if cname.endswith("G2") and n > 1000:
        n = 1 << 11
    bases = O.gen_bases(cid, A4, B4, max(n, 1))[:n]
    scalars = O.gen_scalars(sf(cid), 0xBEEF + n, max(n, 1))[:n]
    pb = A.PreparedBases(cid, bases)
    info = pb.info()
    assert info["n"] == n and info["windows"] >= 1
    assert np.array_equal(aff(cid, pb.msm_bigint(scalars)), oracle_aff(cid, bases, scalars))
    if n >= 33:
        # msm_unchecked semantics: a shorter scalar vector uses the first bases only
        k = n // 3
        assert np.array_equal(aff(cid, pb.msm_bigint(scalars[:k])), oracle_aff(cid, bases[:k], scalars[:k]))
        mont = O.gen_scalars(sf(cid), 5, n, montgomery=True)
        assert np.array_equal(aff(cid, pb.msm(mont)), oracle_aff(cid, bases, mont, montgomery_scalars=True))
        with pytest.raises(A.MsmLengthMismatch) as ei:
            pb.msm(mont[:-1])
        assert ei.value.min_len == n - 1
    pb.free()


@pytest.mark.parametrize("c", [3, 4, 7, 10, 13, 16])
def test_prepared_window_layouts(c, monkeypatch):
    # every window layout the planner can pick (uniform / mixed widths, sparse top window), forced through the knob
    monkeypatch.setenv("ARK_HIP_MSM_C_PREPARED", str(c))
    for cname in ("BLS12_381_G1", "BN254_G1", "BLS12_377_G1"):
        cid = O.CID[cname]
        n = 700
        bases = O.gen_bases(cid, A4, B4, n)
        scalars = O.gen_scalars(sf(cid), 31 + c, n)
        pb = A.PreparedBases(cid, bases)
        assert pb.info()["window_bits"] == c
        assert np.array_equal(aff(cid, pb.msm_bigint(scalars)), oracle_aff(cid, bases, scalars)), (cname, c)
        pb.free()


@pytest.mark.parametrize("cname", ["BN254_G1", "BLS12_381_G1", "BLS12_377_G2"])
def test_prepared_edge_cases(cname):
    cid = O.CID[cname]
    r = P.Curve(cname).r
    fw = O.fe_words(cid)
    n = 64
    bases = O.gen_bases(cid, A4, B4, n)
    rng = np.random.default_rng(3)
    lim = lambda v: np.array(P.to_limbs(v % r, 4), dtype=np.uint64)
    vals = [0, 1, r - 1, 2, r - 2, 255, r - 255, (1 << 64) - 1, r - (1 << 64), (r - 1) // 2, (r + 1) // 2, r, r + 5]
    vals += [int.from_bytes(rng.bytes(40), "little") % r for _ in range(n - len(vals))]
    scalars = np.stack([np.array(P.to_limbs(v, 4), dtype=np.uint64) for v in vals])
    # identity bases, duplicate bases (doubling inside a bucket), P and -P (bucket back to infinity)
    b = bases.copy()
    b[::5] = 0
    b[7] = b[6]
    b[11] = b[10]
    b[11, fw:] = O.basefield_op(cid, "neg", b[10, fw:])
    s = scalars.copy()
    s[6] = s[7] = lim(12345)
    s[10] = s[11] = lim(0xDEADBEEF)
    pb = A.PreparedBases(cid, b)
    assert np.array_equal(aff(cid, pb.msm_bigint(s)), oracle_aff(cid, b, s))
    z = pb.msm_bigint(np.zeros((n, 4), dtype=np.uint64))
    assert np.array_equal(aff(cid, z), np.zeros(2 * fw, dtype=np.uint64))
    # every scalar the same: one run per window takes all points
    same = np.tile(lim(0x1234567 + (1 << 200)), (n, 1))
    assert np.array_equal(aff(cid, pb.msm_bigint(same)), oracle_aff(cid, b, same))
    pb.free()


@pytest.mark.parametrize("cname", ["BLS12_381_G1", "BN254_G1", "BLS12_377_G2"])
def test_prepared_skewed_scalars_heavy_runs(cname):
    # heavy runs in the shared-bucket layout: chunk partials -> per-run sums -> msm_apply_heavy_kernel; several heavy
    # windows landing in ONE bucket (all digits equal) exercise the single-owner rule
    import torch
    cid = O.CID[cname]
    r = P.Curve(cname).r
    n = 1 << (13 if cname.endswith("G2") else 15)
    seed = O.gen_bases(cid, A4, B4, 1 << 10)
    d = H.gpu_extend_bases(cid, seed, n, lambda m: _delta(cid, m))
    bases = d.cpu().numpy().view(np.uint64).reshape(n, -1)
    rng = np.random.default_rng(21)
    lim = lambda v: P.to_limbs(v % r, 4)
    pb = A.PreparedBases(cid, d)
    c = pb.info()["window_bits"]
    rep = sum(3 << (c * j) for j in range(pb.info()["windows"] - 1)) % r  # the same digit in (almost) every window
    cases = {
        "all_equal": [0x1234567 + (1 << 200)] * n,
        "same_digit_every_window": [rep] * n,
        "bool": [int(x) for x in rng.integers(0, 2, size=n)],
        "pm_u8": [int(x) if i % 2 else (r - int(x)) % r for i, x in enumerate(rng.integers(0, 256, size=n))],
        "half_equal_half_random": [7 if i % 2 else int.from_bytes(rng.bytes(40), "little") % r for i in range(n)],
    }
    for name, vals in cases.items():
        scalars = np.array([lim(v) for v in vals], dtype=np.uint64)
        got = pb.msm_bigint(torch.from_numpy(scalars.view(np.int64)).cuda())
        exp = O.msm(cid, bases, scalars, O.SIGNED, 8)
        assert np.array_equal(aff(cid, got), O.to_affine(cid, exp)), (cname, name)
    pb.free()


def _delta(cid, m):
    r = S.R[O.FIELDS[sf(cid)]]
    k = (m * P.from_limbs(B4)) % r
    return O.to_affine(cid, O.scalar_mul(cid, O.generator(cid), np.array(P.to_limbs(k, 4), dtype=np.uint64)))


@pytest.mark.parametrize("cname,logn", [("BLS12_381_G1", 20), ("BLS12_381_G1", 22), ("BN254_G1", 18), ("BLS12_377_G2", 17)])
def test_prepared_large_dlog(cname, logn):
    import torch
    cid = O.CID[cname]
    r = S.R[O.FIELDS[sf(cid)]]
    n = 1 << logn
    bases = S.grow_bases(cid, n, S.A0, S.B0, r)
    pb = A.PreparedBases(cid, bases)
    for seed in (1, 2):
        sc = S.gen_scalars(n, seed + logn, r)
        got = pb.msm_bigint(torch.from_numpy(sc.view(np.int64)).cuda())
        k = S.dlog_of_msm(sc, S.A0, S.B0, r)
        kg = O.to_affine(cid, O.scalar_mul(cid, O.generator(cid), S.limbs4(k)))
        assert np.array_equal(aff(cid, got), kg), (cname, logn, seed)
    pb.free()


def test_async_jobs_in_flight_and_busy():
    import torch
    cid = O.CID["BLS12_381_G1"]
    n = 5000
    bases = O.gen_bases(cid, A4, B4, n)
    d_b = torch.from_numpy(bases.view(np.int64)).cuda()
    scs = [O.gen_scalars(sf(cid), 70 + i, n) for i in range(4)]
    d_s = [torch.from_numpy(s.view(np.int64)).cuda() for s in scs]
    jobs = [A.msm_bigint_async(cid, d_b, d_s[i]) for i in range(4)]
    with pytest.raises(A.ArkHipError) as ei:          # a fifth job: ARK_HIP_ERR_BUSY
        A.msm_bigint_async(cid, d_b, d_s[0])
    assert ei.value.code == -6
    for i in (2, 0, 3, 1):                            # waits in any order
        assert np.array_equal(aff(cid, jobs[i].wait()), oracle_aff(cid, bases, scs[i])), i
    # slots are free again
    assert np.array_equal(aff(cid, A.msm_bigint_async(cid, d_b, d_s[1]).wait()), oracle_aff(cid, bases, scs[1]))


def test_prepared_async_pinned_upload_pipeline():
    # steady state of a prover: one resident SRS, scalar vectors arriving from the host (pinned memory), the upload of
    # MSM k+1 overlapping MSM k
    L = lib()
    cid = O.CID["BLS12_381_G1"]
    n = 20000
    bases = O.gen_bases(cid, A4, B4, n)
    pb = A.PreparedBases(cid, bases)
    ptr = C.c_void_p()
    check(L.ark_hip_host_alloc(6 * n * 32, C.byref(ptr)), "host_alloc")
    pinned = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint64)), shape=(6, n, 4))
    scs = []
    for i in range(6):
        s = O.gen_scalars(sf(cid), 300 + i, n)
        pinned[i] = s
        scs.append(s)
    results = []
    pending = []
    for i in range(6):
        pending.append(pb.msm_bigint_async(pinned[i]))
        if len(pending) == 2:
            results.append(pending.pop(0).wait())
    while pending:
        results.append(pending.pop(0).wait())
    for i in range(6):
        assert np.array_equal(aff(cid, results[i]), oracle_aff(cid, bases, scs[i])), i
    pb.free()
    check(L.ark_hip_host_free(ptr), "host_free")


@pytest.mark.parametrize("cname", ["BLS12_381_G1", "BLS12_377_G2"])
def test_msm_chunks_matches_reference_semantics(cname):
    # VariableBaseMSM::msm_chunks (variable_base/mod.rs:119-150): streams aligned at their END, fixed-size steps
    cid = O.CID[cname]
    nb, ns = 2500, 2100
    bases = O.gen_bases(cid, A4, B4, nb)
    mont = O.gen_scalars(sf(cid), 8, ns, montgomery=True)
    exp = oracle_aff(cid, bases[nb - ns:], mont, montgomery_scalars=True)
    for step in (256, 1000, 4096, 0):
        assert np.array_equal(aff(cid, A.msm_chunks(cid, bases, mont, step)), exp), step
    with pytest.raises(AssertionError):
        A.msm_chunks(cid, bases[:10], mont)
    z = A.msm_chunks(cid, bases, np.zeros((0, 4), dtype=np.uint64))
    assert np.array_equal(aff(cid, z), np.zeros(2 * O.fe_words(cid), dtype=np.uint64))


def test_msm_multi_one_process_many_devices(monkeypatch):
    # ark_hip_msm_sw_multi: one host thread + context per device.  On a one-GPU box ARK_HIP_OVERSUBSCRIBE=1 maps the
    # logical devices onto the same GPU (separate contexts / streams / workspaces), which exercises the same code.
    monkeypatch.setenv("ARK_HIP_OVERSUBSCRIBE", "1")
    cid = O.CID["BLS12_381_G1"]
    n = 10007
    bases = O.gen_bases(cid, A4, B4, n)
    scalars = O.gen_scalars(sf(cid), 12, n)
    exp = oracle_aff(cid, bases, scalars)
    for g in (1, 2, 3, 8):
        assert np.array_equal(aff(cid, A.msm_bigint_multi(cid, g, bases, scalars)), exp), g
    # resident shards through the raw C entry
    L = lib()
    G = 4
    d_b = (C.c_void_p * G)()
    d_s = (C.c_void_p * G)()
    cnt = (C.c_size_t * G)()
    from algebra_amd.dist import shard_bounds
    for g in range(G):
        lo, hi = shard_bounds(n, g, G)
        check(L.ark_hip_set_device(g), "set_device")
        pb, ps = C.c_void_p(), C.c_void_p()
        check(L.ark_hip_malloc((hi - lo) * bases.shape[1] * 8, C.byref(pb)), "malloc")
        check(L.ark_hip_malloc((hi - lo) * 32, C.byref(ps)), "malloc")
        check(L.ark_hip_memcpy_h2d(pb, bases[lo:hi].ctypes.data_as(C.c_void_p), (hi - lo) * bases.shape[1] * 8), "h2d")
        check(L.ark_hip_memcpy_h2d(ps, scalars[lo:hi].ctypes.data_as(C.c_void_p), (hi - lo) * 32), "h2d")
        d_b[g], d_s[g], cnt[g] = pb, ps, hi - lo
    out = np.zeros(18, dtype=np.uint64)
    check(L.ark_hip_msm_sw_multi_device(cid, G, d_b, d_s, cnt, 0, out.ctypes.data_as(C.c_void_p)), "multi_device")
    assert np.array_equal(aff(cid, out), exp)
    for g in range(G):
        check(L.ark_hip_set_device(g), "set_device")
        check(L.ark_hip_free(d_b[g]), "free")
        check(L.ark_hip_free(d_s[g]), "free")
    check(L.ark_hip_set_device(0), "set_device")
    # prepared shards, one per (logical) device: ark_hip_msm_prepared_multi cuts the host scalars at the shard sizes
    G = 3
    shards = []
    for g in range(G):
        lo, hi = shard_bounds(n, g, G)
        check(L.ark_hip_set_device(g), "set_device")
        shards.append(A.PreparedBases(cid, bases[lo:hi]))
    check(L.ark_hip_set_device(0), "set_device")
    hs = (C.c_void_p * G)(*[sh._h for sh in shards])
    out = np.zeros(18, dtype=np.uint64)
    check(L.ark_hip_msm_prepared_multi(G, hs, scalars.ctypes.data_as(C.c_void_p), n, 0, out.ctypes.data_as(C.c_void_p)),
          "prepared_multi")
    assert np.array_equal(aff(cid, out), exp)
    # fewer scalars than bases (msm_unchecked truncates): the last shard gets a short / empty piece
    k = shard_bounds(n, 1, G)[1] + 5
    check(L.ark_hip_msm_prepared_multi(G, hs, scalars.ctypes.data_as(C.c_void_p), k, 0, out.ctypes.data_as(C.c_void_p)),
          "prepared_multi short")
    assert np.array_equal(aff(cid, out), oracle_aff(cid, bases[:k], scalars[:k]))
    assert L.ark_hip_msm_prepared_multi(G, hs, scalars.ctypes.data_as(C.c_void_p), n + 1, 0, out.ctypes.data_as(C.c_void_p)) != 0
    for sh in shards:
        sh.free()


def test_hashmap_pippenger_matches_naive_sum():
    # HashMapPippenger (stream_pippenger.rs:68-128): repeated bases have their Fr scalars added before the MSM
    cid = O.CID["BLS12_381_G1"]
    fid = sf(cid)
    nb = 37
    bases = O.gen_bases(cid, A4, B4, nb)
    rng = np.random.default_rng(5)
    idx = rng.integers(0, nb, size=400)
    mont = O.gen_scalars(fid, 44, 400, montgomery=True)
    p = A.HashMapPippenger(cid, 16)
    for i, s in zip(idx, mont):
        p.add(bases[i], s)
    exp = oracle_aff(cid, bases[idx], mont, montgomery_scalars=True)
    assert np.array_equal(aff(cid, p.finalize()), exp)
    assert np.array_equal(aff(cid, A.HashMapPippenger(cid, 4).finalize()), np.zeros(2 * O.fe_words(cid), dtype=np.uint64))


@pytest.mark.parametrize("cname", O.CURVES)
def test_batch_mul_matches_oracle(cname):
    # ScalarMul::batch_mul / BatchMulPreprocessing (ec/src/scalar_mul/mod.rs:104-251; the doc-test there computes
    # g, s g, s^2 g, ...): v[i] * g for one base, affine results, against the oracle's restatement
    import torch
    cid = O.CID[cname]
    fid = sf(cid)
    n = 300 if cname.endswith("G2") else 2000
    k = np.array([0xC0FFEE, 7, 0, 0], dtype=np.uint64)
    base = O.scalar_mul(cid, O.generator(cid), k)                     # a Projective with z != 1
    canon = O.gen_scalars(fid, 61, n)
    r = S.R[O.FIELDS[fid]]
    canon[0] = 0
    canon[1] = [1, 0, 0, 0]
    canon[2] = P.to_limbs(r - 1, 4)
    mont = O.field_op(fid, "from_bigint", canon).reshape(n, 4)
    exp = O.batch_mul(cid, base, canon)
    t = A.BatchMulPreprocessing(cid, base, n)
    assert np.array_equal(t.batch_mul(mont), exp)                                  # Fr entry (the reference's)
    assert np.array_equal(t.batch_mul(canon, montgomery=False), exp)               # canonical entry
    d = t.batch_mul(torch.from_numpy(mont.view(np.int64)).cuda())
    assert np.array_equal(d.cpu().numpy().view(np.uint64), exp)                    # device-resident entry
    t.free()
    assert np.array_equal(A.batch_mul(cid, base, mont[:5]), exp[:5])
    ident = O.scalar_mul(cid, O.generator(cid), np.zeros(4, dtype=np.uint64))      # base = identity -> all identity
    assert not A.batch_mul(cid, ident, mont[:7]).any()


@pytest.mark.parametrize("window", [0, 4, 7, 16])
@pytest.mark.parametrize("cname", ["BN254_G1", "BLS12_381_G1", "BLS12_377_G2"])
def test_batch_mul_window_geometries(cname, window, monkeypatch):
    # BatchMulPreprocessing::new sizes the table from num_scalars (ec/src/scalar_mul/mod.rs:222-228); the device table's
    # rule: 12-bit rows below 2^24 scalars, 16-bit rows from there (window 0 here = that default, asked for through
    # num_scalars); other row widths through the measurement knob.  The window never changes the result.
    cid = O.CID[cname]
    fid = sf(cid)
    r = S.R[O.FIELDS[fid]]
    n = 200 if cname.endswith("G2") else 700
    if window:
        monkeypatch.setenv("ARK_HIP_BATCHMUL_WINDOW", str(window))
    base = O.scalar_mul(cid, O.generator(cid), np.array([0xBA5E, 3, 0, 0], dtype=np.uint64))
    canon = O.gen_scalars(fid, 67 + window, n)
    canon[0] = 0
    canon[1] = P.to_limbs(r - 1, 4)
    t = A.BatchMulPreprocessing(cid, base, (1 << 24) if window == 0 else n)
    assert np.array_equal(t.batch_mul(canon, montgomery=False), O.batch_mul(cid, base, canon))
    t.free()


@pytest.mark.parametrize("cname", ["BN254_G1", "BLS12_381_G1", "BLS12_377_G1"])
def test_batch_mul_unreduced_scalar_hits_the_doubling_branch(cname):
    # "any BigInt<4> is multiplied exactly": s = d 2^252 + (d 2^252 - r) with 0 <= d 2^252 - r < 2^252 makes the sum of
    # the lower 21 rows EQUAL to the top row's entry (d 2^252 g = (d 2^252 - r) g) -- the addition must double.
    cid = O.CID[cname]
    fid = sf(cid)
    r = S.R[O.FIELDS[fid]]
    d = next(d for d in range(1, 16) if 0 <= (d << 252) - r < (1 << 252))
    s = (d << 252) + ((d << 252) - r)
    assert s < (1 << 256)
    base = O.scalar_mul(cid, O.generator(cid), np.array([0x5EED, 0, 0, 0], dtype=np.uint64))
    sc = np.stack([P.to_limbs(s, 4), P.to_limbs(5, 4), P.to_limbs(s, 4)]).astype(np.uint64)
    baff = O.to_affine(cid, base)
    exp = np.stack([O.to_affine(cid, O.scalar_mul(cid, baff, P.to_limbs(v % r, 4))) for v in (s, 5, s)])
    t = A.BatchMulPreprocessing(cid, base, 3)
    assert np.array_equal(t.batch_mul(sc, montgomery=False).reshape(exp.shape), exp)
    t.free()


def test_two_msm_lanes_under_concurrent_threads_and_batched_fft():
    # Two MSM lanes per device (a job enqueued while another is in flight runs on the second stream / workspace) and
    # up to three FFTs in flight: three host threads mix device-pointer MSMs (synchronous = enqueue + wait, so calls
    # from different threads overlap), pairs of asynchronous prepared jobs waited in reverse order, and batched FFTs.
    # Every result must equal the oracle's.
    import threading
    import torch
    cid = O.CID["BLS12_381_G1"]
    fid = O.FID["BLS12_381_FR"]
    n = 6000
    bases = O.gen_bases(cid, A4, B4, n)
    d_b = torch.from_numpy(bases.view(np.int64)).cuda()
    pb = A.PreparedBases(cid, bases)
    scs = [O.gen_scalars(sf(cid), 500 + i, n - 37 * i) for i in range(4)]
    d_s = [torch.from_numpy(s.view(np.int64)).cuda() for s in scs]
    want = [oracle_aff(cid, bases[: s.shape[0]], s) for s in scs]
    log_n = 11
    dom = A.Radix2EvaluationDomain.new("BLS12_381_FR", 1 << log_n)
    xs = [O.gen_scalars(fid, 40 + i, 1 << log_n, montgomery=True) for i in range(4)]
    fexp = [O.fft(fid, x, log_n, None, False, 2) for x in xs]
    errors = []

    def worker(t):
        try:
            for it in range(10):
                k = (t + it) % 4
                kind = (t + 2 * it) % 3
                if kind == 0:
                    try:
                        got = A.msm_bigint(cid, d_b[: scs[k].shape[0]], d_s[k])
                    except A.ArkHipError as ex:     # four jobs already in flight on the device
                        if ex.code != -6:
                            raise
                        continue
                    if not np.array_equal(aff(cid, got), want[k]):
                        errors.append(("msm", t, it))
                elif kind == 1:
                    k2 = (k + 1) % 4
                    j1 = None
                    try:
                        j1 = pb.msm_bigint_async(d_s[k])
                        j2 = pb.msm_bigint_async(d_s[k2])
                    except A.ArkHipError as ex:     # four jobs in flight on the device: a legal answer under load
                        if j1 is not None:
                            j1.wait()               # a job that was enqueued is always waited for (it holds a slot)
                        if ex.code != -6:
                            raise
                        continue
                    r2, r1 = j2.wait(), j1.wait()
                    if not (np.array_equal(aff(cid, r1), want[k]) and np.array_equal(aff(cid, r2), want[k2])):
                        errors.append(("prepared", t, it))
                else:
                    dev = [torch.from_numpy(x.view(np.int64)).cuda() for x in xs]
                    dom.fft_batch_in_place(dev)
                    for i in range(4):
                        if not np.array_equal(dev[i].cpu().numpy().view(np.uint64).reshape(-1), fexp[i]):
                            errors.append(("fft", t, it, i))
        except Exception as ex:  # noqa: BLE001
            errors.append(("exc", t, repr(ex)))

    th = [threading.Thread(target=worker, args=(t,)) for t in range(3)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    pb.free()
    assert not errors, errors[:5]
