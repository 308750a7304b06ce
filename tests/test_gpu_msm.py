"""MSM parity (GPU): the HIP path through the C ABI vs the oracle's restatement of
msm_bigint_wnaf / the naive sum, mirroring test-templates/src/msm.rs:17-110 and the edge cases of
SURVEY.md 8(d).  Comparison is on into_affine() limbs, bit-exact."""
import numpy as np
import pytest

import algebra_amd as A
import oracle_lib as O
import hip_lib as H
import pyref as P

pytestmark = pytest.mark.gpu

A4 = np.array([0xA11CE, 1, 2, 0], dtype=np.uint64)
B4 = np.array([0xB0B, 3, 0, 0], dtype=np.uint64)


def sf(cid):
    return O.curve_info(cid)[1]


def check(cid, bases, scalars, variant=O.SIGNED, threads=4):
    got = A.msm_bigint(cid, bases, scalars)
    exp = O.msm(cid, bases, scalars, variant, threads)
    assert np.array_equal(A.into_affine(cid, got), O.to_affine(cid, exp))
    return got


@pytest.mark.parametrize("cname", O.CURVES)
@pytest.mark.parametrize("n", [0, 1, 2, 31, 32, 33, 1000, 1 << 12])
def test_msm_random_matches_oracle(cname, n):
    # test_var_base_msm (msm.rs:17-32) at n = 2^10-ish plus the n in {0,1,31,32,33} boundary sizes
    cid = O.CID[cname]
    if cname.endswith("G2") and n > 1000:
        n = 1 << 11
    bases = O.gen_bases(cid, A4, B4, max(n, 1))[:n]
    scalars = O.gen_scalars(sf(cid), 0xA11CE + n, max(n, 1))[:n]
    check(cid, bases, scalars)


@pytest.mark.parametrize("cname", ["BN254_G1", "BLS12_381_G1", "BLS12_377_G2"])
def test_msm_montgomery_scalar_entry(cname):
    # VariableBaseMSM::msm takes Fr elements: the into_bigint pass (mod.rs:60-62) runs on the device
    cid = O.CID[cname]
    n = 777
    bases = O.gen_bases(cid, A4, B4, n)
    mont = O.gen_scalars(sf(cid), 99, n, montgomery=True)
    got = A.msm(cid, bases, mont)
    exp = O.msm(cid, bases, mont, O.SIGNED, 4, montgomery_scalars=True)
    assert np.array_equal(A.into_affine(cid, got), O.to_affine(cid, exp))
    # msm_unchecked truncates to the shorter input (mod.rs:59-64)
    got2 = A.msm_unchecked(cid, bases, mont[:500])
    exp2 = O.msm(cid, bases[:500], mont[:500], O.SIGNED, 4, montgomery_scalars=True)
    assert np.array_equal(A.into_affine(cid, got2), O.to_affine(cid, exp2))


@pytest.mark.parametrize("cname", ["BN254_G1", "BLS12_381_G1", "BLS12_377_G2"])
def test_msm_edge_cases(cname):
    cid = O.CID[cname]
    cvp = P.Curve(cname)
    r = cvp.r
    fw = O.fe_words(cid)
    n = 64
    bases = O.gen_bases(cid, A4, B4, n)
    rng = np.random.default_rng(3)
    lim = lambda v: np.array(P.to_limbs(v % r, 4), dtype=np.uint64)
    # zero scalars, scalar = 1, scalar = r-1, small and negative-small scalars
    vals = [0, 1, r - 1, 2, r - 2, 255, r - 255, (1 << 64) - 1, r - (1 << 64), (r - 1) // 2, (r + 1) // 2]
    vals += [int.from_bytes(rng.bytes(40), "little") % r for _ in range(n - len(vals))]
    scalars = np.stack([lim(v) for v in vals])
    check(cid, bases, scalars)
    # all-zero scalars -> identity
    z = A.msm_bigint(cid, bases, np.zeros((n, 4), dtype=np.uint64))
    assert np.array_equal(A.into_affine(cid, z), np.zeros(2 * fw, dtype=np.uint64))
    # identity bases are skipped
    b2 = bases.copy()
    b2[::3] = 0
    check(cid, b2, scalars)
    # duplicate bases with equal scalars: forces the doubling branch inside a bucket
    b3 = bases.copy()
    b3[1] = b3[0]
    b3[5] = b3[4]
    s3 = scalars.copy()
    s3[0] = s3[1] = lim(12345)
    s3[4] = s3[5]
    check(cid, b3, s3)
    # P and -P with equal scalars: bucket returns to infinity
    b4 = bases.copy()
    b4[3] = b4[2]
    b4[3, fw:] = O.basefield_op(cid, "neg", b4[2, fw:])
    s4 = scalars.copy()
    s4[2] = s4[3] = lim(0xDEADBEEF)
    check(cid, b4, s4)
    # every scalar the same: one bucket per window takes all points
    check(cid, bases, np.tile(lim(0x1234567 + (1 << 200)), (n, 1)))


@pytest.mark.parametrize("cname", ["BN254_G1", "BLS12_381_G1"])
def test_msm_mixed_scalar_classes(cname):
    # test_var_base_msm_mixed_scalars (msm.rs:36-72): +-bool, +-u8, +-u16, +-u32, +-u64, random, shuffled
    cid = O.CID[cname]
    r = P.Curve(cname).r
    rng = np.random.default_rng(8)
    m = 128
    vals = []
    for bits in (1, 8, 16, 32, 64):
        pos = [int(x) for x in rng.integers(0, 1 << min(bits, 63), size=m, dtype=np.uint64)]
        vals += pos + [(r - v) % r for v in pos]
    vals += [int.from_bytes(rng.bytes(40), "little") % r for _ in range(m)]
    perm = rng.permutation(len(vals))
    vals = [vals[i] for i in perm]
    n = len(vals)
    bases = O.gen_bases(cid, A4, B4, n)
    scalars = np.array([P.to_limbs(v, 4) for v in vals], dtype=np.uint64)
    check(cid, bases, scalars)


def test_msm_known_answer_table_bls12_381_g1():
    # reference KAT: k*G for k = 0..999 (curves/bls12_381/src/curves/tests/g1_uncompressed_valid_test_vectors.dat,
    # checked by tests/mod.rs:69-111); MSM over entries 1..44 with all-ones scalars = entry 990
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "bls12_381_g1_multiples.npz"))
    xy = g["xy"]  # canonical limbs [1000, 2, 6]
    fq = O.FID["BLS12_381_FQ"]
    mont = O.field_op(fq, "from_bigint", xy.reshape(-1, 6)).reshape(1000, 12)
    cid = O.CID["BLS12_381_G1"]
    ones = np.zeros((44, 4), dtype=np.uint64)
    ones[:, 0] = 1
    got = A.into_affine(cid, A.msm_bigint(cid, mont[1:45], ones))
    assert np.array_equal(got, mont[990])
    # sum_k k * (kG) over k = 1..13 = (sum k^2) G = 819 G
    sc = np.zeros((13, 4), dtype=np.uint64)
    sc[:, 0] = np.arange(1, 14)
    got = A.into_affine(cid, A.msm_bigint(cid, mont[1:14], sc))
    assert np.array_equal(got, mont[819])


@pytest.mark.parametrize("cname,logn", [("BN254_G1", 16), ("BLS12_381_G1", 18), ("BLS12_377_G2", 14)])
def test_msm_medium_vs_oracle_wnaf(cname, logn):
    # BASELINE config 1 size (BN254 2^16) and friends against the multi-threaded msm_bigint_wnaf restatement
    cid = O.CID[cname]
    n = 1 << logn
    seed = O.gen_bases(cid, A4, B4, 1 << 10)
    import torch
    d = H.gpu_extend_bases(cid, seed, n, lambda m: _delta(cid, m))
    bases = d.cpu().numpy().view(np.uint64).reshape(n, -1)
    # the device-grown bases are the oracle's arithmetic progression
    spot = O.gen_bases(cid, A4, B4, (1 << 10) + 8)
    assert np.array_equal(bases[: (1 << 10) + 8], spot)
    scalars = O.gen_scalars(sf(cid), 4242, n)
    got = A.msm_bigint(cid, d, torch.from_numpy(scalars.view(np.int64)).cuda())
    exp = O.msm(cid, bases, scalars, O.WNAF, 8)
    assert np.array_equal(A.into_affine(cid, got), O.to_affine(cid, exp))
    # and equals k*G with k = sum s_i (a + i b)  (exact discrete-log check)
    k = O.msm_dlog(cid, scalars, A4, B4)
    kg = O.scalar_mul(cid, O.generator(cid), k)
    assert np.array_equal(A.into_affine(cid, got), O.to_affine(cid, kg))


def _delta(cid, m):
    """(m*b) G as affine limbs."""
    r = P.MODULI[O.FIELDS[sf(cid)]][0]
    k = (m * P.from_limbs(B4)) % r
    return O.to_affine(cid, O.scalar_mul(cid, O.generator(cid), np.array(P.to_limbs(k, 4), dtype=np.uint64)))


@pytest.mark.parametrize("logn", [20, 22])
def test_msm_bls12_381_g1_large_dlog(logn):
    # BASELINE config 2: bit-exact at 2^20 / 2^22 through the discrete-log identity
    # (result == k*G, k = sum_i s_i (a + i b) mod r) -- a size-independent exact check.
    import torch
    cid = O.CID["BLS12_381_G1"]
    n = 1 << logn
    seed = O.gen_bases(cid, A4, B4, 1 << 10)
    d = H.gpu_extend_bases(cid, seed, n, lambda m: _delta(cid, m))
    scalars = O.gen_scalars(sf(cid), 777 + logn, n)
    got = A.msm_bigint(cid, d, torch.from_numpy(scalars.view(np.int64)).cuda())
    k = O.msm_dlog(cid, scalars, A4, B4)
    kg = O.scalar_mul(cid, O.generator(cid), k)
    assert np.array_equal(A.into_affine(cid, got), O.to_affine(cid, kg))


@pytest.mark.parametrize("cname", ["BLS12_381_G1", "BN254_G1", "BLS12_377_G2"])
def test_msm_skewed_scalars_heavy_buckets(cname):
    # non-uniform scalars (the reference's bool / u8 / "all equal" shapes, bench-templates/src/macros/ec.rs:244-372):
    # a few buckets receive thousands of points -> heavy-bucket kernels + wave-combined histogram atomics
    cid = O.CID[cname]
    r = P.Curve(cname).r
    n = 1 << (13 if cname.endswith("G2") else 15)
    seed = O.gen_bases(cid, A4, B4, 1 << 10)
    import torch
    d = H.gpu_extend_bases(cid, seed, n, lambda m: _delta(cid, m))
    bases = d.cpu().numpy().view(np.uint64).reshape(n, -1)
    rng = np.random.default_rng(21)
    lim = lambda v: P.to_limbs(v % r, 4)
    cases = {
        "all_equal": [0x1234567 + (1 << 200)] * n,
        "bool": [int(x) for x in rng.integers(0, 2, size=n)],
        "pm_u8": [int(x) if i % 2 else (r - int(x)) % r for i, x in enumerate(rng.integers(0, 256, size=n))],
        "half_equal_half_random": [7 if i % 2 else int.from_bytes(rng.bytes(40), "little") % r for i in range(n)],
    }
    for name, vals in cases.items():
        scalars = np.array([lim(v) for v in vals], dtype=np.uint64)
        got = A.msm_bigint(cid, d, torch.from_numpy(scalars.view(np.int64)).cuda())
        exp = O.msm(cid, bases, scalars, O.SIGNED, 8)
        assert np.array_equal(A.into_affine(cid, got), O.to_affine(cid, exp)), (cname, name)


@pytest.mark.parametrize("cname", O.CURVES)
def test_msm_every_base_twice_doubles_in_most_buckets(cname):
    """Round 6 expanded the equal-points branch of every bucket kernel in place, on the kernels' own carry-free limbs (G1:
    ec28.cuh xyzz_mdbl_lazy / xyzz_dbl_lazy; G2: ec28x2.cuh lazy2_mdbl / lazy2_dbl -- new formulas for the lane-pair form).
    Every base appears TWICE with the same scalar: with few points per bucket most non-empty buckets hold exactly {P, P}, so
    the lane-per-bucket kernel doubles in most of them; a third copy with the negated scalar makes others pass through infinity
    and come back.  Plain entry, the narrow entry and (G1 and G2) a prepared set, against the oracle."""
    cid = O.CID[cname]
    n = 1 << (10 if cname.endswith("G2") else 13)
    b = O.gen_bases(cid, A4, B4, n)
    s = O.gen_scalars(sf(cid), 0xD0B1E, n)
    r = P.Curve(cname).r
    neg = np.array([P.to_limbs((r - P.from_limbs(row)) % r, 4) for row in s[: n // 4]], dtype=np.uint64)
    for bases, scalars in ((np.concatenate([b, b]), np.concatenate([s, s])),
                           (np.concatenate([b, b, b[: n // 4]]), np.concatenate([s, s, neg]))):
        exp = O.to_affine(cid, O.msm(cid, bases, scalars, O.SIGNED, 8))
        assert np.array_equal(A.into_affine(cid, A.msm_bigint(cid, bases, scalars)), exp), cname
        pb = A.PreparedBases(cid, bases)
        got = pb.msm_bigint(scalars)
        pb.free()
        assert np.array_equal(A.into_affine(cid, got), exp), (cname, "prepared")
    # narrow scalars (msm_u16's body): one window, every bucket's run holds its points twice
    small = (np.arange(2 * n, dtype=np.uint64) % np.uint64(n // 2)).astype(np.uint16)
    bb = np.concatenate([b, b])
    sc4 = np.zeros((2 * n, 4), dtype=np.uint64)
    sc4[:, 0] = small
    exp = O.to_affine(cid, O.msm(cid, bb, sc4, O.SIGNED, 8))
    assert np.array_equal(A.into_affine(cid, A.msm_u16(cid, bb, small)), exp), (cname, "u16")


@pytest.mark.parametrize("cname", O.CURVES)
def test_msm_identical_bases_in_heavy_buckets(cname):
    """Thousands of copies of ONE base with ONE scalar (and a block of its inverse): every lane of the heavy-run kernels
    sums the same multiple, so the LDS trees and the chunk combine add EQUAL points (the full addition's doubling branch,
    in the 28-bit form for the Fp384 G1 curves) and P + (-P) (its infinity branch); the lane-per-bucket kernel sees
    the same point twice in a row (the mixed addition's doubling branch).  Checked against the oracle."""
    import torch
    cid = O.CID[cname]
    r = P.Curve(cname).r
    seed = O.gen_bases(cid, A4, B4, 4)
    fw = seed.shape[1] // 2
    for n, nneg in ((6000, 0), (9000, 3000), (70, 35), (5000, 2500)):
        bases = np.tile(seed[1], (n, 1))
        if nneg:
            bases[:nneg, fw:] = O.basefield_op(cid, "neg", seed[1, fw:].reshape(1, -1)).reshape(-1)
        for sval in (1, 0xFFFF, (r - 1) // 3, 0x1234567 + (1 << 200)):
            scalars = np.tile(np.array(P.to_limbs(sval % r, 4), dtype=np.uint64), (n, 1))
            if n == 5000:   # a second base in between: two long runs per bucket
                bases[1::2] = seed[2]
            got = A.msm_bigint(cid, torch.from_numpy(bases.view(np.int64)).cuda(), torch.from_numpy(scalars.view(np.int64)).cuda())
            exp = O.msm(cid, bases, scalars, O.SIGNED, 4)
            assert np.array_equal(A.into_affine(cid, got), O.to_affine(cid, exp)), (cname, n, nneg, hex(sval))
            if not cname.endswith("G2") or n <= 6000:   # (G2 prepared tables: the two smaller cases)
                pb = A.PreparedBases(cid, bases)
                gp = pb.msm_bigint(scalars)
                pb.free()
                assert np.array_equal(A.into_affine(cid, gp), O.to_affine(cid, exp)), (cname, n, nneg, hex(sval), "prepared")


def test_msm_known_answer_table_bls12_381_g2():
    # reference KAT for the Fp2 path: k*G2 for k = 0..999 (g2_uncompressed_valid_test_vectors.dat, tests/mod.rs:113-123)
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "bls12_381_g2_multiples.npz"))
    xy = g["xy"]  # canonical limbs [1000, 2 (x,y), 2 (c0,c1), 6]
    fq = O.FID["BLS12_381_FQ"]
    mont = O.field_op(fq, "from_bigint", xy.reshape(-1, 6)).reshape(1000, 24)
    cid = O.CID["BLS12_381_G2"]
    sc10 = np.zeros((10, 4), dtype=np.uint64)
    sc10[:, 0] = np.arange(10, 0, -1)          # sum_{k=1..10} k (11-k) = 220
    got = A.into_affine(cid, A.msm_bigint(cid, mont[1:11], sc10))
    assert np.array_equal(got, mont[220])
    ones = np.zeros((44, 4), dtype=np.uint64)
    ones[:, 0] = 1
    assert np.array_equal(A.into_affine(cid, A.msm_bigint(cid, mont[1:45], ones)), mont[990])


def test_chunked_pippenger_matches_msm_bigint():
    # test_chunked_pippenger (test-templates/src/msm.rs:112-133): streaming wrapper == msm_bigint
    cid = O.CID["BLS12_381_G1"]
    n = 1 << 10
    bases = O.gen_bases(cid, A4, B4, n)
    scalars = O.gen_scalars(sf(cid), 31337, n)
    whole = A.msm_bigint(cid, bases, scalars)
    p = A.ChunkedPippenger.with_size(cid, 1 << 8)
    for b, s in zip(bases, scalars):
        p.add(b, s)
    assert np.array_equal(A.into_affine(cid, p.finalize()), A.into_affine(cid, whole))
    assert np.array_equal(A.into_affine(cid, A.ChunkedPippenger(cid, 4).finalize()),
                          np.zeros(2 * O.fe_words(cid), dtype=np.uint64))


@pytest.mark.parametrize("cname", ["BLS12_381_G1", "BN254_G1"])
def test_msm_random_sizes_fuzz(cname):
    # sizes that straddle the internal tiles (8192-key sort tiles, 2048-bucket order tiles, window-size changes)
    cid = O.CID[cname]
    rng = np.random.default_rng(2024)
    sizes = [8191, 8192, 8193, 16385, 3, 5, 100, 257, 4097, 12345] + [int(x) for x in rng.integers(1, 20000, size=8)]
    bases_all = O.gen_bases(cid, A4, B4, max(sizes))
    for k, n in enumerate(sizes):
        scalars = O.gen_scalars(sf(cid), 1000 + k, n)
        off = int(rng.integers(0, max(sizes) - n + 1))
        got = A.msm_bigint(cid, bases_all[off:off + n], scalars)
        exp = O.msm(cid, bases_all[off:off + n], scalars, O.WNAF, 8)
        assert np.array_equal(A.into_affine(cid, got), O.to_affine(cid, exp)), (cname, n)


def test_resident_bases_through_the_c_abi_only():
    # what the Rust shim does for a fixed SRS: ark_hip_malloc + ark_hip_memcpy_h2d once, then
    # ark_hip_msm_sw_device with raw device pointers for every scalar vector (no torch involved)
    import ctypes as C
    from algebra_amd._lib import check, lib
    L = lib()
    cid = O.CID["BLS12_381_G1"]
    n = 3000
    bases = O.gen_bases(cid, A4, B4, n)
    d_bases, d_scalars = C.c_void_p(), C.c_void_p()
    check(L.ark_hip_malloc(bases.nbytes, C.byref(d_bases)), "malloc")
    check(L.ark_hip_malloc(n * 32, C.byref(d_scalars)), "malloc")
    check(L.ark_hip_memcpy_h2d(d_bases, bases.ctypes.data_as(C.c_void_p), bases.nbytes), "h2d")
    back = np.zeros_like(bases)
    check(L.ark_hip_memcpy_d2h(back.ctypes.data_as(C.c_void_p), d_bases, bases.nbytes), "d2h")
    assert np.array_equal(back, bases)
    for seed in (1, 2):
        scalars = O.gen_scalars(sf(cid), seed, n)
        check(L.ark_hip_memcpy_h2d(d_scalars, scalars.ctypes.data_as(C.c_void_p), scalars.nbytes), "h2d")
        out = np.zeros(18, dtype=np.uint64)
        check(L.ark_hip_msm_sw_device(cid, d_bases, d_scalars, n, 0, out.ctypes.data_as(C.c_void_p)), "msm")
        assert np.array_equal(A.into_affine(cid, out), O.to_affine(cid, O.msm(cid, bases, scalars, O.SIGNED, 4)))
    check(L.ark_hip_free(d_bases), "free")
    check(L.ark_hip_free(d_scalars), "free")


@pytest.mark.parametrize("cname", O.CURVES)
def test_msm_specialized_small_scalar_entry_points(cname):
    # test_var_base_msm_specialized (test-templates/src/msm.rs:74-110): msm_u1/u8/u16/u32/u64 vs the reference algorithm
    # on the widened scalars.  The scalars cross the C ABI as the reference's own narrow integers (ark_hip_msm_sw_small).
    import torch
    cid = O.CID[cname]
    n = 1 << 9 if cname.endswith("G2") else 5 << 10
    bases = O.gen_bases(cid, A4, B4, n)
    rng = np.random.default_rng(77)
    u64 = rng.integers(0, 1 << 63, size=n, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=n, dtype=np.uint64)
    cases = [(A.msm_u1, rng.integers(0, 2, size=n).astype(bool)),
             (A.msm_u8, rng.integers(0, 1 << 8, size=n, dtype=np.uint64).astype(np.uint8)),
             (A.msm_u16, rng.integers(0, 1 << 16, size=n, dtype=np.uint64).astype(np.uint16)),
             (A.msm_u32, rng.integers(0, 1 << 32, size=n, dtype=np.uint64).astype(np.uint32)),
             (A.msm_u64, u64)]                                   # all 64 bits in play
    for fn, sc in cases:
        if sc.dtype != np.bool_:                                 # extremes: 0, 1, the type's maximum
            sc[0], sc[1], sc[2] = 0, 1, np.iinfo(sc.dtype).max
        big = np.zeros((n, 4), dtype=np.uint64)
        big[:, 0] = sc.astype(np.uint64)
        exp = O.to_affine(cid, O.msm(cid, bases, big, O.SIGNED, 8))
        assert np.array_equal(A.into_affine(cid, fn(cid, bases, sc)), exp), fn.__name__
        # device-resident inputs (ark_hip_msm_sw_small_device)
        d_b = torch.from_numpy(bases.view(np.int64)).cuda()
        host = sc.astype(np.uint8) if sc.dtype == np.bool_ else sc
        signed = {1: np.int8, 2: np.int16, 4: np.int32, 8: np.int64}[host.dtype.itemsize]
        d_s = torch.from_numpy(host.view(signed)).cuda()
        assert np.array_equal(A.into_affine(cid, fn(cid, d_b, d_s)), exp), fn.__name__ + " (device)"
        # ragged / tiny sizes, msm_unchecked-style truncation to the shorter input
        for m in (0, 1, 33):
            e2 = O.to_affine(cid, O.msm(cid, bases[:m], big[:m], O.SIGNED, 2))
            assert np.array_equal(A.into_affine(cid, fn(cid, bases, sc[:m])), e2), (fn.__name__, m)


def test_msm_small_scalars_skewed_and_all_equal():
    # the reference's bench shapes for the narrow entries (bench-templates/src/macros/ec.rs:244-372): every scalar the
    # same value, and one bucket taking almost everything (heavy-run kernels)
    cid = O.CID["BLS12_381_G1"]
    n = 1 << 14
    bases = O.gen_bases(cid, A4, B4, n)
    for fn, dt, val in ((A.msm_u8, np.uint8, 255), (A.msm_u32, np.uint32, 0x80000001), (A.msm_u64, np.uint64, (1 << 64) - 1),
                        (A.msm_u1, np.bool_, True)):
        sc = np.full(n, val, dtype=dt)
        sc[::97] = 0 if dt != np.bool_ else False
        big = np.zeros((n, 4), dtype=np.uint64)
        big[:, 0] = sc.astype(np.uint64)
        exp = O.to_affine(cid, O.msm(cid, bases, big, O.SIGNED, 8))
        assert np.array_equal(A.into_affine(cid, fn(cid, bases, sc)), exp), fn.__name__
