"""Oracle MSM restatement vs (a) the naive sum and (b) independent Python affine arithmetic.
Mirrors test-templates/src/msm.rs:17-110 (random, 11-class mixed scalars) at oracle-friendly sizes,
plus the edge cases of SURVEY.md section 8d."""
import numpy as np
import pytest

import oracle_lib as O
import pyref as P

A4 = np.array([0xA11CE, 1, 2, 0], dtype=np.uint64)
B4 = np.array([0xB0B, 3, 0, 0], dtype=np.uint64)
CURVES = ["BN254_G1", "BLS12_381_G1", "BLS12_377_G1", "BLS12_377_G2", "BLS12_381_G2"]


def scalar_field(cid):
    return O.curve_info(cid)[1]


def aff(cid, jac):
    return O.to_affine(cid, jac)


def test_make_digits_recompose():
    rng = np.random.default_rng(1)
    r = P.MODULI["BLS12_381_FR"][0]
    for c in [3, 7, 13, 15, 16, 17, 20, 23]:
        for s in [0, 1, r - 1, (1 << 254), (1 << 255) - 1 if (1 << 255) - 1 < r else r - 2] + [int.from_bytes(rng.bytes(32), "little") % r for _ in range(20)]:
            d = O.make_digits(P.to_limbs(s, 4), c, 255)
            assert len(d) == -(-255 // c)
            assert sum(int(x) << (c * i) for i, x in enumerate(d)) == s
            assert all(-(1 << (c - 1)) <= int(x) < (1 << (c - 1)) for x in d[:-1])
            assert 0 <= int(d[-1]) <= (1 << c)


def test_window_size_rule():
    # c = ceil(log2 n)*69/100 + 2 ; SURVEY.md section 8(a3): 13/15/17/18/19 for 2^16/2^20/2^22/2^24/2^26
    assert [O.window_size(1 << k) for k in (16, 20, 22, 24, 26)] == [13, 15, 17, 18, 19]
    assert O.window_size(31) == 3 and O.window_size(32) == 5


@pytest.mark.parametrize("cname", CURVES)
def test_gen_bases_and_python_crosscheck(cname):
    cid = O.CID[cname]
    cv = P.Curve(cname)
    n = 12
    bases = O.gen_bases(cid, A4, B4, n)
    g = cv.dec(O.generator(cid))
    a, b = P.from_limbs(A4), P.from_limbs(B4)
    for i in range(n):
        assert O.is_on_curve(cid, bases[i])
        assert cv.dec(bases[i]) == cv.mul(g, a + i * b), i
    scalars = O.gen_scalars(scalar_field(cid), 0xA11CE, n)
    ints = [P.from_limbs(s) for s in scalars]
    assert all(0 <= s < cv.r for s in ints)
    expect = cv.enc(cv.msm([cv.dec(x) for x in bases], ints))
    for variant in (O.NAIVE, O.WNAF, O.SIGNED):
        assert np.array_equal(aff(cid, O.msm(cid, bases, scalars, variant)), expect)
    # known discrete log: result == k*G with k = sum s_i (a + i b)
    k = O.msm_dlog(cid, scalars, A4, B4)
    assert P.from_limbs(k) == sum(s * (a + i * b) for i, s in enumerate(ints)) % cv.r
    assert np.array_equal(aff(cid, O.scalar_mul(cid, O.generator(cid), k)), expect)
    # Montgomery-form scalars through the msm() entry (into_bigint first)
    sm = O.gen_scalars(scalar_field(cid), 0xA11CE, n, montgomery=True)
    assert np.array_equal(aff(cid, O.msm(cid, bases, sm, O.SIGNED, montgomery_scalars=True)), expect)


@pytest.mark.parametrize("cname,n", [("BN254_G1", 1 << 10), ("BLS12_381_G1", 1 << 10), ("BLS12_377_G2", 1 << 8)])
def test_random_msm_variants_agree(cname, n):
    """test_var_base_msm (msm.rs:17-32): fast == naive, n = 2^10 (2^8 for the Fp2 curve to stay in seconds)."""
    cid = O.CID[cname]
    bases = O.gen_bases(cid, A4, B4, n)
    scalars = O.gen_scalars(scalar_field(cid), 42, n)
    ref = aff(cid, O.msm(cid, bases, scalars, O.NAIVE))
    assert np.array_equal(aff(cid, O.msm(cid, bases, scalars, O.WNAF, threads=1)), ref)
    assert np.array_equal(aff(cid, O.msm(cid, bases, scalars, O.WNAF, threads=4)), ref)
    assert np.array_equal(aff(cid, O.msm(cid, bases, scalars, O.SIGNED, threads=4)), ref)
    k = O.msm_dlog(cid, scalars, A4, B4)
    assert np.array_equal(aff(cid, O.scalar_mul(cid, O.generator(cid), k)), ref)


def mixed_scalars(r, m, rng):
    """11 x m scalars: +-bool, +-u8, +-u16, +-u32, +-u64, random -- shuffled (msm.rs:36-72)."""
    out = []
    for bits in (1, 8, 16, 32, 64):
        vals = [int(rng.integers(0, 1 << min(bits, 63))) << max(0, bits - 63) | int(rng.integers(0, 2)) if bits == 64
                else int(rng.integers(0, 1 << bits)) for _ in range(m)]
        out += vals
        out += [(-v) % r for v in (int(rng.integers(0, 1 << min(bits, 62))) for _ in range(m))]
    out += [int.from_bytes(rng.bytes(32), "little") % r for _ in range(m)]
    rng.shuffle(out)
    return out


@pytest.mark.parametrize("cname", ["BN254_G1", "BLS12_381_G1"])
def test_mixed_scalars(cname):
    cid = O.CID[cname]
    r = P.MODULI[P.CURVE_PARAMS[cname][1]][0]
    rng = np.random.default_rng(5)
    m = 48
    ints = mixed_scalars(r, m, rng)
    n = len(ints)
    bases = O.gen_bases(cid, A4, B4, n)
    scalars = np.stack([P.to_limbs(s, 4) for s in ints])
    ref = aff(cid, O.msm(cid, bases, scalars, O.NAIVE))
    assert np.array_equal(aff(cid, O.msm(cid, bases, scalars, O.SIGNED, threads=3)), ref)
    assert np.array_equal(aff(cid, O.msm(cid, bases, scalars, O.WNAF, threads=3)), ref)


@pytest.mark.parametrize("cname", ["BN254_G1", "BLS12_381_G1", "BLS12_377_G2"])
def test_edge_cases(cname):
    cid = O.CID[cname]
    cv = P.Curve(cname)
    fw = O.fe_words(cid)
    r = cv.r
    bases = O.gen_bases(cid, A4, B4, 40)
    zero_aff = np.zeros(2 * fw, dtype=np.uint64)
    for n in (0, 1, 31, 32, 33):  # empty, single, both sides of the c = 3 / c = ln+2 switch
        sc = O.gen_scalars(scalar_field(cid), 9, n)
        ref = aff(cid, O.msm(cid, bases[:n], sc, O.NAIVE)) if n else zero_aff
        for v in (O.WNAF, O.SIGNED):
            assert np.array_equal(aff(cid, O.msm(cid, bases[:n], sc, v)), ref), (n, v)
    # zero scalars, scalar 1, scalar r-1, identity bases, duplicates (doubling branch), P and -P (cancellation)
    b = bases[:34].copy()
    b[5] = 0                      # identity base
    b[7] = b[6]                   # duplicate base, same scalar -> bucket doubling branch
    b[9] = b[8]
    b[9, fw:] = O.basefield_op(cid, "neg", b[8, fw:])  # -P with the same scalar -> bucket cancels
    ints = [int(x) for x in np.random.default_rng(11).integers(2, 1 << 62, size=34)]
    ints[0], ints[1], ints[2], ints[3] = 0, 1, r - 1, r - 2
    ints[7] = ints[6]
    ints[9] = ints[8]
    sc = np.stack([P.to_limbs(s, 4) for s in ints])
    ref = aff(cid, O.msm(cid, b, sc, O.NAIVE))
    pts = [cv.dec(x) for x in b]
    assert np.array_equal(ref, cv.enc(cv.msm(pts, ints)))
    for v in (O.WNAF, O.SIGNED):
        assert np.array_equal(aff(cid, O.msm(cid, b, sc, v, threads=2)), ref), v
    # all scalars equal, all bases equal -> every window funnels into one bucket
    bb = np.repeat(bases[3:4], 33, axis=0)
    ss = np.repeat(P.to_limbs(0x123456789ABCDEF0123, 4)[None, :], 33, axis=0)
    ref = aff(cid, O.msm(cid, bb, ss, O.NAIVE))
    assert np.array_equal(ref, cv.enc(cv.mul(cv.dec(bases[3]), 33 * 0x123456789ABCDEF0123)))
    for v in (O.WNAF, O.SIGNED):
        assert np.array_equal(aff(cid, O.msm(cid, bb, ss, v)), ref)
    # unequal lengths: msm_unchecked truncates to the shorter slice (variable_base/mod.rs:243-245)
    sc5 = sc[:5]
    assert np.array_equal(aff(cid, O.msm(cid, b, sc5, O.SIGNED)), aff(cid, O.msm(cid, b[:5], sc5, O.NAIVE)))
