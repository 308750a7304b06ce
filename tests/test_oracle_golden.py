"""Pin the oracle against the reference's own known-answer data:
  * k*G tables for BLS12-381 G1 and G2, k = 0..999 (zkcrypto vectors the reference checks at
    curves/bls12_381/src/curves/tests/mod.rs:69-123)
  * RFC 9380 points on BLS12-377 G2 (curves/bls12_377/src/curves/tests/*.json): pins the Fq2
    multiplication / non-residue convention of the curve BASELINE config 5 uses
Fixtures were extracted by tools/make_golden.py (committed; /root/reference is absent on the GPU box).
"""
import os

import numpy as np
import pytest

import oracle_lib as O
import pyref as P

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_multiples(curve_name):
    tag = "g1" if curve_name.endswith("G1") else "g2"
    z = np.load(os.path.join(G, "bls12_381_%s_multiples.npz" % tag))
    cid = O.CID[curve_name]
    fw = O.fe_words(cid)
    xy = z["xy"].reshape(1000, 2 * fw)  # canonical limbs (c0|c1 per coordinate for G2)
    bf = O.curve_info(cid)[0]
    mont = O.field_op(bf, "from_bigint", xy).reshape(1000, 2 * fw)
    return cid, fw, mont, z["infinity"]


@pytest.mark.parametrize("curve", ["BLS12_381_G1", "BLS12_381_G2"])
def test_generator_and_chain(curve):
    cid, fw, mont, inf = load_multiples(curve)
    assert inf[0] == 1 and inf[1:].sum() == 0
    g = O.generator(cid)
    assert np.array_equal(mont[1], g)  # entry 1 == the generator constants (g1.rs:199-205 / g2.rs:229-241)
    # k*G by repeated Projective += generator must reproduce the whole table (group.rs:450-538, affine.rs:374-396)
    gj = np.concatenate([g, O.field_const(O.curve_info(cid)[0], 1), np.zeros(fw - O.field_limbs(O.curve_info(cid)[0]), dtype=np.uint64)])
    acc = gj.copy()
    for k in range(2, 200):
        acc = O.point_op(cid, "jac_add", acc, gj)
        assert np.array_equal(O.to_affine(cid, acc), mont[k]), k
    # the doubling formula: 2^j * G
    acc = gj.copy()
    k = 1
    while 2 * k < 1000:
        acc = O.point_op(cid, "jac_double", acc)
        k *= 2
        assert np.array_equal(O.to_affine(cid, acc), mont[k]), k


@pytest.mark.parametrize("curve", ["BLS12_381_G1", "BLS12_381_G2"])
def test_bucket_formulas_against_table(curve):
    """XYZZ bucket arithmetic (bucket.rs) walks the table: bucket += G repeatedly, bucket += bucket, -= affine."""
    cid, fw, mont, _ = load_multiples(curve)
    one = np.zeros(fw, dtype=np.uint64)
    bf = O.curve_info(cid)[0]
    one[:O.field_limbs(bf)] = O.field_const(bf, 1)
    bzero = np.concatenate([one, one, np.zeros(2 * fw, dtype=np.uint64)])
    b = bzero.copy()
    for k in range(1, 60):  # k=1 copy branch, k=2 doubling branch (same point), then generic madd
        b = O.point_op(cid, "bkt_add_aff", b, mont[1])
        j = O.point_op(cid, "bkt_to_jac", np.zeros(3 * fw, dtype=np.uint64), b)
        assert np.array_equal(O.to_affine(cid, j), mont[k]), k
    # bucket(59G) + bucket(59G) -> doubling branch of add-2008-s ; then + itself again etc.
    b2 = O.point_op(cid, "bkt_add_bkt", b, b)
    j = O.point_op(cid, "bkt_to_jac", np.zeros(3 * fw, dtype=np.uint64), b2)
    assert np.array_equal(O.to_affine(cid, j), mont[118])
    b3 = O.point_op(cid, "bkt_add_bkt", b2, b)
    j = O.point_op(cid, "bkt_to_jac", np.zeros(3 * fw, dtype=np.uint64), b3)
    assert np.array_equal(O.to_affine(cid, j), mont[177])
    # subtraction down to the identity: 177G - 100G - 77G = 0 (inverse-point branch)
    b4 = O.point_op(cid, "bkt_sub_aff", b3, mont[100])
    j = O.point_op(cid, "bkt_to_jac", np.zeros(3 * fw, dtype=np.uint64), b4)
    assert np.array_equal(O.to_affine(cid, j), mont[77])
    b5 = O.point_op(cid, "bkt_sub_aff", b4, mont[77])
    assert not b5[2 * fw:].any()  # Bucket::zero: zz = zzz = 0
    # identity operands are no-ops
    assert np.array_equal(O.point_op(cid, "bkt_add_aff", b3, np.zeros(2 * fw, dtype=np.uint64)), b3)
    assert np.array_equal(O.point_op(cid, "bkt_add_bkt", b3, bzero), b3)


@pytest.mark.parametrize("curve", ["BLS12_381_G1", "BLS12_381_G2"])
@pytest.mark.parametrize("variant", [O.NAIVE, O.WNAF, O.SIGNED])
def test_msm_kat_from_table(curve, variant):
    """bases = k*G for k = 1..m, scalars s_k  =>  MSM = (sum k*s_k)*G, looked up in the table."""
    cid, fw, mont, _ = load_multiples(curve)
    # all-ones over k = 1..44 -> 990*G
    bases = mont[1:45]
    scalars = np.zeros((44, 4), dtype=np.uint64)
    scalars[:, 0] = 1
    out = O.msm(cid, bases, scalars, variant, threads=2)
    assert np.array_equal(O.to_affine(cid, out), mont[990])
    # mixed small scalars incl. zeros; 40 bases so the windowed path (c = ln(n)+2) is taken
    rng = np.random.default_rng(7)
    ks = rng.integers(1, 25, size=40)
    ss = rng.integers(0, 2, size=40)
    ss[:3] = [0, 1, 2]
    total = int((ks * ss).sum())
    assert 0 < total < 1000
    scalars = np.zeros((40, 4), dtype=np.uint64)
    scalars[:, 0] = ss.astype(np.uint64)
    out = O.msm(cid, mont[ks], scalars, variant, threads=2)
    assert np.array_equal(O.to_affine(cid, out), mont[total])
    # negative scalars: s = r - t  (exercises the NegU* groups of msm_signed and high windows elsewhere)
    r = P.MODULI["BLS12_381_FR"][0]
    pos = [(500, 1), (300, 1), (7, 3)]  # 500 + 300 + 21 = 821
    neg = [(20, 2), (1, 1)]             # -41 -> 780
    bs = np.stack([mont[k] for k, _ in pos + neg])
    sc = np.stack([P.to_limbs(s, 4) for _, s in pos] + [P.to_limbs(r - s, 4) for _, s in neg])
    out = O.msm(cid, bs, sc, variant, threads=1)
    assert np.array_equal(O.to_affine(cid, out), mont[780])


def test_bls12_377_g2_h2c_points_on_curve():
    """15 RFC 9380 points satisfy y^2 = x^3 + (0, b1) over Fq[u]/(u^2+5) -- with the oracle's Fp2 and with Python ints."""
    z = np.load(os.path.join(G, "bls12_377_g2_h2c_points.npz"))["xy"]
    cid = O.CID["BLS12_377_G2"]
    bf = O.curve_info(cid)[0]
    cv = P.Curve("BLS12_377_G2")
    assert len(z) == 15
    for pt in z:
        mont = O.field_op(bf, "from_bigint", pt.reshape(-1))
        assert O.is_on_curve(cid, mont)
        x = (P.from_limbs(pt[0, 0]), P.from_limbs(pt[0, 1]))
        y = (P.from_limbs(pt[1, 0]), P.from_limbs(pt[1, 1]))
        assert cv.on_curve((x, y))
        # a corrupted point is rejected
        bad = mont.copy()
        bad[0] ^= np.uint64(1)
        assert not O.is_on_curve(cid, bad)
    # Fq2 arithmetic of the oracle agrees with Python on those coordinates
    a = O.field_op(bf, "from_bigint", z[0, 0].reshape(-1))
    b = O.field_op(bf, "from_bigint", z[1, 1].reshape(-1))
    F = cv.F
    ai, bi = F.dec(a), F.dec(b)
    assert np.array_equal(O.basefield_op(cid, "mul", a, b), F.enc(F.mul(ai, bi)))
    assert np.array_equal(O.basefield_op(cid, "sqr", a), F.enc(F.mul(ai, ai)))
    assert np.array_equal(O.basefield_op(cid, "inv", a), F.enc(F.inv(ai)))
    assert np.array_equal(O.basefield_op(cid, "sub", a, b), F.enc(F.sub(ai, bi)))


@pytest.mark.parametrize("cname", ["BN254_G1", "BLS12_377_G1", "BLS12_377_G2"])
def test_generators_on_curve(cname):
    cid = O.CID[cname]
    assert O.is_on_curve(cid, O.generator(cid))
    cv = P.Curve(cname)
    g = cv.dec(O.generator(cid))
    assert cv.on_curve(g)
    assert cv.mul(g, cv.r - 1) == cv.neg(g)  # generator has order r
