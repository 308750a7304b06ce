"""The transform over GROUP elements (round 5; VERDICT r4 missing #3): EvaluationDomain::fft_in_place / ifft_in_place for
T = Projective<P> (poly/src/domain/mod.rs:332-362 with radix2/fft.rs:74-119; the reference's own test: poly/src/test.rs:57
transforms G1Projective coefficients and compares with the transform of their discrete logs).  Same idea here, with the
ORACLE on both sides of the comparison: coefficients P_i = [a_i] G with known a_i (an arithmetic progression, some of them
zero = the identity), so FFT(P)_j = [FFT_F(a)_j] G -- the field transform from the oracle's FFT, the scalar multiplications
from the oracle's scalar_mul, compared after into_affine (Projective representatives differ, as for the MSM)."""
import numpy as np
import pytest

import algebra_amd as A
from algebra_amd import curves as cv
import oracle_lib as O

pytestmark = pytest.mark.gpu

A4 = np.array([0xA11CE, 1, 2, 0], dtype=np.uint64)
B4 = np.array([0xB0B, 3, 0, 0], dtype=np.uint64)


def _int(limbs):
    return sum(int(v) << (64 * k) for k, v in enumerate(limbs))


def _limbs(v, n=4):
    return np.array([(v >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for k in range(n)], dtype=np.uint64)


def _setup(cname, n, zero_at=()):
    """n Jacobian points P_i = [a + i b] G (z = 1 in Montgomery form), identities where asked, and their scalars"""
    cid = O.CID[cname]
    bf, sf, ext = O.curve_info(cid)
    fw = O.fe_words(cid)
    r = _int(O.field_const(sf, 0))
    aff = O.gen_bases(cid, A4, B4, n)
    one = O.field_const(bf, 1)                                     # R mod p: the Montgomery form of 1
    pts = np.zeros((n, 3 * fw), dtype=np.uint64)
    pts[:, :2 * fw] = aff
    pts[:, 2 * fw:2 * fw + one.size] = one                         # (Fp2: c0 = 1, c1 = 0)
    scal = [(_int(A4) + i * _int(B4)) % r for i in range(n)]
    for i in zero_at:
        pts[i] = 0                                                 # z = 0: the identity
        scal[i] = 0
    return cid, sf, r, pts, scal


def _expected(cid, sf, scal, log_n, offset, inverse):
    """affine [FFT_F(scal)_j] G through the oracle"""
    n = 1 << log_n
    sc = np.array([_limbs(v) for v in scal], dtype=np.uint64)
    mont = O.field_op(sf, "from_bigint", sc).reshape(n, 4)
    out = O.fft(sf, mont, log_n, offset, inverse, 2).reshape(n, 4)
    canon = O.field_op(sf, "into_bigint", out).reshape(n, 4)
    g = O.generator(cid)
    return np.stack([O.to_affine(cid, O.scalar_mul(cid, g, canon[j])) for j in range(n)])


@pytest.mark.parametrize("cname", ["BLS12_381_G1", "BN254_G1", "BLS12_377_G2"])
@pytest.mark.parametrize("log_n", [0, 1, 2, 3, 5])
def test_group_fft_matches_the_transform_of_the_discrete_logs(cname, log_n):
    n = 1 << log_n
    cid, sf, r, pts, scal = _setup(cname, n, zero_at=(n - 1,) if n >= 4 else ())
    fname = [k for k, v in O.FID.items() if v == sf][0]
    dom = A.Radix2EvaluationDomain.new(fname, n)
    gen = O.field_const(sf, 3)
    for d, off in ((dom, None), (dom.get_coset(gen), gen)):
        got = d.fft_group_in_place(cname, pts.copy())
        assert np.array_equal(A.into_affine(cid, got), _expected(cid, sf, scal, log_n, off, False))
        inv = d.fft_group_in_place(cname, pts.copy(), inverse=True)
        assert np.array_equal(A.into_affine(cid, inv), _expected(cid, sf, scal, log_n, off, True))
        back = d.fft_group_in_place(cname, got, inverse=True)          # ifft(fft(P)) = P
        assert np.array_equal(A.into_affine(cid, back), A.into_affine(cid, pts))


def test_group_fft_2_10_device_resident_and_linear():
    import torch
    cname, log_n = "BLS12_381_G1", 10
    n = 1 << log_n
    cid, sf, r, pts, scal = _setup(cname, n, zero_at=(0, 17, n - 2))
    dom = A.Radix2EvaluationDomain.new("BLS12_381_FR", n)
    d = torch.from_numpy(pts.view(np.int64)).cuda()
    got = dom.fft_group_in_place(cname, d).cpu().numpy().view(np.uint64).reshape(n, -1)
    assert np.array_equal(A.into_affine(cid, got), _expected(cid, sf, scal, log_n, None, False))
    # all identities in, all identities out; a single point at index 0 spreads to every output unchanged
    z = np.zeros_like(pts)
    assert not A.into_affine(cid, dom.fft_group_in_place(cname, z.copy())).any()
    e0 = np.zeros_like(pts)
    e0[0] = pts[1]
    spread = A.into_affine(cid, dom.fft_group_in_place(cname, e0))
    assert np.array_equal(spread, np.tile(A.into_affine(cid, pts[1:2]).reshape(1, -1), (n, 1)))


@pytest.mark.parametrize("cname", ["BLS12_381_G1", "BLS12_377_G2"])
def test_group_fft_in_several_slabs(cname, monkeypatch):
    # a transform above 2^18 lanes runs its launches in slabs that share one set of window tables: the same code at 2^8 with
    # slabs of 64 lanes (ARK_HIP_GFFT_SLAB_LOG, read per call) -- plain, coset and inverse against the single-slab results,
    # which the first test pins to the oracle
    import torch
    log_n = 8
    n = 1 << log_n
    cid, sf, r, pts, scal = _setup(cname, n, zero_at=(3, n - 1))
    fname = [k for k, v in O.FID.items() if v == sf][0]
    dom = A.Radix2EvaluationDomain.new(fname, n)
    for d in (dom, dom.get_coset(O.field_const(sf, 3))):
        whole_f = d.fft_group_in_place(cname, pts.copy())
        whole_i = d.fft_group_in_place(cname, pts.copy(), inverse=True)
        monkeypatch.setenv("ARK_HIP_GFFT_SLAB_LOG", "6")
        slab_f = d.fft_group_in_place(cname, pts.copy())
        slab_i = d.fft_group_in_place(cname, pts.copy(), inverse=True)
        dd = torch.from_numpy(pts.view(np.int64).copy()).cuda()
        slab_dev = d.fft_group_in_place(cname, dd).cpu().numpy().view(np.uint64).reshape(n, -1)
        monkeypatch.delenv("ARK_HIP_GFFT_SLAB_LOG")
        assert np.array_equal(slab_f, whole_f) and np.array_equal(slab_i, whole_i) and np.array_equal(slab_dev, whole_f)
        back = d.fft_group_in_place(cname, whole_f.copy(), inverse=True)
        assert np.array_equal(A.into_affine(cid, back), A.into_affine(cid, pts))


def test_a_domain_of_another_field_is_refused():
    """`ark_hip_radix2_domain` carries no field id -- a BN254 domain handed to a BLS12-381 curve's transform over
    group elements used to be reinterpreted in the wrong field and give garbage points.  The Python mirror compares the field
    ids; the C entry checks that the domain's generator is a size-th root of unity in the CURVE's scalar field."""
    import ctypes as C
    from algebra_amd._lib import lib
    cid = O.CID["BLS12_381_G1"]
    n = 1 << 6
    fw = O.fe_words(cid)
    pts = np.zeros((n, 3 * fw), dtype=np.uint64)          # identities: z = 0
    wrong = A.Radix2EvaluationDomain.new("BN254_FR", n)
    with pytest.raises(ValueError):
        wrong.fft_group_in_place("BLS12_381_G1", pts)
    rc = lib().ark_hip_fft_group_in_place(cid, C.byref(wrong._s), pts.ctypes.data_as(C.c_void_p), 0)
    assert rc == -1                                        # ARK_HIP_ERR_ARG, points untouched
    assert not pts.any()
    right = A.Radix2EvaluationDomain.new("BLS12_381_FR", n)
    assert lib().ark_hip_fft_group_in_place(cid, C.byref(right._s), pts.ctypes.data_as(C.c_void_p), 0) == 0
