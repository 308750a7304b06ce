"""Device-resident Fr vectors: the pointwise algebra the C ABI gained in round 5
(ark_hip_fr_add / sub / neg / scale_device, ark_hip_memcpy_d2d, ark_hip_memset_device) and the chain
evaluate_over_domain -> pointwise -> interpolate (poly/src/polynomial/univariate/mod.rs:305-360,
poly/src/evaluations/univariate/mod.rs:40-50, :104-180) through the Python mirror of `ark_hip::DeviceVec`, against the
oracle's transforms and host field arithmetic, limb for limb.  The same chain through the compiled C++ mirror at 2^20 is
tests/test_gpu_cpp_mirror.py."""
import numpy as np
import pytest

import algebra_amd as A
import oracle_lib as O

pytestmark = pytest.mark.gpu

FR = ["BN254_FR", "BLS12_381_FR", "BLS12_377_FR"]


def rand_fr(fid, n, seed):
    return O.gen_scalars(fid, seed, max(n, 1), montgomery=True)[:n]


@pytest.mark.parametrize("fname", FR)
@pytest.mark.parametrize("n", [0, 1, 63, 256, 4097])
def test_pointwise_ops_match_the_oracle(fname, n):
    fid = O.FID[fname]
    a, b = rand_fr(fid, n, 1), rand_fr(fid, n, 2)
    if n > 3:
        p = O.field_const(fid, 0)                      # the modulus: plant 0, p - 1 and equal operands
        a[0] = 0
        b[0] = 0
        pm1 = np.array(p, dtype=np.uint64)
        pm1[0] -= np.uint64(1)
        a[1] = O.field_op(fid, "from_bigint", pm1.reshape(1, 4)).reshape(4)     # p - 1 as a Montgomery residue
        b[2] = a[2]
    k = rand_fr(fid, 1, 3)[0]
    da, db = A.DeviceVec.from_host(fname, a), A.DeviceVec.from_host(fname, b)
    assert len(da) == n and np.array_equal(da.to_host(), a)
    s = da.clone()
    s += db
    assert np.array_equal(s.to_host().reshape(-1), O.field_op(fid, "add", a, b)) if n else len(s) == 0
    d = da.clone()
    d -= db
    m = da.clone()
    m *= db
    ng = da.clone().negate()
    sc = da.clone().scale(k)
    if n:
        assert np.array_equal(d.to_host().reshape(-1), O.field_op(fid, "sub", a, b))
        assert np.array_equal(m.to_host().reshape(-1), O.field_op(fid, "mul", a, b))
        assert np.array_equal(ng.to_host().reshape(-1), O.field_op(fid, "neg", a))
        assert np.array_equal(sc.to_host().reshape(-1), O.field_op(fid, "mul", a, np.tile(k, (n, 1))))
    if n:
        q = da.clone()
        q /= db                                                                     # b[0] = 0 (n > 3): the quotient there is zero
        binv = O.field_op(fid, "inv", b).reshape(-1, 4)
        zero_rows = ~b.any(axis=1)
        binv[zero_rows] = 0
        assert np.array_equal(q.to_host().reshape(-1), O.field_op(fid, "mul", a, binv))
        assert np.array_equal(db.clone().batch_inverse().to_host(), binv)
    assert np.array_equal(da.to_host(), a) and np.array_equal(db.to_host(), b)     # operands untouched
    with pytest.raises(ValueError):
        da += A.DeviceVec(fname, n + 1)


def test_vec_semantics():
    fname = "BLS12_381_FR"
    fid = O.FID[fname]
    z = A.DeviceVec(fname, 37)
    assert not z.to_host().any()
    a = rand_fr(fid, 100, 9)
    v = A.DeviceVec.from_host(fname, a)
    v.resize_zeroed(300)                                # grows beyond capacity: prefix kept, tail zero
    h = v.to_host()
    assert h.shape == (300, 4) and np.array_equal(h[:100], a) and not h[100:].any()
    v.resize_zeroed(40)
    assert np.array_equal(v.to_host(), a[:40])
    v.resize_zeroed(64)                                 # regrow within capacity: the stale tail must be zero again
    h = v.to_host()
    assert np.array_equal(h[:40], a[:40]) and not h[40:].any()
    v.free()
    assert len(v) == 0


@pytest.mark.parametrize("fname,log_n", [("BLS12_381_FR", 10), ("BLS12_381_FR", 16), ("BN254_FR", 12), ("BLS12_377_FR", 13),
                                         ("BLS12_381_FR", 20)])
def test_chain_evaluate_pointwise_interpolate(fname, log_n):
    """c = IFFT((FFT(a) * FFT(b) + FFT(b)) * k - FFT(a)) with two uploads and one download; b is short enough for the
    degree-aware path, a is not; also over a coset domain."""
    fid = O.FID[fname]
    n = 1 << log_n
    a, b = rand_fr(fid, n // 2 - 3, 11), rand_fr(fid, n // 8 + 1, 12)
    k = rand_fr(fid, 1, 13)[0]
    gen = O.field_const(fid, 3)
    base = A.Radix2EvaluationDomain.new(fname, n)
    for dom, off in ((base, None), (base.get_coset(gen), gen)):
        pad = lambda x: np.concatenate([x, np.zeros((n - x.shape[0], 4), dtype=np.uint64)])  # noqa: E731
        ea, eb = O.fft(fid, pad(a), log_n, off, False, 4), O.fft(fid, pad(b), log_n, off, False, 4)
        e = O.field_op(fid, "mul", ea, eb)
        e = O.field_op(fid, "add", e, eb)
        e = O.field_op(fid, "mul", e, np.tile(k, (n, 1)))
        e = O.field_op(fid, "sub", e, ea)
        want = O.fft(fid, e.reshape(n, 4), log_n, off, True, 4)
        da = A.DeviceVec.from_host(fname, a).evaluate_over_domain(dom)
        db = A.DeviceVec.from_host(fname, b).evaluate_over_domain(dom)
        assert len(da) == n and len(db) == n
        keep = da.clone()
        da *= db
        da += db
        da.scale(k)
        da -= keep
        assert np.array_equal(da.clone().to_host().reshape(-1), e)
        got = da.interpolate(dom).to_host()
        assert np.array_equal(got.reshape(-1), want)
    with pytest.raises(ValueError):
        A.DeviceVec.from_host(fname, rand_fr(fid, n + 1, 5)).evaluate_over_domain(base)
