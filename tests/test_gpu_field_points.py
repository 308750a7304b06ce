"""Device field and point arithmetic vs the oracle (GPU).  Every formula the MSM / FFT kernels use is
exercised on its own through the C-ABI test hooks, on random inputs plus the edge values of
SURVEY.md section 7 step 3 (0, 1, p-1, R, values that need the final conditional subtraction)."""
import numpy as np
import pytest

import oracle_lib as O
import hip_lib as H
import pyref as P

pytestmark = pytest.mark.gpu


def _field_inputs(fid, n, seed):
    name = O.FIELDS[fid]
    p = P.MODULI[name][0]
    words = O.field_limbs(fid)
    rng = np.random.default_rng(seed)
    vals = [0, 1, 2, p - 1, p - 2, (p - 1) // 2, (p + 1) // 2, P.from_limbs(O.field_const(fid, 1)),
            (1 << (p.bit_length() - 1)), (1 << (p.bit_length() - 1)) - 1]
    vals = [v % p for v in vals]
    while len(vals) < n:
        vals.append(int.from_bytes(rng.bytes(8 * words + 8), "little") % p)
    a = np.array([P.to_limbs(v, words) for v in vals], dtype=np.uint64)
    b = a[rng.permutation(len(vals))].copy()
    return a, b


@pytest.mark.parametrize("fname", O.FIELDS)
def test_field_ops_match_oracle(fname):
    fid = O.FID[fname]
    a, b = _field_inputs(fid, 4096, 11 + fid)
    ops = ["add", "sub", "mul", "sqr", "neg", "dbl"]
    if fname.endswith("FR"):
        ops += ["into_bigint", "from_bigint"]
    for op in ops:
        two = op in ("add", "sub", "mul")
        got = H.field_op(fid, op, a, b if two else None)
        exp = O.field_op(fid, op, a, b if two else None)
        assert np.array_equal(got.reshape(-1), exp.reshape(-1)), (fname, op)


@pytest.mark.parametrize("fname", O.FIELDS)
def test_lazy_28bit_device_forms_match_oracle(fname):
    """fp28.cuh's product / square / sum of two products as the DEVICE runs them (one asm multiply-add chain per column),
    on every field: operands enter by the shifted repack, results leave through shr_mod + one conditional subtraction --
    the path of a bucket through the accumulate kernels -- and must equal the reference's canonical product."""
    fid = O.FID[fname]
    a, b = _field_inputs(fid, 4096, 77 + fid)
    mul = O.field_op(fid, "mul", a, b)
    assert np.array_equal(H.field_op(fid, "lazy_mul", a, b).reshape(-1), mul.reshape(-1)), fname
    assert np.array_equal(H.field_op(fid, "lazy_sqr", a, None).reshape(-1), O.field_op(fid, "sqr", a, None).reshape(-1)), fname
    assert np.array_equal(H.field_op(fid, "lazy_sop2", a, b).reshape(-1), O.field_op(fid, "dbl", mul, None).reshape(-1)), fname


@pytest.mark.parametrize("cname", ["BLS12_377_G2", "BLS12_381_G2"])
def test_fp2_ops_match_oracle(cname):
    cid = O.CID[cname]
    bf = O.curve_info(cid)[0]
    a0, b0 = _field_inputs(bf, 2048, 5)
    a1, b1 = _field_inputs(bf, 2048, 6)
    a = np.concatenate([a0, a1[::-1]], axis=1)
    b = np.concatenate([b0[::-1], b1], axis=1)
    for op in ["add", "sub", "mul", "sqr", "neg", "dbl"]:
        two = op in ("add", "sub", "mul")
        got = H.basefield_op(cid, op, a, b if two else None)
        exp = O.basefield_op(cid, op, a, b if two else None)
        assert np.array_equal(got.reshape(-1), exp.reshape(-1)), (cname, op)


A4 = np.array([0xA11CE, 1, 2, 0], dtype=np.uint64)
B4 = np.array([0xB0B, 3, 0, 0], dtype=np.uint64)


def _norm(cid, xyzz):
    """XYZZ rows -> affine via the oracle (bucket -> jacobian -> affine)."""
    out = []
    for row in xyzz:
        jac = O.point_op(cid, "bkt_to_jac", np.zeros(3 * O.fe_words(cid), dtype=np.uint64), row)
        out.append(O.to_affine(cid, jac))
    return np.stack(out)


def _canonical(cid, xyzz):
    """every coordinate reduced once more by the oracle (x + 0): unchanged exactly when it already was canonical."""
    flat = np.ascontiguousarray(xyzz, dtype=np.uint64).reshape(-1)
    return O.basefield_op(cid, "add", flat, np.zeros_like(flat)).reshape(xyzz.shape)


@pytest.mark.parametrize("cname", O.CURVES)
def test_point_ops_match_oracle(cname):
    cid = O.CID[cname]
    fw = O.fe_words(cid)
    n = 24
    bases = O.gen_bases(cid, A4, B4, n)
    zero_aff = np.zeros(2 * fw, dtype=np.uint64)
    # buckets: start from ZERO (1,1,0,0) and accumulate a few points with the oracle
    fone = np.zeros(fw, dtype=np.uint64)
    r = O.field_const(O.curve_info(cid)[0], 1)
    fone[: r.size] = r
    bzero = np.concatenate([fone, fone, np.zeros(2 * fw, dtype=np.uint64)])
    bk = []
    for i in range(n):
        acc = bzero.copy()
        for j in range(i % 4):
            acc = O.point_op(cid, "bkt_add_aff", acc, bases[(i + j) % n])
        bk.append(acc)
    bk = np.stack(bk)
    # operands: generic, identity, the same point (doubling branch), the inverse (-> infinity)
    other = bases[::-1].copy()
    other[0] = zero_aff
    for i in (1, 5, 9):           # bucket holding exactly P, adding P  -> doubling
        bk[i] = O.point_op(cid, "bkt_add_aff", bzero, bases[i])
        other[i] = bases[i]
    for i in (2, 6):              # bucket holding P, adding -P -> infinity
        bk[i] = O.point_op(cid, "bkt_add_aff", bzero, bases[i])
        o = bases[i].copy()
        o[fw:] = O.basefield_op(cid, "neg", bases[i][fw:])
        other[i] = o
    for kind in ("bkt_add_aff", "bkt_sub_aff"):
        got = H.point_op(cid, kind, bk, other)
        exp = np.stack([O.point_op(cid, kind, bk[i], other[i]) for i in range(n)])
        assert np.array_equal(_norm(cid, got), _norm(cid, exp)), (cname, kind)
        # the same addition as the accumulate kernels run it: carry-free limbs (14 x 28 / 9 x 29 bits; G2: one lane pair
        # per point), stored bucket in -> stored (canonical) bucket out, doubling and infinity branches included
        lz = H.point_op(cid, kind.replace("bkt_", "lazy_"), bk, other)
        assert np.array_equal(_norm(cid, lz), _norm(cid, exp)), (cname, kind, "lazy")
        assert np.array_equal(lz.reshape(n, 4, fw)[:, :, :], _canonical(cid, lz).reshape(n, 4, fw)), (cname, kind, "canonical limbs")
    # bucket + bucket, including equal operands and the identity
    ob = bk[::-1].copy()
    ob[3] = bk[3]
    ob[4] = bzero
    got = H.point_op(cid, "bkt_add_bkt", bk, ob)
    exp = np.stack([O.point_op(cid, "bkt_add_bkt", bk[i], ob[i]) for i in range(n)])
    assert np.array_equal(_norm(cid, got), _norm(cid, exp)), (cname, "bkt_add_bkt")
    ob[7] = bk[7].copy()                                        # inverse operands: infinity
    ob[7][fw:2 * fw] = O.basefield_op(cid, "neg", bk[7][fw:2 * fw])
    exp = np.stack([O.point_op(cid, "bkt_add_bkt", bk[i], ob[i]) for i in range(n)])
    for kind in ("lazy_add_bkt", "lazy_add_acc"):               # the reduction / heavy-run kernels' full addition
        lz = H.point_op(cid, kind, bk, ob)
        assert np.array_equal(_norm(cid, lz), _norm(cid, exp)), (cname, kind)
    got = H.point_op(cid, "bkt_double", bk)
    exp = np.stack([O.point_op(cid, "bkt_double", bk[i]) for i in range(n)])
    assert np.array_equal(_norm(cid, got), _norm(cid, exp)), (cname, "bkt_double")
    got = H.point_op(cid, "aff_double_to_bkt", other)
    exp = np.stack([O.point_op(cid, "aff_double_to_bkt", bzero, other[i]) for i in range(n)])
    assert np.array_equal(_norm(cid, got), _norm(cid, exp)), (cname, "aff_double_to_bkt")
    got = H.point_op(cid, "bkt_to_jac", bk)
    for i in range(n):
        assert np.array_equal(O.to_affine(cid, got[i]), _norm(cid, bk[i:i + 1])[0])


@pytest.mark.parametrize("cname", O.CURVES)
def test_normalize_batch_on_device(cname):
    # CurveGroup::normalize_batch == per-point into_affine (test-templates/src/groups.rs batch normalisation check)
    import torch
    import algebra_amd as A
    cid = O.CID[cname]
    n = 40
    bases = O.gen_bases(cid, A4, B4, n)
    sc = O.gen_scalars(O.curve_info(cid)[1], 3, n)
    pts = np.stack([O.scalar_mul(cid, bases[i], sc[i]) for i in range(n)])
    pts[7] = O.msm(cid, bases[:0], sc[:0], O.NAIVE)            # an identity in the batch
    d = torch.from_numpy(pts.view(np.int64)).cuda()
    got = A.normalize_batch(cid, d).cpu().numpy().view(np.uint64).reshape(n, -1)
    assert np.array_equal(got, O.to_affine(cid, pts))
    # the host-pointer entry the Rust hook behind CurveGroup::normalize_batch binds
    assert np.array_equal(A.normalize_batch(cid, pts), O.to_affine(cid, pts))
