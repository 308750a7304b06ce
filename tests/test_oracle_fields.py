"""Oracle field arithmetic vs Python ints: random + edge values (0, 1, p-1, R, values needing the
final conditional subtraction). Mirrors the property tests of test-templates/src/fields.rs:56-503."""
import numpy as np
import pytest

import oracle_lib as O
import pyref as P


def edge_and_random(p, n, rng, count=64):
    vals = [0, 1, 2, p - 1, p - 2, (1 << (64 * n)) % p, (p + 1) // 2, (1 << (p.bit_length() - 1)), (1 << 64) - 1, 1 << 64]
    vals += [int.from_bytes(rng.bytes(8 * n), "little") % p for _ in range(count)]
    return vals


@pytest.mark.parametrize("fid,name", list(enumerate(P.FIELD_ORDER)))
def test_field_ops(fid, name):
    p = P.MODULI[name][0]
    n = P.nlimbs(p)
    rng = np.random.default_rng(fid)
    xs = edge_and_random(p, n, rng)
    ys = list(reversed(xs))
    A = np.stack([P.to_mont(x, p) for x in xs])
    B = np.stack([P.to_mont(y, p) for y in ys])
    enc = lambda vs: np.stack([P.to_mont(v, p) for v in vs]).reshape(-1)
    assert np.array_equal(O.field_op(fid, "add", A, B), enc([(x + y) % p for x, y in zip(xs, ys)]))
    assert np.array_equal(O.field_op(fid, "sub", A, B), enc([(x - y) % p for x, y in zip(xs, ys)]))
    assert np.array_equal(O.field_op(fid, "mul", A, B), enc([x * y % p for x, y in zip(xs, ys)]))
    assert np.array_equal(O.field_op(fid, "sqr", A), enc([x * x % p for x in xs]))
    assert np.array_equal(O.field_op(fid, "neg", A), enc([(-x) % p for x in xs]))
    assert np.array_equal(O.field_op(fid, "dbl", A), enc([2 * x % p for x in xs]))
    assert np.array_equal(O.field_op(fid, "inv", A), enc([pow(x, -1, p) if x else 0 for x in xs]))
    canon = np.stack([P.to_limbs(x, n) for x in xs]).reshape(-1)
    assert np.array_equal(O.field_op(fid, "into_bigint", A), canon)
    assert np.array_equal(O.field_op(fid, "from_bigint", canon), A.reshape(-1))


@pytest.mark.parametrize("cname", ["BLS12_377_G2", "BLS12_381_G2"])
def test_fp2_ops(cname):
    cid = O.CID[cname]
    cv = P.Curve(cname)
    F = cv.F
    rng = np.random.default_rng(3)
    p = cv.p
    xs = [(0, 0), (1, 0), (0, 1), (p - 1, p - 1), (5, p - 5)] + [
        (int.from_bytes(rng.bytes(48), "little") % p, int.from_bytes(rng.bytes(48), "little") % p) for _ in range(24)]
    ys = list(reversed(xs))
    A = np.stack([F.enc(x) for x in xs])
    B = np.stack([F.enc(y) for y in ys])
    enc = lambda vs: np.stack([F.enc(v) for v in vs]).reshape(-1)
    assert np.array_equal(O.basefield_op(cid, "add", A, B), enc([F.add(x, y) for x, y in zip(xs, ys)]))
    assert np.array_equal(O.basefield_op(cid, "sub", A, B), enc([F.sub(x, y) for x, y in zip(xs, ys)]))
    assert np.array_equal(O.basefield_op(cid, "mul", A, B), enc([F.mul(x, y) for x, y in zip(xs, ys)]))
    assert np.array_equal(O.basefield_op(cid, "sqr", A), enc([F.mul(x, x) for x in xs]))
    assert np.array_equal(O.basefield_op(cid, "neg", A), enc([F.neg(x) for x in xs]))
    nz = [x for x in xs if x != (0, 0)]
    assert np.array_equal(O.basefield_op(cid, "inv", np.stack([F.enc(x) for x in nz])), enc([F.inv(x) for x in nz]))
