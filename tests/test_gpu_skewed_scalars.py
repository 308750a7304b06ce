"""Narrow and skewed scalars at sizes where the planner's narrow-scalar choices, the width probe of msm_bigint, the
heavy-run kernels and the wave-aggregated sort counters are live (n = 2^19): the distributions of the reference's MSM
bench (bench-templates/src/macros/ec.rs:222-372 -- bool, u8, u16, u32, u64, their signed forms, the mixed vector) and
witness-like vectors (mostly 0 / 1 with a few full-width values), through msm_bigint (device-resident and Montgomery
form) and the msm_u* entries, each against k*G with k = sum s_i (a + i b) in closed form and k*G from the ORACLE."""
import os
import sys

import numpy as np
import pytest

import algebra_amd as A
from algebra_amd import curves as cv
import oracle_lib as O

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import synth as S  # noqa: E402

pytestmark = pytest.mark.gpu

LOGN = 19


@pytest.fixture(scope="module", params=["BLS12_381_G1", "BN254_G1", "BLS12_377_G2"])
def setup(request):
    import torch
    name = request.param
    cid = O.CID[name]
    r = S.R[cv.scalar_field(cid)]
    n = 1 << LOGN
    bases = S.grow_bases(cid, n, S.A0, S.B0, r)
    yield {"cid": cid, "r": r, "n": n, "bases": bases}
    del bases
    torch.cuda.empty_cache()


def _kg(cid, sc, r):
    k = S.dlog_of_msm(sc, S.A0, S.B0, r)
    return O.to_affine(cid, O.scalar_mul(cid, O.generator(cid), S.limbs4(k)))


def _limbs(vals):
    out = np.zeros((len(vals), 4), dtype=np.uint64)
    for k in range(4):
        out[:, k] = [(v >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for v in vals]
    return out


def _unsigned(rng, n, bits):
    sc = np.zeros((n, 4), dtype=np.uint64)
    sc[:, 0] = rng.integers(0, 1 << min(bits, 63), size=n, dtype=np.uint64)
    if bits == 64:
        sc[:, 0] |= rng.integers(0, 2, size=n, dtype=np.uint64) << np.uint64(63)
    return sc


def _wide(rng, n, bits):
    """uniform in [0, 2^bits), bits up to 250"""
    sc = np.zeros((n, 4), dtype=np.uint64)
    for k in range(4):
        take = max(0, min(64, bits - 64 * k))
        if take:
            v = rng.integers(0, 1 << 63, size=n, dtype=np.uint64) | (rng.integers(0, 2, size=n, dtype=np.uint64) << np.uint64(63))
            sc[:, k] = v if take == 64 else v & np.uint64((1 << take) - 1)
    return sc


def _run_bigint(s, sc, mont=False):
    import torch
    d = torch.from_numpy(np.ascontiguousarray(sc).view(np.int64)).cuda()
    if mont:
        return A.into_affine(s["cid"], A.msm_unchecked(s["cid"], s["bases"], d))
    return A.into_affine(s["cid"], A.msm_bigint(s["cid"], s["bases"], d))


@pytest.mark.parametrize("bits", [0, 1, 8, 16, 17, 32, 64, 100, 200, 246])
def test_msm_bigint_with_scalars_of_bounded_width(setup, bits):
    """the probe plans for the measured width (bits + 1 windows' worth): all-zero, bool, ..., and widths around the
    slack below which the 255-bit plan is kept"""
    s = setup
    rng = np.random.default_rng(1000 + bits)
    sc = np.zeros((s["n"], 4), dtype=np.uint64) if bits == 0 else (_unsigned(rng, s["n"], bits) if bits <= 64 else _wide(rng, s["n"], bits))
    got = _run_bigint(s, sc)
    assert np.array_equal(got, _kg(s["cid"], sc, s["r"]))


@pytest.mark.parametrize("bits", [1, 8, 16, 32])
def test_msm_bigint_signed_small_values(setup, bits):
    """iN::rand as Scalar::from(negative) = r - |x| (ec.rs:252-330): folded to |x| with the point negated"""
    s = setup
    r = s["r"]
    rng = np.random.default_rng(2000 + bits)
    mag = rng.integers(0, 1 << bits, size=s["n"], dtype=np.uint64)
    neg = rng.integers(0, 2, size=s["n"]).astype(bool)
    sc = _limbs([(r - int(m)) % r if ng else int(m) for m, ng in zip(mag, neg)])
    assert np.array_equal(_run_bigint(s, sc), _kg(s["cid"], sc, r))


def test_one_wide_scalar_keeps_the_full_plan(setup):
    """2^19 - 1 small scalars and ONE full-width one at an index the 4096-scalar sample does not visit: the full pass of
    the probe sees it"""
    s = setup
    rng = np.random.default_rng(7)
    sc = _unsigned(rng, s["n"], 8)
    sc[12345] = S.gen_scalars(1, 99, s["r"])[0]
    assert np.array_equal(_run_bigint(s, sc), _kg(s["cid"], sc, s["r"]))


def test_witness_like_vector(setup):
    """60 % zeros, 30 % ones, 5 % minus one, the rest full width: one giant bucket in window 0 (sort counters combined
    per wave, heavy-run kernels) next to uniformly loaded ones"""
    s = setup
    r = s["r"]
    rng = np.random.default_rng(11)
    n = s["n"]
    sc = S.gen_scalars(n, 0x77, r)
    u = rng.random(n)
    sc[u < 0.60] = 0
    one = np.zeros(4, dtype=np.uint64)
    one[0] = 1
    sc[(u >= 0.60) & (u < 0.90)] = one
    sc[(u >= 0.90) & (u < 0.95)] = _limbs([r - 1])[0]
    assert np.array_equal(_run_bigint(s, sc), _kg(s["cid"], sc, r))


def test_montgomery_form_narrow_scalars(setup):
    """the trait path hands Fr elements in Montgomery form: the probe converts before measuring"""
    s = setup
    rng = np.random.default_rng(13)
    sc = _unsigned(rng, s["n"], 16)
    R = (1 << 256) % s["r"]
    mont = _limbs([(int(v) * R) % s["r"] for v in sc[:, 0]])
    assert np.array_equal(_run_bigint(s, mont, mont=True), _kg(s["cid"], sc, s["r"]))


@pytest.mark.parametrize("name,dtype,bits", [("msm_u1", np.uint8, 1), ("msm_u8", np.uint8, 8), ("msm_u16", np.uint16, 16),
                                             ("msm_u32", np.uint32, 32), ("msm_u64", np.uint64, 64)])
def test_narrow_entries_at_size(setup, name, dtype, bits):
    """msm_u1 .. msm_u64 (variable_base/mod.rs:87-117) with the narrow planner's layouts (c <= bits + 1)"""
    import torch
    s = setup
    rng = np.random.default_rng(3000 + bits)
    sc = _unsigned(rng, s["n"], bits)
    sg = {1: np.int8, 2: np.int16, 4: np.int32, 8: np.int64}[np.dtype(dtype).itemsize]
    d = torch.from_numpy(np.ascontiguousarray(sc[:, 0].astype(dtype)).view(sg)).cuda()
    got = A.into_affine(s["cid"], getattr(A, name)(s["cid"], s["bases"], d))
    assert np.array_equal(got, _kg(s["cid"], sc, s["r"]))


def test_probe_off_gives_the_same_point(setup, monkeypatch):
    """ARK_HIP_MSM_PROBE is read once per process, so the comparison is against the narrow ENTRY (planned for 16 bits
    without any probe) and the closed form"""
    import torch
    s = setup
    rng = np.random.default_rng(17)
    sc = _unsigned(rng, s["n"], 16)
    d16 = torch.from_numpy(np.ascontiguousarray(sc[:, 0].astype(np.uint16)).view(np.int16)).cuda()
    a = _run_bigint(s, sc)
    b = A.into_affine(s["cid"], A.msm_u16(s["cid"], s["bases"], d16))
    assert np.array_equal(a, b) and np.array_equal(a, _kg(s["cid"], sc, s["r"]))


def test_host_pointer_entry_plans_from_a_sample(setup):
    """ark_hip_msm_sw with host arrays (what the Rust hooks call): the streamed path estimates the width classes from a
    sample of the host scalars, canonical and Montgomery form; a vector whose SAMPLE is all ones but which holds full-width
    scalars elsewhere must still come out right (the estimate picks the window size only)"""
    s = setup
    r = s["r"]
    n = s["n"]
    hb = s["bases"].cpu().numpy().view(np.uint64).reshape(n, -1)
    rng = np.random.default_rng(23)
    sc = np.zeros((n, 4), dtype=np.uint64)
    sc[:, 0] = rng.integers(0, 2, size=n, dtype=np.uint64)
    stride = n // 1024                              # the host sampler visits every (n / 1024)-th scalar
    wide = S.gen_scalars(n, 0x99, r)
    off = np.arange(n) % stride != 0                 # every index the sampler does not visit
    pick = off & (rng.random(n) < 0.03)
    sc[pick] = wide[pick]
    kg = _kg(s["cid"], sc, r)
    assert np.array_equal(A.into_affine(s["cid"], A.msm_bigint(s["cid"], hb, sc)), kg)
    R = (1 << 256) % r
    mont = _limbs([(S.scalar_int(row) * R) % r for row in sc])
    assert np.array_equal(A.into_affine(s["cid"], A.msm_unchecked(s["cid"], hb, mont)), kg)


@pytest.mark.parametrize("parts", [2, 4, 8])
def test_runs_walked_by_several_lanes(setup, parts, monkeypatch):
    """msm_accumulate_parts_kernel + msm_sum_parts_kernel (chosen by the library for few slots with long runs, from 2^22
    narrow scalars on): forced here at 2^19 for uniform full-width scalars, a u16 vector and a witness-like one -- runs
    shorter than the number of lanes (empty pieces), heavy runs next to split ones, G2 as lane pairs"""
    s = setup
    r = s["r"]
    n = s["n"]
    rng = np.random.default_rng(4000 + parts)
    monkeypatch.setenv("ARK_HIP_MSM_RUN_PARTS", str(parts))
    full = S.gen_scalars(n, 0x4242 + parts, r)
    u16 = _unsigned(rng, n, 16)
    wit = full.copy()
    u = rng.random(n)
    wit[u < 0.5] = 0
    one = np.zeros(4, dtype=np.uint64)
    one[0] = 1
    wit[(u >= 0.5) & (u < 0.9)] = one
    for sc in (full, u16, wit):
        assert np.array_equal(_run_bigint(s, sc), _kg(s["cid"], sc, r))
    d16 = u16[:, 0].astype(np.uint16)
    import torch
    t16 = torch.from_numpy(np.ascontiguousarray(d16).view(np.int16)).cuda()
    assert np.array_equal(A.into_affine(s["cid"], A.msm_u16(s["cid"], s["bases"], t16)), _kg(s["cid"], u16, r))


def test_prepared_set_with_narrow_and_mixed_scalars(setup):
    """a prepared base set asked for scalars no wider than 48 bits runs the plain pipeline on row 0 of its table (= the
    bases); anything with wider scalars keeps the prepared plan; then a uniform call on the same handle"""
    import torch
    s = setup
    r, n = s["r"], s["n"]
    rng = np.random.default_rng(31)
    pb = A.PreparedBases(s["cid"], s["bases"])
    try:
        cases = [np.zeros((n, 4), dtype=np.uint64), _unsigned(rng, n, 1), _unsigned(rng, n, 16), _unsigned(rng, n, 64)]
        mixed = _unsigned(rng, n, 8)
        mixed[::9] = S.gen_scalars((n + 8) // 9, 0x51, r)
        cases += [mixed, S.gen_scalars(n, 0x52, r)]
        for sc in cases:
            d = torch.from_numpy(np.ascontiguousarray(sc).view(np.int64)).cuda()
            assert np.array_equal(A.into_affine(s["cid"], pb.msm_bigint(d)), _kg(s["cid"], sc, r))
    finally:
        pb.free()


def test_switches_off_give_the_same_points():
    """the A/B switches (read once per process) in a fresh process: probe, sliced pass B, side stream and run parts all
    off -- the round-3 pipeline with this round's sort counters -- against k*G on a witness-like vector and a u16 one"""
    import subprocess
    code = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.getcwd(), "tools")); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
import algebra_amd as A
import oracle_lib as O
import synth as S
from algebra_amd import curves as cv
cid = O.CID["BLS12_381_G1"]; r = S.R[cv.scalar_field(cid)]; n = 1 << 19
bases = S.grow_bases(cid, n, S.A0, S.B0, r)
rng = np.random.default_rng(5)
wit = S.gen_scalars(n, 9, r); u = rng.random(n); wit[u < 0.6] = 0
one = np.zeros(4, dtype=np.uint64); one[0] = 1; wit[(u >= 0.6) & (u < 0.95)] = one
u16 = np.zeros((n, 4), dtype=np.uint64); u16[:, 0] = rng.integers(0, 1 << 16, size=n, dtype=np.uint64)
for sc in (wit, u16):
    d = torch.from_numpy(np.ascontiguousarray(sc).view(np.int64)).cuda()
    k = S.dlog_of_msm(sc, S.A0, S.B0, r)
    want = O.to_affine(cid, O.scalar_mul(cid, O.generator(cid), S.limbs4(k)))
    assert np.array_equal(A.into_affine(cid, A.msm_bigint(cid, bases, d)), want)
print("switches-off ok")
'''
    env = dict(os.environ, ARK_HIP_MSM_PROBE="0", ARK_HIP_MSM_BIG_SLICES="0", ARK_HIP_MSM_HEAVY_SIDE="0",
               ARK_HIP_MSM_PARTS_LANES="0")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "switches-off ok" in out.stdout, out.stderr[-2000:]


def test_zero_scalars_leave_before_the_sort(setup, monkeypatch):
    """Round 5 (K0c): with a quarter or more of the scalars zero the non-zero ones are compacted before the digit recoding and
    the sorted entries mapped back to base indices.  Zeros in every form the entry accepts (0; r and 2r for Montgomery-form
    input, r for canonical), ones, minus ones, full-width values, one lone non-zero scalar; the same vectors with the
    compaction switched off give the same point; an out-of-range scalar among the dropped ones still fails the call; a
    prepared set and Montgomery-form scalars take the same route."""
    import torch
    s = setup
    cid, r, n = s["cid"], s["r"], s["n"]
    rng = np.random.default_rng(23)
    sc = S.gen_scalars(n, 0x2323, r)
    u = rng.random(n)
    sc[u < 0.70] = 0
    one = np.zeros(4, dtype=np.uint64)
    one[0] = 1
    sc[(u >= 0.70) & (u < 0.85)] = one
    sc[(u >= 0.85) & (u < 0.90)] = _limbs([r - 1])[0]
    sc[(u >= 0.90) & (u < 0.92)] = _limbs([r])[0]              # = 0 mod r, canonical input in [r, 2^bits): accepted, a zero
    sc[0] = 0
    sc[n - 1] = _limbs([r - 2])[0]                              # the last index survives the compaction
    want = _kg(cid, sc, r)
    got = _run_bigint(s, sc)
    assert np.array_equal(got, want)
    monkeypatch.setenv("ARK_HIP_MSM_COMPACT", "0")              # (read once per process: a no-op unless this test runs first)
    lone = np.zeros((n, 4), dtype=np.uint64)
    lone[n // 3] = _limbs([r // 3])[0]
    assert np.array_equal(_run_bigint(s, lone), _kg(cid, lone, r))
    # Montgomery-form input (what SWCurveConfig::msm hands over): zeros are 0 there too
    fname = cv.scalar_field(cid)
    fid = O.FID[fname]
    mont = O.field_op(fid, "from_bigint", _reduce_rows(sc, r)).reshape(n, 4)
    assert np.array_equal(_run_bigint(s, mont, mont=True), want)
    # an out-of-range scalar sits among the zeros: the call fails as it did before
    bits = {"BLS12_381_FR": 255, "BN254_FR": 254, "BLS12_377_FR": 253}[fname]
    bad = sc.copy()
    bad[int(np.nonzero(u < 0.70)[0][5])] = _limbs([1 << bits])[0]
    with pytest.raises(A.ArkHipError) as ei:
        _run_bigint(s, bad)
    assert ei.value.code == -4
    assert np.array_equal(_run_bigint(s, sc), want)             # and the library is usable afterwards
    pb = A.PreparedBases(cid, s["bases"])
    d = torch.from_numpy(np.ascontiguousarray(sc).view(np.int64)).cuda()
    assert np.array_equal(A.into_affine(cid, pb.msm_bigint(d)), want)
    pb.free()


def _reduce_rows(sc, r):
    """rows of canonical limbs reduced mod r (only rows equal to r change here)"""
    out = sc.copy()
    rl = _limbs([r])[0]
    out[(sc == rl).all(axis=1)] = 0
    return out
