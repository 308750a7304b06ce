"""CPU-side checks of the product library (no GPU needed): it loads, exports every symbol the header
declares, and its host-side pieces (domain constants, projective sum, into_affine -- the same
templated formulas the kernels use, compiled for the host) agree with the oracle."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import algebra_amd as A
from algebra_amd import _lib
import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
A4 = np.array([0xA11CE, 1, 2, 0], dtype=np.uint64)
B4 = np.array([0xB0B, 3, 0, 0], dtype=np.uint64)


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "ark_hip.h")).read()
    i, j = hdr.index("#ifdef ARK_HIP_TEST_HOOKS"), hdr.index("#endif /* ARK_HIP_TEST_HOOKS */")
    public = set(re.findall(r"\b(ark_hip_[a-z0-9_]+)\s*\(", hdr[:i] + hdr[j:]))
    hooks = set(re.findall(r"\b(ark_hip_[a-z0-9_]+)\s*\(", hdr[i:j]))
    assert len(public) >= 20 and hooks and all(h.startswith("ark_hip_test_") for h in hooks)
    assert not any(p.startswith("ark_hip_test_") for p in public)
    L = _lib.lib()  # raises if the .so is missing
    for name in public:
        assert hasattr(L, name), name
    assert public == set(_lib.SYMBOLS), public ^ set(_lib.SYMBOLS)
    assert hooks == set(_lib.TEST_SYMBOLS), hooks ^ set(_lib.TEST_SYMBOLS)
    assert b"gfx950" in L.ark_hip_version()
    T = _lib.test_lib()
    for name in public | hooks:
        assert hasattr(T, name), name


def test_shipped_library_exports_the_c_abi_and_nothing_else():
    """`nm -D libark_hip.so | grep test_` is empty -- the hooks live in libark_hip_test.so -- and with
    -fvisibility=hidden the dynamic symbol table holds the functions of include/ark_hip.h, not the library's C++ internals."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    syms = [l.split()[-1] for l in out.splitlines() if l.split() and l.split()[-2] in ("T", "t", "W", "w", "B", "D", "V", "u")]
    assert not [s for s in syms if "test_" in s], [s for s in syms if "test_" in s][:5]
    ours = [s for s in syms if s.startswith("ark_hip_")]
    assert set(ours) == set(_lib.SYMBOLS), set(ours) ^ set(_lib.SYMBOLS)
    # nothing of namespace arkhip leaks (mangled names start with _ZN6arkhip)
    assert not [s for s in syms if "6arkhip" in s], [s for s in syms if "6arkhip" in s][:5]
    outt = subprocess.run(["nm", "-D", "--defined-only", _lib.TEST_LIB_PATH], capture_output=True, text=True, check=True).stdout
    for h in _lib.TEST_SYMBOLS:
        assert (" T " + h) in outt, h


def test_curve_info_matches_oracle():
    import ctypes as C
    for cid in range(5):
        fw, sf, bf, ext = (C.c_int() for _ in range(4))
        assert _lib.lib().ark_hip_curve_info(cid, C.byref(fw), C.byref(sf), C.byref(bf), C.byref(ext)) == 0
        ob, os_, oe = O.curve_info(cid)
        assert (fw.value, sf.value, bf.value, ext.value) == (O.fe_words(cid), os_, ob, oe)


@pytest.mark.parametrize("fname", ["BN254_FR", "BLS12_381_FR", "BLS12_377_FR"])
def test_domain_new_matches_oracle(fname):
    fid = O.FID[fname]
    for log_n in [0, 1, 2, 5, 10, 16, 22, 28]:
        d = A.Radix2EvaluationDomain.new(fname, 1 << log_n)
        g, gi, si = O.domain(fid, log_n)
        assert d.size() == 1 << log_n and d.log_size_of_group() == log_n
        assert np.array_equal(d.group_gen(), g)
        assert np.array_equal(d.group_gen_inv(), gi)
        assert np.array_equal(d.size_inv(), si)
        one = O.field_const(fid, 1)
        assert np.array_equal(d.coset_offset(), one) and np.array_equal(d.coset_offset_pow_size(), one)
        # size_as_field_element = F::from(size)
        sz = np.array([1 << log_n, 0, 0, 0], dtype=np.uint64)
        assert np.array_equal(d.size_as_field_element(), O.field_op(fid, "from_bigint", sz))
    # next_power_of_two semantics and the None case (radix2/mod.rs:55-64)
    assert A.Radix2EvaluationDomain.new(fname, 0).size() == 1
    assert A.Radix2EvaluationDomain.new(fname, 1000).size() == 1024
    two_adicity = {"BN254_FR": 28, "BLS12_381_FR": 32, "BLS12_377_FR": 47}[fname]
    if two_adicity < 40:
        assert A.Radix2EvaluationDomain.new(fname, (1 << two_adicity) + 1) is None
    # coset: offset = GENERATOR as in poly/benches/fft.rs:107
    gen = O.field_const(fid, 3)
    d = A.Radix2EvaluationDomain.new(fname, 64).get_coset(gen)
    assert np.array_equal(d.coset_offset(), gen)
    assert np.array_equal(O.field_op(fid, "mul", d.coset_offset(), d.coset_offset_inv()), O.field_const(fid, 1))
    p = gen
    for _ in range(6):
        p = O.field_op(fid, "sqr", p)
    assert np.array_equal(d.coset_offset_pow_size(), p)
    assert A.Radix2EvaluationDomain.new(fname, 64).get_coset(np.zeros(4, dtype=np.uint64)) is None


@pytest.mark.parametrize("cname", O.CURVES)
def test_host_sum_and_into_affine_match_oracle(cname):
    cid = O.CID[cname]
    bases = O.gen_bases(cid, A4, B4, 6)
    sc = O.gen_scalars(O.curve_info(cid)[1], 7, 6)
    pts = np.stack([O.scalar_mul(cid, bases[i], sc[i]) for i in range(6)])
    # into_affine
    got = A.into_affine(cid, pts)
    assert np.array_equal(got, O.to_affine(cid, pts))
    # identity -> (0,0)
    ident = O.msm(cid, bases[:0], sc[:0], O.NAIVE)
    assert np.array_equal(A.into_affine(cid, ident), np.zeros(2 * O.fe_words(cid), dtype=np.uint64))
    # sum of projective points == naive msm
    total = A.sum_projective(cid, pts)
    expect = O.msm(cid, bases, sc, O.NAIVE)
    assert np.array_equal(A.into_affine(cid, total), O.to_affine(cid, expect))
    # P + (-P) = identity ; P + P = 2P (doubling branch) ; with identity operand
    negp = pts[0].copy()
    fw = O.fe_words(cid)
    negp[fw:2 * fw] = O.basefield_op(cid, "neg", pts[0][fw:2 * fw])
    z = A.sum_projective(cid, np.stack([pts[0], negp]))
    assert np.array_equal(A.into_affine(cid, z), np.zeros(2 * fw, dtype=np.uint64))
    dbl = A.sum_projective(cid, np.stack([pts[0], pts[0], ident]))
    two = np.array([2, 0, 0, 0], dtype=np.uint64)
    s2 = O.field_op(O.curve_info(cid)[1], "mul", O.field_op(O.curve_info(cid)[1], "from_bigint", sc[0]),
                    O.field_op(O.curve_info(cid)[1], "from_bigint", two))
    s2 = O.field_op(O.curve_info(cid)[1], "into_bigint", s2)
    assert np.array_equal(A.into_affine(cid, dbl), O.to_affine(cid, O.scalar_mul(cid, bases[0], s2)))


def test_length_mismatch_is_reported_like_the_reference():
    # variable_base/mod.rs:73-77: Err(min(len)) ; raised before any device work
    bases = np.zeros((3, 12), dtype=np.uint64)
    scalars = np.zeros((2, 4), dtype=np.uint64)
    with pytest.raises(A.MsmLengthMismatch) as e:
        A.msm("BLS12_381_G1", bases, scalars)
    assert e.value.min_len == 2


def test_no_gpu_means_loud_failure_not_a_cpu_fallback():
    # The product has no CPU compute path: without a visible GPU every compute entry point returns
    # ARK_HIP_ERR_NO_DEVICE (-5) and the Python mirror raises.  (Skipped where a GPU is present.)
    L = _lib.lib()
    if L.ark_hip_device_count() > 0:
        pytest.skip("a GPU is visible here")
    cid = O.CID["BLS12_381_G1"]
    bases = O.gen_bases(cid, A4, B4, 4)
    scalars = O.gen_scalars(O.curve_info(cid)[1], 1, 4)
    with pytest.raises(A.ArkHipError) as e:
        A.msm_bigint(cid, bases, scalars)
    assert e.value.code == -5
    d = A.Radix2EvaluationDomain.new("BLS12_381_FR", 8)      # domain constants are host arithmetic: fine
    with pytest.raises(A.ArkHipError) as e:
        d.fft(np.zeros((8, 4), dtype=np.uint64))
    assert e.value.code == -5
    # the round-5 entries refuse the same way: device-resident vectors, the pointwise algebra, the transform over points
    import ctypes as C
    with pytest.raises(A.ArkHipError) as e:
        A.DeviceVec.from_host("BLS12_381_FR", np.zeros((4, 4), dtype=np.uint64))
    assert e.value.code == -5
    dummy = C.c_void_p(64)
    for rc in (L.ark_hip_fr_add_device(3, dummy, dummy, dummy, 1), L.ark_hip_fr_div_device(3, dummy, dummy, dummy, 1),
               L.ark_hip_fr_inverse_device(3, dummy, dummy, 1), L.ark_hip_memcpy_d2d(dummy, dummy, 32),
               L.ark_hip_memset_device(dummy, 0, 32)):
        assert rc == -5
    with pytest.raises(A.ArkHipError) as e:
        d.fft_group_in_place("BLS12_381_G1", np.zeros((8, 18), dtype=np.uint64))
    assert e.value.code == -5


def test_cpp_host_mirror_compiles_and_links(tmp_path):
    # include/ark_hip.hpp is plain C++17 over the C ABI: g++ alone must build a program against it (run on the GPU by
    # tests/test_gpu_cpp_mirror.py; here, without a GPU, the program must refuse to compute: exit code 2)
    import shutil
    import subprocess
    if not shutil.which("g++"):
        pytest.skip("g++ not available")
    exe = str(tmp_path / "hmc")
    subprocess.check_call(["g++", "-std=c++17", "-O0", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "oracle"),
                           os.path.join(ROOT, "tests", "cpp", "host_mirror_check.cpp"), "-o", exe,
                           "-L", os.path.join(ROOT, "algebra_amd"), "-lark_hip", "-L", os.path.join(ROOT, "oracle"),
                           "-lark_oracle", "-Wl,-rpath," + os.path.join(ROOT, "algebra_amd"),
                           "-Wl,-rpath," + os.path.join(ROOT, "oracle")], timeout=300)
    if _lib.lib().ark_hip_device_count() == 0:
        assert subprocess.run([exe], capture_output=True, timeout=120).returncode == 2


def test_msm_plan_is_host_only_and_sane():
    # ark_hip_msm_plan: pure host arithmetic (no GPU): the widths of the W windows must cover the scalar, wider windows
    # for larger inputs, and a prepared base set never plans more windows than the plain path
    import ctypes as C
    from algebra_amd._lib import lib
    L = lib()
    bits = {0: 254, 1: 255, 2: 253, 3: 253, 4: 255}
    for curve in range(5):
        prev = 0
        for logn in (0, 4, 10, 16, 20, 24, 26):
            got = {}
            for prepared in (0, 1):
                c, w = C.c_int(), C.c_int()
                assert L.ark_hip_msm_plan(curve, 1 << logn, prepared, C.byref(c), C.byref(w)) == 0
                assert 3 <= c.value <= 26 and w.value >= 1
                assert c.value * w.value >= bits[curve], (curve, logn, prepared, c.value, w.value)
                got[prepared] = (c.value, w.value)
            assert got[1][1] <= got[0][1] + 1
            if logn >= 16:
                assert got[0][0] >= prev
                prev = got[0][0]
    assert L.ark_hip_msm_plan(7, 100, 0, None, None) == -1
    # window sizes that were measured to be the fastest on MI355X (profiles/r2_msm_sweeps.txt): a change of the planner's
    # cost model must not silently move them
    measured = {(1, 24, 0): (20, 13), (1, 24, 1): (22, 12), (1, 25, 1): (22, 12), (1, 26, 1): (24, 11), (1, 23, 1): (22, 12),
                (1, 20, 1): (19, 14), (1, 16, 1): (17, 15), (0, 24, 1): (22, 12), (0, 23, 1): (22, 12), (3, 22, 1): (20, 13),
                (3, 16, 1): (17, 15), (4, 16, 1): (17, 15), (3, 12, 1): (17, 15), (3, 22, 0): (19, 14),
                # small plain MSMs: narrow windows with split runs (profiles/r5_small_n_run_parts.txt)
                (1, 16, 0): (12, 22), (1, 14, 0): (12, 22), (1, 12, 0): (10, 26), (1, 8, 0): (6, 43), (0, 16, 0): (12, 22),
                (3, 14, 0): (11, 23), (3, 12, 0): (10, 26), (1, 17, 0): (15, 17), (3, 16, 0): (14, 19)}
    for (curve, logn, prepared), want in measured.items():
        c, w = C.c_int(), C.c_int()
        assert L.ark_hip_msm_plan(curve, 1 << logn, prepared, C.byref(c), C.byref(w)) == 0
        assert (c.value, w.value) == want, (curve, logn, prepared, c.value, w.value)


def test_msm_plan_follows_the_width_classes():
    # ark_hip_msm_plan_widths: the plan msm_bigint uses after its width probe (the GPU form of msm_signed's partition,
    # variable_base/mod.rs:251-336).  Host arithmetic only.
    import ctypes as C
    import numpy as np
    from algebra_amd._lib import lib
    L = lib()
    TOP = (0, 1, 8, 16, 32, 64, 128, 192, 256)
    fbits = {0: 254, 1: 255, 2: 253, 3: 253, 4: 255}

    def plan(curve, n, max_bits, counts):
        arr = (C.c_uint32 * 9)(*[int(x) for x in counts])
        c, w = C.c_int(), C.c_int()
        assert L.ark_hip_msm_plan_widths(curve, n, max_bits, arr, C.byref(c), C.byref(w)) == 0
        return c.value, w.value

    def uniform(curve, n):
        c, w = C.c_int(), C.c_int()
        assert L.ark_hip_msm_plan(curve, n, 0, C.byref(c), C.byref(w)) == 0
        return c.value, w.value

    for curve in range(5):
        for logn in (19, 22, 24, 26):
            n = 1 << logn
            # every scalar full width: the plan of ark_hip_msm_plan
            assert plan(curve, n, fbits[curve], [0] * 8 + [n]) == uniform(curve, n)
            # one class only: a single window of exactly the buckets its digits reach where that fits, never c > bits + 1
            for k, top in enumerate(TOP[1:6], start=1):
                cnt = [0] * 9
                cnt[k] = n
                c, w = plan(curve, n, top, cnt)
                assert c <= max(3, top + 1) and c * w >= top + 1 and (w - 1) * c < top + 1 + w, (curve, logn, top, c, w)
            assert plan(curve, n, 16, [0, 0, 0, n, 0, 0, 0, 0, 0]) == (17, 1)
            # a witness: few full-width scalars among zeros and ones -> the window size of a much smaller MSM, same coverage
            wit = [n * 6 // 10, n * 35 // 100, 0, 0, 0, 0, 0, 0, n - n * 6 // 10 - n * 35 // 100]
            c, w = plan(curve, n, fbits[curve], wit)
            cu, wu = uniform(curve, n)
            assert c <= cu and c * w >= fbits[curve] and (c < cu or logn < 22), (curve, logn, c, w, cu, wu)
            # all zero: still a valid (tiny) plan
            c, w = plan(curve, n, 0, [n] + [0] * 8)
            assert c >= 3 and w == 1
    # random histograms: the windows always cover the widest scalar, a wider maximum never plans fewer total bits
    rng = np.random.default_rng(5)
    for _ in range(300):
        curve = int(rng.integers(0, 5))
        n = 1 << int(rng.integers(19, 27))
        wts = rng.random(9) * (rng.random(9) < 0.5)
        if wts.sum() == 0:
            wts[int(rng.integers(0, 9))] = 1.0
        cnt = np.floor(wts / wts.sum() * n).astype(np.int64)
        top_class = max(k for k in range(9) if cnt[k] > 0)
        lo = TOP[top_class - 1] + 1 if top_class else 0
        hi = min(TOP[top_class], fbits[curve] - 1) if top_class else 0   # folded magnitudes are below r/2
        max_bits = int(rng.integers(lo, max(lo, hi) + 1))
        c, w = plan(curve, n, max_bits, cnt)
        assert 3 <= c <= 26 and w >= 1 and c * w >= min(max_bits + 1, fbits[curve]), (curve, n, max_bits, cnt.tolist(), c, w)
    assert L.ark_hip_msm_plan_widths(9, 1, 0, (C.c_uint32 * 9)(), None, None) == -1
    assert L.ark_hip_msm_plan_widths(1, 1, 0, None, None, None) == -1


def test_synth_discrete_log_identity_is_exact_for_large_indices():
    # tools/synth.py::dlog_of_msm backs every at-size parity check (k*G identity): its chunked uint64 sums must stay exact
    # for the largest index ranges used (a 2^28-pair job overflowed the first version's chunks of 32)
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import numpy as np
    import synth as S
    r = S.R["BLS12_381_FR"]
    sc = S.gen_scalars(1500, 1, r)
    for first in (0, (1 << 26) - 700, (1 << 28) - 1500):
        want = 0
        for i in range(sc.shape[0]):
            s = sum(int(sc[i, j]) << (64 * j) for j in range(4))
            want += s * (S.A0 + (first + i) * S.B0)
        assert S.dlog_of_msm(sc, S.A0, S.B0, r, first_index=first) == want % r
    # the large-n summation path (n > 2^26 switches to chunks of 4): entries just below 2^61 must not wrap
    x = np.full(1003, (1 << 61) - 1, dtype=np.uint64)
    assert S._exact_sum(x, 4) == 1003 * ((1 << 61) - 1)
    assert S._exact_sum(np.full(77, (1 << 58) - 1, dtype=np.uint64)) == 77 * ((1 << 58) - 1)


@pytest.mark.parametrize("cname", O.CURVES)
def test_msm_host_tail_folds_bit_sums_like_the_window_combine(cname):
    # The serial tail of every MSM (msm_finish -> msm_host_fold: the window combine of
    # ec/src/scalar_mul/variable_base/mod.rs:489-502 over bit-sliced bucket sums), run on caller-supplied parts:
    #   out = sum_w 2^(off_w) (A_w + 2^l0 sum_b 2^b U_(w,b))  ==  one naive MSM of the same points with those weights.
    # Parts are bucket-form points (x t^2, y t^3, t^2, t^3) with a different t each; some are the identity (zz = 0).
    import ctypes as C
    cid = O.CID[cname]
    fw = O.fe_words(cid)
    for windows, nbits, l0, widths in ((3, 4, 2, [7, 7, 6]), (1, 5, 0, [20]), (4, 3, 3, [6, 6, 6, 5]),
                                       (11, 4, 2, [9, 9, 9, 9, 9, 9, 9, 8, 8, 8, 8])):   # (>= 4 windows over Fp2: the threaded tail)
        npts = windows * (nbits + 1)
        aff = O.gen_bases(cid, A4, B4, npts + 1)
        t_src = O.gen_bases(cid, B4, A4, npts)                      # x coordinates of other points: the t values
        parts = np.zeros((windows, nbits + 1, 4 * fw), dtype=np.uint64)
        weights, pts = [], []
        off = 0
        for w in range(windows):
            for q in range(nbits + 1):
                k = w * (nbits + 1) + q
                if k % 5 == 3:
                    continue                                        # identity part: all-zero cell
                x, y = aff[k][:fw], aff[k][fw:]
                t = t_src[k][:fw]
                t2 = O.basefield_op(cid, "mul", t, t)
                t3 = O.basefield_op(cid, "mul", t2, t)
                parts[w, q, :fw] = O.basefield_op(cid, "mul", x, t2)
                parts[w, q, fw:2 * fw] = O.basefield_op(cid, "mul", y, t3)
                parts[w, q, 2 * fw:3 * fw] = t2
                parts[w, q, 3 * fw:] = t3
                pos = off if q == nbits else off + l0 + q
                weights.append(1 << pos)
                pts.append(aff[k])
            off += widths[w]
        out = np.zeros(3 * fw, dtype=np.uint64)
        wid = (C.c_int * windows)(*widths)
        parts = np.ascontiguousarray(parts)
        rc = _lib.test_lib().ark_hip_test_msm_host_fold(cid, parts.ctypes.data_as(C.c_void_p), windows, nbits, l0, wid,
                                                   out.ctypes.data_as(C.c_void_p))
        assert rc == 0
        sc = np.array([[(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)] for v in weights], dtype=np.uint64)
        want = O.to_affine(cid, O.msm(cid, np.stack(pts), sc, O.NAIVE))
        assert np.array_equal(A.into_affine(cid, out), want), (cname, windows, nbits, l0)
    assert _lib.test_lib().ark_hip_test_msm_host_fold(cid, None, 1, 1, 0, None, None) != 0      # argument check


def _tag(a):
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1)
    out = (C.c_uint64 * 2)()
    assert _lib.test_lib().ark_hip_test_base_hash(a.ctypes.data_as(C.c_void_p), a.size, out) == 0
    return int(out[0]) | (int(out[1]) << 64)


def test_verified_cache_tag_is_keyed_128_bits_and_sees_what_the_round4_hash_missed():
    """the round-4 validation hash -- four unkeyed 64-bit multiply-rotate lanes with published
    constants -- admitted collisions constructible from the source: flip any bits of word i and repair the lane's state with
    word i + 4.  The round-5 tag (NH under a per-process key) must tell such a pair apart, must see every single-bit edit,
    transpositions, length changes, edits in any 64 KiB block, and must be stable within the process."""
    M64 = (1 << 64) - 1
    K = 0xff51afd7ed558ccd

    def step(h, v):
        h = ((h ^ v) * K) & M64
        return ((h << 31) | (h >> 33)) & M64

    rng = np.random.default_rng(5)
    words = 3 * 8192 + 77                      # three full blocks and a ragged one
    a = rng.integers(0, 1 << 63, size=words, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=words, dtype=np.uint64)
    t0 = _tag(a)
    assert t0 == _tag(a.copy()) and t0 >> 64 != 0 and t0 & M64 != 0
    # the constructible collision of the old hash, lane 0 of block 1 (seed = block index 1): words i = 8192 and i + 4
    i = 8192
    h_before = 1 ^ 0x9e3779b97f4a7c15          # lane 0's state entering the block
    v0, v4 = int(a[i]), int(a[i + 4])
    v0x = v0 ^ 0x0123456789abcdef
    h1, h1x = step(h_before, v0), step(h_before, v0x)
    v4x = v4 ^ h1 ^ h1x                        # (h1x ^ v4x) == (h1 ^ v4): the old lane state is identical from here on
    assert step(h1, v4) == step(h1x, v4x)      # ... which is exactly why the old hash could not see the edit
    b = a.copy()
    b[i] = np.uint64(v0x)
    b[i + 4] = np.uint64(v4x)
    assert not np.array_equal(a, b) and _tag(b) != t0
    # every kind of local edit, in every block
    for pos in (0, 1, 5, 8191, 8192, 2 * 8192 + 3, words - 1):
        for bit in (0, 31, 63):
            c = a.copy()
            c[pos] ^= np.uint64(1 << bit)
            assert _tag(c) != t0, (pos, bit)
    c = a.copy()
    c[[10, 11]] = c[[11, 10]]
    assert _tag(c) != t0
    c = a.copy()
    c[[3, 3 + 8192]] = c[[3 + 8192, 3]]        # the same position of two blocks swapped
    assert _tag(c) != t0
    assert _tag(a[:-1]) != t0 and _tag(np.concatenate([a, np.zeros(1, dtype=np.uint64)])) != t0
    z = np.zeros(8192 * 2, dtype=np.uint64)
    assert _tag(z) != _tag(z[:8192]) and _tag(z) != 0
    assert len({_tag(rng.integers(0, 1 << 62, size=64, dtype=np.uint64)) for _ in range(200)}) == 200


@pytest.mark.parametrize("cname", O.CURVES)
def test_host_field_arithmetic_on_edge_values(cname):
    # The MSM's serial tail runs on the HOST builds of the base field's arithmetic (fp.cuh: products, additions, subtractions,
    # doublings and negation on 64-bit limbs -- montgomery_backend.rs:129-246 restated twice, once per side).  Edge values where a
    # carry / borrow / the final subtraction decides: 0, 1, 2, p - 1, p - 2, (p - 1) / 2, (p + 1) / 2, limbs of all ones below p,
    # and seeded random elements, every pair of them through add / sub / mul and each one through sqr / neg / dbl -- against the oracle.
    import ctypes as C
    cid = O.CID[cname]
    bf, sf, ext = O.curve_info(cid)
    fw = O.fe_words(cid)
    comp = fw // ext                                   # words per Fp component
    p = sum(int(v) << (64 * k) for k, v in enumerate(O.field_const(bf, 0)))
    rng = np.random.default_rng(20240925)
    ints = [0, 1, 2, p - 1, p - 2, (p - 1) // 2, (p + 1) // 2, (1 << (64 * (comp - 1))) - 1, ((1 << (p.bit_length() - 1)) - 1),
            1 << 64, (1 << 64) - 1, p - (1 << 64)] + [int.from_bytes(rng.bytes(8 * comp), "little") % p for _ in range(6)]

    def elem(vals):                                     # ext components, canonical integers -> Montgomery limbs as the library holds them
        raw = np.array([[(v >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for k in range(comp)] for v in vals], dtype=np.uint64).reshape(-1)
        return raw
    # treat the integers as Montgomery residues directly: every N-limb value below p is a valid element
    elems = []
    for i, v in enumerate(ints):
        elems.append(elem([v] + [ints[(i * 7 + 3) % len(ints)]] * (ext - 1)))
    m = len(elems)
    a = np.stack([elems[i] for i in range(m) for _ in range(m)])
    b = np.stack([elems[j] for _ in range(m) for j in range(m)])
    L = _lib.test_lib()
    for op in ("add", "sub", "mul"):
        got = np.zeros_like(a)
        assert L.ark_hip_test_host_basefield_op(cid, O.OPS[op], a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p),
                                                got.ctypes.data_as(C.c_void_p), a.shape[0]) == 0
        assert np.array_equal(got.reshape(-1), O.basefield_op(cid, op, a, b)), (cname, op)
    one = np.stack(elems)
    for op in ("sqr", "neg", "dbl"):
        got = np.zeros_like(one)
        assert L.ark_hip_test_host_basefield_op(cid, O.OPS[op], one.ctypes.data_as(C.c_void_p), None,
                                                got.ctypes.data_as(C.c_void_p), one.shape[0]) == 0
        assert np.array_equal(got.reshape(-1), O.basefield_op(cid, op, one)), (cname, op)
    assert L.ark_hip_test_host_basefield_op(cid, 0, one.ctypes.data_as(C.c_void_p), None, None, 1) != 0   # argument check


def test_msm_plan_is_defined_for_every_size():
    # the cost model divides by bucket counts, takes exp / sqrt / ldexp of loads and rounds: every size from 1 to 2^28 pairs -- the rule
    # boundaries of the small-n plan (256, 20 480, 24 576, 73 728) and their neighbours included -- must give a layout that covers the scalar,
    # keeps the sort's 32-bit positions and does not depend on anything but (curve, n, prepared)
    import ctypes as C
    L = _lib.lib()
    bits = {0: 254, 1: 255, 2: 253, 3: 253, 4: 255}
    sizes = sorted(set([1, 2, 3, 5, 17, 100, 255, 256, 257, 1000, 4095, 4096, 20479, 20480, 20481, 24575, 24576, 24577, 65535, 65536, 65537,
                        73727, 73728, 73729, 100000] + [(1 << k) + d for k in range(17, 29) for d in (-1, 0, 1)]))
    for curve in range(5):
        for prepared in (0, 1):
            for n in sizes:
                if n > (1 << 28):
                    continue
                c, w = C.c_int(), C.c_int()
                assert L.ark_hip_msm_plan(curve, n, prepared, C.byref(c), C.byref(w)) == 0
                assert 3 <= c.value <= 26 and 1 <= w.value <= 90, (curve, prepared, n, c.value, w.value)
                assert c.value * w.value >= bits[curve], (curve, prepared, n, c.value, w.value)
                assert n * w.value < (1 << 32), (curve, prepared, n, c.value, w.value)
                c2, w2 = C.c_int(), C.c_int()
                assert L.ark_hip_msm_plan(curve, n, prepared, C.byref(c2), C.byref(w2)) == 0 and (c2.value, w2.value) == (c.value, w.value)


def test_helper_pool_is_one_bounded_set_of_threads_under_concurrent_callers():
    """VERDICT r5 weak #5 / ADVICE r5 (medium): the host tails no longer create threads per call.  Sixteen host threads run the
    Fp2 host tail (eleven windows: the pooled path) and the verified cache's hashing pass at once, on the CPU only; every
    result equals the single-threaded one, and the pool's thread count is what it was before (csrc/hostpool.hpp)."""
    import threading
    L = _lib.test_lib()   # the hooks' library: a separate instance of the product's objects, with a pool of its own
    out = (C.c_int * 2)()
    assert L.ark_hip_host_threads(out) == 0
    helpers, created = out[0], out[1]
    assert 0 <= helpers <= 16 and created == helpers
    assert helpers <= max(0, len(os.sched_getaffinity(0)) - 1)      # the caller is a worker too
    cid = O.CID["BLS12_377_G2"]
    fw = O.fe_words(cid)
    windows, nbits, l0 = 11, 4, 2
    widths = [9, 9, 9, 9, 9, 9, 9, 8, 8, 8, 8]
    npts = windows * (nbits + 1)
    aff = O.gen_bases(cid, A4, B4, npts)
    parts = np.zeros((windows, nbits + 1, 4 * fw), dtype=np.uint64)
    t_src = O.gen_bases(cid, B4, A4, npts)                          # x coordinates of other points: the t values
    for w in range(windows):
        for q in range(nbits + 1):
            k = w * (nbits + 1) + q
            x, y, t = aff[k][:fw], aff[k][fw:], t_src[k][:fw]
            t2 = O.basefield_op(cid, "mul", t, t)
            t3 = O.basefield_op(cid, "mul", t2, t)
            parts[w, q, :fw] = O.basefield_op(cid, "mul", x, t2)    # bucket form (x t^2, y t^3, t^2, t^3)
            parts[w, q, fw:2 * fw] = O.basefield_op(cid, "mul", y, t3)
            parts[w, q, 2 * fw:3 * fw] = t2
            parts[w, q, 3 * fw:] = t3
    parts = np.ascontiguousarray(parts)
    wid = (C.c_int * windows)(*widths)

    def fold():
        o = np.zeros(3 * fw, dtype=np.uint64)
        assert L.ark_hip_test_msm_host_fold(cid, parts.ctypes.data_as(C.c_void_p), windows, nbits, l0, wid,
                                            o.ctypes.data_as(C.c_void_p)) == 0
        return A.into_affine(cid, o)

    rng = np.random.default_rng(11)
    buf = rng.integers(0, 1 << 62, size=(8 << 20) // 8, dtype=np.uint64)    # 8 MiB: 128 blocks, several ranges
    want_pt, want_tag = fold(), _tag(buf)
    errors = []

    def worker(t):
        try:
            for it in range(6):
                if not np.array_equal(fold(), want_pt):
                    errors.append(("fold", t, it))
                if _tag(buf) != want_tag:
                    errors.append(("tag", t, it))
        except Exception as ex:  # noqa: BLE001
            errors.append(("exc", t, repr(ex)))

    th = [threading.Thread(target=worker, args=(t,)) for t in range(16)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors[:5]
    assert L.ark_hip_host_threads(out) == 0 and (out[0], out[1]) == (helpers, created)
