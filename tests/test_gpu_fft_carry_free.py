"""The carry-free pass kernel of the FFT (fft_pass29_kernel: 9 x 29-bit limbs, ark_hip_fft_set_kernel(1)) -- kept beside the
default saturated kernel (it measured no faster: DESIGN.md section 5) and held to the same parity: forward / inverse / coset
against the oracle on all three scalar fields from 2^0 to 2^14, the degree-aware path, a multi-pass size (2^17: passes of
8 + 8 + 1 stages exercise the odd single-stage round, the 9-limb ping buffer and the exact final reduction) and the
BASELINE size 2^22 limb for limb; and both kernels against each other."""
import numpy as np
import pytest

import algebra_amd as A
import oracle_lib as O
from algebra_amd._lib import check, lib

pytestmark = pytest.mark.gpu
FR = ["BN254_FR", "BLS12_381_FR", "BLS12_377_FR"]


@pytest.fixture(autouse=True)
def carry_free_kernel():
    check(lib().ark_hip_fft_set_kernel(1), "fft_set_kernel")
    yield
    check(lib().ark_hip_fft_set_kernel(-1), "fft_set_kernel")


def rand_fr(fid, n, seed):
    return O.gen_scalars(fid, seed, max(n, 1), montgomery=True)[:n]


@pytest.mark.parametrize("fname", FR)
@pytest.mark.parametrize("log_n", [0, 1, 2, 3, 5, 8, 10, 11, 12, 13, 14])
def test_carry_free_fft_matches_oracle(fname, log_n):
    fid = O.FID[fname]
    x = rand_fr(fid, 1 << log_n, 300 + log_n)
    d = A.Radix2EvaluationDomain.new(fname, 1 << log_n)
    gen = O.field_const(fid, 3)
    dc = d.get_coset(gen)
    assert np.array_equal(d.fft(x).reshape(-1), O.fft(fid, x, log_n, None, False, 4))
    assert np.array_equal(d.ifft(x).reshape(-1), O.fft(fid, x, log_n, None, True, 4))
    assert np.array_equal(dc.fft(x).reshape(-1), O.fft(fid, x, log_n, gen, False, 4))
    assert np.array_equal(dc.ifft(x).reshape(-1), O.fft(fid, x, log_n, gen, True, 4))


@pytest.mark.parametrize("fname", FR)
def test_carry_free_edge_values_and_degree_aware(fname):
    fid = O.FID[fname]
    log_n = 12
    n = 1 << log_n
    d = A.Radix2EvaluationDomain.new(fname, n)
    # all p - 1 (the largest residue: every sum is at the top of its bound), all zero, one hot
    pm1 = np.tile(O.field_op(fid, "neg", O.field_const(fid, 1)), (n, 1))   # -1 in Montgomery form
    for x in (pm1, np.zeros((n, 4), dtype=np.uint64), np.concatenate([O.field_const(fid, 1)[None, :], np.zeros((n - 1, 4), dtype=np.uint64)])):
        assert np.array_equal(d.fft(x).reshape(-1), O.fft(fid, x, log_n, None, False, 4))
        assert np.array_equal(d.ifft(x).reshape(-1), O.fft(fid, x, log_n, None, True, 4))
    # degree-aware path (len * 4 <= size): power-of-two and ragged lengths
    for ln in (1, 5, 64, 1000):
        c = rand_fr(fid, ln, 9 + ln)
        full = np.concatenate([c, np.zeros((n - ln, 4), dtype=np.uint64)])
        assert np.array_equal(d.fft(c).reshape(-1), O.fft(fid, full, log_n, None, False, 4)), (fname, ln)


def test_carry_free_multi_pass_sizes_and_both_kernels_agree():
    import torch
    fname = "BLS12_381_FR"
    fid = O.FID[fname]
    for log_n in (17, 22):
        n = 1 << log_n
        x = rand_fr(fid, n, 4000 + log_n)
        d = A.Radix2EvaluationDomain.new(fname, n)
        gen = O.field_const(fid, 3)
        dc = d.get_coset(gen)
        dx = torch.from_numpy(x.view(np.int64)).cuda()
        got = d.fft(dx).cpu().numpy().view(np.uint64).reshape(-1)
        assert np.array_equal(got, O.fft(fid, x, log_n, None, False, 16)), log_n
        goti = dc.ifft(dx).cpu().numpy().view(np.uint64).reshape(-1)
        assert np.array_equal(goti, O.fft(fid, x, log_n, gen, True, 16)), log_n
        check(lib().ark_hip_fft_set_kernel(0), "fft_set_kernel")        # the default kernel on the same input
        sat = d.fft(dx).cpu().numpy().view(np.uint64).reshape(-1)
        check(lib().ark_hip_fft_set_kernel(1), "fft_set_kernel")
        assert np.array_equal(sat, got)
