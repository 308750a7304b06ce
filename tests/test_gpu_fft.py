"""Radix-2 FFT / IFFT parity (GPU): HIP path through the C ABI vs the oracle (serial Cooley-Tukey
restatement and direct Horner evaluation), mirroring poly/src/domain/radix2/mod.rs:351-536 and
poly/src/test.rs:12-60.  Outputs are compared limb-for-limb (field elements are canonical)."""
import numpy as np
import pytest

import algebra_amd as A
import oracle_lib as O

pytestmark = pytest.mark.gpu

FR = ["BN254_FR", "BLS12_381_FR", "BLS12_377_FR"]


def rand_fr(fid, n, seed):
    return O.gen_scalars(fid, seed, max(n, 1), montgomery=True)[:n]


@pytest.mark.parametrize("fname", FR)
@pytest.mark.parametrize("log_n", list(range(0, 15)))
def test_fft_matches_oracle_all_small_sizes(fname, log_n):
    fid = O.FID[fname]
    n = 1 << log_n
    x = rand_fr(fid, n, 100 + log_n)
    d = A.Radix2EvaluationDomain.new(fname, n)
    got = d.fft(x)
    assert np.array_equal(got.reshape(-1), O.fft(fid, x, log_n, None, False, 4))
    goti = d.ifft(x)
    assert np.array_equal(goti.reshape(-1), O.fft(fid, x, log_n, None, True, 4))
    # coset (offset = GENERATOR, poly/benches/fft.rs:107)
    gen = O.field_const(fid, 3)
    dc = d.get_coset(gen)
    assert np.array_equal(dc.fft(x).reshape(-1), O.fft(fid, x, log_n, gen, False, 4))
    assert np.array_equal(dc.ifft(x).reshape(-1), O.fft(fid, x, log_n, gen, True, 4))


@pytest.mark.parametrize("fname", FR)
def test_fft_is_polynomial_evaluation(fname):
    # test_fft_correctness (radix2/mod.rs:351-391): fft(p)[i] == p(domain.element(i)), degree 31 on 32/64
    fid = O.FID[fname]
    coeffs = rand_fr(fid, 32, 5)
    gen = O.field_const(fid, 3)
    for log_n in (5, 6):
        d = A.Radix2EvaluationDomain.new(fname, 1 << log_n)
        assert np.array_equal(d.fft(coeffs).reshape(-1), O.dft_naive(fid, coeffs, log_n, None))  # zero-extended input
        dc = d.get_coset(gen)
        assert np.array_equal(dc.fft(coeffs).reshape(-1), O.dft_naive(fid, coeffs, log_n, gen))
        back = dc.ifft(dc.fft(coeffs))
        assert np.array_equal(back[:32], coeffs) and not back[32:].any()


def test_fft_ifft_identity_small_vector():
    # test_fft_ifft_identity on [1..8] (radix2/mod.rs:581-600)
    fid = O.FID["BLS12_381_FR"]
    v = np.zeros((8, 4), dtype=np.uint64)
    v[:, 0] = np.arange(1, 9)
    x = O.field_op(fid, "from_bigint", v).reshape(8, 4)
    d = A.Radix2EvaluationDomain.new("BLS12_381_FR", 8)
    assert np.array_equal(d.ifft(d.fft(x)), x)


@pytest.mark.parametrize("fname,log_n", [("BLS12_381_FR", 16), ("BLS12_381_FR", 20), ("BN254_FR", 17),
                                         ("BLS12_377_FR", 19)])
def test_fft_large_matches_oracle(fname, log_n):
    fid = O.FID[fname]
    n = 1 << log_n
    x = rand_fr(fid, n, 31 + log_n)
    d = A.Radix2EvaluationDomain.new(fname, n)
    assert np.array_equal(d.fft(x).reshape(-1), O.fft(fid, x, log_n, None, False, 8))
    gen = O.field_const(fid, 3)
    dc = d.get_coset(gen)
    assert np.array_equal(dc.ifft(x).reshape(-1), O.fft(fid, x, log_n, gen, True, 8))


def test_fft_2_22_bls12_381_device_resident():
    # BASELINE config 3: BLS12-381 Fr, domain 2^22, on device memory; vs the oracle and round trip
    import torch
    fname, log_n = "BLS12_381_FR", 22
    fid = O.FID[fname]
    n = 1 << log_n
    x = rand_fr(fid, n, 2222)
    d = A.Radix2EvaluationDomain.new(fname, n)
    dx = torch.from_numpy(x.view(np.int64)).cuda()
    y = d.fft(dx)
    exp = O.fft(fid, x, log_n, None, False, 8)
    assert np.array_equal(y.cpu().numpy().view(np.uint64).reshape(-1), exp)
    back = d.ifft_in_place(y)
    assert torch.equal(back, dx)
    # coset round trip at full size
    dc = d.get_coset(O.field_const(fid, 3))
    assert torch.equal(dc.ifft(dc.fft(dx)), dx)
    # linearity: fft(a + b) = fft(a) + fft(b) on a slice
    b = rand_fr(fid, n, 3333)
    ab = O.field_op(fid, "add", x, b).reshape(n, 4)
    ya = y = d.fft(dx).cpu().numpy().view(np.uint64).reshape(n, 4)
    yb = d.fft(b).reshape(n, 4)
    yab = d.fft(ab).reshape(n, 4)
    assert np.array_equal(O.field_op(fid, "add", ya, yb).reshape(n, 4), yab)


@pytest.mark.parametrize("fname", FR)
def test_polynomial_multiplication_on_device(fname):
    # &DensePolynomial * &DensePolynomial (dense.rs:641-656): FFT, pointwise product, IFFT without leaving the GPU;
    # checked against the schoolbook product computed with the oracle's field arithmetic
    fid = O.FID[fname]
    for la, lb in [(1, 1), (3, 5), (17, 16), (100, 29)]:
        a = rand_fr(fid, la, 71 + la)
        b = rand_fr(fid, lb, 93 + lb)
        got = A.poly_mul(fname, a, b)
        exp = np.zeros((la + lb - 1, 4), dtype=np.uint64)
        for i in range(la):
            prod = O.field_op(fid, "mul", np.tile(a[i], (lb, 1)), b).reshape(lb, 4)
            exp[i:i + lb] = O.field_op(fid, "add", exp[i:i + lb], prod).reshape(lb, 4)
        assert np.array_equal(got, exp), (fname, la, lb)
        # the ONE host-pointer entry the Rust hook behind `&a * &b` binds (ark_hip_poly_mul): same coefficients
        assert np.array_equal(A.poly_mul_host(fname, a, b), exp), (fname, la, lb, "host entry")
    assert A.poly_mul(fname, np.zeros((0, 4), dtype=np.uint64), rand_fr(fid, 3, 1)).shape == (0, 4)
    # DensePolynomial::is_zero: an empty or all-zero factor gives the zero polynomial (dense.rs:646-648)
    assert A.poly_mul_host(fname, np.zeros((0, 4), dtype=np.uint64), rand_fr(fid, 3, 1)).shape == (0, 4)
    assert A.poly_mul_host(fname, rand_fr(fid, 3, 1), np.zeros((5, 4), dtype=np.uint64)).shape == (0, 4)
    # leading zeros of the product are dropped (from_coefficients_vec): (x - 1)(x + 1) in characteristic != 2 keeps 3,
    # a factor padded with zero coefficients keeps none of the padding
    a = rand_fr(fid, 6, 5)
    a[4:] = 0
    b = rand_fr(fid, 9, 6)
    got = A.poly_mul_host(fname, a, b)
    assert got.shape[0] == 4 + 9 - 1 and np.array_equal(got, A.poly_mul(fname, a[:4], b))


def test_polynomial_multiplication_host_entry_at_size():
    """2^20 x 2^20 coefficients through ark_hip_poly_mul (domain 2^21: one H2D of 64 MiB, one D2H of 64 MiB) against the
    device-resident composition, and linearity: (a + a') b = a b + a' b."""
    fname = "BLS12_381_FR"
    fid = O.FID[fname]
    n = 1 << 20
    a, a2, b = rand_fr(fid, n, 1), rand_fr(fid, n, 2), rand_fr(fid, n - 7, 3)
    ab = A.poly_mul_host(fname, a, b)
    assert ab.shape[0] == 2 * n - 8
    assert np.array_equal(ab, A.poly_mul(fname, a, b))
    s = O.field_op(fid, "add", a, a2).reshape(n, 4)
    lhs = A.poly_mul_host(fname, s, b)
    rhs = O.field_op(fid, "add", ab, A.poly_mul_host(fname, a2, b)).reshape(-1, 4)
    assert np.array_equal(lhs, rhs)
