"""Oracle radix-2 FFT vs Horner evaluation and the Python textbook DFT.
Mirrors poly/src/domain/radix2/mod.rs:351-391 (fft == evaluate at domain elements, subgroup + coset),
:430-536 (sizes 2^0..2^9 against a serial reference) and poly/src/test.rs:12-60 (ifft o fft = id)."""
import numpy as np
import pytest

import oracle_lib as O
import pyref as P

FR = ["BN254_FR", "BLS12_381_FR", "BLS12_377_FR"]


def rand_elems(field, n, seed):
    return O.gen_scalars(O.FID[field], seed, n, montgomery=True)


@pytest.mark.parametrize("field", FR)
def test_domain_constants(field):
    fid = O.FID[field]
    p = P.MODULI[field][0]
    for log_n in (0, 1, 5, 12, 22):
        g, gi, si = O.domain(fid, log_n)
        w = P.root_of_unity(field, log_n)
        assert P.from_mont(g, p) == w and P.from_mont(gi, p) == pow(w, -1, p)
        assert P.from_mont(si, p) == pow(1 << log_n, -1, p)
        assert pow(w, 1 << log_n, p) == 1 and (log_n == 0 or pow(w, 1 << (log_n - 1), p) == p - 1)
    s = P.two_adicity(p)[0]
    assert O.domain(fid, s) is not None
    assert O.domain(fid, s + 1) is None  # Radix2EvaluationDomain::new -> None (radix2/mod.rs:62-64)


@pytest.mark.parametrize("field", FR)
@pytest.mark.parametrize("log_n", [0, 1, 2, 3, 5, 6])
def test_fft_matches_python_dft(field, log_n):
    fid = O.FID[field]
    p, gen = P.MODULI[field]
    n = 1 << log_n
    x = rand_elems(field, n, 100 + log_n)
    ints = [P.from_mont(v, p) for v in x]
    enc = lambda vs: np.stack([P.to_mont(v, p) for v in vs]).reshape(-1)
    assert np.array_equal(O.fft(fid, x, log_n), enc(P.dft(field, ints, log_n)))
    assert np.array_equal(O.fft(fid, x, log_n, inverse=True), enc(P.dft(field, ints, log_n, inverse=True)))
    off = P.to_mont(gen, p)  # coset offset = F::GENERATOR as in poly/benches/fft.rs:107
    assert np.array_equal(O.fft(fid, x, log_n, offset=off), enc(P.dft(field, ints, log_n, offset=gen)))
    assert np.array_equal(O.fft(fid, x, log_n, offset=off, inverse=True), enc(P.dft(field, ints, log_n, offset=gen, inverse=True)))


@pytest.mark.parametrize("log_n", list(range(0, 11)))
def test_fft_vs_horner_and_roundtrip(log_n):
    field = "BLS12_381_FR"
    fid = O.FID[field]
    p, gen = P.MODULI[field]
    n = 1 << log_n
    x = rand_elems(field, n, 7 + log_n)
    off = P.to_mont(gen, p)
    for o in (None, off):
        y = O.fft(fid, x, log_n, offset=o, threads=2)
        assert np.array_equal(y, O.dft_naive(fid, x, log_n, offset=o))
        assert np.array_equal(O.fft(fid, y, log_n, offset=o, inverse=True, threads=2), x.reshape(-1))
    # zero-padded shorter input == evaluating the shorter polynomial (radix2/mod.rs:144 resize)
    if n >= 4:
        short = x[: n // 2 + 1]
        padded = np.zeros((n, 4), dtype=np.uint64)
        padded[: len(short)] = short
        assert np.array_equal(O.fft(fid, padded, log_n), O.dft_naive(fid, short, log_n))


def test_fft_linearity_and_threads_large():
    """2^14: threaded == serial, fft(a+b) == fft(a)+fft(b), roundtrip (size-independent properties)."""
    field = "BLS12_381_FR"
    fid = O.FID[field]
    log_n = 14
    n = 1 << log_n
    a, b = rand_elems(field, n, 1), rand_elems(field, n, 2)
    fa, fb = O.fft(fid, a, log_n), O.fft(fid, b, log_n)
    assert np.array_equal(O.fft(fid, a, log_n, threads=4), fa)
    s = O.field_op(fid, "add", a, b)
    assert np.array_equal(O.fft(fid, s, log_n), O.field_op(fid, "add", fa, fb))
    assert np.array_equal(O.fft(fid, fa, log_n, inverse=True, threads=4), a.reshape(-1))


def test_fft_ifft_identity_1_to_8():
    """radix2/mod.rs:581-600 test_fft_ifft_identity on [1..8]."""
    field = "BLS12_381_FR"
    fid = O.FID[field]
    p = P.MODULI[field][0]
    x = np.stack([P.to_mont(i, p) for i in range(1, 9)])
    y = O.fft(fid, x, 3)
    assert np.array_equal(O.fft(fid, y, 3, inverse=True), x.reshape(-1))
