"""Synthetic MSM inputs with an exactly known answer, shared by bench.py, tools/ and tests/.

Bases P_i = (a + i*b)G are grown on the GPU from the generator (P[i + m] = P[i] + (m b)G), scalars are uniform in
[0, r) by top-limb-masked rejection (ff/src/fields/models/fp/mod.rs:525-547 style).  The MSM of such inputs is
k*G with k = sum_i s_i (a + i b) mod r -- an exact big-integer identity, so a result of any size can be checked
bit for bit with one small scalar multiplication.  Everything here goes through the product's C ABI only."""
import ctypes as C

import numpy as np

R = {"BN254_FR": 21888242871839275222246405745257275088548364400416034343698204186575808495617,
     "BLS12_381_FR": 52435875175126190479447740508185965837690552500527637822603658699938581184513,
     "BLS12_377_FR": 8444461749428370424248824938781546531375899335154063827935233455917409239041}
A0 = 0xA11CE + (1 << 64) + (2 << 128)
B0 = 0xB0B + (3 << 64)


def limbs4(v):
    return np.array([(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)


def scalar_int(row):
    """one BigInt<4> row as a Python integer."""
    return sum(int(row[i]) << (64 * i) for i in range(4))


def gen_scalars(n, seed, r):
    """uniform in [0, r) as canonical BigInt<4> limbs [n, 4]."""
    bits = r.bit_length()
    top_mask = np.uint64((1 << (bits - 192)) - 1)
    rng = np.random.default_rng(seed)
    rl = [int(x) for x in limbs4(r)]

    def draw(m):
        a = rng.integers(0, 1 << 64, size=(m, 4), dtype=np.uint64)
        a[:, 3] &= top_mask
        return a

    def ge_r(a):
        ge = np.ones(a.shape[0], dtype=bool)   # equal so far -> counts as >=
        decided = np.zeros(a.shape[0], dtype=bool)
        for k in (3, 2, 1, 0):
            gt = a[:, k] > np.uint64(rl[k])
            lt = a[:, k] < np.uint64(rl[k])
            ge = np.where(~decided & lt, False, ge)
            decided |= gt | lt
        return ge

    out = draw(n)
    bad = np.nonzero(ge_r(out))[0]
    while bad.size:
        new = draw(bad.size)
        out[bad] = new
        bad = bad[ge_r(new)]
    return out


def _exact_sum(x, chunk=32):
    """exact integer sum of a uint64 array whose entries are < 2^63 / chunk (a chunk's sum stays < 2^63)."""
    pad = (-x.size) % chunk
    if pad:
        x = np.concatenate([x, np.zeros(pad, dtype=np.uint64)])
    return int(x.reshape(-1, chunk).sum(axis=1, dtype=np.uint64).astype(object).sum())


def dlog_of_msm(scalars, a, b, r, first_index=0):
    """k = sum_i s_i (a + (first_index + i) b) mod r, exact: the discrete log of the MSM of P_i = (a + i b)G."""
    n = scalars.shape[0]
    assert n <= 1 << 29
    idx = np.arange(n, dtype=np.uint64)  # idx * 32-bit half < 2^61: summed in chunks of 4 (2^28 pairs overflowed chunks of 32)
    s_sum = 0
    is_sum = 0
    for k in range(4):
        for half, sh in ((scalars[:, k] & np.uint64(0xFFFFFFFF), 0), (scalars[:, k] >> np.uint64(32), 32)):
            w = 1 << (64 * k + sh)
            s_sum += w * _exact_sum(half)
            is_sum += w * _exact_sum(half * idx, 32 if n <= (1 << 26) else 4)
    return (s_sum * (a + first_index * b) + is_sum * b) % r


def mul_gen(cid, k, r):
    """k*G as affine limbs, via a 1-point MSM on the GPU."""
    import algebra_amd as A
    from algebra_amd import curves as cv
    from algebra_amd._lib import check, lib
    gen = np.zeros(cv.affine_words(cid), dtype=np.uint64)
    check(lib().ark_hip_curve_generator(cid, gen.ctypes.data_as(C.c_void_p)), "generator")
    return A.into_affine(cid, A.msm_bigint(cid, gen.reshape(1, -1), limbs4(k % r).reshape(1, 4)))


def grow_bases(cid, n, a, b, r):
    """P_i = (a + i b)G for i < n as a CUDA uint8 tensor in the reference's Affine layout."""
    import torch
    from algebra_amd import curves as cv
    from algebra_amd._lib import check, lib
    L = lib()
    ab = cv.affine_bytes(cid)
    bases = torch.zeros(n * ab, dtype=torch.uint8, device="cuda")
    bases[:ab] = torch.from_numpy(mul_gen(cid, a, r).view(np.uint8)).cuda()
    torch.cuda.synchronize()
    m = 1
    while m < n:
        cnt = min(m, n - m)
        d = np.ascontiguousarray(mul_gen(cid, m * b, r))
        check(L.ark_hip_sw_add_affine_device(cid, bases.data_ptr(), bases.data_ptr() + m * ab, cnt,
                                             d.ctypes.data_as(C.c_void_p)), "sw_add_affine_device")
        m += cnt
    return bases
