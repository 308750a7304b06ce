"""Pure-Python big-integer reference, independent of the C oracle and of the HIP code.

Used only for small cases: it re-derives field constants from the reference's decimal literals,
implements textbook affine short-Weierstrass arithmetic over Fp / Fp2 and a textbook DFT, and
converts between integers and the reference's Montgomery little-endian u64 limb layout.
"""
import numpy as np

MODULI = {
    # curves/bn254/src/fields/{fq,fr}.rs:4-5
    "BN254_FQ": (21888242871839275222246405745257275088696311157297823662689037894645226208583, 3),
    "BN254_FR": (21888242871839275222246405745257275088548364400416034343698204186575808495617, 5),
    # curves/bls12_381/src/fields/{fq,fr}.rs:4-5
    "BLS12_381_FQ": (4002409555221667393417789825735904156556882819939007885332058136124031650490837864442687629129015664037894272559787, 2),
    "BLS12_381_FR": (52435875175126190479447740508185965837690552500527637822603658699938581184513, 7),
    # curves/bls12_377/src/fields/fq.rs:4-5, fr.rs:24-25
    "BLS12_377_FQ": (258664426012969094010652733694893533536393512754914660539884262666720468348340822774968888139573360124440321458177, 15),
    "BLS12_377_FR": (8444461749428370424248824938781546531375899335154063827935233455917409239041, 22),
}
FIELD_ORDER = ["BN254_FQ", "BN254_FR", "BLS12_381_FQ", "BLS12_381_FR", "BLS12_377_FQ", "BLS12_377_FR"]

# curve -> (base field, scalar field, ext degree, beta, b)
CURVE_PARAMS = {
    "BN254_G1": ("BN254_FQ", "BN254_FR", 1, None, 3),
    "BLS12_381_G1": ("BLS12_381_FQ", "BLS12_381_FR", 1, None, 4),
    "BLS12_377_G1": ("BLS12_377_FQ", "BLS12_377_FR", 1, None, 1),
    "BLS12_377_G2": ("BLS12_377_FQ", "BLS12_377_FR", 2, -5,
                     (0, 155198655607781456406391640216936120121836107652948796323930557600032281009004493664981332883744016074664192874906)),
    "BLS12_381_G2": ("BLS12_381_FQ", "BLS12_381_FR", 2, -1, (4, 4)),
}
CURVE_ORDER = ["BN254_G1", "BLS12_381_G1", "BLS12_377_G1", "BLS12_377_G2", "BLS12_381_G2"]


def nlimbs(p):
    return (p.bit_length() + 63) // 64


def R_of(p):
    return (1 << (64 * nlimbs(p))) % p


def to_limbs(x, n):
    return np.array([(x >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(n)], dtype=np.uint64)


def from_limbs(a):
    x = 0
    for i, v in enumerate(np.asarray(a).reshape(-1).tolist()):
        x |= int(v) << (64 * i)
    return x


def to_mont(x, p):
    return to_limbs(x * R_of(p) % p, nlimbs(p))


def from_mont(a, p):
    return from_limbs(a) * pow(R_of(p), -1, p) % p


# ---- generic field element helpers: Fp elements are ints, Fp2 elements are (c0, c1) tuples ----
class Fld:
    def __init__(self, p, beta=None):
        self.p, self.beta = p, beta

    def add(self, a, b):
        if self.beta is None:
            return (a + b) % self.p
        return ((a[0] + b[0]) % self.p, (a[1] + b[1]) % self.p)

    def sub(self, a, b):
        if self.beta is None:
            return (a - b) % self.p
        return ((a[0] - b[0]) % self.p, (a[1] - b[1]) % self.p)

    def mul(self, a, b):
        p = self.p
        if self.beta is None:
            return a * b % p
        return ((a[0] * b[0] + self.beta * a[1] * b[1]) % p, (a[0] * b[1] + a[1] * b[0]) % p)

    def inv(self, a):
        p = self.p
        if self.beta is None:
            return pow(a, -1, p)
        n = pow((a[0] * a[0] - self.beta * a[1] * a[1]) % p, -1, p)
        return (a[0] * n % p, (-a[1] * n) % p)

    def neg(self, a):
        if self.beta is None:
            return (-a) % self.p
        return ((-a[0]) % self.p, (-a[1]) % self.p)

    def zero(self):
        return 0 if self.beta is None else (0, 0)

    def from_int(self, v):
        if self.beta is None:
            return v % self.p
        if isinstance(v, tuple):
            return (v[0] % self.p, v[1] % self.p)
        return (v % self.p, 0)

    def enc(self, a):
        """element -> Montgomery limbs (c0|c1 for Fp2)"""
        if self.beta is None:
            return to_mont(a, self.p)
        return np.concatenate([to_mont(a[0], self.p), to_mont(a[1], self.p)])

    def dec(self, limbs):
        n = nlimbs(self.p)
        limbs = np.asarray(limbs).reshape(-1)
        if self.beta is None:
            return from_mont(limbs[:n], self.p)
        return (from_mont(limbs[:n], self.p), from_mont(limbs[n:2 * n], self.p))


class Curve:
    """y^2 = x^3 + b over Fp or Fp2; points are None (identity) or (x, y)."""

    def __init__(self, name):
        bf, sf, ext, beta, b = CURVE_PARAMS[name]
        self.name = name
        self.p = MODULI[bf][0]
        self.r = MODULI[sf][0]
        self.F = Fld(self.p, beta)
        self.b = self.F.from_int(b)
        self.fw = nlimbs(self.p) * ext

    def on_curve(self, P):
        if P is None:
            return True
        F = self.F
        x, y = P
        return F.mul(y, y) == F.add(F.mul(F.mul(x, x), x), self.b)

    def add(self, P, Q):
        F = self.F
        if P is None:
            return Q
        if Q is None:
            return P
        x1, y1 = P
        x2, y2 = Q
        if x1 == x2:
            if F.add(y1, y2) == F.zero():
                return None
            lam = F.mul(F.mul(F.from_int(3), F.mul(x1, x1)), F.inv(F.add(y1, y1)))
        else:
            lam = F.mul(F.sub(y2, y1), F.inv(F.sub(x2, x1)))
        x3 = F.sub(F.sub(F.mul(lam, lam), x1), x2)
        y3 = F.sub(F.mul(lam, F.sub(x1, x3)), y1)
        return (x3, y3)

    def neg(self, P):
        return None if P is None else (P[0], self.F.neg(P[1]))

    def mul(self, P, k):
        k %= self.r
        acc = None
        for bit in bin(k)[2:] if k else "":
            acc = self.add(acc, acc)
            if bit == "1":
                acc = self.add(acc, P)
        return acc

    def msm(self, pts, scalars):
        acc = None
        for P, s in zip(pts, scalars):
            acc = self.add(acc, self.mul(P, s))
        return acc

    def enc(self, P):
        """affine point -> reference layout x|y Montgomery limbs, identity = zeros"""
        if P is None:
            return np.zeros(2 * self.fw, dtype=np.uint64)
        return np.concatenate([self.F.enc(P[0]), self.F.enc(P[1])])

    def dec(self, limbs):
        limbs = np.asarray(limbs).reshape(-1)
        if not limbs.any():
            return None
        return (self.F.dec(limbs[:self.fw]), self.F.dec(limbs[self.fw:]))


def two_adicity(p):
    s, t = 0, p - 1
    while t % 2 == 0:
        t //= 2
        s += 1
    return s, t


def root_of_unity(field, log_n):
    """F::get_root_of_unity(2^log_n) per ff/src/fields/fft_friendly.rs:67-81"""
    p, g = MODULI[field]
    s, t = two_adicity(p)
    assert log_n <= s
    w = pow(g, t, p)
    for _ in range(s - log_n):
        w = w * w % p
    return w


def dft(field, coeffs, log_n, offset=1, inverse=False):
    """textbook O(n^2) evaluation / interpolation contract of SURVEY.md section 3 (mathematical contract)"""
    p = MODULI[field][0]
    n = 1 << log_n
    g = root_of_unity(field, log_n)
    c = list(coeffs) + [0] * (n - len(coeffs))
    if not inverse:
        out = []
        for k in range(n):
            x = offset * pow(g, k, p) % p
            acc = 0
            for a in reversed(c):
                acc = (acc * x + a) % p
            out.append(acc)
        return out
    gi, ni, oi = pow(g, -1, p), pow(n, -1, p), pow(offset, -1, p)
    out = []
    for j in range(n):
        acc = 0
        for k in range(n):
            acc = (acc + c[k] * pow(gi, j * k, p)) % p
        out.append(acc * ni % p * pow(oi, j, p) % p)
    return out
