"""Field / curve ids and element sizes of the C ABI (include/ark_hip.h), named as in the reference's
curve crates (curves/bn254, curves/bls12_381, curves/bls12_377)."""

FIELDS = ["BN254_FQ", "BN254_FR", "BLS12_381_FQ", "BLS12_381_FR", "BLS12_377_FQ", "BLS12_377_FR"]
CURVES = ["BN254_G1", "BLS12_381_G1", "BLS12_377_G1", "BLS12_377_G2", "BLS12_381_G2"]
FIELD_ID = {n: i for i, n in enumerate(FIELDS)}
CURVE_ID = {n: i for i, n in enumerate(CURVES)}

# u64 words per base-field element, scalar field, base field, extension degree
CURVE_INFO = {
    "BN254_G1": (4, "BN254_FR", "BN254_FQ", 1),
    "BLS12_381_G1": (6, "BLS12_381_FR", "BLS12_381_FQ", 1),
    "BLS12_377_G1": (6, "BLS12_377_FR", "BLS12_377_FQ", 1),
    "BLS12_377_G2": (12, "BLS12_377_FR", "BLS12_377_FQ", 2),
    "BLS12_381_G2": (12, "BLS12_381_FR", "BLS12_381_FQ", 2),
}
SCALAR_WORDS = 4  # BigInt<4> / Fr

# scalar-field moduli (curves/bn254/src/fields/fr.rs:4-5, curves/bls12_381/src/fields/fr.rs:4-5,
# curves/bls12_377/src/fields/fr.rs:24-25): used only for host-side bookkeeping (HashMapPippenger's Fr additions)
SCALAR_MODULUS = {
    "BN254_FR": 21888242871839275222246405745257275088548364400416034343698204186575808495617,
    "BLS12_381_FR": 52435875175126190479447740508185965837690552500527637822603658699938581184513,
    "BLS12_377_FR": 8444461749428370424248824938781546531375899335154063827935233455917409239041,
}


def curve_id(curve):
    return curve if isinstance(curve, int) else CURVE_ID[curve]


def curve_name(curve):
    return CURVES[curve] if isinstance(curve, int) else curve


def field_id(field):
    return field if isinstance(field, int) else FIELD_ID[field]


def fe_words(curve):
    return CURVE_INFO[curve_name(curve)][0]


def affine_words(curve):
    return 2 * fe_words(curve)


def projective_words(curve):
    return 3 * fe_words(curve)


def affine_bytes(curve):
    return 8 * affine_words(curve)


def scalar_field(curve):
    return CURVE_INFO[curve_name(curve)][1]
