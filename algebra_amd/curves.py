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
