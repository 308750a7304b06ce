/*
 * TEST INFRASTRUCTURE (oracle) -- not part of the product path.
 *
 * Prime-field arithmetic in Montgomery form, restated from the reference:
 *   ff/src/fields/models/fp/montgomery_backend.rs:129-171  add / sub / double / neg
 *   ff/src/fields/models/fp/montgomery_backend.rs:214-233  no-carry CIOS multiply
 *   ff/src/fields/models/fp/montgomery_backend.rs:380-412  from_bigint / into_bigint
 *   ff/src/fields/models/fp/mod.rs:134-151                  subtract_modulus / is_geq_modulus
 *   ff/src/biginteger/arithmetic.rs:6-113                   mac / mac_with_carry / adc / sbb
 * All six moduli of BASELINE.json's configs have a spare top bit, so the "no-carry" CIOS form
 * (CAN_USE_NO_CARRY_MUL_OPT, montgomery_backend.rs:63-77) is the branch the reference takes.
 *
 * "Template": include with NL defined to the limb count (4 or 6); emits fp<NL>_* functions.
 */
#ifndef NL
#error "define NL before including fp_tmpl.h"
#endif

#define FP_CAT_(a, b) a##b
#define FP_CAT(a, b) FP_CAT_(a, b)
#define FPN(name) FP_CAT(FP_CAT(FP_CAT(fp, NL), _), name)

/* ff/src/biginteger/arithmetic.rs:74-80  mac_with_carry: a + b*c + carry */
#ifndef ORACLE_MAC_DEFINED
#define ORACLE_MAC_DEFINED
static inline u64 mac_with_carry(u64 a, u64 b, u64 c, u64 *carry) {
    unsigned __int128 t = (unsigned __int128)a + (unsigned __int128)b * c + *carry;
    *carry = (u64)(t >> 64);
    return (u64)t;
}
static inline u64 adc64(u64 a, u64 b, u64 *carry) {
    unsigned __int128 t = (unsigned __int128)a + b + *carry;
    *carry = (u64)(t >> 64);
    return (u64)t;
}
static inline u64 sbb64(u64 a, u64 b, u64 *borrow) {
    unsigned __int128 t = (unsigned __int128)a - b - *borrow;
    *borrow = (u64)(t >> 64) & 1;
    return (u64)t;
}
#endif

static inline int FPN(is_zero)(const u64 *a) {
    u64 x = 0;
    for (int i = 0; i < NL; i++) x |= a[i];
    return x == 0;
}
static inline int FPN(eq)(const u64 *a, const u64 *b) {
    u64 x = 0;
    for (int i = 0; i < NL; i++) x |= a[i] ^ b[i];
    return x == 0;
}
static inline void FPN(copy)(u64 *r, const u64 *a) {
    for (int i = 0; i < NL; i++) r[i] = a[i];
}
/* a >= b as little-endian integers (biginteger/mod.rs Ord impl) */
static inline int FPN(geq)(const u64 *a, const u64 *b) {
    for (int i = NL - 1; i >= 0; i--) {
        if (a[i] > b[i]) return 1;
        if (a[i] < b[i]) return 0;
    }
    return 1;
}
/* fp/mod.rs:139-151 subtract_modulus */
static inline void FPN(subtract_modulus)(const ark_field_consts *F, u64 *a) {
    if (FPN(geq)(a, F->p)) {
        u64 borrow = 0;
        for (int i = 0; i < NL; i++) a[i] = sbb64(a[i], F->p[i], &borrow);
    }
}
/* montgomery_backend.rs:129-139 */
static inline void FPN(add)(const ark_field_consts *F, u64 *r, const u64 *a, const u64 *b) {
    u64 carry = 0;
    for (int i = 0; i < NL; i++) r[i] = adc64(a[i], b[i], &carry);
    FPN(subtract_modulus)(F, r); /* MODULUS_HAS_SPARE_BIT: no carry out possible */
}
/* montgomery_backend.rs:141-149 */
static inline void FPN(sub)(const ark_field_consts *F, u64 *r, const u64 *a, const u64 *b) {
    u64 t[NL];
    FPN(copy)(t, a);
    if (!FPN(geq)(a, b)) { /* b > a: add the modulus first */
        u64 carry = 0;
        for (int i = 0; i < NL; i++) t[i] = adc64(t[i], F->p[i], &carry);
    }
    u64 borrow = 0;
    for (int i = 0; i < NL; i++) r[i] = sbb64(t[i], b[i], &borrow);
}
/* montgomery_backend.rs:151-161 */
static inline void FPN(dbl)(const ark_field_consts *F, u64 *r, const u64 *a) { FPN(add)(F, r, a, a); }
/* montgomery_backend.rs:163-171 */
static inline void FPN(neg)(const ark_field_consts *F, u64 *r, const u64 *a) {
    if (FPN(is_zero)(a)) {
        FPN(copy)(r, a);
        return;
    }
    u64 borrow = 0;
    u64 t[NL];
    for (int i = 0; i < NL; i++) t[i] = sbb64(F->p[i], a[i], &borrow);
    FPN(copy)(r, t);
}
/* montgomery_backend.rs:214-233: CIOS with the no-carry optimisation, then subtract_modulus */
static inline void FPN(mul)(const ark_field_consts *F, u64 *out, const u64 *a, const u64 *b) {
    u64 r[NL];
    for (int i = 0; i < NL; i++) r[i] = 0;
    for (int i = 0; i < NL; i++) {
        u64 carry1 = 0;
        r[0] = mac_with_carry(r[0], a[0], b[i], &carry1);
        u64 k = r[0] * F->inv;
        u64 carry2 = 0;
        (void)mac_with_carry(r[0], k, F->p[0], &carry2); /* mac_discard */
        for (int j = 1; j < NL; j++) {
            r[j] = mac_with_carry(r[j], a[j], b[i], &carry1);
            r[j - 1] = mac_with_carry(r[j], k, F->p[j], &carry2);
        }
        r[NL - 1] = carry1 + carry2;
    }
    FPN(subtract_modulus)(F, r);
    FPN(copy)(out, r);
}
static inline void FPN(sqr)(const ark_field_consts *F, u64 *r, const u64 *a) { FPN(mul)(F, r, a, a); }
/* montgomery_backend.rs:396-412 into_bigint: Montgomery reduction of (a, 0) */
static inline void FPN(into_bigint)(const ark_field_consts *F, u64 *out, const u64 *a) {
    u64 r[NL];
    FPN(copy)(r, a);
    for (int i = 0; i < NL; i++) {
        u64 k = r[i] * F->inv;
        u64 carry = 0;
        (void)mac_with_carry(r[i], k, F->p[0], &carry);
        for (int j = 1; j < NL; j++) r[(j + i) % NL] = mac_with_carry(r[(j + i) % NL], k, F->p[j], &carry);
        r[i] = carry;
    }
    FPN(copy)(out, r);
}
/* montgomery_backend.rs:380-391 from_bigint: r * R2 (caller guarantees r < p) */
static inline void FPN(from_bigint)(const ark_field_consts *F, u64 *out, const u64 *a) {
    if (FPN(is_zero)(a)) {
        FPN(copy)(out, a);
        return;
    }
    FPN(mul)(F, out, a, F->r2);
}
/* a^e for a little-endian exponent of ne limbs (ff/src/fields/mod.rs pow: square-and-multiply) */
static inline void FPN(pow)(const ark_field_consts *F, u64 *out, const u64 *a, const u64 *e, int ne) {
    u64 acc[NL], base[NL];
    FPN(copy)(acc, F->r); /* ONE = R */
    FPN(copy)(base, a);
    int started = 0;
    for (int i = ne * 64 - 1; i >= 0; i--) {
        int bit = (e[i / 64] >> (i % 64)) & 1;
        if (started) FPN(sqr)(F, acc, acc);
        if (bit) {
            FPN(mul)(F, acc, acc, base);
            started = 1;
        }
    }
    FPN(copy)(out, acc);
}
/* Inverse. The reference uses a binary extended Euclid (montgomery_backend.rs:319-378); the inverse
 * of a non-zero element is unique, so Fermat's a^(p-2) yields the identical canonical limbs.
 * Returns 0 for a == 0 (reference: None). */
static inline int FPN(inv)(const ark_field_consts *F, u64 *out, const u64 *a) {
    if (FPN(is_zero)(a)) return 0;
    u64 e[NL];
    u64 borrow = 0;
    for (int i = 0; i < NL; i++) e[i] = sbb64(F->p[i], i == 0 ? 2 : 0, &borrow);
    FPN(pow)(F, out, a, e, NL);
    return 1;
}

#undef FPN
