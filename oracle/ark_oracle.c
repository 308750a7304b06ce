/*
 * TEST INFRASTRUCTURE (oracle) -- CPU restatement of arkworks-rs/algebra's MSM + radix-2 FFT hot
 * path in plain C. Parity pinned by tests/test_oracle_golden.py (reference KAT tables).
 * Never linked into or loaded by the product library. See ark_oracle.h for the API and
 * fp_tmpl.h / ec_tmpl.h for per-function reference citations.
 *
 * FFT restates (all under /root/reference/poly/src/domain):
 *   radix2/mod.rs:55-83     Radix2EvaluationDomain::new (group_gen, size_inv)
 *   radix2/fft.rs:74-88     in_order_fft_in_place / in_order_ifft_in_place
 *   radix2/fft.rs:90-119    fft_helper_in_place / ifft_helper_in_place
 *   radix2/fft.rs:124-129   roots_of_unity (serial: utils.rs:34-51 compute_powers_serial)
 *   radix2/fft.rs:190-210   butterfly_fn_io / butterfly_fn_oi
 *   radix2/fft.rs:252-349   io_helper / oi_helper (incl. roots compaction)
 *   radix2/fft.rs:368-380   bitrev / derange
 *   mod.rs:115-128          distribute_powers(_and_mul_by_const)
 *   ff/src/fields/fft_friendly.rs:67-81  get_root_of_unity
 */
#include "ark_oracle.h"

#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef uint64_t u64;
#include "constants.h"

#define SCALAR_LIMBS 4

typedef struct {
    const ark_field_consts *F; /* base prime field */
    const ark_field_consts *S; /* scalar field */
    int ext, beta_abs;
    const u64 *b, *gx, *gy;
} ark_curve_ctx;

/* ---------- prime fields, N = 4 and N = 6 ---------- */
#define NL 4
#include "fp_tmpl.h"
#undef NL
#define NL 6
#include "fp_tmpl.h"
#undef NL

/* ---------- bigint helpers (ff/src/biginteger/mod.rs:336-360 num_bits, is_zero) ---------- */
static inline int bigint_is_zero(const u64 *a, int n) {
    u64 x = 0;
    for (int i = 0; i < n; i++) x |= a[i];
    return x == 0;
}
static inline int bigint_num_bits(const u64 *a, int n) {
    for (int i = n - 1; i >= 0; i--)
        if (a[i]) return 64 * i + (64 - __builtin_clzll(a[i]));
    return 0;
}
/* ark_std::log2 = ceil(log2(x)) (ark-std 0.6, un-vendored; call site ec/src/scalar_mul/mod.rs:24) */
static inline unsigned ark_log2(size_t x) {
    if (x <= 1) return 0;
    return 64 - (unsigned)__builtin_clzll((u64)x - 1);
}
/* ec/src/scalar_mul/mod.rs:22-25 */
static inline size_t ln_without_floats(size_t a) { return (size_t)ark_log2(a) * 69 / 100; }

/* ec/src/scalar_mul/variable_base/mod.rs:754-794 make_digits */
static int make_digits(const u64 *scalar, int nlimbs, int w, int num_bits, int64_t *out) {
    u64 radix = (u64)1 << w, window_mask = radix - 1, carry = 0;
    if (num_bits == 0) num_bits = bigint_num_bits(scalar, nlimbs);
    int digits_count = (num_bits + w - 1) / w;
    for (int i = 0; i < digits_count; i++) {
        int bit_offset = i * w, u64_idx = bit_offset / 64, bit_idx = bit_offset % 64;
        u64 bit_buf;
        if (bit_idx < 64 - w || u64_idx == nlimbs - 1)
            bit_buf = scalar[u64_idx] >> bit_idx;
        else
            bit_buf = (scalar[u64_idx] >> bit_idx) | (scalar[1 + u64_idx] << (64 - bit_idx));
        u64 coef = carry + (bit_buf & window_mask);
        carry = (coef + radix / 2) >> w;
        int64_t digit = (int64_t)coef - (int64_t)(carry << w);
        if (i == digits_count - 1) digit += (int64_t)(carry << w);
        out[i] = digit;
    }
    return digits_count;
}

/* ---------- base-field op tables: Fp (N=4), Fp (N=6), Fp2 over N=6 ---------- */
#define DEF_FP_WRAP(N)                                                                                                  \
    static inline void f##N##_add(const ark_curve_ctx *C, u64 *r, const u64 *a, const u64 *b) { fp##N##_add(C->F, r, a, b); } \
    static inline void f##N##_sub(const ark_curve_ctx *C, u64 *r, const u64 *a, const u64 *b) { fp##N##_sub(C->F, r, a, b); } \
    static inline void f##N##_mul(const ark_curve_ctx *C, u64 *r, const u64 *a, const u64 *b) { fp##N##_mul(C->F, r, a, b); } \
    static inline void f##N##_neg(const ark_curve_ctx *C, u64 *r, const u64 *a, const u64 *b) { (void)b; fp##N##_neg(C->F, r, a); } \
    static inline void f##N##_inv(const ark_curve_ctx *C, u64 *r, const u64 *a, const u64 *b) { (void)b; fp##N##_inv(C->F, r, a); }
DEF_FP_WRAP(4)
DEF_FP_WRAP(6)

/* Fp2 = Fp[u]/(u^2 - beta), beta = -beta_abs (quadratic_extension.rs:268-320, 626-670; fp2.rs:6-52).
 * Elements are canonical, so any correct formula yields the reference's limbs. */
static inline void fp6_mul_by_beta(const ark_curve_ctx *C, u64 *r, const u64 *a) { /* r = beta * a */
    u64 t[6], acc[6];
    fp6_copy(acc, a);
    for (int k = 1; k < C->beta_abs; k++) {
        fp6_add(C->F, t, acc, a);
        fp6_copy(acc, t);
    }
    fp6_neg(C->F, r, acc);
}
static inline void f12_add(const ark_curve_ctx *C, u64 *r, const u64 *a, const u64 *b) {
    fp6_add(C->F, r, a, b);
    fp6_add(C->F, r + 6, a + 6, b + 6);
}
static inline void f12_sub(const ark_curve_ctx *C, u64 *r, const u64 *a, const u64 *b) {
    fp6_sub(C->F, r, a, b);
    fp6_sub(C->F, r + 6, a + 6, b + 6);
}
static inline void f12_neg(const ark_curve_ctx *C, u64 *r, const u64 *a, const u64 *b) {
    (void)b;
    fp6_neg(C->F, r, a);
    fp6_neg(C->F, r + 6, a + 6);
}
/* quadratic_extension.rs:646-654: c0 = a0*b0 + beta*a1*b1 ; c1 = a0*b1 + a1*b0 */
static inline void f12_mul(const ark_curve_ctx *C, u64 *r, const u64 *a, const u64 *b) {
    u64 t0[6], t1[6], t2[6], c0[6], c1[6];
    fp6_mul(C->F, t0, a, b);
    fp6_mul(C->F, t1, a + 6, b + 6);
    fp6_mul_by_beta(C, t2, t1);
    fp6_add(C->F, c0, t0, t2);
    fp6_mul(C->F, t0, a, b + 6);
    fp6_mul(C->F, t1, a + 6, b);
    fp6_add(C->F, c1, t0, t1);
    fp6_copy(r, c0);
    fp6_copy(r + 6, c1);
}
/* quadratic_extension.rs:322-339 (Guide to Pairing-based Cryptography, Alg. 5.19) */
static inline void f12_inv(const ark_curve_ctx *C, u64 *r, const u64 *a, const u64 *b) {
    (void)b;
    u64 v0[6], v1[6], t[6];
    fp6_sqr(C->F, v1, a + 6);
    fp6_mul_by_beta(C, t, v1);
    fp6_sqr(C->F, v0, a);
    fp6_sub(C->F, v0, v0, t); /* c0^2 - beta*c1^2 */
    fp6_inv(C->F, v1, v0);
    fp6_mul(C->F, r, a, v1);
    fp6_mul(C->F, t, a + 6, v1);
    fp6_neg(C->F, r + 6, t);
}

#define EC g4
#define FW 4
#define FE(op) f4_##op
#include "ec_tmpl.h"
#undef EC
#undef FW
#undef FE
#define EC g6
#define FW 6
#define FE(op) f6_##op
#include "ec_tmpl.h"
#undef EC
#undef FW
#undef FE
#define EC g12
#define FW 12
#define FE(op) f12_##op
#include "ec_tmpl.h"
#undef EC
#undef FW
#undef FE

/* ---------- contexts ---------- */
#define NCURVES 5
#define NFIELDS 6
static int get_curve(int curve, ark_curve_ctx *C) {
    if (curve < 0 || curve >= NCURVES) return -1;
    const ark_curve_consts *cc = &ARK_CURVES[curve];
    C->F = &ARK_FIELDS[cc->base_field];
    C->S = &ARK_FIELDS[cc->scalar_field];
    C->ext = cc->ext_degree;
    C->beta_abs = cc->beta_abs;
    C->b = cc->b;
    C->gx = cc->gx;
    C->gy = cc->gy;
    return C->F->n * C->ext; /* FW */
}
/* the constants tables store Fp2 as 6+6 words but padded layout for N=4 would differ; ext curves are N=6 only */

int ark_oracle_field_limbs(int field) { return (field < 0 || field >= NFIELDS) ? -1 : ARK_FIELDS[field].n; }
int ark_oracle_curve_fe_words(int curve) {
    ark_curve_ctx C;
    return get_curve(curve, &C);
}
int ark_oracle_curve_info(int curve, int *base_field, int *scalar_field, int *ext_degree) {
    if (curve < 0 || curve >= NCURVES) return -1;
    *base_field = ARK_CURVES[curve].base_field;
    *scalar_field = ARK_CURVES[curve].scalar_field;
    *ext_degree = ARK_CURVES[curve].ext_degree;
    return 0;
}
int ark_oracle_field_const(int field, int which, u64 *out) {
    if (field < 0 || field >= NFIELDS) return -1;
    const ark_field_consts *F = &ARK_FIELDS[field];
    const u64 *src = which == 0 ? F->p : which == 1 ? F->r : which == 2 ? F->r2 : which == 3 ? F->gen : F->root;
    memcpy(out, src, F->n * 8);
    return F->n;
}
int ark_oracle_curve_generator(int curve, u64 *out_xy) {
    ark_curve_ctx C;
    int fw = get_curve(curve, &C);
    if (fw < 0) return -1;
    memcpy(out_xy, C.gx, fw * 8);
    memcpy(out_xy + fw, C.gy, fw * 8);
    return 0;
}

int ark_oracle_field_op(int field, int op, const u64 *a, const u64 *b, u64 *r, size_t n) {
    if (field < 0 || field >= NFIELDS) return -1;
    const ark_field_consts *F = &ARK_FIELDS[field];
    int N = F->n;
    for (size_t i = 0; i < n; i++) {
        const u64 *x = a + i * N, *y = b ? b + i * N : NULL;
        u64 *o = r + i * N;
#define DISPATCH(NN)                                              \
    switch (op) {                                                 \
    case 0: fp##NN##_add(F, o, x, y); break;                      \
    case 1: fp##NN##_sub(F, o, x, y); break;                      \
    case 2: fp##NN##_mul(F, o, x, y); break;                      \
    case 3: fp##NN##_sqr(F, o, x); break;                         \
    case 4: fp##NN##_neg(F, o, x); break;                         \
    case 5: fp##NN##_dbl(F, o, x); break;                         \
    case 6: if (!fp##NN##_inv(F, o, x)) memset(o, 0, NN * 8); break; \
    case 7: fp##NN##_into_bigint(F, o, x); break;                 \
    case 8: fp##NN##_from_bigint(F, o, x); break;                 \
    default: return -2;                                           \
    }
        if (N == 4) {
            DISPATCH(4)
        } else {
            DISPATCH(6)
        }
#undef DISPATCH
    }
    return 0;
}

int ark_oracle_basefield_op(int curve, int op, const u64 *a, const u64 *b, u64 *r, size_t n) {
    ark_curve_ctx C;
    int fw = get_curve(curve, &C);
    if (fw < 0) return -1;
    for (size_t i = 0; i < n; i++) {
        const u64 *x = a + i * fw, *y = b ? b + i * fw : x;
        u64 *o = r + i * fw;
#define DISPATCH(P)                               \
    switch (op) {                                 \
    case 0: P##_add(&C, o, x, y); break;          \
    case 1: P##_sub(&C, o, x, y); break;          \
    case 2: P##_mul(&C, o, x, y); break;          \
    case 3: P##_mul(&C, o, x, x); break;          \
    case 4: P##_neg(&C, o, x, x); break;          \
    case 5: P##_add(&C, o, x, x); break;          \
    case 6: P##_inv(&C, o, x, x); break;          \
    default: return -2;                           \
    }
        if (fw == 4) {
            DISPATCH(f4)
        } else if (fw == 6) {
            DISPATCH(f6)
        } else {
            DISPATCH(f12)
        }
#undef DISPATCH
    }
    return 0;
}

#define BY_FW(fw, CALL4, CALL6, CALL12) \
    do {                                \
        if ((fw) == 4) {                \
            CALL4;                      \
        } else if ((fw) == 6) {         \
            CALL6;                      \
        } else {                        \
            CALL12;                     \
        }                               \
    } while (0)

int ark_oracle_point_op(int curve, int kind, u64 *acc, const u64 *other) {
    ark_curve_ctx C;
    int fw = get_curve(curve, &C);
    if (fw < 0) return -1;
#define PO(P)                                                   \
    switch (kind) {                                             \
    case 0: P##_jac_add(&C, acc, other); break;                 \
    case 1: P##_jac_double(&C, acc); break;                     \
    case 2: P##_bkt_add_affine(&C, acc, other, 0); break;       \
    case 3: P##_bkt_add_affine(&C, acc, other, 1); break;       \
    case 4: P##_bkt_add_bkt(&C, acc, other); break;             \
    case 5: P##_bkt_double(&C, acc); break;                     \
    case 6: P##_bkt_to_jac(&C, acc, other); break;              \
    case 7: P##_aff_double_to_bucket(&C, acc, other); break;    \
    default: return -2;                                         \
    }
    BY_FW(fw, PO(g4), PO(g6), PO(g12));
#undef PO
    return 0;
}

int ark_oracle_to_affine(int curve, const u64 *jac, u64 *out_xy, size_t n) {
    ark_curve_ctx C;
    int fw = get_curve(curve, &C);
    if (fw < 0) return -1;
    for (size_t i = 0; i < n; i++)
        BY_FW(fw, g4_jac_to_aff(&C, out_xy + i * 8, jac + i * 12), g6_jac_to_aff(&C, out_xy + i * 12, jac + i * 18),
              g12_jac_to_aff(&C, out_xy + i * 24, jac + i * 36));
    return 0;
}

int ark_oracle_scalar_mul(int curve, const u64 *base_xy, const u64 *scalar4, u64 *out_jac) {
    ark_curve_ctx C;
    int fw = get_curve(curve, &C);
    if (fw < 0) return -1;
    BY_FW(fw, g4_scalar_mul(&C, out_jac, base_xy, scalar4, 4), g6_scalar_mul(&C, out_jac, base_xy, scalar4, 4),
          g12_scalar_mul(&C, out_jac, base_xy, scalar4, 4));
    return 0;
}

/* ScalarMul::batch_mul (ec/src/scalar_mul/mod.rs:104-107): out[i] = scalars[i] * base, affine */
int ark_oracle_batch_mul(int curve, const u64 *base_jac, const u64 *scalars, size_t n, u64 *out_xy) {
    ark_curve_ctx C;
    int fw = get_curve(curve, &C);
    if (fw < 0) return -1;
    const int bits = ARK_FIELDS[ARK_CURVES[curve].scalar_field].bits;
    BY_FW(fw, g4_batch_mul(&C, out_xy, base_jac, scalars, n, bits), g6_batch_mul(&C, out_xy, base_jac, scalars, n, bits),
          g12_batch_mul(&C, out_xy, base_jac, scalars, n, bits));
    return 0;
}

/* y^2 == x^3 + b (a = 0); identity (0,0) counts as on-curve  (affine.rs is_on_curve) */
int ark_oracle_is_on_curve(int curve, const u64 *xy) {
    ark_curve_ctx C;
    int fw = get_curve(curve, &C);
    if (fw < 0) return -1;
    u64 l[12], r[12];
    u64 acc = 0;
    for (int i = 0; i < 2 * fw; i++) acc |= xy[i];
    if (!acc) return 1;
#define OC(P)                          \
    P##_mul(&C, l, xy + fw, xy + fw);  \
    P##_mul(&C, r, xy, xy);            \
    P##_mul(&C, r, r, xy);             \
    P##_add(&C, r, r, C.b);
    BY_FW(fw, OC(f4), OC(f6), OC(f12));
#undef OC
    return memcmp(l, r, fw * 8) == 0;
}

static int msm_dispatch(const ark_curve_ctx *C, int fw, const u64 *bases, const u64 *scalars, size_t n, int variant,
                        int threads, u64 *out) {
    if (threads < 1) threads = 1;
#define MS(P)                                                              \
    switch (variant) {                                                     \
    case 0: P##_msm_naive(C, out, bases, scalars, n); break;               \
    case 1: P##_msm_wnaf(C, out, bases, scalars, n, threads); break;       \
    case 2: P##_msm_signed(C, out, bases, scalars, n, threads); break;     \
    default: return -2;                                                    \
    }
    BY_FW(fw, MS(g4), MS(g6), MS(g12));
#undef MS
    return 0;
}

int ark_oracle_msm(int curve, const u64 *bases, const u64 *scalars, size_t n, int variant, int threads, u64 *out_jac) {
    ark_curve_ctx C;
    int fw = get_curve(curve, &C);
    if (fw < 0) return -1;
#ifdef _OPENMP
    omp_set_max_active_levels(2);
#endif
    return msm_dispatch(&C, fw, bases, scalars, n, variant, threads, out_jac);
}

/* VariableBaseMSM::msm_unchecked (variable_base/mod.rs:59-64): into_bigint every scalar, then msm_bigint */
int ark_oracle_msm_fr(int curve, const u64 *bases, const u64 *scalars_mont, size_t n, int variant, int threads,
                      u64 *out_jac) {
    ark_curve_ctx C;
    int fw = get_curve(curve, &C);
    if (fw < 0) return -1;
    u64 *big = (u64 *)malloc((n ? n : 1) * 4 * 8);
    for (size_t i = 0; i < n; i++) fp4_into_bigint(C.S, big + i * 4, scalars_mont + i * 4);
    int rc = ark_oracle_msm(curve, bases, big, n, variant, threads, out_jac);
    free(big);
    return rc;
}

int ark_oracle_make_digits(const u64 *scalar4, int c, int num_bits, int64_t *out) {
    return make_digits(scalar4, 4, c, num_bits, out);
}
int ark_oracle_window_size(size_t n) { return n < 32 ? 3 : (int)ln_without_floats(n) + 2; }

int ark_oracle_gen_bases(int curve, const u64 *a4, const u64 *b4, size_t n, u64 *out_xy) {
    ark_curve_ctx C;
    int fw = get_curve(curve, &C);
    if (fw < 0) return -1;
    BY_FW(fw, g4_gen_bases(&C, out_xy, a4, b4, n), g6_gen_bases(&C, out_xy, a4, b4, n),
          g12_gen_bases(&C, out_xy, a4, b4, n));
    return 0;
}

/* SplitMix64 stream; uniform in [0, p) by masking the top limb to the modulus bit length and
 * rejecting (same scheme as ff/src/fields/models/fp/mod.rs:521-548 UniformRand for Fp) */
static inline u64 splitmix64(u64 *s) {
    u64 z = (*s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
int ark_oracle_gen_scalars(int field, u64 seed, size_t n, int montgomery, u64 *out) {
    if (field < 0 || field >= NFIELDS) return -1;
    const ark_field_consts *F = &ARK_FIELDS[field];
    if (F->n != 4) return -2;
    int top_bits = F->bits - 64 * 3;
    u64 mask = top_bits >= 64 ? ~(u64)0 : (((u64)1 << top_bits) - 1);
    u64 s = seed;
    for (size_t i = 0; i < n; i++) {
        u64 v[4];
        do {
            for (int k = 0; k < 4; k++) v[k] = splitmix64(&s);
            v[3] &= mask;
        } while (fp4_geq(v, F->p));
        if (montgomery)
            fp4_from_bigint(F, out + i * 4, v);
        else
            memcpy(out + i * 4, v, 32);
    }
    return 0;
}

int ark_oracle_msm_dlog(int curve, const u64 *scalars, size_t n, const u64 *a4, const u64 *b4, u64 *out_k4) {
    ark_curve_ctx C;
    if (get_curve(curve, &C) < 0) return -1;
    const ark_field_consts *S = C.S;
    u64 am[4], bm[4], cur[4], acc[4] = {0, 0, 0, 0}, sm[4], t[4];
    fp4_from_bigint(S, am, a4);
    fp4_from_bigint(S, bm, b4);
    fp4_copy(cur, am);
    for (size_t i = 0; i < n; i++) {
        fp4_from_bigint(S, sm, scalars + i * 4);
        fp4_mul(S, t, sm, cur);
        fp4_add(S, acc, acc, t);
        fp4_add(S, cur, cur, bm);
    }
    fp4_into_bigint(S, out_k4, acc);
    return 0;
}

/* ======================= radix-2 FFT (scalar fields, N = 4) ======================= */
/* fft.rs:368-371 */
static inline u64 bitrev(u64 a, unsigned log_len) {
    u64 r = 0;
    for (unsigned i = 0; i < log_len; i++) r |= ((a >> i) & 1) << (log_len - 1 - i);
    return r;
}
/* fft.rs:373-380 */
static void derange(u64 *xi, size_t len, unsigned log_len) {
    if (len < 2) return;
    for (u64 idx = 1; idx < len - 1; idx++) {
        u64 r = bitrev(idx, log_len);
        if (idx < r) {
            u64 t[4];
            memcpy(t, xi + idx * 4, 32);
            memcpy(xi + idx * 4, xi + r * 4, 32);
            memcpy(xi + r * 4, t, 32);
        }
    }
}
/* fft_friendly.rs:67-81 get_root_of_unity(n): omega = ROOT^(2^(TWO_ADICITY - log n)) */
static int get_root_of_unity(const ark_field_consts *F, unsigned log_n, u64 *omega) {
    if ((int)log_n > F->two_adicity) return -1;
    fp4_copy(omega, F->root);
    for (int i = (int)log_n; i < F->two_adicity; i++) fp4_sqr(F, omega, omega);
    return 0;
}
/* utils.rs:34-51 compute_powers_serial: [1, g, ..., g^(size-1)] */
static void compute_powers(const ark_field_consts *F, u64 *out, size_t size, const u64 *g) {
    u64 v[4];
    fp4_copy(v, F->r);
    for (size_t i = 0; i < size; i++) {
        fp4_copy(out + i * 4, v);
        fp4_mul(F, v, v, g);
    }
}
/* domain/mod.rs:115-128 distribute_powers_and_mul_by_const: x[i] *= c * g^i
 * (the parallel variant, :131-148, computes the same products chunk-wise; values are identical) */
static void distribute_powers_and_mul_by_const(const ark_field_consts *F, u64 *x, size_t n, const u64 *g, const u64 *c,
                                               int threads) {
    size_t nchunk = (size_t)threads;
    size_t per = (n + nchunk - 1) / nchunk;
    if (per < 1024) per = 1024;
#pragma omp parallel for num_threads(threads)
    for (size_t lo = 0; lo < n; lo += per) {
        size_t hi = lo + per > n ? n : lo + per;
        u64 e[1] = {lo}, pw[4];
        fp4_pow(F, pw, g, e, 1);
        fp4_mul(F, pw, pw, c);
        for (size_t i = lo; i < hi; i++) {
            fp4_mul(F, x + i * 4, x + i * 4, pw);
            fp4_mul(F, pw, pw, g);
        }
    }
}
/* fft.rs:190-198 */
static inline void butterfly_io(const ark_field_consts *F, u64 *lo, u64 *hi, const u64 *root) {
    u64 neg[4];
    fp4_sub(F, neg, lo, hi);
    fp4_add(F, lo, lo, hi);
    fp4_mul(F, hi, neg, root);
}
/* fft.rs:201-210 */
static inline void butterfly_oi(const ark_field_consts *F, u64 *lo, u64 *hi, const u64 *root) {
    u64 neg[4];
    fp4_mul(F, hi, hi, root);
    fp4_sub(F, neg, lo, hi);
    fp4_add(F, lo, lo, hi);
    fp4_copy(hi, neg);
}
#define MIN_NUM_CHUNKS_FOR_COMPACTION ((size_t)1 << 7) /* fft.rs:354 */

/* fft.rs:213-250 apply_butterfly: every chunk of 2*gap, butterfly j uses roots[j*step] */
static void apply_butterfly(const ark_field_consts *F, int io, u64 *xi, size_t n, const u64 *roots, size_t step,
                            size_t gap, int threads) {
    size_t chunk = 2 * gap, num_chunks = n / chunk;
    if (num_chunks >= (size_t)threads || threads == 1) {
#pragma omp parallel for num_threads(threads) if (threads > 1 && n > 1024)
        for (size_t c = 0; c < num_chunks; c++) {
            u64 *lo = xi + c * chunk * 4, *hi = lo + gap * 4;
            for (size_t j = 0; j < gap; j++) {
                if (io)
                    butterfly_io(F, lo + j * 4, hi + j * 4, roots + j * step * 4);
                else
                    butterfly_oi(F, lo + j * 4, hi + j * 4, roots + j * step * 4);
            }
        }
    } else {
        for (size_t c = 0; c < num_chunks; c++) {
            u64 *lo = xi + c * chunk * 4, *hi = lo + gap * 4;
#pragma omp parallel for num_threads(threads) if (gap > 1024)
            for (size_t j = 0; j < gap; j++) {
                if (io)
                    butterfly_io(F, lo + j * 4, hi + j * 4, roots + j * step * 4);
                else
                    butterfly_oi(F, lo + j * 4, hi + j * 4, roots + j * step * 4);
            }
        }
    }
}
/* fft.rs:252-295 io_helper (DIF: in-order in, bit-reversed out) */
static void io_helper(const ark_field_consts *F, u64 *xi, size_t n, const u64 *root, int threads) {
    size_t nroots = n / 2;
    u64 *roots = (u64 *)malloc((nroots ? nroots : 1) * 32);
    compute_powers(F, roots, nroots, root);
    size_t step = 1;
    int first = 1;
    for (size_t gap = n / 2; gap > 0; gap /= 2) {
        size_t num_chunks = n / (2 * gap);
        if (num_chunks >= MIN_NUM_CHUNKS_FOR_COMPACTION) {
            if (!first) { /* roots = roots.step_by(step * 2) */
                size_t m = (nroots + step * 2 - 1) / (step * 2);
                for (size_t i = 0; i < m; i++) memmove(roots + i * 4, roots + i * step * 2 * 4, 32);
                nroots = m;
            }
            step = 1;
        } else {
            step = num_chunks;
        }
        first = 0;
        apply_butterfly(F, 1, xi, n, roots, step, gap, threads);
    }
    free(roots);
}
/* fft.rs:297-349 oi_helper (DIT: bit-reversed in, in-order out) */
static void oi_helper(const ark_field_consts *F, u64 *xi, size_t n, const u64 *root, size_t start_gap, int threads) {
    size_t nroots = n / 2;
    u64 *roots_cache = (u64 *)malloc((nroots ? nroots : 1) * 32);
    compute_powers(F, roots_cache, nroots, root);
    size_t cmax = nroots / 2 < nroots / MIN_NUM_CHUNKS_FOR_COMPACTION ? nroots / 2 : nroots / MIN_NUM_CHUNKS_FOR_COMPACTION;
    u64 *compacted = (u64 *)malloc((cmax ? cmax : 1) * 32);
    for (size_t gap = start_gap; gap < n; gap *= 2) {
        size_t num_chunks = n / (2 * gap);
        if (num_chunks >= MIN_NUM_CHUNKS_FOR_COMPACTION && gap < n / 2) {
            for (size_t i = 0; i < gap; i++) memcpy(compacted + i * 4, roots_cache + i * num_chunks * 4, 32);
            apply_butterfly(F, 0, xi, n, compacted, 1, gap, threads);
        } else {
            apply_butterfly(F, 0, xi, n, roots_cache, num_chunks, gap, threads);
        }
    }
    free(compacted);
    free(roots_cache);
}

int ark_oracle_domain(int field, unsigned log_n, u64 *group_gen, u64 *group_gen_inv, u64 *size_inv) {
    if (field < 0 || field >= NFIELDS) return -1;
    const ark_field_consts *F = &ARK_FIELDS[field];
    if (F->n != 4) return -2;
    if (get_root_of_unity(F, log_n, group_gen)) return -3; /* radix2/mod.rs:62-64: log n > TWO_ADICITY -> None */
    fp4_inv(F, group_gen_inv, group_gen);
    u64 sz[4] = {(u64)1 << log_n, 0, 0, 0}, szm[4];
    fp4_from_bigint(F, szm, sz); /* F::from(size) */
    fp4_inv(F, size_inv, szm);
    return 0;
}

int ark_oracle_fft(int field, u64 *data, unsigned log_n, const u64 *offset, int inverse, int threads) {
    if (field < 0 || field >= NFIELDS) return -1;
    const ark_field_consts *F = &ARK_FIELDS[field];
    if (F->n != 4) return -2;
    if (threads < 1) threads = 1;
    u64 g[4], ginv[4], sinv[4];
    int rc = ark_oracle_domain(field, log_n, g, ginv, sinv);
    if (rc) return rc;
    size_t n = (size_t)1 << log_n;
    int coset = offset && !fp4_eq(offset, F->r);
    if (!inverse) {
        /* fft.rs:74-79 in_order_fft_in_place */
        if (coset) distribute_powers_and_mul_by_const(F, data, n, offset, F->r, threads);
        io_helper(F, data, n, g, threads);
        derange(data, n, log_n);
    } else {
        /* fft.rs:81-88 in_order_ifft_in_place */
        derange(data, n, log_n);
        oi_helper(F, data, n, ginv, 1, threads);
        if (!coset) {
#pragma omp parallel for num_threads(threads) if (n > 1024)
            for (size_t i = 0; i < n; i++) fp4_mul(F, data + i * 4, data + i * 4, sinv);
        } else {
            u64 oinv[4];
            fp4_inv(F, oinv, offset);
            distribute_powers_and_mul_by_const(F, data, n, oinv, sinv, threads);
        }
    }
    return 0;
}

/* radix2/mod.rs:351-391 test_fft_correctness oracle: evaluate by Horner at offset * g^k */
int ark_oracle_dft_naive(int field, const u64 *coeffs, size_t ncoeffs, unsigned log_n, const u64 *offset, u64 *out) {
    if (field < 0 || field >= NFIELDS) return -1;
    const ark_field_consts *F = &ARK_FIELDS[field];
    if (F->n != 4) return -2;
    u64 g[4], ginv[4], sinv[4], pt[4];
    int rc = ark_oracle_domain(field, log_n, g, ginv, sinv);
    if (rc) return rc;
    size_t n = (size_t)1 << log_n;
    if (offset)
        fp4_copy(pt, offset);
    else
        fp4_copy(pt, F->r);
    for (size_t k = 0; k < n; k++) {
        u64 acc[4] = {0, 0, 0, 0};
        for (size_t j = ncoeffs; j-- > 0;) {
            fp4_mul(F, acc, acc, pt);
            fp4_add(F, acc, acc, coeffs + j * 4);
        }
        fp4_copy(out + k * 4, acc);
        fp4_mul(F, pt, pt, g);
    }
    return 0;
}
