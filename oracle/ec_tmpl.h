/*
 * TEST INFRASTRUCTURE (oracle) -- not part of the product path.
 *
 * Short-Weierstrass group law (a = 0 curves only: every curve in BASELINE.json's configs has
 * COEFF_A = 0) and Pippenger MSM, restated from the reference:
 *   ec/src/models/short_weierstrass/bucket.rs:112-146   Bucket::double_in_place   (dbl-2008-s-1)
 *   ec/src/models/short_weierstrass/bucket.rs:168-244   Bucket += / -= Affine      (madd-2008-s)
 *   ec/src/models/short_weierstrass/bucket.rs:256-337   Bucket += &Bucket          (add-2008-s)
 *   ec/src/models/short_weierstrass/bucket.rs:345-359   Projective += &Bucket
 *   ec/src/models/short_weierstrass/bucket.rs:389-397   From<Bucket> for Projective
 *   ec/src/models/short_weierstrass/affine.rs:169-201   Affine::double_to_bucket   (mdbl-2008-s-1)
 *   ec/src/models/short_weierstrass/affine.rs:374-396   From<Projective> for Affine
 *   ec/src/models/short_weierstrass/group.rs:171-221    Projective::double_in_place (a = 0 branch)
 *   ec/src/models/short_weierstrass/group.rs:450-538    Projective += &Projective  (add-2007-bl)
 *   ec/src/scalar_mul/variable_base/mod.rs:242-347      msm_signed
 *   ec/src/scalar_mul/variable_base/mod.rs:373-434      msm_binary, msm_u8..u64
 *   ec/src/scalar_mul/variable_base/mod.rs:437-558      msm_bigint_wnaf(_parallel)
 *   ec/src/scalar_mul/variable_base/mod.rs:657-751      msm_serial
 *   ec/src/scalar_mul/variable_base/mod.rs:754-794      make_digits
 *   ec/src/scalar_mul/mod.rs:22-25                      ln_without_floats
 *   test-templates/src/msm.rs:8-15                      naive_var_base_msm
 *
 * "Template": include with
 *    EC     -- function-name prefix (e.g. g1_4)
 *    FW     -- u64 words per base-field element (4, 6 or 12)
 *    FE(op) -- name of the field op  void op(const ark_curve_ctx*, u64* r, const u64* a, const u64* b)
 * Points are plain u64 arrays: affine = [x|y] (identity = all zero, ZeroFlag = (),
 * short_weierstrass/mod.rs:224-230), Jacobian = [x|y|z], bucket (XYZZ) = [x|y|zz|zzz].
 */
#define EC_CAT_(a, b) a##b
#define EC_CAT(a, b) EC_CAT_(a, b)
#define ECN(name) EC_CAT(EC_CAT(EC, _), name)

#define AX(p) (p)
#define AY(p) ((p) + FW)
#define JZ(p) ((p) + 2 * FW)
#define BZZ(p) ((p) + 2 * FW)
#define BZZZ(p) ((p) + 3 * FW)

static inline int ECN(fe_is_zero)(const u64 *a) {
    u64 x = 0;
    for (int i = 0; i < FW; i++) x |= a[i];
    return x == 0;
}
static inline int ECN(fe_eq)(const u64 *a, const u64 *b) {
    u64 x = 0;
    for (int i = 0; i < FW; i++) x |= a[i] ^ b[i];
    return x == 0;
}
static inline void ECN(fe_copy)(u64 *r, const u64 *a) { memcpy(r, a, FW * 8); }
static inline void ECN(fe_zero)(u64 *r) { memset(r, 0, FW * 8); }
static inline void ECN(fe_one)(const ark_curve_ctx *C, u64 *r) {
    memset(r, 0, FW * 8);
    memcpy(r, C->F->r, C->F->n * 8); /* ONE = R (montgomery_backend.rs:21); Fp2 one = (R, 0) */
}

/* affine identity flag: (0,0)  -- affine.rs:91-104 */
static inline int ECN(aff_is_zero)(const u64 *p) { return ECN(fe_is_zero)(AX(p)) && ECN(fe_is_zero)(AY(p)); }
/* Bucket::ZERO = (1, 1, 0, 0)  -- bucket.rs:78-83 */
static inline void ECN(bkt_set_zero)(const ark_curve_ctx *C, u64 *b) {
    ECN(fe_one)(C, AX(b));
    ECN(fe_one)(C, AY(b));
    ECN(fe_zero)(BZZ(b));
    ECN(fe_zero)(BZZZ(b));
}
static inline int ECN(bkt_is_zero)(const u64 *b) { return ECN(fe_is_zero)(BZZ(b)) && ECN(fe_is_zero)(BZZZ(b)); }
/* Projective::zero = (1, 1, 0)  -- group.rs:145-151 */
static inline void ECN(jac_set_zero)(const ark_curve_ctx *C, u64 *p) {
    ECN(fe_one)(C, AX(p));
    ECN(fe_one)(C, AY(p));
    ECN(fe_zero)(JZ(p));
}
static inline int ECN(jac_is_zero)(const u64 *p) { return ECN(fe_is_zero)(JZ(p)); }

/* affine.rs:169-201 double_to_bucket (a = 0) */
static void ECN(aff_double_to_bucket)(const ark_curve_ctx *C, u64 *out, const u64 *p) {
    if (ECN(aff_is_zero)(p)) {
        ECN(bkt_set_zero)(C, out);
        return;
    }
    u64 u[FW], v[FW], w[FW], s[FW], m[FW], t[FW], x[FW], y[FW];
    FE(add)(C, u, AY(p), AY(p)); /* U = 2*Y1 */
    FE(mul)(C, v, u, u);         /* V = U^2 */
    FE(mul)(C, w, u, v);         /* W = U*V */
    FE(mul)(C, s, AX(p), v);     /* S = X1*V */
    FE(mul)(C, m, AX(p), AX(p)); /* M = 3*X1^2 (+a, a = 0) */
    FE(add)(C, t, m, m);
    FE(add)(C, m, m, t);
    FE(mul)(C, x, m, m); /* X3 = M^2 - 2*S */
    FE(add)(C, t, s, s);
    FE(sub)(C, x, x, t);
    FE(sub)(C, t, s, x); /* Y3 = M*(S-X3) - W*Y1 */
    FE(mul)(C, y, m, t);
    FE(mul)(C, t, w, AY(p));
    FE(sub)(C, y, y, t);
    ECN(fe_copy)(AX(out), x);
    ECN(fe_copy)(AY(out), y);
    ECN(fe_copy)(BZZ(out), v);
    ECN(fe_copy)(BZZZ(out), w);
}

/* bucket.rs:112-146 Bucket::double_in_place (a = 0) */
static void ECN(bkt_double)(const ark_curve_ctx *C, u64 *b) {
    u64 u[FW], v[FW], w[FW], s[FW], m[FW], t[FW], x[FW], y[FW];
    FE(add)(C, u, AY(b), AY(b));
    FE(mul)(C, v, u, u);
    FE(mul)(C, w, u, v);
    FE(mul)(C, s, AX(b), v);
    FE(mul)(C, m, AX(b), AX(b));
    FE(add)(C, t, m, m);
    FE(add)(C, m, m, t);
    FE(mul)(C, x, m, m);
    FE(add)(C, t, s, s);
    FE(sub)(C, x, x, t);
    /* Y3 = M*(S-X3) - W*Y1   (reference: sum_of_products([m, -w], [s - x3, y1])) */
    FE(sub)(C, t, s, x);
    FE(mul)(C, y, m, t);
    FE(mul)(C, t, w, AY(b));
    FE(sub)(C, y, y, t);
    FE(mul)(C, BZZ(b), BZZ(b), v);
    FE(mul)(C, BZZZ(b), BZZZ(b), w);
    ECN(fe_copy)(AX(b), x);
    ECN(fe_copy)(AY(b), y);
}

/* bucket.rs:168-238 Bucket += Affine ; negate_other != 0 gives Bucket -= Affine (bucket.rs:240-244) */
static void ECN(bkt_add_affine)(const ark_curve_ctx *C, u64 *b, const u64 *other, int negate_other) {
    if (ECN(aff_is_zero)(other)) return; /* other.xy() == None */
    u64 oy[FW];
    if (negate_other)
        FE(neg)(C, oy, AY(other), AY(other));
    else
        ECN(fe_copy)(oy, AY(other));
    if (ECN(bkt_is_zero)(b)) {
        ECN(fe_copy)(AX(b), AX(other));
        ECN(fe_copy)(AY(b), oy);
        ECN(fe_one)(C, BZZ(b));
        ECN(fe_one)(C, BZZZ(b));
        return;
    }
    u64 u2[FW], s2[FW];
    FE(mul)(C, u2, AX(other), BZZ(b)); /* U2 = X2*ZZ1 */
    FE(mul)(C, s2, oy, BZZZ(b));       /* S2 = Y2*ZZZ1 */
    if (ECN(fe_eq)(AX(b), u2)) {
        if (ECN(fe_eq)(AY(b), s2)) {
            u64 o2[2 * FW];
            ECN(fe_copy)(AX(o2), AX(other));
            ECN(fe_copy)(AY(o2), oy);
            ECN(aff_double_to_bucket)(C, b, o2);
        } else {
            ECN(bkt_set_zero)(C, b);
        }
        return;
    }
    u64 p[FW], r[FW], pp[FW], ppp[FW], q[FW], t[FW], x[FW], y[FW];
    FE(sub)(C, p, u2, AX(b));
    FE(sub)(C, r, s2, AY(b));
    FE(mul)(C, pp, p, p);
    FE(mul)(C, ppp, pp, p);
    FE(mul)(C, q, AX(b), pp);
    FE(mul)(C, x, r, r); /* X3 = R^2 - PPP - 2Q */
    FE(sub)(C, x, x, ppp);
    FE(add)(C, t, q, q);
    FE(sub)(C, x, x, t);
    FE(sub)(C, q, q, x); /* Y3 = R*(Q-X3) - Y1*PPP */
    FE(mul)(C, y, r, q);
    FE(mul)(C, t, AY(b), ppp);
    FE(sub)(C, y, y, t);
    FE(mul)(C, BZZ(b), BZZ(b), pp);
    FE(mul)(C, BZZZ(b), BZZZ(b), ppp);
    ECN(fe_copy)(AX(b), x);
    ECN(fe_copy)(AY(b), y);
}

/* bucket.rs:256-337 Bucket += &Bucket */
static void ECN(bkt_add_bkt)(const ark_curve_ctx *C, u64 *a, const u64 *o) {
    if (ECN(bkt_is_zero)(a)) {
        memcpy(a, o, 4 * FW * 8);
        return;
    }
    if (ECN(bkt_is_zero)(o)) return;
    u64 u1[FW], u2[FW], s1[FW], s2[FW];
    FE(mul)(C, u1, AX(a), BZZ(o));
    FE(mul)(C, u2, AX(o), BZZ(a));
    FE(mul)(C, s1, AY(a), BZZZ(o));
    FE(mul)(C, s2, AY(o), BZZZ(a));
    if (ECN(fe_eq)(u1, u2)) {
        if (ECN(fe_eq)(s1, s2))
            ECN(bkt_double)(C, a);
        else
            ECN(bkt_set_zero)(C, a);
        return;
    }
    u64 p[FW], r[FW], pp[FW], ppp[FW], q[FW], t[FW], x[FW], y[FW];
    FE(sub)(C, p, u2, u1);
    FE(sub)(C, r, s2, s1);
    FE(mul)(C, pp, p, p);
    FE(mul)(C, ppp, pp, p);
    FE(mul)(C, q, u1, pp);
    FE(mul)(C, x, r, r);
    FE(sub)(C, x, x, ppp);
    FE(add)(C, t, q, q);
    FE(sub)(C, x, x, t);
    FE(sub)(C, q, q, x);
    FE(mul)(C, y, r, q);
    FE(mul)(C, t, s1, ppp);
    FE(sub)(C, y, y, t);
    FE(mul)(C, BZZ(a), BZZ(a), pp);
    FE(mul)(C, BZZ(a), BZZ(a), BZZ(o));
    FE(mul)(C, BZZZ(a), BZZZ(a), ppp);
    FE(mul)(C, BZZZ(a), BZZZ(a), BZZZ(o));
    ECN(fe_copy)(AX(a), x);
    ECN(fe_copy)(AY(a), y);
}

/* bucket.rs:389-397 From<Bucket> for Projective: (X*ZZ, Y*ZZZ, ZZ) */
static void ECN(bkt_to_jac)(const ark_curve_ctx *C, u64 *j, const u64 *b) {
    if (ECN(bkt_is_zero)(b)) {
        ECN(jac_set_zero)(C, j);
        return;
    }
    FE(mul)(C, AX(j), AX(b), BZZ(b));
    FE(mul)(C, AY(j), AY(b), BZZZ(b));
    ECN(fe_copy)(JZ(j), BZZ(b));
}

/* group.rs:171-221 Projective::double_in_place, COEFF_A == 0 branch, extension degree 1 or 2 */
static void ECN(jac_double)(const ark_curve_ctx *C, u64 *p) {
    if (ECN(jac_is_zero)(p)) return;
    u64 a[FW], b[FW], c[FW], d[FW], e[FW], t[FW];
    FE(mul)(C, a, AX(p), AX(p)); /* A = X1^2 */
    FE(mul)(C, b, AY(p), AY(p)); /* B = Y1^2 */
    FE(mul)(C, c, b, b);         /* C = B^2 */
    FE(mul)(C, d, AX(p), b);     /* D = 4*X1*B */
    FE(add)(C, d, d, d);
    FE(add)(C, d, d, d);
    FE(add)(C, e, a, a); /* E = 3*A */
    FE(add)(C, e, e, a);
    FE(mul)(C, JZ(p), JZ(p), AY(p)); /* Z3 = 2*Y1*Z1 */
    FE(add)(C, JZ(p), JZ(p), JZ(p));
    FE(mul)(C, AX(p), e, e); /* X3 = E^2 - 2*D */
    FE(add)(C, t, d, d);
    FE(sub)(C, AX(p), AX(p), t);
    FE(sub)(C, t, d, AX(p)); /* Y3 = E*(D-X3) - 8*C */
    FE(mul)(C, AY(p), t, e);
    FE(add)(C, c, c, c);
    FE(add)(C, c, c, c);
    FE(add)(C, c, c, c);
    FE(sub)(C, AY(p), AY(p), c);
}

/* group.rs:450-538 Projective += &Projective */
static void ECN(jac_add)(const ark_curve_ctx *C, u64 *a, const u64 *o) {
    if (ECN(jac_is_zero)(a)) {
        memcpy(a, o, 3 * FW * 8);
        return;
    }
    if (ECN(jac_is_zero)(o)) return;
    u64 z1z1[FW], z2z2[FW], u1[FW], u2[FW], s1[FW], s2[FW];
    FE(mul)(C, z1z1, JZ(a), JZ(a));
    FE(mul)(C, z2z2, JZ(o), JZ(o));
    FE(mul)(C, u1, AX(a), z2z2);
    FE(mul)(C, u2, AX(o), z1z1);
    FE(mul)(C, s1, AY(a), JZ(o));
    FE(mul)(C, s1, s1, z2z2);
    FE(mul)(C, s2, AY(o), JZ(a));
    FE(mul)(C, s2, s2, z1z1);
    if (ECN(fe_eq)(u1, u2)) {
        if (ECN(fe_eq)(s1, s2))
            ECN(jac_double)(C, a);
        else
            ECN(jac_set_zero)(C, a);
        return;
    }
    u64 h[FW], i[FW], j[FW], r[FW], v[FW], t[FW], x[FW], y[FW];
    FE(sub)(C, h, u2, u1);
    FE(add)(C, i, h, h); /* I = (2H)^2 */
    FE(mul)(C, i, i, i);
    FE(neg)(C, j, h, h); /* J = -H*I */
    FE(mul)(C, j, j, i);
    FE(sub)(C, r, s2, s1); /* r = 2*(S2-S1) */
    FE(add)(C, r, r, r);
    FE(mul)(C, v, u1, i);
    FE(mul)(C, x, r, r); /* X3 = r^2 + J - 2V */
    FE(add)(C, x, x, j);
    FE(add)(C, t, v, v);
    FE(sub)(C, x, x, t);
    FE(sub)(C, v, v, x); /* Y3 = r*(V-X3) + 2*S1*J */
    FE(mul)(C, y, r, v);
    FE(add)(C, t, s1, s1);
    FE(mul)(C, t, t, j);
    FE(add)(C, y, y, t);
    FE(mul)(C, JZ(a), JZ(a), JZ(o)); /* Z3 = 2*Z1*Z2*H */
    FE(add)(C, JZ(a), JZ(a), JZ(a));
    FE(mul)(C, JZ(a), JZ(a), h);
    ECN(fe_copy)(AX(a), x);
    ECN(fe_copy)(AY(a), y);
}
/* Projective -= &Projective */
static void ECN(jac_sub)(const ark_curve_ctx *C, u64 *a, const u64 *o) {
    u64 n[3 * FW];
    memcpy(n, o, sizeof n);
    FE(neg)(C, AY(n), AY(n), AY(n));
    ECN(jac_add)(C, a, n);
}
/* bucket.rs:345-359 Projective += &Bucket */
static void ECN(jac_add_bkt)(const ark_curve_ctx *C, u64 *a, const u64 *b) {
    u64 j[3 * FW];
    if (ECN(bkt_is_zero)(b)) return;
    ECN(bkt_to_jac)(C, j, b);
    if (ECN(jac_is_zero)(a)) {
        memcpy(a, j, sizeof j);
        return;
    }
    ECN(jac_add)(C, a, j);
}
/* Affine -> Projective (x, y, 1) ; identity -> zero()   (affine.rs From<Affine> for Projective) */
static void ECN(aff_to_jac)(const ark_curve_ctx *C, u64 *j, const u64 *p) {
    if (ECN(aff_is_zero)(p)) {
        ECN(jac_set_zero)(C, j);
        return;
    }
    ECN(fe_copy)(AX(j), AX(p));
    ECN(fe_copy)(AY(j), AY(p));
    ECN(fe_one)(C, JZ(j));
}
/* affine.rs:374-396 From<Projective> for Affine */
static void ECN(jac_to_aff)(const ark_curve_ctx *C, u64 *out, const u64 *p) {
    if (ECN(jac_is_zero)(p)) {
        memset(out, 0, 2 * FW * 8);
        return;
    }
    u64 zi[FW], zi2[FW], zi3[FW];
    FE(inv)(C, zi, JZ(p), JZ(p));
    FE(mul)(C, zi2, zi, zi);
    FE(mul)(C, zi3, zi2, zi);
    FE(mul)(C, AX(out), AX(p), zi2);
    FE(mul)(C, AY(out), AY(p), zi3);
}

/* ec/src/scalar_mul/mod.rs:41-51 double_and_add_affine (MSB first, skipping leading zeros) */
static void ECN(scalar_mul)(const ark_curve_ctx *C, u64 *out_jac, const u64 *base_aff, const u64 *scalar, int nlimbs) {
    u64 res[3 * FW], bj[3 * FW];
    ECN(jac_set_zero)(C, res);
    ECN(aff_to_jac)(C, bj, base_aff);
    int started = 0;
    for (int i = nlimbs * 64 - 1; i >= 0; i--) {
        int bit = (scalar[i / 64] >> (i % 64)) & 1;
        if (started) ECN(jac_double)(C, res);
        if (bit) {
            ECN(jac_add)(C, res, bj);
            started = 1;
        }
    }
    memcpy(out_jac, res, sizeof res);
}

/* test-templates/src/msm.rs:8-15 naive_var_base_msm (scalars as canonical bigints) */
static void ECN(msm_naive)(const ark_curve_ctx *C, u64 *out_jac, const u64 *bases, const u64 *scalars, size_t n) {
    u64 acc[3 * FW], t[3 * FW];
    ECN(jac_set_zero)(C, acc);
    for (size_t i = 0; i < n; i++) {
        ECN(scalar_mul)(C, t, bases + i * 2 * FW, scalars + i * SCALAR_LIMBS, SCALAR_LIMBS);
        ECN(jac_add)(C, acc, t);
    }
    memcpy(out_jac, acc, sizeof acc);
}

/* lowest = first window sum; fold the rest high -> low with c doublings each (mod.rs:486-502) */
static void ECN(combine_windows)(const ark_curve_ctx *C, u64 *out_jac, const u64 *window_sums /* buckets */, int nwin, int c) {
    u64 lowest[3 * FW], total[3 * FW];
    ECN(bkt_to_jac)(C, lowest, window_sums);
    ECN(jac_set_zero)(C, total);
    for (int w = nwin - 1; w >= 1; w--) {
        ECN(jac_add_bkt)(C, total, window_sums + (size_t)w * 4 * FW);
        for (int k = 0; k < c; k++) ECN(jac_double)(C, total);
    }
    ECN(jac_add)(C, lowest, total);
    memcpy(out_jac, lowest, sizeof lowest);
}

/* running-sum bucket reduction (mod.rs:478-484): res = sum_k (k+1) * buckets[k] */
static void ECN(reduce_buckets)(const ark_curve_ctx *C, u64 *res_bkt, const u64 *buckets, size_t nb) {
    u64 running[4 * FW];
    ECN(bkt_set_zero)(C, running);
    for (size_t k = nb; k-- > 0;) {
        ECN(bkt_add_bkt)(C, running, buckets + k * 4 * FW);
        ECN(bkt_add_bkt)(C, res_bkt, running);
    }
}

/* mod.rs:437-503 msm_bigint_wnaf_parallel; `threads` OpenMP threads over the windows */
static void ECN(msm_wnaf_parallel)(const ark_curve_ctx *C, u64 *out_jac, const u64 *bases, const u64 *scalars, size_t size, int threads) {
    int c = size < 32 ? 3 : (int)ln_without_floats(size) + 2;
    int num_bits = C->S->bits;
    int digits_count = (num_bits + c - 1) / c;
    int64_t *digits = (int64_t *)malloc(sizeof(int64_t) * size * digits_count);
    for (size_t i = 0; i < size; i++) make_digits(scalars + i * SCALAR_LIMBS, SCALAR_LIMBS, c, num_bits, digits + i * digits_count);
    u64 *window_sums = (u64 *)malloc((size_t)digits_count * 4 * FW * 8);
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1)
    for (int w = 0; w < digits_count; w++) {
        size_t nb = (size_t)1 << c;
        u64 *buckets = (u64 *)malloc(nb * 4 * FW * 8);
        for (size_t k = 0; k < nb; k++) ECN(bkt_set_zero)(C, buckets + k * 4 * FW);
        for (size_t i = 0; i < size; i++) {
            int64_t d = digits[i * digits_count + w];
            if (d > 0)
                ECN(bkt_add_affine)(C, buckets + (size_t)(d - 1) * 4 * FW, bases + i * 2 * FW, 0);
            else if (d < 0)
                ECN(bkt_add_affine)(C, buckets + (size_t)(-d - 1) * 4 * FW, bases + i * 2 * FW, 1);
        }
        u64 *res = window_sums + (size_t)w * 4 * FW;
        ECN(bkt_set_zero)(C, res);
        ECN(reduce_buckets)(C, res, buckets, nb);
        free(buckets);
    }
    ECN(combine_windows)(C, out_jac, window_sums, digits_count, c);
    free(window_sums);
    free(digits);
}

/* mod.rs:512-558 msm_bigint_wnaf: threads/2 base-range chunks, each summed with msm_bigint_wnaf_parallel (:437-503)
 * on its own 2-thread pool, chunk sums added up.  The ARITHMETIC here is exactly that -- the same chunk boundaries, the
 * same per-chunk window size c = ln_without_floats(chunk) + 2, the same digits, bucket sums and window combine -- but
 * the SCHEDULE is flat: rayon's nested pools keep every core busy in the reference (work stealing), whereas nested
 * OpenMP regions run serially by default, which left half the cores idle and serialised the digit pass (the round-2
 * baseline, 1.2e6 scalar-muls/s on 256 cores).  Three flat phases over all `threads` threads:
 *   1. digits of every scalar (window-major per chunk, int32: c <= 22 bits),
 *   2. one task per (chunk, window): bucket accumulation + running-sum reduction, buckets in a per-thread arena,
 *   3. one task per chunk: window combine; then the chunk sums are added in order. */
static void ECN(msm_wnaf)(const ark_curve_ctx *C, u64 *out_jac, const u64 *bases, const u64 *scalars, size_t size, int threads) {
    ECN(jac_set_zero)(C, out_jac);
    if (size == 0) return;
    if (threads < 1) threads = 1;
    size_t num_chunks = threads < 2 ? 1 : (size_t)threads / 2;
    size_t chunk = size / num_chunks;
    if (chunk == 0) chunk = size;
    size_t nchunks = (size + chunk - 1) / chunk;
    const int num_bits = C->S->bits;
    /* per-chunk geometry (the last chunk may be shorter and so may use a smaller c) */
    int *cs = (int *)malloc(nchunks * sizeof(int));
    int *dcs = (int *)malloc(nchunks * sizeof(int));
    size_t *doff = (size_t *)malloc((nchunks + 1) * sizeof(size_t)); /* start of chunk k's digits */
    size_t *toff = (size_t *)malloc((nchunks + 1) * sizeof(size_t)); /* first (chunk, window) task of chunk k */
    doff[0] = 0;
    toff[0] = 0;
    for (size_t k = 0; k < nchunks; k++) {
        size_t lo = k * chunk, hi = lo + chunk > size ? size : lo + chunk, len = hi - lo;
        cs[k] = len < 32 ? 3 : (int)ln_without_floats(len) + 2;
        dcs[k] = (num_bits + cs[k] - 1) / cs[k];
        doff[k + 1] = doff[k] + len * (size_t)dcs[k];
        toff[k + 1] = toff[k] + (size_t)dcs[k];
    }
    const size_t ntasks = toff[nchunks];
    int32_t *digits = (int32_t *)malloc(doff[nchunks] * sizeof(int32_t));
    u64 *window_sums = (u64 *)malloc(ntasks * 4 * FW * 8);
    u64 *partial = (u64 *)malloc(nchunks * 3 * FW * 8);
#pragma omp parallel num_threads(threads)
    {
        /* phase 1: digits[doff[k] + w * len + i] = digit w of scalar lo + i */
#pragma omp for schedule(static)
        for (size_t i = 0; i < size; i++) {
            size_t k = i / chunk;
            if (k >= nchunks) k = nchunks - 1;
            size_t lo = k * chunk, hi = lo + chunk > size ? size : lo + chunk, len = hi - lo;
            int64_t d[96];
            make_digits(scalars + i * SCALAR_LIMBS, SCALAR_LIMBS, cs[k], num_bits, d);
            for (int w = 0; w < dcs[k]; w++) digits[doff[k] + (size_t)w * len + (i - lo)] = (int32_t)d[w];
        }
        /* phase 2: (chunk, window) tasks */
        u64 *arena = NULL;
        size_t arena_nb = 0;
#pragma omp for schedule(dynamic, 1)
        for (size_t t = 0; t < ntasks; t++) {
            size_t k = 0; /* chunks have (almost) all the same window count: start near t / dcs[0] */
            k = t / (size_t)dcs[0];
            if (k >= nchunks) k = nchunks - 1;
            while (toff[k] > t) k--;
            while (toff[k + 1] <= t) k++;
            const int w = (int)(t - toff[k]);
            size_t lo = k * chunk, hi = lo + chunk > size ? size : lo + chunk, len = hi - lo;
            const size_t nb = (size_t)1 << cs[k];
            if (arena_nb < nb) {
                free(arena);
                arena = (u64 *)malloc(nb * 4 * FW * 8);
                arena_nb = nb;
            }
            for (size_t b = 0; b < nb; b++) ECN(bkt_set_zero)(C, arena + b * 4 * FW);
            const int32_t *dg = digits + doff[k] + (size_t)w * len;
            const u64 *bs = bases + lo * 2 * FW;
            for (size_t i = 0; i < len; i++) {
                int32_t d = dg[i];
                if (d > 0)
                    ECN(bkt_add_affine)(C, arena + (size_t)(d - 1) * 4 * FW, bs + i * 2 * FW, 0);
                else if (d < 0)
                    ECN(bkt_add_affine)(C, arena + (size_t)(-(int64_t)d - 1) * 4 * FW, bs + i * 2 * FW, 1);
            }
            u64 *res = window_sums + t * 4 * FW;
            ECN(bkt_set_zero)(C, res);
            ECN(reduce_buckets)(C, res, arena, nb);
        }
        free(arena);
        /* phase 3: window combine per chunk */
#pragma omp for schedule(dynamic, 1)
        for (size_t k = 0; k < nchunks; k++)
            ECN(combine_windows)(C, partial + k * 3 * FW, window_sums + toff[k] * 4 * FW, dcs[k], cs[k]);
    }
    for (size_t k = 0; k < nchunks; k++) ECN(jac_add)(C, out_jac, partial + k * 3 * FW);
    free(partial);
    free(window_sums);
    free(digits);
    free(toff);
    free(doff);
    free(dcs);
    free(cs);
}

/* mod.rs:657-751 msm_serial over u64-sized scalars (unsigned windows, scalar == 1 fast path) */
static void ECN(msm_serial_u64)(const ark_curve_ctx *C, u64 *out_jac, const u64 *bases, const u64 *scalars, size_t n, int scalar_bits) {
    int c = n < 32 ? 3 : (int)ln_without_floats(n) + 2;
    size_t nb = ((size_t)1 << c) - 1;
    int nwin = (scalar_bits + c - 1) / c; /* (0..size_of::<u64>()*8).step_by(c) uses 64; windows above the type width are empty */
    nwin = (64 + c - 1) / c;
    (void)scalar_bits;
    u64 *window_sums = (u64 *)malloc((size_t)nwin * 4 * FW * 8);
    u64 *buckets = (u64 *)malloc((nb ? nb : 1) * 4 * FW * 8);
    for (int w = 0; w < nwin; w++) {
        int w_start = w * c;
        u64 *res = window_sums + (size_t)w * 4 * FW;
        ECN(bkt_set_zero)(C, res);
        for (size_t k = 0; k < nb; k++) ECN(bkt_set_zero)(C, buckets + k * 4 * FW);
        for (size_t i = 0; i < n; i++) {
            u64 s = scalars[i];
            if (s == 0) continue;
            if (s == 1) {
                if (w_start == 0) ECN(bkt_add_affine)(C, res, bases + i * 2 * FW, 0);
            } else {
                s >>= w_start;
                s %= ((u64)1 << c);
                if (s != 0) ECN(bkt_add_affine)(C, buckets + (size_t)(s - 1) * 4 * FW, bases + i * 2 * FW, 0);
            }
        }
        ECN(reduce_buckets)(C, res, buckets, nb);
    }
    ECN(combine_windows)(C, out_jac, window_sums, nwin, c);
    free(buckets);
    free(window_sums);
}

/* mod.rs:349-369 preamble + :392-434 msm_u8..u64: per-thread chunks of msm_serial, summed */
static void ECN(msm_small)(const ark_curve_ctx *C, u64 *out_jac, const u64 *bases, const u64 *scalars, size_t n, int threads) {
    ECN(jac_set_zero)(C, out_jac);
    if (n == 0) return;
    size_t chunk = n / (size_t)(threads > 0 ? threads : 1);
    if (chunk == 0) chunk = n;
    for (size_t lo = 0; lo < n; lo += chunk) {
        size_t hi = lo + chunk > n ? n : lo + chunk;
        u64 part[3 * FW];
        ECN(msm_serial_u64)(C, part, bases + lo * 2 * FW, scalars + lo, hi - lo, 64);
        ECN(jac_add)(C, out_jac, part);
    }
}
/* mod.rs:373-390 msm_binary */
static void ECN(msm_binary)(const ark_curve_ctx *C, u64 *out_jac, const u64 *bases, const u64 *flags, size_t n, int threads) {
    ECN(jac_set_zero)(C, out_jac);
    if (n == 0) return;
    size_t chunk = n / (size_t)(threads > 0 ? threads : 1);
    if (chunk == 0) chunk = n;
    for (size_t lo = 0; lo < n; lo += chunk) {
        size_t hi = lo + chunk > n ? n : lo + chunk;
        u64 res[4 * FW];
        ECN(bkt_set_zero)(C, res);
        for (size_t i = lo; i < hi; i++)
            if (flags[i]) ECN(bkt_add_affine)(C, res, bases + i * 2 * FW, 0);
        ECN(jac_add_bkt)(C, out_jac, res);
    }
}

/* mod.rs:242-347 msm_signed: classify scalars by magnitude of s and r - s, dispatch, add - sub */
static void ECN(msm_signed)(const ark_curve_ctx *C, u64 *out_jac, const u64 *bases, const u64 *scalars, size_t size, int threads) {
    enum { U1, NEGU1, U8, NEGU8, U16, NEGU16, U32, NEGU32, U64_, NEGU64, BIG, NGROUPS };
    size_t cnt[NGROUPS] = {0};
    uint8_t *grp = (uint8_t *)malloc(size ? size : 1);
    u64 *val = (u64 *)malloc((size ? size : 1) * 8);
    for (size_t i = 0; i < size; i++) {
        const u64 *s = scalars + i * SCALAR_LIMBS;
        if (bigint_is_zero(s, SCALAR_LIMBS)) {
            grp[i] = 255;
            continue;
        }
        int nb = bigint_num_bits(s, SCALAR_LIMBS), g;
        u64 v = s[0];
        if (nb <= 1) g = U1;
        else if (nb <= 8) g = U8;
        else if (nb <= 16) g = U16;
        else if (nb <= 32) g = U32;
        else if (nb <= 64) g = U64_;
        else {
            u64 neg[SCALAR_LIMBS];
            u64 borrow = 0;
            for (int k = 0; k < SCALAR_LIMBS; k++) neg[k] = sbb64(C->S->p[k], s[k], &borrow);
            int nn = bigint_num_bits(neg, SCALAR_LIMBS);
            v = neg[0];
            if (nn <= 1) g = NEGU1;
            else if (nn <= 8) g = NEGU8;
            else if (nn <= 16) g = NEGU16;
            else if (nn <= 32) g = NEGU32;
            else if (nn <= 64) g = NEGU64;
            else g = BIG;
        }
        grp[i] = (uint8_t)g;
        val[i] = v;
        cnt[g]++;
    }
    u64 add_res[3 * FW], sub_res[3 * FW], part[3 * FW];
    ECN(jac_set_zero)(C, add_res);
    ECN(jac_set_zero)(C, sub_res);
    for (int g = 0; g < NGROUPS; g++) {
        size_t m = cnt[g];
        if (m == 0) continue;
        u64 *gb = (u64 *)malloc(m * 2 * FW * 8);
        u64 *gs = (u64 *)malloc(m * SCALAR_LIMBS * 8);
        size_t k = 0;
        for (size_t i = 0; i < size; i++) {
            if (grp[i] != g) continue;
            memcpy(gb + k * 2 * FW, bases + i * 2 * FW, 2 * FW * 8);
            if (g == BIG)
                memcpy(gs + k * SCALAR_LIMBS, scalars + i * SCALAR_LIMBS, SCALAR_LIMBS * 8);
            else
                gs[k] = val[i];
            k++;
        }
        if (g == U1 || g == NEGU1) {
            for (size_t q = 0; q < m; q++) gs[q] = (gs[q] == 1);
            ECN(msm_binary)(C, part, gb, gs, m, threads);
        } else if (g == BIG) {
            ECN(msm_wnaf)(C, part, gb, gs, m, threads); /* NEGATION_IS_CHEAP = true for SW (group.rs:643) */
        } else {
            ECN(msm_small)(C, part, gb, gs, m, threads);
        }
        ECN(jac_add)(C, (g == BIG || (g % 2) == 0) ? add_res : sub_res, part);
        free(gb);
        free(gs);
    }
    ECN(jac_sub)(C, add_res, sub_res);
    memcpy(out_jac, add_res, sizeof add_res);
    free(grp);
    free(val);
}

/* group.rs:302-319 normalize_batch via ff/src/fields/mod.rs:358-385 batch_inversion (Montgomery's trick) */
static void ECN(normalize_batch)(const ark_curve_ctx *C, u64 *out_aff, const u64 *jac, size_t n) {
    u64 *prod = (u64 *)malloc((n ? n : 1) * FW * 8);
    u64 acc[FW], inv[FW], t[FW];
    ECN(fe_one)(C, acc);
    for (size_t i = 0; i < n; i++) {
        const u64 *z = JZ(jac + i * 3 * FW);
        if (!ECN(fe_is_zero)(z)) FE(mul)(C, acc, acc, z);
        ECN(fe_copy)(prod + i * FW, acc);
    }
    FE(inv)(C, inv, acc, acc);
    for (size_t i = n; i-- > 0;) {
        const u64 *p = jac + i * 3 * FW;
        u64 *o = out_aff + i * 2 * FW;
        if (ECN(fe_is_zero)(JZ(p))) {
            memset(o, 0, 2 * FW * 8);
            continue;
        }
        u64 zi[FW], zi2[FW];
        if (i == 0) {
            ECN(fe_copy)(zi, inv);
        } else {
            /* previous non-skipped prefix product is prod[i-1] (prefix carries through zeros) */
            FE(mul)(C, zi, inv, prod + (i - 1) * FW);
        }
        FE(mul)(C, inv, inv, JZ(p));
        FE(mul)(C, zi2, zi, zi);
        FE(mul)(C, AX(o), AX(p), zi2);
        FE(mul)(C, t, zi2, zi);
        FE(mul)(C, AY(o), AY(p), t);
    }
    free(prod);
}

/* ec/src/scalar_mul/mod.rs:156-245 BatchMulPreprocessing::new(base, num_scalars) + batch_mul: the table of
 * multiples inner * 2^(window*outer) * base (window = compute_window_size, :222-228), windowed_mul (:235-251) per scalar
 * (canonical bigints here: the reference's into_bigint() is the caller's), batch_convert_to_mul_base at the end. */
static void ECN(batch_mul)(const ark_curve_ctx *C, u64 *out_aff, const u64 *base_jac, const u64 *scalars, size_t n,
                           int max_scalar_size) {
    const size_t window = n < 32 ? 3 : ln_without_floats(n);
    const size_t in_window = (size_t)1 << window;
    const size_t outerc = ((size_t)max_scalar_size + window - 1) / window;
    const size_t last_in_window = (size_t)1 << ((size_t)max_scalar_size - (outerc - 1) * window);
    u64 *table = (u64 *)malloc(outerc * in_window * 2 * FW * 8); /* affine (batch_convert_to_mul_base) */
    u64 g_outer[3 * FW];
    memcpy(g_outer, base_jac, 3 * FW * 8);
    for (size_t outer = 0; outer < outerc; outer++) {
        const size_t cur = outer == outerc - 1 ? last_in_window : in_window;
        u64 g_inner[3 * FW];
        ECN(jac_set_zero)(C, g_inner);
        for (size_t inner = 0; inner < in_window; inner++) {
            u64 *cell = table + (outer * in_window + inner) * 2 * FW;
            if (inner < cur) {
                ECN(jac_to_aff)(C, cell, g_inner);
                ECN(jac_add)(C, g_inner, g_outer);
            } else {
                memset(cell, 0, 2 * FW * 8); /* T::zero() */
            }
        }
        for (size_t k = 0; k < window; k++) ECN(jac_double)(C, g_outer);
    }
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) {
        const u64 *sc = scalars + i * SCALAR_LIMBS;
        u64 res[3 * FW];
        ECN(jac_set_zero)(C, res); /* T::from(table[0][0]) = zero */
        for (size_t outer = 0; outer < outerc; outer++) {
            size_t inner = 0;
            for (size_t k = 0; k < window; k++) {
                const size_t bit = outer * window + k;
                if (bit < (size_t)max_scalar_size && ((sc[bit / 64] >> (bit % 64)) & 1)) inner |= (size_t)1 << k;
            }
            u64 t[3 * FW];
            ECN(aff_to_jac)(C, t, table + (outer * in_window + inner) * 2 * FW);
            ECN(jac_add)(C, res, t);
        }
        ECN(jac_to_aff)(C, out_aff + i * 2 * FW, res);
    }
    free(table);
}

/* bases P_i = (a + i*b) * G, i = 0..n-1, as affine points (SURVEY.md section 8d synthetic inputs) */
static void ECN(gen_bases)(const ark_curve_ctx *C, u64 *out_aff, const u64 *a, const u64 *b, size_t n) {
    u64 g[2 * FW], cur[3 * FW], step[3 * FW];
    ECN(fe_copy)(AX(g), C->gx);
    ECN(fe_copy)(AY(g), C->gy);
    ECN(scalar_mul)(C, cur, g, a, SCALAR_LIMBS);
    ECN(scalar_mul)(C, step, g, b, SCALAR_LIMBS);
    const size_t CH = 4096;
    u64 *buf = (u64 *)malloc(CH * 3 * FW * 8);
    for (size_t lo = 0; lo < n; lo += CH) {
        size_t m = lo + CH > n ? n - lo : CH;
        for (size_t i = 0; i < m; i++) {
            memcpy(buf + i * 3 * FW, cur, 3 * FW * 8);
            ECN(jac_add)(C, cur, step);
        }
        ECN(normalize_batch)(C, out_aff + lo * 2 * FW, buf, m);
    }
    free(buf);
}

#undef AX
#undef AY
#undef JZ
#undef BZZ
#undef BZZZ
#undef ECN
