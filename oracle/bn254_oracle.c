/*
 * oracle/bn254_oracle.c -- TEST INFRASTRUCTURE ONLY.  Never linked, loaded or called by the
 * product path (only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it).
 *
 * Plain-C CPU restatement of the reference hot path (tokamak-network/interactive-zkp-study):
 * BN254 G1/G2 "MSM" as the reference computes it -- a loop of scalar multiplications followed
 * by additions (zkp/plonk/kzg.py:59-65, zkp/groth16/proving.py:23-75) -- and the radix-2
 * recursive NTT of zkp/plonk/polynomial.py:292-378.  The group/field arithmetic itself lives
 * in the third-party py-ecc==7.0.1 (requirements.txt:14), which is not vendored in the
 * reference tree; its published affine algorithm is restated in oracle/py_ref.py, and this
 * file computes the same group elements with Jacobian coordinates + one final inversion
 * (group law is exact, canonical affine outputs are identical).
 *
 * Deliberately independent of the product code: 4 x 64-bit limbs with unsigned __int128,
 * its own Montgomery constants computed at start-up from the moduli below.
 *
 * PARITY: "unpinned" for EC coordinates / NTT vectors w.r.t. the real py_ecc (not importable
 * here, and the reference's tests hold no expected coordinates: SURVEY.md section 8c); pinned
 * against oracle/py_ref.py, the F_r KATs of zkp/groth16/backend.py:355,363-367 and the
 * relational identities of the reference's tests (see tests/test_oracle.py).
 *
 * ABI: field elements are 4 x uint64 little-endian canonical residues; G1 affine = x||y
 * (8 limbs), G2 affine = x.c0||x.c1||y.c0||y.c1 (16 limbs); infinity = all-zero coordinates.
 */
#include <stdint.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef struct { uint64_t l[4]; } fe;
typedef struct { uint64_t m[4]; uint64_t inv; fe r1, r2; } field_t;

static field_t FP, FR;
static int g_init = 0;

static const uint64_t P_LIMBS[4] = {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
static const uint64_t R_LIMBS[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};

static int fe_is_zero(const fe *a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0; }
static int fe_eq(const fe *a, const fe *b) { return !memcmp(a, b, sizeof(fe)); }

static int geq(const uint64_t a[4], const uint64_t b[4]) {
    for (int i = 3; i >= 0; i--) {
        if (a[i] > b[i]) return 1;
        if (a[i] < b[i]) return 0;
    }
    return 1;
}
static void sub_nb(uint64_t r[4], const uint64_t a[4], const uint64_t b[4]) {
    u128 br = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)a[i] - b[i] - br;
        r[i] = (uint64_t)d;
        br = (d >> 64) & 1;
    }
}
static void f_add(const field_t *F, fe *r, const fe *a, const fe *b) {
    u128 c = 0;
    fe t;
    for (int i = 0; i < 4; i++) {
        c += (u128)a->l[i] + b->l[i];
        t.l[i] = (uint64_t)c;
        c >>= 64;
    }
    if (geq(t.l, F->m)) sub_nb(t.l, t.l, F->m);
    *r = t;
}
static void f_sub(const field_t *F, fe *r, const fe *a, const fe *b) {
    fe t;
    if (geq(a->l, b->l)) {
        sub_nb(t.l, a->l, b->l);
    } else {
        fe u;
        sub_nb(u.l, b->l, a->l);
        sub_nb(t.l, F->m, u.l);
    }
    *r = t;
}
static void f_neg(const field_t *F, fe *r, const fe *a) {
    if (fe_is_zero(a)) { *r = *a; return; }
    sub_nb(r->l, F->m, a->l);
}
/* Montgomery product, radix 2^256, 64-bit limbs (separate operand scanning). */
static void f_mul(const field_t *F, fe *r, const fe *a, const fe *b) {
    uint64_t t[9] = {0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) {
            c += (u128)a->l[j] * b->l[i] + t[i + j];
            t[i + j] = (uint64_t)c;
            c >>= 64;
        }
        t[i + 4] = (uint64_t)c;
    }
    for (int i = 0; i < 4; i++) {
        uint64_t m = t[i] * F->inv;
        u128 c = 0;
        for (int j = 0; j < 4; j++) {
            c += (u128)m * F->m[j] + t[i + j];
            t[i + j] = (uint64_t)c;
            c >>= 64;
        }
        for (int k = i + 4; k < 9 && c; k++) {
            c += t[k];
            t[k] = (uint64_t)c;
            c >>= 64;
        }
    }
    fe o = {{t[4], t[5], t[6], t[7]}};
    if (t[8] || geq(o.l, F->m)) sub_nb(o.l, o.l, F->m);
    *r = o;
}
static void f_pow(const field_t *F, fe *r, const fe *a, const uint64_t e[4]) {
    fe acc = F->r1;
    for (int i = 3; i >= 0; i--)
        for (int b = 63; b >= 0; b--) {
            f_mul(F, &acc, &acc, &acc);
            if ((e[i] >> b) & 1) f_mul(F, &acc, &acc, a);
        }
    *r = acc;
}
static void f_inv(const field_t *F, fe *r, const fe *a) {
    uint64_t e[4];
    uint64_t two[4] = {2, 0, 0, 0};
    sub_nb(e, F->m, two);
    f_pow(F, r, a, e);
}
static void f_to_mont(const field_t *F, fe *r, const fe *a) { f_mul(F, r, a, &F->r2); }
static void f_from_mont(const field_t *F, fe *r, const fe *a) {
    fe one = {{1, 0, 0, 0}};
    f_mul(F, r, a, &one);
}

static void field_init(field_t *F, const uint64_t m[4]) {
    memcpy(F->m, m, 32);
    uint64_t inv = 1;
    for (int i = 0; i < 6; i++) inv *= 2 - m[0] * inv; /* Newton: m^-1 mod 2^64 */
    F->inv = (uint64_t)0 - inv;
    /* r1 = 2^256 mod m by 256 modular doublings of 1; r2 = 2^512 mod m by 256 more. */
    fe x = {{1, 0, 0, 0}};
    for (int i = 0; i < 512; i++) {
        f_add(F, &x, &x, &x);
        if (i == 255) F->r1 = x;
    }
    F->r2 = x;
}
static void ensure_init(void) {
    if (g_init) return;
    field_init(&FP, P_LIMBS);
    field_init(&FR, R_LIMBS);
    g_init = 1;
}

/* ------------------------------------------------------------------ F_p^2 (py_ecc FQ2, i^2 = -1) */
typedef struct { fe c0, c1; } fe2;
static int fe2_is_zero(const fe2 *a) { return fe_is_zero(&a->c0) && fe_is_zero(&a->c1); }
static void f2_add(fe2 *r, const fe2 *a, const fe2 *b) { f_add(&FP, &r->c0, &a->c0, &b->c0); f_add(&FP, &r->c1, &a->c1, &b->c1); }
static void f2_sub(fe2 *r, const fe2 *a, const fe2 *b) { f_sub(&FP, &r->c0, &a->c0, &b->c0); f_sub(&FP, &r->c1, &a->c1, &b->c1); }
static void f2_neg(fe2 *r, const fe2 *a) { f_neg(&FP, &r->c0, &a->c0); f_neg(&FP, &r->c1, &a->c1); }
static void f2_mul(fe2 *r, const fe2 *a, const fe2 *b) {
    fe t0, t1, t2, t3;
    f_mul(&FP, &t0, &a->c0, &b->c0);
    f_mul(&FP, &t1, &a->c1, &b->c1);
    f_mul(&FP, &t2, &a->c0, &b->c1);
    f_mul(&FP, &t3, &a->c1, &b->c0);
    f_sub(&FP, &r->c0, &t0, &t1);
    f_add(&FP, &r->c1, &t2, &t3);
}
static void f2_inv(fe2 *r, const fe2 *a) {
    fe n0, n1, d;
    f_mul(&FP, &n0, &a->c0, &a->c0);
    f_mul(&FP, &n1, &a->c1, &a->c1);
    f_add(&FP, &d, &n0, &n1);
    f_inv(&FP, &d, &d);
    f_mul(&FP, &r->c0, &a->c0, &d);
    fe t;
    f_mul(&FP, &t, &a->c1, &d);
    f_neg(&FP, &r->c1, &t);
}

/* ------------------------------------------------------------------ Jacobian points, generic via macros
 * (X, Y, Z), x = X/Z^2, y = Y/Z^3; infinity Z = 0.  dbl-2009-l and add-2007-bl (a = 0).         */
#define DEFINE_GROUP(NAME, T, ISZ, ADD, SUB, MUL, NEGF)                                              \
    typedef struct { T x, y, z; } NAME##_jac;                                                        \
    static void NAME##_dbl(NAME##_jac *r, const NAME##_jac *p) {                                     \
        if (ISZ(&p->z) || ISZ(&p->y)) { memset(r, 0, sizeof(*r)); return; }                          \
        T a, b, c, d, e, f, t, x3, y3, z3;                                                           \
        MUL(&a, &p->x, &p->x);                                                                       \
        MUL(&b, &p->y, &p->y);                                                                       \
        MUL(&c, &b, &b);                                                                             \
        ADD(&t, &p->x, &b); MUL(&t, &t, &t); SUB(&t, &t, &a); SUB(&t, &t, &c); ADD(&d, &t, &t);      \
        ADD(&e, &a, &a); ADD(&e, &e, &a);                                                            \
        MUL(&f, &e, &e);                                                                             \
        SUB(&x3, &f, &d); SUB(&x3, &x3, &d);                                                         \
        SUB(&t, &d, &x3); MUL(&y3, &e, &t);                                                          \
        ADD(&t, &c, &c); ADD(&t, &t, &t); ADD(&t, &t, &t); SUB(&y3, &y3, &t);                        \
        MUL(&z3, &p->y, &p->z); ADD(&z3, &z3, &z3);                                                  \
        r->x = x3; r->y = y3; r->z = z3;                                                             \
    }                                                                                                \
    static void NAME##_add(NAME##_jac *r, const NAME##_jac *p, const NAME##_jac *q) {                \
        if (ISZ(&p->z)) { *r = *q; return; }                                                         \
        if (ISZ(&q->z)) { *r = *p; return; }                                                         \
        T z1z1, z2z2, u1, u2, s1, s2, h, rr, t, hh, hhh, v, x3, y3, z3;                              \
        MUL(&z1z1, &p->z, &p->z); MUL(&z2z2, &q->z, &q->z);                                          \
        MUL(&u1, &p->x, &z2z2); MUL(&u2, &q->x, &z1z1);                                              \
        MUL(&t, &q->z, &z2z2); MUL(&s1, &p->y, &t);                                                  \
        MUL(&t, &p->z, &z1z1); MUL(&s2, &q->y, &t);                                                  \
        SUB(&h, &u2, &u1); SUB(&rr, &s2, &s1);                                                       \
        if (ISZ(&h)) {                                                                               \
            if (ISZ(&rr)) { NAME##_dbl(r, p); return; }                                              \
            memset(r, 0, sizeof(*r)); return;                                                        \
        }                                                                                            \
        MUL(&hh, &h, &h); MUL(&hhh, &hh, &h); MUL(&v, &u1, &hh);                                     \
        MUL(&x3, &rr, &rr); SUB(&x3, &x3, &hhh); SUB(&x3, &x3, &v); SUB(&x3, &x3, &v);               \
        SUB(&t, &v, &x3); MUL(&y3, &rr, &t); MUL(&t, &s1, &hhh); SUB(&y3, &y3, &t);                  \
        MUL(&z3, &p->z, &q->z); MUL(&z3, &z3, &h);                                                   \
        r->x = x3; r->y = y3; r->z = z3;                                                             \
    }                                                                                                \
    /* k*P: MSB-first double-and-add (same group element as py_ecc's recursive multiply) */         \
    static void NAME##_mul(NAME##_jac *r, const NAME##_jac *p, const uint64_t k[4]) {                \
        NAME##_jac acc; memset(&acc, 0, sizeof(acc));                                                \
        for (int i = 3; i >= 0; i--)                                                                 \
            for (int b = 63; b >= 0; b--) {                                                          \
                NAME##_dbl(&acc, &acc);                                                              \
                if ((k[i] >> b) & 1) NAME##_add(&acc, &acc, p);                                      \
            }                                                                                        \
        *r = acc;                                                                                    \
    }

static void fp_add_(fe *r, const fe *a, const fe *b) { f_add(&FP, r, a, b); }
static void fp_sub_(fe *r, const fe *a, const fe *b) { f_sub(&FP, r, a, b); }
static void fp_mul_(fe *r, const fe *a, const fe *b) { f_mul(&FP, r, a, b); }
static void fp_neg_(fe *r, const fe *a) { f_neg(&FP, r, a); }

DEFINE_GROUP(g1, fe, fe_is_zero, fp_add_, fp_sub_, fp_mul_, fp_neg_)
DEFINE_GROUP(g2, fe2, fe2_is_zero, f2_add, f2_sub, f2_mul, f2_neg)

/* affine canonical <-> Jacobian Montgomery */
static void g1_load(g1_jac *p, const uint64_t xy[8]) {
    fe x, y;
    memcpy(&x, xy, 32); memcpy(&y, xy + 4, 32);
    if (fe_is_zero(&x) && fe_is_zero(&y)) { memset(p, 0, sizeof(*p)); return; }
    f_to_mont(&FP, &p->x, &x); f_to_mont(&FP, &p->y, &y); p->z = FP.r1;
}
static void g1_store(uint64_t xy[8], const g1_jac *p) {
    if (fe_is_zero(&p->z)) { memset(xy, 0, 64); return; }
    fe zi, zi2, zi3, x, y;
    f_inv(&FP, &zi, &p->z);
    f_mul(&FP, &zi2, &zi, &zi); f_mul(&FP, &zi3, &zi2, &zi);
    f_mul(&FP, &x, &p->x, &zi2); f_mul(&FP, &y, &p->y, &zi3);
    f_from_mont(&FP, &x, &x); f_from_mont(&FP, &y, &y);
    memcpy(xy, &x, 32); memcpy(xy + 4, &y, 32);
}
static void g2_load(g2_jac *p, const uint64_t v[16]) {
    fe2 x, y;
    memcpy(&x, v, 64); memcpy(&y, v + 8, 64);
    if (fe2_is_zero(&x) && fe2_is_zero(&y)) { memset(p, 0, sizeof(*p)); return; }
    f_to_mont(&FP, &p->x.c0, &x.c0); f_to_mont(&FP, &p->x.c1, &x.c1);
    f_to_mont(&FP, &p->y.c0, &y.c0); f_to_mont(&FP, &p->y.c1, &y.c1);
    p->z.c0 = FP.r1; memset(&p->z.c1, 0, 32);
}
static void g2_store(uint64_t v[16], const g2_jac *p) {
    if (fe2_is_zero(&p->z)) { memset(v, 0, 128); return; }
    fe2 zi, zi2, zi3, x, y;
    f2_inv(&zi, &p->z);
    f2_mul(&zi2, &zi, &zi); f2_mul(&zi3, &zi2, &zi);
    f2_mul(&x, &p->x, &zi2); f2_mul(&y, &p->y, &zi3);
    f_from_mont(&FP, &x.c0, &x.c0); f_from_mont(&FP, &x.c1, &x.c1);
    f_from_mont(&FP, &y.c0, &y.c0); f_from_mont(&FP, &y.c1, &y.c1);
    memcpy(v, &x, 64); memcpy(v + 8, &y, 64);
}

/* =========================================================================== exported API */

/* out = a (op) b in F_r / F_p, canonical in/out.  which: 0 = F_p, 1 = F_r; op: 0 add 1 sub 2 mul 3 inv(a) */
void orc_field_op(int which, int op, const uint64_t a[4], const uint64_t b[4], uint64_t out[4]) {
    ensure_init();
    const field_t *F = which ? &FR : &FP;
    fe x, y, r;
    memcpy(&x, a, 32);
    if (b) memcpy(&y, b, 32); else memset(&y, 0, 32);
    f_to_mont(F, &x, &x); f_to_mont(F, &y, &y);
    switch (op) {
        case 0: f_add(F, &r, &x, &y); break;
        case 1: f_sub(F, &r, &x, &y); break;
        case 2: f_mul(F, &r, &x, &y); break;
        default: f_inv(F, &r, &x); break;
    }
    f_from_mont(F, &r, &r);
    memcpy(out, &r, 32);
}

/* bn128.multiply(P, k) on G1 -- call sites zkp/plonk/field.py:88, zkp/groth16/setup.py:16-68 */
void orc_g1_mul(const uint64_t p[8], const uint64_t k[4], uint64_t out[8]) {
    ensure_init();
    g1_jac a, r;
    g1_load(&a, p);
    g1_mul(&r, &a, k);
    g1_store(out, &r);
}
/* bn128.add(P, Q) on G1 -- zkp/plonk/field.py:103 */
void orc_g1_add(const uint64_t p[8], const uint64_t q[8], uint64_t out[8]) {
    ensure_init();
    g1_jac a, b, r;
    g1_load(&a, p); g1_load(&b, q);
    g1_add(&r, &a, &b);
    g1_store(out, &r);
}
void orc_g2_mul(const uint64_t p[16], const uint64_t k[4], uint64_t out[16]) {
    ensure_init();
    g2_jac a, r;
    g2_load(&a, p);
    g2_mul(&r, &a, k);
    g2_store(out, &r);
}
void orc_g2_add(const uint64_t p[16], const uint64_t q[16], uint64_t out[16]) {
    ensure_init();
    g2_jac a, b, r;
    g2_load(&a, p); g2_load(&b, q);
    g2_add(&r, &a, &b);
    g2_store(out, &r);
}

/* The reference's "MSM": result = None; for i: if c_i != 0: result = add(result, multiply(P_i, c_i))
 * (zkp/plonk/kzg.py:59-65; same shape as the inner loops of zkp/groth16/proving.py:27-31). */
void orc_g1_msm(const uint64_t *scalars, const uint64_t *points, size_t n, uint64_t out[8]) {
    ensure_init();
    g1_jac acc, p, t;
    memset(&acc, 0, sizeof(acc));
    for (size_t i = 0; i < n; i++) {
        const uint64_t *k = scalars + 4 * i;
        if (!(k[0] | k[1] | k[2] | k[3])) continue;
        g1_load(&p, points + 8 * i);
        g1_mul(&t, &p, k);
        g1_add(&acc, &acc, &t);
    }
    g1_store(out, &acc);
}
void orc_g2_msm(const uint64_t *scalars, const uint64_t *points, size_t n, uint64_t out[16]) {
    ensure_init();
    g2_jac acc, p, t;
    memset(&acc, 0, sizeof(acc));
    for (size_t i = 0; i < n; i++) {
        const uint64_t *k = scalars + 4 * i;
        if (!(k[0] | k[1] | k[2] | k[3])) continue;
        g2_load(&p, points + 16 * i);
        g2_mul(&t, &p, k);
        g2_add(&acc, &acc, &t);
    }
    g2_store(out, &acc);
}

/* Bucket-method (Pippenger) G1 MSM, unsigned c-bit windows -- the textbook algorithm, serial, used as
 * (i) a second, structurally different checker for large inputs where orc_g1_msm is too slow and
 * (ii) bench.py's "compiled C, one core" CPU line.  Same group element as orc_g1_msm. */
static unsigned window_digit(const uint64_t k[4], unsigned lo, unsigned c) {
    unsigned word = lo >> 6, off = lo & 63;
    uint64_t v = k[word] >> off;
    if (off + c > 64 && word < 3) v |= k[word + 1] << (64 - off);
    return (unsigned)(v & ((1ULL << c) - 1));
}
int orc_g1_msm_bucket(const uint64_t *scalars, const uint64_t *points, size_t n, unsigned c, uint64_t out[8]) {
    ensure_init();
    if (c < 1 || c > 20) return -1;
    const unsigned windows = (254 + c - 1) / c;
    const size_t nb = ((size_t)1 << c) - 1;
    g1_jac *buckets = (g1_jac *)malloc(sizeof(g1_jac) * nb);
    g1_jac *pts = (g1_jac *)malloc(sizeof(g1_jac) * (n ? n : 1));
    if (!buckets || !pts) { free(buckets); free(pts); return -2; }
    for (size_t i = 0; i < n; i++) g1_load(&pts[i], points + 8 * i);
    g1_jac total, run, wsum;
    memset(&total, 0, sizeof(total));
    for (int w = (int)windows - 1; w >= 0; w--) {
        for (unsigned d = 0; d < c; d++) g1_dbl(&total, &total);
        memset(buckets, 0, sizeof(g1_jac) * nb);
        for (size_t i = 0; i < n; i++) {
            unsigned d = window_digit(scalars + 4 * i, (unsigned)w * c, c);
            if (d) g1_add(&buckets[d - 1], &buckets[d - 1], &pts[i]);
        }
        memset(&run, 0, sizeof(run));
        memset(&wsum, 0, sizeof(wsum));
        for (size_t b = nb; b-- > 0;) {       /* sum_b (b+1) * bucket[b] by running sums */
            g1_add(&run, &run, &buckets[b]);
            g1_add(&wsum, &wsum, &run);
        }
        g1_add(&total, &total, &wsum);
    }
    g1_store(out, &total);
    free(buckets);
    free(pts);
    return 0;
}

/* The same textbook bucket method in G2 (proof_b's query, zkp/groth16/proving.py:35-45, at sizes where orc_g2_msm's
 * per-term double-and-add would take minutes).  Same group element as orc_g2_msm (tests/test_oracle.py). */
int orc_g2_msm_bucket(const uint64_t *scalars, const uint64_t *points, size_t n, unsigned c, uint64_t out[16]) {
    ensure_init();
    if (c < 1 || c > 20) return -1;
    const unsigned windows = (254 + c - 1) / c;
    const size_t nb = ((size_t)1 << c) - 1;
    g2_jac *buckets = (g2_jac *)malloc(sizeof(g2_jac) * nb);
    g2_jac *pts = (g2_jac *)malloc(sizeof(g2_jac) * (n ? n : 1));
    if (!buckets || !pts) { free(buckets); free(pts); return -2; }
    for (size_t i = 0; i < n; i++) g2_load(&pts[i], points + 16 * i);
    g2_jac total, run, wsum;
    memset(&total, 0, sizeof(total));
    for (int w = (int)windows - 1; w >= 0; w--) {
        for (unsigned d = 0; d < c; d++) g2_dbl(&total, &total);
        memset(buckets, 0, sizeof(g2_jac) * nb);
        for (size_t i = 0; i < n; i++) {
            unsigned d = window_digit(scalars + 4 * i, (unsigned)w * c, c);
            if (d) g2_add(&buckets[d - 1], &buckets[d - 1], &pts[i]);
        }
        memset(&run, 0, sizeof(run));
        memset(&wsum, 0, sizeof(wsum));
        for (size_t b = nb; b-- > 0;) {
            g2_add(&run, &run, &buckets[b]);
            g2_add(&wsum, &wsum, &run);
        }
        g2_add(&total, &total, &wsum);
    }
    g2_store(out, &total);
    free(buckets);
    free(pts);
    return 0;
}

/* Fixed-base batch in G2: out[i] = k_i * P  (sigma22 zkp/groth16/setup.py:65-69) */
void orc_g2_fixed_base(const uint64_t p[16], const uint64_t *scalars, size_t n, uint64_t *out) {
    for (size_t i = 0; i < n; i++) orc_g2_mul(p, scalars + 4 * i, out + 16 * i);
}

/* The same bucket method with the windows spread over `threads` POSIX threads (windows are independent; the window sums
 * are combined by one Horner pass) -- bench.py's "all the cores a one-GPU box gives us" CPU line. */
typedef struct {
    const uint64_t *scalars;
    const g1_jac *pts;
    size_t n;
    unsigned c, windows, first, stride;
    g1_jac *wsum;
    int rc;
} msm_mt_job;
static void *msm_mt_worker(void *arg) {
    msm_mt_job *J = (msm_mt_job *)arg;
    const size_t nb = ((size_t)1 << J->c) - 1;
    g1_jac *buckets = (g1_jac *)malloc(sizeof(g1_jac) * nb);
    if (!buckets) { J->rc = -2; return NULL; }
    for (unsigned w = J->first; w < J->windows; w += J->stride) {
        g1_jac run, wsum;
        memset(buckets, 0, sizeof(g1_jac) * nb);
        for (size_t i = 0; i < J->n; i++) {
            unsigned d = window_digit(J->scalars + 4 * i, w * J->c, J->c);
            if (d) g1_add(&buckets[d - 1], &buckets[d - 1], &J->pts[i]);
        }
        memset(&run, 0, sizeof(run));
        memset(&wsum, 0, sizeof(wsum));
        for (size_t b = nb; b-- > 0;) {
            g1_add(&run, &run, &buckets[b]);
            g1_add(&wsum, &wsum, &run);
        }
        J->wsum[w] = wsum;
    }
    free(buckets);
    return NULL;
}
int orc_g1_msm_bucket_mt(const uint64_t *scalars, const uint64_t *points, size_t n, unsigned c, unsigned threads, uint64_t out[8]) {
    ensure_init();
    if (c < 1 || c > 20 || threads < 1 || threads > 64) return -1;
    const unsigned windows = (254 + c - 1) / c;
    if (threads > windows) threads = windows;
    g1_jac *pts = (g1_jac *)malloc(sizeof(g1_jac) * (n ? n : 1));
    g1_jac *wsum = (g1_jac *)calloc(windows, sizeof(g1_jac));
    if (!pts || !wsum) { free(pts); free(wsum); return -2; }
    for (size_t i = 0; i < n; i++) g1_load(&pts[i], points + 8 * i);
    pthread_t tid[64];
    msm_mt_job jobs[64];
    int rc = 0;
    for (unsigned t = 0; t < threads; t++) {
        msm_mt_job j = {scalars, pts, n, c, windows, t, threads, wsum, 0};
        jobs[t] = j;
        if (pthread_create(&tid[t], NULL, msm_mt_worker, &jobs[t])) { jobs[t].rc = -3; msm_mt_worker(&jobs[t]); tid[t] = 0; }
    }
    for (unsigned t = 0; t < threads; t++) {
        if (tid[t]) pthread_join(tid[t], NULL);
        if (jobs[t].rc && jobs[t].rc != -3) rc = jobs[t].rc;
    }
    g1_jac total;
    memset(&total, 0, sizeof(total));
    for (int w = (int)windows - 1; w >= 0; w--) {
        for (unsigned d = 0; d < c; d++) g1_dbl(&total, &total);
        g1_add(&total, &total, &wsum[w]);
    }
    g1_store(out, &total);
    free(pts);
    free(wsum);
    return rc;
}

/* Fixed-base batch: out[i] = k_i * P  (SRS.generate zkp/plonk/srs.py:77-82, sigma12 setup.py:18-23) */
void orc_g1_fixed_base(const uint64_t p[8], const uint64_t *scalars, size_t n, uint64_t *out) {
    for (size_t i = 0; i < n; i++) orc_g1_mul(p, scalars + 4 * i, out + 8 * i);
}

/* fft(coeffs, omega): zkp/plonk/polynomial.py:292-341 -- recursive radix-2 DIT, natural order
 * in and out.  v: n Montgomery-form elements, w: omega (Montgomery), tmp: n scratch. */
static void fft_rec(fe *v, size_t n, const fe *w, fe *tmp) {
    if (n == 1) return;
    size_t h = n / 2;
    for (size_t i = 0; i < h; i++) { tmp[i] = v[2 * i]; tmp[h + i] = v[2 * i + 1]; }
    fe w2;
    f_mul(&FR, &w2, w, w);
    fft_rec(tmp, h, &w2, v);
    fft_rec(tmp + h, h, &w2, v + h);
    fe wk = FR.r1;
    for (size_t k = 0; k < h; k++) {
        fe t;
        f_mul(&FR, &t, &wk, &tmp[h + k]);
        f_add(&FR, &v[k], &tmp[k], &t);
        f_sub(&FR, &v[k + h], &tmp[k], &t);
        f_mul(&FR, &wk, &wk, w);
    }
}
/* data: n = 2^log_n canonical F_r elements, transformed in place with the given omega
 * (canonical).  inverse != 0: ifft of zkp/plonk/polynomial.py:344-378 (omega^-1, then * n^-1). */
int orc_ntt(uint64_t *data, unsigned log_n, const uint64_t omega[4], int inverse) {
    ensure_init();
    size_t n = (size_t)1 << log_n;
    fe *v = (fe *)malloc(n * sizeof(fe)), *tmp = (fe *)malloc(n * sizeof(fe));
    if (!v || !tmp) { free(v); free(tmp); return -1; }
    for (size_t i = 0; i < n; i++) { memcpy(&v[i], data + 4 * i, 32); f_to_mont(&FR, &v[i], &v[i]); }
    fe w;
    memcpy(&w, omega, 32);
    f_to_mont(&FR, &w, &w);
    if (inverse) f_inv(&FR, &w, &w);
    fft_rec(v, n, &w, tmp);
    if (inverse) {
        fe nn = {{(uint64_t)n, 0, 0, 0}}, ninv;
        f_to_mont(&FR, &nn, &nn);
        f_inv(&FR, &ninv, &nn);
        for (size_t i = 0; i < n; i++) f_mul(&FR, &v[i], &v[i], &ninv);
    }
    for (size_t i = 0; i < n; i++) { f_from_mont(&FR, &v[i], &v[i]); memcpy(data + 4 * i, &v[i], 32); }
    free(v); free(tmp);
    return 0;
}

/* Horner evaluation (Polynomial.evaluate, zkp/plonk/polynomial.py:85-106), canonical in/out. */
void orc_fr_horner(const uint64_t *coeffs, size_t n, const uint64_t x[4], uint64_t out[4]) {
    ensure_init();
    fe acc = {{0, 0, 0, 0}}, xm, c;
    memcpy(&xm, x, 32);
    f_to_mont(&FR, &xm, &xm);
    for (size_t i = n; i-- > 0;) {
        memcpy(&c, coeffs + 4 * i, 32);
        f_to_mont(&FR, &c, &c);
        f_mul(&FR, &acc, &acc, &xm);
        f_add(&FR, &acc, &acc, &c);
    }
    f_from_mont(&FR, &acc, &acc);
    memcpy(out, &acc, 32);
}

/* sum_i a_i * b_i in F_r (closed-form MSM checks: (sum s_i k_i) * G). */
void orc_fr_dot(const uint64_t *a, const uint64_t *b, size_t n, uint64_t out[4]) {
    ensure_init();
    fe acc = {{0, 0, 0, 0}}, x, y, t;
    for (size_t i = 0; i < n; i++) {
        memcpy(&x, a + 4 * i, 32); memcpy(&y, b + 4 * i, 32);
        f_to_mont(&FR, &x, &x);
        f_mul(&FR, &t, &x, &y); /* (xR)*y/R = x*y canonical */
        f_add(&FR, &acc, &acc, &t);
    }
    memcpy(out, &acc, 32);
}

/* y = M x over F_r for a CSR matrix (row_ptr u32[rows+1], col u32[nnz], vals nnz x 4 limbs canonical): the per-constraint dot
 * products A.w, B.w, C.w of an R1CS -- what the reference's `_multiply_vec_matrix` (zkp/groth16/poly_utils.py:52-59) computes
 * from its dense matrices.  Canonical in and out. */
void orc_fr_spmv(const uint32_t *row_ptr, const uint32_t *col, const uint64_t *vals, const uint64_t *x, size_t rows, uint64_t *y) {
    ensure_init();
    for (size_t k = 0; k < rows; k++) {
        fe acc = {{0, 0, 0, 0}}, a, b, t;
        for (uint32_t e = row_ptr[k]; e < row_ptr[k + 1]; e++) {
            memcpy(&a, vals + 4 * (size_t)e, 32); memcpy(&b, x + 4 * (size_t)col[e], 32);
            f_to_mont(&FR, &a, &a);
            f_mul(&FR, &t, &a, &b); /* (aR) * b / R = a * b canonical */
            f_add(&FR, &acc, &acc, &t);
        }
        memcpy(y + 4 * k, &acc, 32);
    }
}

