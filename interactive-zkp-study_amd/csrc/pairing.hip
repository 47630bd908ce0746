// pairing.hip -- HOST-side BN254 optimal-ate pairing for the verifiers (Groth16 verify: 4 pairings,
// KZG opening check: 2 pairings).  A handful of pairings per proof is never a GPU target (SURVEY.md
// section 8 f1); this file contains no device code.
//
// Replaces py_ecc.bn128.pairing as the reference uses it: zkp/groth16/verifying.py:17-40,
// zkp/plonk/field.py:118-138, zkp/plonk/kzg.py:117-160.  The algorithm is py_ecc's
// (bn128_pairing.py): Miller loop over ate_loop_count = 6u+2 with affine line functions, the two
// Frobenius correction lines, final exponentiation by (p^12-1)/r (split into its Frobenius-friendly factors, same value) -- computed here on the sextic
// twist: G2 stays in F_p^2, F_p^12 = F_p^2[w]/(w^6 - xi), xi = 9 + i, and a line through twisted
// points evaluates to the sparse element  -y_P + (m x_P) w + (y_1 - m x_1) w^3.
// zk_pairing returns the value in py_ecc's basis (12 coefficients of F_p[w]/(w^12 - 18 w^6 + 82),
// i -> w^6 - 9), so it can be compared with the reference's FQ12 objects coefficient by coefficient.
#include <string.h>
#include "common.h"
#include "curve.h"
#include "host_field.h"
#include "msm.h"

namespace zk {
namespace {

inline HFp2 fp2_conj(const HFp2 &a) { return HFp2{a.c0, fe_neg(a.c1)}; }
inline HFp2 fp2_mul_xi(const HFp2 &a) {  // (a0 + a1 i)(9 + i)
    HFp n0 = fe_dbl(fe_dbl(fe_dbl(a.c0))), n1 = fe_dbl(fe_dbl(fe_dbl(a.c1)));
    n0 = fe_add(n0, a.c0);  // 9 a0
    n1 = fe_add(n1, a.c1);  // 9 a1
    return HFp2{fe_sub(n0, a.c1), fe_add(a.c0, n1)};
}
inline HFp2 fp2_from_fp(const HFp &a) { return HFp2{a, HFp::zero()}; }
inline HFp2 fp2_mul_fp(const HFp2 &a, const HFp &k) { return HFp2{fe_mul(a.c0, k), fe_mul(a.c1, k)}; }

struct Fp12 {
    HFp2 c[6];  // sum c[k] w^k, w^6 = xi
    static Fp12 one() {
        Fp12 r;
        for (auto &x : r.c) x = HFp2::zero();
        r.c[0] = HFp2::one();
        return r;
    }
    bool equals(const Fp12 &b) const {
        for (int k = 0; k < 6; k++)
            if (!c[k].equals(b.c[k])) return false;
        return true;
    }
};

Fp12 f12_mul(const Fp12 &a, const Fp12 &b) {
    HFp2 t[11];
    for (auto &x : t) x = HFp2::zero();
    for (int i = 0; i < 6; i++) {
        if (a.c[i].is_zero()) continue;
        for (int j = 0; j < 6; j++) {
            if (b.c[j].is_zero()) continue;
            t[i + j] = fe_add(t[i + j], fe_mul(a.c[i], b.c[j]));
        }
    }
    Fp12 r;
    for (int k = 0; k < 6; k++) r.c[k] = k < 5 ? fe_add(t[k], fp2_mul_xi(t[k + 6])) : t[5];
    return r;
}

Fp12 f12_pow_words(const Fp12 &a, const uint64_t *e, int nwords) {
    Fp12 r = Fp12::one();
    bool started = false;
    for (int i = nwords - 1; i >= 0; i--)
        for (int b = 63; b >= 0; b--) {
            if (started) r = f12_mul(r, r);
            if ((e[i] >> b) & 1) {
                r = started ? f12_mul(r, a) : a;
                started = true;
            }
        }
    return r;
}

struct G2Aff {
    HFp2 x, y;
    bool inf;
};
struct G1Aff {
    HFp x, y;
    bool inf;
};

// line through T1, T2 (twist points, affine) evaluated at P; also returns T1 + T2 in *sum.
Fp12 line_and_add(const G2Aff &t1, const G2Aff &t2, const G1Aff &p, G2Aff *sum) {
    Fp12 l;
    for (auto &x : l.c) x = HFp2::zero();
    HFp2 m;
    if (!t1.x.equals(t2.x)) {
        m = fe_mul(fe_sub(t2.y, t1.y), fe_inv(fe_sub(t2.x, t1.x)));
    } else if (t1.y.equals(t2.y)) {
        const HFp2 x2 = fe_sqr(t1.x);
        m = fe_mul(fe_add(fe_dbl(x2), x2), fe_inv(fe_dbl(t1.y)));
    } else {  // vertical line: x_P - x_1 w^2, sum is infinity
        l.c[0] = fp2_from_fp(p.x);
        l.c[2] = fe_neg(t1.x);
        sum->inf = true;
        return l;
    }
    l.c[0] = fp2_from_fp(fe_neg(p.y));
    l.c[1] = fp2_mul_fp(m, p.x);
    l.c[3] = fe_sub(t1.y, fe_mul(m, t1.x));
    const HFp2 nx = fe_sub(fe_sub(fe_sqr(m), t1.x), t2.x);
    const HFp2 ny = fe_sub(fe_mul(m, fe_sub(t1.x, nx)), t1.y);
    sum->x = nx;
    sum->y = ny;
    sum->inf = false;
    return l;
}

HFp fp_from_words32(const uint32_t *w) { return fe_to_mont(HFp::from_words(w)); }

// The running point of the Miller loop in homogeneous projective coordinates (x = X/Z, y = Y/Z): no inversion per step.
// Each line below is the affine line of line_and_add times an F_p^2 factor (2 Y Z resp. x_2 Z - X); factors from the
// subfield F_p^2 are annihilated by the (p^6 - 1) part of the final exponentiation, so the pairing VALUE is unchanged.
struct G2Proj {
    HFp2 x, y, z;
};
inline HFp2 fp2_small(unsigned k) { return HFp2{hfe_from_u64<FpTag>(k), HFp::zero()}; }

// tangent at R evaluated at P, R <- 2R.   line * (2 Y Z):  -2 Y Z y_P + 3 X^2 x_P w + (3 b' Z^2 - Y^2) w^3
Fp12 line_dbl_proj(G2Proj &r, const G1Aff &p, const HFp2 &b3) {
    const HFp2 xx = fe_sqr(r.x), yy = fe_sqr(r.y), zz = fe_sqr(r.z), s = fe_mul(r.y, r.z);
    const HFp2 w = fe_add(fe_dbl(xx), xx);                        // 3 X^2
    Fp12 l;
    for (auto &c : l.c) c = HFp2::zero();
    l.c[0] = fe_neg(fp2_mul_fp(fe_dbl(s), p.y));
    l.c[1] = fp2_mul_fp(w, p.x);
    l.c[3] = fe_sub(fe_mul(b3, zz), yy);
    // dbl-1998-cmo-2 (a = 0): B = X Y S, H = W^2 - 8B, X3 = 2 H S, Y3 = W (4B - H) - 8 Y^2 S^2, Z3 = 8 S^3
    const HFp2 b = fe_mul(fe_mul(r.x, r.y), s);
    const HFp2 b4 = fe_dbl(fe_dbl(b));
    const HFp2 h = fe_sub(fe_sqr(w), fe_dbl(b4));
    const HFp2 ss = fe_sqr(s);
    const HFp2 yyss8 = fe_dbl(fe_dbl(fe_dbl(fe_mul(yy, ss))));
    r.x = fe_dbl(fe_mul(h, s));
    r.y = fe_sub(fe_mul(w, fe_sub(b4, h)), yyss8);
    r.z = fe_dbl(fe_dbl(fe_dbl(fe_mul(ss, s))));
    return l;
}
// chord through R and the affine point Q evaluated at P, R <- R + Q (R != +-Q, checked by the caller).
//   line * (x_2 Z - X):  -v y_P + u x_P w + (v y_2 - u x_2) w^3   with u = y_2 Z - Y, v = x_2 Z - X
Fp12 line_add_proj(G2Proj &r, const G2Aff &q, const G1Aff &p, const HFp2 &u, const HFp2 &v) {
    Fp12 l;
    for (auto &c : l.c) c = HFp2::zero();
    l.c[0] = fe_neg(fp2_mul_fp(v, p.y));
    l.c[1] = fp2_mul_fp(u, p.x);
    l.c[3] = fe_sub(fe_mul(v, q.y), fe_mul(u, q.x));
    // madd-1998-cmo: A = u^2 Z - v^3 - 2 v^2 X, X3 = v A, Y3 = u (v^2 X - A) - v^3 Y, Z3 = v^3 Z
    const HFp2 vv = fe_sqr(v), vvv = fe_mul(vv, v), rr = fe_mul(vv, r.x);
    const HFp2 a = fe_sub(fe_sub(fe_mul(fe_sqr(u), r.z), vvv), fe_dbl(rr));
    const HFp2 y3 = fe_sub(fe_mul(u, fe_sub(rr, a)), fe_mul(vvv, r.y));
    r.x = fe_mul(v, a);
    r.y = y3;
    r.z = fe_mul(vvv, r.z);
    return l;
}

Fp12 miller_loop(const G2Aff &q, const G1Aff &p) {
    if (q.inf || p.inf) return Fp12::one();
    static const uint32_t fx0[8] = ZK_FROB_X_C0, fx1[8] = ZK_FROB_X_C1, fy0[8] = ZK_FROB_Y_C0, fy1[8] = ZK_FROB_Y_C1;
    static const HFp2 frob_x{fp_from_words32(fx0), fp_from_words32(fx1)}, frob_y{fp_from_words32(fy0), fp_from_words32(fy1)};
    static const HFp2 b3 = fe_mul(fp2_small(9), fe_inv(HFp2{hfe_from_u64<FpTag>(9), HFp::one()}));  // 3 b' = 9 / xi
    const uint64_t ate = ZK_ATE_LOOP_LOW64;  // low 64 bits of 6u+2; the 65th bit is the start R = Q
    G2Proj r{q.x, q.y, HFp2::one()};
    Fp12 f = Fp12::one();
    // one chord step; the degenerate positions (R = +-Q, R at infinity) cannot occur for points of prime order inside the
    // loop, but inputs are not trusted: they go through the affine routine, which handles every case
    auto add_step = [&](const G2Aff &t) {
        if (r.z.is_zero()) {  // R = O: the line is 1 up to a subfield factor, R <- T
            r = G2Proj{t.x, t.y, HFp2::one()};
            return;
        }
        const HFp2 u = fe_sub(fe_mul(t.y, r.z), r.y), v = fe_sub(fe_mul(t.x, r.z), r.x);
        if (v.is_zero()) {
            const HFp2 zi = fe_inv(r.z);
            const G2Aff ra{fe_mul(r.x, zi), fe_mul(r.y, zi), false};
            G2Aff nr;
            f = f12_mul(f, line_and_add(ra, t, p, &nr));
            r = nr.inf ? G2Proj{HFp2::zero(), HFp2::one(), HFp2::zero()} : G2Proj{nr.x, nr.y, HFp2::one()};
            return;
        }
        f = f12_mul(f, line_add_proj(r, t, p, u, v));
    };
    for (int i = 63; i >= 0; i--) {
        f = f12_mul(f, f);
        if (!r.z.is_zero() && !r.y.is_zero()) f = f12_mul(f, line_dbl_proj(r, p, b3));
        else r = G2Proj{HFp2::zero(), HFp2::one(), HFp2::zero()};
        if ((ate >> i) & 1) add_step(q);
    }
    // Frobenius images of Q on the twist: pi(x, y) = (conj(x) * xi^((p-1)/3), conj(y) * xi^((p-1)/2))
    G2Aff q1{fe_mul(fp2_conj(q.x), frob_x), fe_mul(fp2_conj(q.y), frob_y), false};
    G2Aff q2{fe_mul(fp2_conj(q1.x), frob_x), fe_neg(fe_mul(fp2_conj(q1.y), frob_y)), false};
    add_step(q1);
    add_step(q2);
    return f;
}

// ---- final exponentiation f^((p^12-1)/r), same value as py_ecc's plain power, computed as
//   (p^12-1)/r = (p^6-1) (p^2+1) h,   h = (p^4-p^2+1)/r = l0 + l1 p + l2 p^2 + p^3   (base-p digits, bn254_params.h)
// with the Frobenius maps of F_p^12 = F_p^2[w]/(w^6 - xi):  (c w^k)^p = conj(c) g^k w^k,  g = xi^((p-1)/6).
struct FrobTable {
    HFp2 g[6];  // g^k
    FrobTable() {
        static const uint32_t c0[8] = ZK_FROB_W_C0, c1[8] = ZK_FROB_W_C1;
        const HFp2 gw{fp_from_words32(c0), fp_from_words32(c1)};
        g[0] = HFp2::one();
        for (int k = 1; k < 6; k++) g[k] = fe_mul(g[k - 1], gw);
    }
};
Fp12 f12_frob(const Fp12 &a) {  // a^p
    static const FrobTable T;
    Fp12 r;
    for (int k = 0; k < 6; k++) r.c[k] = fe_mul(fp2_conj(a.c[k]), T.g[k]);
    return r;
}
Fp12 f12_conj6(const Fp12 &a) {  // a^(p^6): w -> -w
    Fp12 r = a;
    for (int k = 1; k < 6; k += 2) r.c[k] = fe_neg(a.c[k]);
    return r;
}
// 1/a: the maps a -> a^(p^2j), j = 1..5, are the F_p^2-automorphisms of F_p^12, so a * prod_j a^(p^2j) is the norm, in F_p^2.
Fp12 f12_inv(const Fp12 &a) {
    Fp12 s = f12_frob(f12_frob(a)), g = s;
    for (int j = 2; j <= 5; j++) {
        s = f12_frob(f12_frob(s));
        g = f12_mul(g, s);
    }
    const HFp2 ninv = fe_inv(f12_mul(a, g).c[0]);
    for (auto &x : g.c) x = fe_mul(x, ninv);
    return g;
}

Fp12 final_exp(const Fp12 &f) {
    Fp12 t = f12_mul(f12_conj6(f), f12_inv(f));   // f^(p^6 - 1)
    t = f12_mul(f12_frob(f12_frob(t)), t);         // ^(p^2 + 1)
    // t^h by simultaneous square-and-multiply over the three 254-bit digits; the digit of p^3 is 1
    static const uint64_t L[3][4] = {ZK_HARD_L0, ZK_HARD_L1, ZK_HARD_L2};
    Fp12 base[3] = {t, f12_frob(t), f12_frob(f12_frob(t))};
    const Fp12 top = f12_frob(base[2]);
    Fp12 table[8];  // table[m] = prod_{bit i of m} base[i]
    table[0] = Fp12::one();
    for (int m = 1; m < 8; m++) {
        const int low = m & -m, i = low == 1 ? 0 : low == 2 ? 1 : 2;
        table[m] = (m == low) ? base[i] : f12_mul(table[m ^ low], base[i]);
    }
    Fp12 r = Fp12::one();
    bool started = false;
    for (int bit = 255; bit >= 0; bit--) {
        if (started) r = f12_mul(r, r);
        const int m = (int)((L[0][bit >> 6] >> (bit & 63)) & 1) | (int)(((L[1][bit >> 6] >> (bit & 63)) & 1) << 1) |
                      (int)(((L[2][bit >> 6] >> (bit & 63)) & 1) << 2);
        if (m) {
            r = started ? f12_mul(r, table[m]) : table[m];
            started = true;
        }
    }
    return f12_mul(r, top);
}

G1Aff load_g1(const uint64_t *p) {
    G1Aff a;
    a.x = read_fe_canonical_fp(p);
    a.y = read_fe_canonical_fp(p + 4);
    a.inf = a.x.is_zero() && a.y.is_zero();
    return a;
}
G2Aff load_g2(const uint64_t *p) {
    G2Aff a;
    a.x = HFp2{read_fe_canonical_fp(p), read_fe_canonical_fp(p + 4)};
    a.y = HFp2{read_fe_canonical_fp(p + 8), read_fe_canonical_fp(p + 12)};
    a.inf = a.x.is_zero() && a.y.is_zero();
    return a;
}

// Input validation, as py_ecc's pairing() does before anything else (bn128_pairing.py: `assert is_on_curve(Q, b2)`,
// `assert is_on_curve(P, b)`; infinity passes): coordinates must be canonical (< p) and the point must satisfy its curve
// equation -- y^2 = x^3 + 3 on G1, y^2 = x^3 + 3/(9 + i) on the twist.  An off-curve "point" would otherwise run through the
// Miller loop and give a meaningless value, so a verifier would accept or reject on garbage.  Like the reference, this does
// NOT test that a twist point lies in the order-r subgroup (G1 has cofactor 1, so on-curve is enough there).
bool fp_words_canonical(const uint64_t *w) {
    const HFp m = HFp::modulus();
    for (int k = 3; k >= 0; k--) {
        if (w[k] < m.l[k]) return true;
        if (w[k] > m.l[k]) return false;
    }
    return false;
}
bool g1_input_ok(const uint64_t *p) {
    if (!fp_words_canonical(p) || !fp_words_canonical(p + 4)) return false;
    const G1Aff a = load_g1(p);
    if (a.inf) return true;
    const HFp rhs = fe_add(fe_mul(fe_sqr(a.x), a.x), hfe_from_u64<FpTag>(3));
    return fe_sqr(a.y).equals(rhs);
}
bool g2_input_ok(const uint64_t *p) {
    for (int k = 0; k < 4; k++)
        if (!fp_words_canonical(p + 4 * k)) return false;
    const G2Aff a = load_g2(p);
    if (a.inf) return true;
    static const HFp2 b2 = fp2_mul_fp(fe_inv(HFp2{hfe_from_u64<FpTag>(9), HFp::one()}), hfe_from_u64<FpTag>(3));  // 3 / (9 + i)
    const HFp2 rhs = fe_add(fe_mul(fe_sqr(a.x), a.x), b2);
    return fe_sqr(a.y).equals(rhs);
}

}  // namespace

// prod_i e(P_i, Q_i) with ONE final exponentiation; out (nullable): 12 canonical F_p coefficients
// in py_ecc's F_p[w]/(w^12 - 18 w^6 + 82) basis; *is_one: whether the product is the identity.
int pairing_product(const uint64_t *g1_points, const uint64_t *g2_points, size_t n, uint64_t *out, int *is_one) {
    for (size_t i = 0; i < n; i++) {
        if (!g1_input_ok(g1_points + 8 * i)) return invalid("zk_pairing: a G1 input is not a canonical point of y^2 = x^3 + 3");
        if (!g2_input_ok(g2_points + 16 * i)) return invalid("zk_pairing: a G2 input is not a canonical point of the twist y^2 = x^3 + 3/(9+i)");
    }
    Fp12 f = Fp12::one();
    for (size_t i = 0; i < n; i++) f = f12_mul(f, miller_loop(load_g2(g2_points + 16 * i), load_g1(g1_points + 8 * i)));
    f = final_exp(f);
    if (is_one) *is_one = f.equals(Fp12::one()) ? 1 : 0;
    if (out) {
        // c[k] = a + b i with i = w^6 - 9  ->  coefficient of w^k: a - 9 b, of w^(k+6): b
        for (int k = 0; k < 6; k++) {
            HFp nine_b = fe_add(fe_dbl(fe_dbl(fe_dbl(f.c[k].c1))), f.c[k].c1);
            HFp lo = fe_from_mont(fe_sub(f.c[k].c0, nine_b)), hi = fe_from_mont(f.c[k].c1);
            memcpy(out + 4 * k, lo.l, 32);
            memcpy(out + 4 * (k + 6), hi.l, 32);
        }
    }
    return ZK_OK;
}

}  // namespace zk
