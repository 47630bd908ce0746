// host_field.h -- host-only F_p / F_p^2 / F_r arithmetic with 4 x 64-bit limbs (unsigned __int128
// products), Montgomery radix 2^256.  The device code uses 9 x 29-bit limbs with radix 2^261, so a
// device element crosses over with one multiplication by a constant (HFe::from_dev / to_dev).
// Used for the O(1)-size host epilogues of the GPU pipelines (window-sum Horner fold of the MSM,
// multi-GPU partial folding, NTT twiddle-table generation, the pairing), where a single GPU thread
// would be latency-bound.  curve.h's templates work on these types unchanged.
#pragma once
#include <stdint.h>
#include <string.h>
#include "field.h"

namespace zk {

typedef unsigned __int128 u128_t;

template <class Tag> struct HostConst;
#define ZK_DEFINE_HOST_CONST(TAG, PFX)                                             \
    template <> struct HostConst<TAG> {                                            \
        static const uint32_t *mod32() { static const uint32_t v[8] = PFX##_MOD32; return v; }      \
        static const uint32_t *r1() { static const uint32_t v[8] = PFX##_H_R1; return v; }          \
        static const uint32_t *r2() { static const uint32_t v[8] = PFX##_H_R2; return v; }          \
        static const uint32_t *from_dev() { static const uint32_t v[8] = PFX##_H_FROM_DEV; return v; } \
        static const uint32_t *to_dev() { static const uint32_t v[8] = PFX##_H_TO_DEV; return v; }  \
    };
ZK_DEFINE_HOST_CONST(FpTag, ZK_FP)
ZK_DEFINE_HOST_CONST(FrTag, ZK_FR)

template <class Tag> struct HFe;
template <class Tag> inline HFe<Tag> fe_mul(const HFe<Tag> &a, const HFe<Tag> &b);

template <class Tag> struct HFe {
    uint64_t l[4];
    static HFe from_words(const uint32_t *w) {
        HFe r;
        for (int i = 0; i < 4; i++) r.l[i] = (uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32);
        return r;
    }
    static HFe zero() { return HFe{{0, 0, 0, 0}}; }
    static HFe one() { return from_words(HostConst<Tag>::r1()); }
    static HFe modulus() { return from_words(HostConst<Tag>::mod32()); }
    static HFe r2() { return from_words(HostConst<Tag>::r2()); }
    static uint64_t inv64() {  // -m^-1 mod 2^64 by Newton iteration
        uint64_t m0 = modulus().l[0], inv = 1;
        for (int i = 0; i < 6; i++) inv *= 2 - m0 * inv;
        return (uint64_t)0 - inv;
    }
    bool is_zero() const { return (l[0] | l[1] | l[2] | l[3]) == 0; }
    bool maybe_zero() const { return is_zero(); }   // curve.h's cheap pre-test; host elements are canonical, so it is exact here
    bool equals(const HFe &b) const { return !memcmp(l, b.l, 32); }
    // Device element (9x29 limbs, lazy, Montgomery radix 2^261) <-> host element (radix 2^256).
    static HFe from_dev(const Fe<Tag> &a) {
        uint32_t w[8];
        fe_to_words(fe_reduce_full(a), w);               // V = x * 2^261 mod m, canonical
        return fe_mul(from_words(w), from_words(HostConst<Tag>::from_dev()));  // V * 2^251 * 2^-256 = x * 2^256
    }
    Fe<Tag> to_dev() const {
        const HFe v = fe_mul(*this, from_words(HostConst<Tag>::to_dev()));      // h * 2^261 * 2^-256 = x * 2^261
        uint32_t w[8];
        for (int i = 0; i < 4; i++) { w[2 * i] = (uint32_t)v.l[i]; w[2 * i + 1] = (uint32_t)(v.l[i] >> 32); }
        return fe_from_words<Tag>(w);
    }
};

template <class Tag> inline bool hfe_geq_mod(const HFe<Tag> &a) {
    const HFe<Tag> m = HFe<Tag>::modulus();
    for (int i = 3; i >= 0; i--) {
        if (a.l[i] > m.l[i]) return true;
        if (a.l[i] < m.l[i]) return false;
    }
    return true;
}
template <class Tag> inline void hfe_sub_mod(HFe<Tag> &a) {
    const HFe<Tag> m = HFe<Tag>::modulus();
    u128_t br = 0;
    for (int i = 0; i < 4; i++) {
        u128_t d = (u128_t)a.l[i] - m.l[i] - br;
        a.l[i] = (uint64_t)d;
        br = (d >> 64) & 1;
    }
}
template <class Tag> inline HFe<Tag> fe_add(const HFe<Tag> &a, const HFe<Tag> &b) {
    HFe<Tag> r;
    u128_t c = 0;
    for (int i = 0; i < 4; i++) {
        c += (u128_t)a.l[i] + b.l[i];
        r.l[i] = (uint64_t)c;
        c >>= 64;
    }
    if (hfe_geq_mod(r)) hfe_sub_mod(r);
    return r;
}
template <class Tag> inline HFe<Tag> fe_sub(const HFe<Tag> &a, const HFe<Tag> &b) {
    HFe<Tag> r;
    u128_t br = 0;
    for (int i = 0; i < 4; i++) {
        u128_t d = (u128_t)a.l[i] - b.l[i] - br;
        r.l[i] = (uint64_t)d;
        br = (d >> 64) & 1;
    }
    if (br) {
        const HFe<Tag> m = HFe<Tag>::modulus();
        u128_t c = 0;
        for (int i = 0; i < 4; i++) {
            c += (u128_t)r.l[i] + m.l[i];
            r.l[i] = (uint64_t)c;
            c >>= 64;
        }
    }
    return r;
}
template <class Tag> inline HFe<Tag> fe_neg(const HFe<Tag> &a) {
    if (a.is_zero()) return a;
    return fe_sub(HFe<Tag>::zero(), a);
}
template <class Tag> inline HFe<Tag> fe_dbl(const HFe<Tag> &a) { return fe_add(a, a); }
template <class Tag> inline HFe<Tag> fe_triple(const HFe<Tag> &a) { return fe_add(fe_add(a, a), a); }
// curve.h passes value-bound hints K for the lazy device representation; host elements are always canonical.
template <int K, class Tag> inline HFe<Tag> fe_sub(const HFe<Tag> &a, const HFe<Tag> &b) { return fe_sub(a, b); }
template <int K, class Tag> inline HFe<Tag> fe_neg(const HFe<Tag> &a) { return fe_neg(a); }
template <int K, class Tag> inline HFe<Tag> fe_sub_once(const HFe<Tag> &a, const HFe<Tag> &b) { return fe_sub(a, b); }
template <int K, class Tag> inline HFe<Tag> fe_neg_once(const HFe<Tag> &a) { return fe_neg(a); }
template <class Tag> inline HFe<Tag> fe_tidy(const HFe<Tag> &a) { return a; }
template <int K, class Tag> inline HFe<Tag> fe_mul_minus(const HFe<Tag> &a, const HFe<Tag> &b, const HFe<Tag> &c) { return fe_sub(fe_mul(a, b), c); }
template <int K, class Tag> inline HFe<Tag> fe_sub2(const HFe<Tag> &a, const HFe<Tag> &b, const HFe<Tag> &c) { return fe_sub(fe_sub(a, b), fe_dbl(c)); }
template <class Tag> inline HFe<Tag> fe_mul(const HFe<Tag> &a, const HFe<Tag> &b) {
    static const HFe<Tag> m = HFe<Tag>::modulus();
    static const uint64_t inv = HFe<Tag>::inv64();
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u128_t c = 0;
        for (int j = 0; j < 4; j++) {
            c += (u128_t)a.l[j] * b.l[i] + t[j];
            t[j] = (uint64_t)c;
            c >>= 64;
        }
        c += t[4];
        t[4] = (uint64_t)c;
        t[5] = (uint64_t)(c >> 64);
        uint64_t q = t[0] * inv;
        c = (u128_t)q * m.l[0] + t[0];
        c >>= 64;
        for (int j = 1; j < 4; j++) {
            c += (u128_t)q * m.l[j] + t[j];
            t[j - 1] = (uint64_t)c;
            c >>= 64;
        }
        c += t[4];
        t[3] = (uint64_t)c;
        t[4] = t[5] + (uint64_t)(c >> 64);
    }
    HFe<Tag> r{{t[0], t[1], t[2], t[3]}};
    if (t[4] || hfe_geq_mod(r)) hfe_sub_mod(r);
    return r;
}
template <class Tag> inline HFe<Tag> fe_sqr(const HFe<Tag> &a) { return fe_mul(a, a); }
template <int K, class Tag> inline HFe<Tag> fe_mulsub(const HFe<Tag> &a, const HFe<Tag> &b, const HFe<Tag> &c, const HFe<Tag> &d) {
    return fe_sub(fe_mul(a, b), fe_mul(c, d));
}
template <class Tag> inline HFe<Tag> fe_pow(const HFe<Tag> &a, const uint64_t e[4]) {
    HFe<Tag> r = HFe<Tag>::one();
    for (int i = 3; i >= 0; i--)
        for (int b = 63; b >= 0; b--) {
            r = fe_sqr(r);
            if ((e[i] >> b) & 1) r = fe_mul(r, a);
        }
    return r;
}
template <class Tag> inline HFe<Tag> fe_inv(const HFe<Tag> &a) {
    HFe<Tag> m = HFe<Tag>::modulus();
    uint64_t e[4] = {m.l[0] - 2, m.l[1], m.l[2], m.l[3]};  // low limb of p and r is > 2
    return fe_pow(a, e);
}
template <class Tag> inline HFe<Tag> fe_to_mont(const HFe<Tag> &a) { return fe_mul(a, HFe<Tag>::r2()); }
template <class Tag> inline HFe<Tag> fe_from_mont(const HFe<Tag> &a) { return fe_mul(a, HFe<Tag>{{1, 0, 0, 0}}); }
template <class Tag> inline HFe<Tag> hfe_from_u64(uint64_t v) { return fe_to_mont(HFe<Tag>{{v, 0, 0, 0}}); }

typedef HFe<FpTag> HFp;
typedef HFe<FrTag> HFr;

struct HFp2 {
    HFp c0, c1;
    static HFp2 zero() { return HFp2{HFp::zero(), HFp::zero()}; }
    static HFp2 one() { return HFp2{HFp::one(), HFp::zero()}; }
    bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
    bool maybe_zero() const { return is_zero(); }
    bool equals(const HFp2 &b) const { return c0.equals(b.c0) && c1.equals(b.c1); }
    static HFp2 from_dev(const Fp2 &a) { return HFp2{HFp::from_dev(a.c0), HFp::from_dev(a.c1)}; }
    Fp2 to_dev() const { return Fp2{c0.to_dev(), c1.to_dev()}; }
};
inline HFp2 fe_add(const HFp2 &a, const HFp2 &b) { return HFp2{fe_add(a.c0, b.c0), fe_add(a.c1, b.c1)}; }
inline HFp2 fe_sub(const HFp2 &a, const HFp2 &b) { return HFp2{fe_sub(a.c0, b.c0), fe_sub(a.c1, b.c1)}; }
inline HFp2 fe_neg(const HFp2 &a) { return HFp2{fe_neg(a.c0), fe_neg(a.c1)}; }
inline HFp2 fe_dbl(const HFp2 &a) { return HFp2{fe_dbl(a.c0), fe_dbl(a.c1)}; }
inline HFp2 fe_triple(const HFp2 &a) { return HFp2{fe_triple(a.c0), fe_triple(a.c1)}; }
template <int K> inline HFp2 fe_sub(const HFp2 &a, const HFp2 &b) { return fe_sub(a, b); }
template <int K> inline HFp2 fe_neg(const HFp2 &a) { return fe_neg(a); }
template <int K> inline HFp2 fe_sub_once(const HFp2 &a, const HFp2 &b) { return fe_sub(a, b); }
template <int K> inline HFp2 fe_neg_once(const HFp2 &a) { return fe_neg(a); }
inline HFp2 fe_tidy(const HFp2 &a) { return a; }
template <int K> inline HFp2 fe_sub2(const HFp2 &a, const HFp2 &b, const HFp2 &c) { return fe_sub(fe_sub(a, b), fe_dbl(c)); }
inline HFp2 fe_mul(const HFp2 &a, const HFp2 &b) {
    HFp v0 = fe_mul(a.c0, b.c0), v1 = fe_mul(a.c1, b.c1);
    HFp s = fe_mul(fe_add(a.c0, a.c1), fe_add(b.c0, b.c1));
    return HFp2{fe_sub(v0, v1), fe_sub(fe_sub(s, v0), v1)};
}
inline HFp2 fe_sqr(const HFp2 &a) {
    HFp t = fe_mul(a.c0, a.c1);
    HFp u = fe_mul(fe_add(a.c0, a.c1), fe_sub(a.c0, a.c1));
    return HFp2{u, fe_dbl(t)};
}
template <int K> inline HFp2 fe_mulsub(const HFp2 &a, const HFp2 &b, const HFp2 &c, const HFp2 &d) { return fe_sub(fe_mul(a, b), fe_mul(c, d)); }
inline HFp2 fe_inv(const HFp2 &a) {
    HFp d = fe_inv(fe_add(fe_sqr(a.c0), fe_sqr(a.c1)));
    return HFp2{fe_mul(a.c0, d), fe_neg(fe_mul(a.c1, d))};
}
inline HFp2 fe_to_mont(const HFp2 &a) { return HFp2{fe_to_mont(a.c0), fe_to_mont(a.c1)}; }
inline HFp2 fe_from_mont(const HFp2 &a) { return HFp2{fe_from_mont(a.c0), fe_from_mont(a.c1)}; }

// Device field type -> matching host type.
template <class F> struct HostOf;
template <> struct HostOf<Fp> { typedef HFp type; };
template <> struct HostOf<Fp2> { typedef HFp2 type; };
template <> struct HostOf<Fr> { typedef HFr type; };

}  // namespace zk
