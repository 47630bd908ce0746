// field.h -- 254-bit prime-field arithmetic for BN254 (F_p and F_r), Montgomery form,
// radix 2^256, 8 x 32-bit little-endian limbs.  One definition for host and gfx950 device
// code: on the device every limb is a VGPR and the inner products lower to v_mad_u64_u32.
//
// Replaces (as arithmetic) what the reference gets from py_ecc's FQ / its own FR subclass:
//   zkp/plonk/field.py:36-51 (FR), zkp/groth16/proving.py:20-21.
#pragma once
#include <stdint.h>
#include "bn254_params.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define ZK_HD __host__ __device__ __forceinline__
#else
#define ZK_HD inline
#endif

namespace zk {

struct FpTag {};  // base field   p
struct FrTag {};  // scalar field r

template <class Tag> struct FieldConst;

template <> struct FieldConst<FpTag> {
    static ZK_HD uint32_t mod(int i) {
        constexpr uint32_t m[8] = ZK_FP_MOD;
        return m[i];
    }
    static ZK_HD uint32_t r1(int i) {
        constexpr uint32_t m[8] = ZK_FP_R1;
        return m[i];
    }
    static ZK_HD uint32_t r2(int i) {
        constexpr uint32_t m[8] = ZK_FP_R2;
        return m[i];
    }
    static constexpr uint32_t inv32 = ZK_FP_INV32;
};

template <> struct FieldConst<FrTag> {
    static ZK_HD uint32_t mod(int i) {
        constexpr uint32_t m[8] = ZK_FR_MOD;
        return m[i];
    }
    static ZK_HD uint32_t r1(int i) {
        constexpr uint32_t m[8] = ZK_FR_R1;
        return m[i];
    }
    static ZK_HD uint32_t r2(int i) {
        constexpr uint32_t m[8] = ZK_FR_R2;
        return m[i];
    }
    static constexpr uint32_t inv32 = ZK_FR_INV32;
};

// A field element; value semantics, limbs in l[0] (least significant) .. l[7].
template <class Tag> struct alignas(16) Fe {
    uint32_t l[8];
    typedef FieldConst<Tag> C;

    static ZK_HD Fe zero() {
        Fe r;
#pragma unroll
        for (int i = 0; i < 8; i++) r.l[i] = 0;
        return r;
    }
    static ZK_HD Fe one() {  // Montgomery form of 1
        Fe r;
#pragma unroll
        for (int i = 0; i < 8; i++) r.l[i] = C::r1(i);
        return r;
    }
    ZK_HD bool is_zero() const {
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) o |= l[i];
        return o == 0;
    }
    ZK_HD bool equals(const Fe &b) const {
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) o |= (l[i] ^ b.l[i]);
        return o == 0;
    }
};

// r = a - mod if a >= mod (a < 2*mod)
template <class Tag> ZK_HD void fe_reduce_once(Fe<Tag> &a) {
    typedef FieldConst<Tag> C;
    uint32_t t[8];
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t d = (uint64_t)a.l[i] - C::mod(i) - borrow;
        t[i] = (uint32_t)d;
        borrow = (d >> 63) & 1;
    }
    if (!borrow) {
#pragma unroll
        for (int i = 0; i < 8; i++) a.l[i] = t[i];
    }
}

template <class Tag> ZK_HD Fe<Tag> fe_add(const Fe<Tag> &a, const Fe<Tag> &b) {
    Fe<Tag> r;
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (uint64_t)a.l[i] + b.l[i];
        r.l[i] = (uint32_t)c;
        c >>= 32;
    }
    fe_reduce_once(r);  // a+b < 2p < 2^255: no carry out of limb 7
    return r;
}

template <class Tag> ZK_HD Fe<Tag> fe_sub(const Fe<Tag> &a, const Fe<Tag> &b) {
    typedef FieldConst<Tag> C;
    Fe<Tag> r;
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t d = (uint64_t)a.l[i] - b.l[i] - borrow;
        r.l[i] = (uint32_t)d;
        borrow = (d >> 63) & 1;
    }
    uint32_t mask = (uint32_t)0 - (uint32_t)borrow;  // add the modulus back when a < b
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (uint64_t)r.l[i] + (C::mod(i) & mask);
        r.l[i] = (uint32_t)c;
        c >>= 32;
    }
    return r;
}

template <class Tag> ZK_HD Fe<Tag> fe_neg(const Fe<Tag> &a) {
    if (a.is_zero()) return a;
    typedef FieldConst<Tag> C;
    Fe<Tag> r;
    uint64_t borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t d = (uint64_t)C::mod(i) - a.l[i] - borrow;
        r.l[i] = (uint32_t)d;
        borrow = (d >> 63) & 1;
    }
    return r;
}

template <class Tag> ZK_HD Fe<Tag> fe_dbl(const Fe<Tag> &a) { return fe_add(a, a); }

// Montgomery product a*b*2^-256 mod m, CIOS with 32-bit limbs.  The modulus is < 2^254, so
// the running sum never needs more than 9 limbs and one conditional subtraction finishes.
template <class Tag> ZK_HD Fe<Tag> fe_mul(const Fe<Tag> &a, const Fe<Tag> &b) {
    typedef FieldConst<Tag> C;
    uint32_t t[9];
#pragma unroll
    for (int i = 0; i < 9; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t carry = 0;
        const uint32_t bi = b.l[i];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            uint64_t s = (uint64_t)a.l[j] * bi + t[j] + carry;
            t[j] = (uint32_t)s;
            carry = s >> 32;
        }
        uint32_t t8 = t[8] + (uint32_t)carry;
        const uint32_t m = t[0] * C::inv32;
        uint64_t s = (uint64_t)m * C::mod(0) + t[0];
        carry = s >> 32;
#pragma unroll
        for (int j = 1; j < 8; j++) {
            s = (uint64_t)m * C::mod(j) + t[j] + carry;
            t[j - 1] = (uint32_t)s;
            carry = s >> 32;
        }
        s = (uint64_t)t8 + carry;
        t[7] = (uint32_t)s;
        t[8] = (uint32_t)(s >> 32);
    }
    Fe<Tag> r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = t[i];
    fe_reduce_once(r);
    return r;
}

template <class Tag> ZK_HD Fe<Tag> fe_sqr(const Fe<Tag> &a) { return fe_mul(a, a); }

template <class Tag> ZK_HD Fe<Tag> fe_to_mont(const Fe<Tag> &a) {
    typedef FieldConst<Tag> C;
    Fe<Tag> r2;
#pragma unroll
    for (int i = 0; i < 8; i++) r2.l[i] = C::r2(i);
    return fe_mul(a, r2);
}

template <class Tag> ZK_HD Fe<Tag> fe_from_mont(const Fe<Tag> &a) {
    Fe<Tag> one;
#pragma unroll
    for (int i = 0; i < 8; i++) one.l[i] = (i == 0);
    return fe_mul(a, one);
}

// a^e for a 256-bit exponent given as limbs (square-and-multiply, MSB first).
template <class Tag> ZK_HD Fe<Tag> fe_pow(const Fe<Tag> &a, const uint32_t e[8]) {
    Fe<Tag> r = Fe<Tag>::one();
    for (int i = 7; i >= 0; i--) {
        for (int b = 31; b >= 0; b--) {
            r = fe_sqr(r);
            if ((e[i] >> b) & 1) r = fe_mul(r, a);
        }
    }
    return r;
}

// Inverse by Fermat: a^(m-2).  inv(0) = 0.
template <class Tag> ZK_HD Fe<Tag> fe_inv(const Fe<Tag> &a) {
    typedef FieldConst<Tag> C;
    uint32_t e[8];
    uint64_t borrow = 2;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t d = (uint64_t)C::mod(i) - borrow;
        e[i] = (uint32_t)d;
        borrow = (d >> 63) & 1;
    }
    return fe_pow(a, e);
}

typedef Fe<FpTag> Fp;
typedef Fe<FrTag> Fr;

// ---------------------------------------------------------------------------------------
// F_p^2 = F_p[i]/(i^2+1); element c0 + c1*i.  (py_ecc FQ2, coeffs [c0, c1].)
struct Fp2 {
    Fp c0, c1;
    static ZK_HD Fp2 zero() { return Fp2{Fp::zero(), Fp::zero()}; }
    static ZK_HD Fp2 one() { return Fp2{Fp::one(), Fp::zero()}; }
    ZK_HD bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
    ZK_HD bool equals(const Fp2 &b) const { return c0.equals(b.c0) && c1.equals(b.c1); }
};

ZK_HD Fp2 fe_add(const Fp2 &a, const Fp2 &b) { return Fp2{fe_add(a.c0, b.c0), fe_add(a.c1, b.c1)}; }
ZK_HD Fp2 fe_sub(const Fp2 &a, const Fp2 &b) { return Fp2{fe_sub(a.c0, b.c0), fe_sub(a.c1, b.c1)}; }
ZK_HD Fp2 fe_neg(const Fp2 &a) { return Fp2{fe_neg(a.c0), fe_neg(a.c1)}; }
ZK_HD Fp2 fe_dbl(const Fp2 &a) { return Fp2{fe_dbl(a.c0), fe_dbl(a.c1)}; }
// Karatsuba: 3 base-field products.
ZK_HD Fp2 fe_mul(const Fp2 &a, const Fp2 &b) {
    Fp v0 = fe_mul(a.c0, b.c0);
    Fp v1 = fe_mul(a.c1, b.c1);
    Fp s = fe_mul(fe_add(a.c0, a.c1), fe_add(b.c0, b.c1));
    return Fp2{fe_sub(v0, v1), fe_sub(fe_sub(s, v0), v1)};
}
// (c0+c1 i)^2 = (c0+c1)(c0-c1) + 2 c0 c1 i : 2 base-field products.
ZK_HD Fp2 fe_sqr(const Fp2 &a) {
    Fp t = fe_mul(a.c0, a.c1);
    Fp u = fe_mul(fe_add(a.c0, a.c1), fe_sub(a.c0, a.c1));
    return Fp2{u, fe_dbl(t)};
}
ZK_HD Fp2 fe_inv(const Fp2 &a) {
    Fp d = fe_inv(fe_add(fe_sqr(a.c0), fe_sqr(a.c1)));
    return Fp2{fe_mul(a.c0, d), fe_neg(fe_mul(a.c1, d))};
}
ZK_HD Fp2 fe_to_mont(const Fp2 &a) { return Fp2{fe_to_mont(a.c0), fe_to_mont(a.c1)}; }
ZK_HD Fp2 fe_from_mont(const Fp2 &a) { return Fp2{fe_from_mont(a.c0), fe_from_mont(a.c1)}; }

}  // namespace zk
