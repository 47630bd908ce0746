// field.h -- 254-bit prime-field arithmetic for BN254 (F_p and F_r) on gfx950.
//
// Representation: 9 limbs x 29 bits (reduced radix), Montgomery radix 2^261.  On CDNA4 the
// 32x32+64 multiply-add v_mad_u64_u32 issues at nearly the rate of a 64-bit add (measured
// ~5 vs ~4.4 cycles per wave-instruction, tools/ubench.hip), so the cheapest modular product is
// the one with the fewest NON-multiply instructions: with 29-bit limbs a whole column of the
// product (<= 18 terms of < 2^60) accumulates in one 64-bit register pair with no carry
// handling at all -- 171 multiply-adds plus ~30 shifts/masks per Montgomery product, against
// 136 multiply-adds plus ~470 carry instructions for saturated 8 x 32-bit limbs.
//
// Values are kept "lazy": limbs are normalised to < 2^29 (top limb holds the excess), but the
// VALUE is only bounded by a small multiple of the modulus (< 16 m); nothing is conditionally
// subtracted on the hot path.  Contracts (m = modulus, R = 2^261, R/m ~ 169):
//   fe_mul / fe_sqr   inputs: limbs < 2^30.5 (a sum of two normalised elements may be fed
//                     unnormalised), values va, vb with va*vb < 169 m^2;  output: normalised,
//                     value < va*vb/R + m  (< 2m under the contract).
//   fe_add            value a+b, normalised.         fe_sub_k<K>: a - b + K*m (needs b <= K*m).
//   fe_wreduce<M>     value < M*m (M <= 16) -> < 2m.  fe_reduce_full: < 16m -> canonical < m.
//   fe_sub_lazy<K> / fe_neg_lazy<K>   a - b + (K+1)*m / (K+1)*m - a with NO carry propagation (limbs up to 3 * 2^29):
//                     only for a value that is used once, as a multiplication operand.
//   fe_sub2<K>        a - b - 2c + K*m in one pass.   fe_mul_minus<K>: a*b - c out of one reduction (c rides in the
//                     upper columns of the product).  fe_dot<N> / fe_mulsub<K>: sums of products, one reduction.
// The accumulate kernel is instruction-issue-bound, so these exist to delete non-multiply instructions.
// curve.h documents the bound of every intermediate of its formulas against these contracts;
// tests/hostmath runs the same code on the CPU with ZK_FIELD_DEBUG bound tracking.
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

// Host-only contract checker (tests/hostmath builds with -DZK_FIELD_DEBUG): every element carries an
// upper bound of its value (in units of the modulus) and of its limbs, every operation checks its
// precondition from the header comment above and aborts on a violation.
#if defined(ZK_FIELD_DEBUG) && !defined(__HIPCC__)
#include <stdio.h>
#include <stdlib.h>
#define ZK_DBG(...) __VA_ARGS__
#define ZK_DBG_ASSERT(cond, what)                                                        \
    do {                                                                                 \
        if (!(cond)) {                                                                   \
            fprintf(stderr, "ZK_FIELD_DEBUG: contract violated: %s (%s:%d)\n", what, __FILE__, __LINE__); \
            abort();                                                                     \
        }                                                                                \
    } while (0)
#else
#define ZK_DBG(...)
#define ZK_DBG_ASSERT(cond, what)
#endif

namespace zk {

struct FpTag {};  // base field   p
struct FrTag {};  // scalar field r

constexpr int NL = ZK_NLIMBS;          // 9
constexpr int LB = ZK_LIMB_BITS;       // 29
constexpr uint32_t LMASK = (1u << LB) - 1u;

template <class Tag> struct FieldConst;

#define ZK_DEFINE_FIELD_CONST(TAG, PFX)                                                   \
    template <> struct FieldConst<TAG> {                                                  \
        static ZK_HD uint32_t mod(int i) {                                                \
            constexpr uint32_t m[NL] = PFX##_MOD;                                         \
            return m[i];                                                                  \
        }                                                                                 \
        static ZK_HD uint32_t r1(int i) {                                                 \
            constexpr uint32_t m[NL] = PFX##_R1;                                          \
            return m[i];                                                                  \
        }                                                                                 \
        static ZK_HD uint32_t r2(int i) {                                                 \
            constexpr uint32_t m[NL] = PFX##_R2;                                          \
            return m[i];                                                                  \
        }                                                                                 \
        static ZK_HD uint32_t kp(int k, int i) { /* limb i of k * modulus, k <= 16 */      \
            constexpr uint32_t m[17][NL] = PFX##_KP;                                      \
            return m[k][i];                                                               \
        }                                                                                 \
        /* k * modulus again, with 2^29 lent to every limb below the top one (and taken back from the   */ \
        /* limb above): kpl(k, i) - b.l[i] is non-negative for every normalised b <= (k-1) * modulus     */ \
        static ZK_HD uint32_t kpl(int k, int i) {                                         \
            return kp(k, i) + (i < NL - 1 ? (1u << LB) : 0u) - (i > 0 ? 1u : 0u);         \
        }                                                                                 \
        static constexpr uint32_t inv = PFX##_INV29;                                      \
    };
ZK_DEFINE_FIELD_CONST(FpTag, ZK_FP)
ZK_DEFINE_FIELD_CONST(FrTag, ZK_FR)

// A field element; value semantics.  l[0..7] < 2^29 when normalised, l[8] holds the rest.
template <class Tag> struct alignas(4) Fe {
    uint32_t l[NL];
    ZK_DBG(double vb = 16.0;    /* value < vb * m                      */
           double lmax = 4.0;   /* every limb < lmax * 2^29 (limbs 0..7) */)
    typedef FieldConst<Tag> C;
    static constexpr int CANON_WORDS = 8;  // 32-bit words of the canonical (ABI) encoding

    static ZK_HD Fe zero() {
        Fe r;
#pragma unroll
        for (int i = 0; i < NL; i++) r.l[i] = 0;
        ZK_DBG(r.vb = 0; r.lmax = 1;)
        return r;
    }
    static ZK_HD Fe one() {  // Montgomery form of 1
        Fe r;
#pragma unroll
        for (int i = 0; i < NL; i++) r.l[i] = C::r1(i);
        ZK_DBG(r.vb = 1; r.lmax = 1;)
        return r;
    }
    ZK_HD bool raw_is_zero() const {  // all limbs zero (exact integer zero)
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < NL; i++) o |= l[i];
        return o == 0;
    }
    ZK_HD bool is_zero() const;              // value == 0 mod m (any lazy representative)
    ZK_HD bool maybe_zero() const;           // false: certainly non-zero (is_zero's three-instruction reject); true: run is_zero()
    ZK_HD bool equals(const Fe &b) const;    // values equal mod m
};

// Carry propagation of limbs that may be >= 2^29 or (as int32) negative; the total value must be
// non-negative and < 2^(232+31).
template <class Tag> ZK_HD void fe_normalize(Fe<Tag> &a) {
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < NL - 1; i++) {
        const int32_t t = (int32_t)a.l[i] + c;
        a.l[i] = (uint32_t)t & LMASK;
        c = t >> LB;  // arithmetic shift: negative limbs borrow
    }
    a.l[NL - 1] = (uint32_t)((int32_t)a.l[NL - 1] + c);
    ZK_DBG(a.lmax = 1;)
}

// a - K*m if that is >= 0, else a.  Input normalised.
template <int K, class Tag> ZK_HD void fe_cond_sub(Fe<Tag> &a) {
    typedef FieldConst<Tag> C;
    uint32_t d[NL];
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < NL - 1; i++) {
        const int32_t t = (int32_t)a.l[i] - (int32_t)C::kp(K, i) + c;
        d[i] = (uint32_t)t & LMASK;
        c = t >> LB;
    }
    const int32_t top = (int32_t)a.l[NL - 1] - (int32_t)C::kp(K, NL - 1) + c;
    d[NL - 1] = (uint32_t)top;
    if (top >= 0) {
#pragma unroll
        for (int i = 0; i < NL; i++) a.l[i] = d[i];
    }
    ZK_DBG_ASSERT(a.lmax <= 1, "fe_cond_sub needs normalised limbs");
    ZK_DBG(a.vb = (a.vb > K) ? ((a.vb - K > K) ? a.vb - K : (double)K) : a.vb;)
}

// value < M*m (M <= 16)  ->  value < 2m
template <int M, class Tag> ZK_HD void fe_wreduce(Fe<Tag> &a) {
    ZK_DBG_ASSERT(a.vb <= M, "fe_wreduce<M>: value bound exceeds M");
    if (M > 8) fe_cond_sub<8>(a);
    if (M > 4) fe_cond_sub<4>(a);
    if (M > 2) fe_cond_sub<2>(a);
}
// value < 16m -> canonical representative < m
template <class Tag> ZK_HD Fe<Tag> fe_reduce_full(Fe<Tag> a) {
    ZK_DBG_ASSERT(a.vb <= 16, "fe_reduce_full: value bound exceeds 16 m");
    fe_normalize(a);
    fe_cond_sub<8>(a);
    fe_cond_sub<4>(a);
    fe_cond_sub<2>(a);
    fe_cond_sub<1>(a);
    return a;
}

template <class Tag> ZK_HD bool Fe<Tag>::is_zero() const {
    // Quick reject in three instructions: a value k*m with k < 16 has k*m_0 in its low 29 bits, so
    // l_0 * m_0^-1 mod 2^29 must come out below 16 (C::inv is -m_0^-1).  The low 29 bits of l[0] are those
    // of the value whether or not the limbs are normalised.
    ZK_DBG_ASSERT(vb <= 16, "is_zero: value bound exceeds 16 m");
    if ((((0u - l[0]) * C::inv) & LMASK) >= 16u) return false;
    return fe_reduce_full(*this).raw_is_zero();
}
template <class Tag> ZK_HD bool Fe<Tag>::maybe_zero() const { return (((0u - l[0]) * C::inv) & LMASK) < 16u; }
template <class Tag> ZK_HD bool Fe<Tag>::equals(const Fe &b) const {
    const Fe x = fe_reduce_full(*this), y = fe_reduce_full(b);
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) o |= (x.l[i] ^ y.l[i]);
    return o == 0;
}

template <class Tag> ZK_HD Fe<Tag> fe_add(const Fe<Tag> &a, const Fe<Tag> &b) {
    Fe<Tag> r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.l[i] = a.l[i] + b.l[i];
    ZK_DBG_ASSERT(a.lmax + b.lmax <= 6, "fe_add: limb overflow");
    fe_normalize(r);
    ZK_DBG(r.vb = a.vb + b.vb;)
    ZK_DBG_ASSERT(r.vb <= 1024, "fe_add: value too large");
    return r;
}
// limb-wise sum without carry propagation: only as a direct operand of fe_mul / fe_sqr
template <class Tag> ZK_HD Fe<Tag> fe_add_lazy(const Fe<Tag> &a, const Fe<Tag> &b) {
    Fe<Tag> r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.l[i] = a.l[i] + b.l[i];
    ZK_DBG(r.vb = a.vb + b.vb; r.lmax = a.lmax + b.lmax;)
    return r;
}
// a - b + K*m; requires value(b) <= K*m.  Result normalised, value < value(a) + K*m.
template <int K, class Tag> ZK_HD Fe<Tag> fe_sub_k(const Fe<Tag> &a, const Fe<Tag> &b) {
    typedef FieldConst<Tag> C;
    Fe<Tag> r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.l[i] = a.l[i] + C::kp(K, i) - b.l[i];
    ZK_DBG_ASSERT(b.vb <= K, "fe_sub_k<K>: subtrahend may exceed K*m");
    ZK_DBG_ASSERT(a.lmax <= 2 && b.lmax <= 3, "fe_sub_k: limb overflow");
    fe_normalize(r);
    ZK_DBG(r.vb = a.vb + K;)
    return r;
}
// K*m - a; requires value(a) <= K*m.
template <int K, class Tag> ZK_HD Fe<Tag> fe_neg_k(const Fe<Tag> &a) {
    typedef FieldConst<Tag> C;
    Fe<Tag> r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.l[i] = C::kp(K, i) - a.l[i];
    ZK_DBG_ASSERT(a.vb <= K && a.lmax <= 3, "fe_neg_k<K>: operand may exceed K*m");
    fe_normalize(r);
    ZK_DBG(r.vb = K;)
    return r;
}
// a + b brought below 2m, for normalised a, b < 2m: the same representative as fe_add followed by fe_wreduce<4>, with ONE carry
// chain instead of two.  Whether 2m has to come off is read from the top limbs before any carry is propagated: with
// t = a_8 + b_8 - (2m)_8, the lower limbs of the sum add less than 2 * 2^232 and those of 2m less than 2^232, so
//   t >= 1  =>  a + b - 2m > 0  (subtract),     t <= -2  =>  a + b - 2m < 0  (leave),
// and only t in {-1, 0} (probability 2^-22 for a uniform top limb) needs the full comparison: that case takes the two-chain path.
template <class Tag> ZK_HD Fe<Tag> fe_add_r2(const Fe<Tag> &a, const Fe<Tag> &b) {
    typedef FieldConst<Tag> C;
    ZK_DBG_ASSERT(a.vb <= 2 && b.vb <= 2 && a.lmax <= 1 && b.lmax <= 1, "fe_add_r2: operands must be normalised and < 2m");
    const int32_t t = (int32_t)(a.l[NL - 1] + b.l[NL - 1]) - (int32_t)C::kp(2, NL - 1);
    Fe<Tag> r;
    if ((uint32_t)(t + 1) <= 1u) {
        r = fe_add(a, b);
        fe_cond_sub<2>(r);
    } else {
        const uint32_t take = t > 0 ? 0xffffffffu : 0u;
#pragma unroll
        for (int i = 0; i < NL; i++) r.l[i] = a.l[i] + b.l[i] - (C::kp(2, i) & take);
        fe_normalize(r);
        ZK_DBG(r.vb = 2;)
    }
    return r;
}
// a - b brought into [0, 2m), for normalised a, b < 2m: the same representative as fe_sub_k<2> followed by fe_wreduce<4>.  With
// t = a_8 - b_8 the lower limbs differ by less than 2^232 either way:  t >= 1 => a - b > 0,  t <= -1 => a - b < 0 (add 2m);
// t == 0 takes the two-chain path.
template <class Tag> ZK_HD Fe<Tag> fe_sub_r2(const Fe<Tag> &a, const Fe<Tag> &b) {
    typedef FieldConst<Tag> C;
    ZK_DBG_ASSERT(a.vb <= 2 && b.vb <= 2 && a.lmax <= 1 && b.lmax <= 1, "fe_sub_r2: operands must be normalised and < 2m");
    const int32_t t = (int32_t)a.l[NL - 1] - (int32_t)b.l[NL - 1];
    Fe<Tag> r;
    if (t == 0) {
        r = fe_sub_k<2>(a, b);
        fe_cond_sub<2>(r);
    } else {
        const uint32_t give = t < 0 ? 0xffffffffu : 0u;
#pragma unroll
        for (int i = 0; i < NL; i++) r.l[i] = a.l[i] - b.l[i] + (C::kp(2, i) & give);
        fe_normalize(r);
        ZK_DBG(r.vb = 2;)
    }
    return r;
}
// Unnormalised forms for values that go straight into ONE multiplication (fe_mul / fe_dot / fe_mulsub take limbs
// up to a few 2^29: 9 * la * lb + 10 < 64 per product column): no carry propagation at all.
//   fe_sub_lazy<K>: a - b + (K+1)*m, needs b normalised and <= K*m; limbs < (la + 2) * 2^29.
//   fe_neg_lazy<K>: (K+1)*m - a,     same conditions on a;           limbs < 2 * 2^29.
template <int K, class Tag> ZK_HD Fe<Tag> fe_sub_lazy(const Fe<Tag> &a, const Fe<Tag> &b) {
    typedef FieldConst<Tag> C;
    static_assert(K + 1 <= 16, "fe_sub_lazy: no constant for (K+1)*m");
    Fe<Tag> r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.l[i] = a.l[i] + C::kpl(K + 1, i) - b.l[i];
    ZK_DBG_ASSERT(b.vb <= K && b.lmax <= 1, "fe_sub_lazy<K>: subtrahend must be normalised and <= K*m");
    ZK_DBG_ASSERT(a.lmax <= 2, "fe_sub_lazy: limb overflow");
    ZK_DBG(r.vb = a.vb + K + 1; r.lmax = a.lmax + 2;)
    return r;
}
template <int K, class Tag> ZK_HD Fe<Tag> fe_neg_lazy(const Fe<Tag> &a) {
    typedef FieldConst<Tag> C;
    static_assert(K + 1 <= 16, "fe_neg_lazy: no constant for (K+1)*m");
    Fe<Tag> r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.l[i] = C::kpl(K + 1, i) - a.l[i];
    ZK_DBG_ASSERT(a.vb <= K && a.lmax <= 1, "fe_neg_lazy<K>: operand must be normalised and <= K*m");
    ZK_DBG(r.vb = K + 1; r.lmax = 2;)
    return r;
}
// a - b - 2c + K*m with one carry propagation (the X coordinate of every addition); needs b + 2c <= K*m.
template <int K, class Tag> ZK_HD Fe<Tag> fe_sub2(const Fe<Tag> &a, const Fe<Tag> &b, const Fe<Tag> &c) {
    typedef FieldConst<Tag> C;
    Fe<Tag> r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.l[i] = a.l[i] + C::kp(K, i) - b.l[i] - 2u * c.l[i];
    ZK_DBG_ASSERT(b.vb + 2 * c.vb <= K, "fe_sub2<K>: b + 2c may exceed K*m");
    ZK_DBG_ASSERT(a.lmax <= 2 && b.lmax <= 1 && c.lmax <= 1, "fe_sub2: limb overflow");
    fe_normalize(r);
    ZK_DBG(r.vb = a.vb + K;)
    return r;
}
template <class Tag> ZK_HD Fe<Tag> fe_dbl(const Fe<Tag> &a) {
    Fe<Tag> r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.l[i] = a.l[i] << 1;
    ZK_DBG_ASSERT(a.lmax <= 3, "fe_dbl: limb overflow");
    fe_normalize(r);
    ZK_DBG(r.vb = 2 * a.vb;)
    return r;
}
// 2a with no carry propagation (limbs < 2 * 2^29 for a normalised a): only as a direct operand of fe_mul
template <class Tag> ZK_HD Fe<Tag> fe_dbl_lazy(const Fe<Tag> &a) {
    Fe<Tag> r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.l[i] = a.l[i] << 1;
    ZK_DBG_ASSERT(a.lmax <= 2, "fe_dbl_lazy: limb overflow");
    ZK_DBG(r.vb = 2 * a.vb; r.lmax = 2 * a.lmax;)
    return r;
}
template <class Tag> ZK_HD Fe<Tag> fe_triple(const Fe<Tag> &a) {
    Fe<Tag> r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.l[i] = a.l[i] * 3u;
    ZK_DBG_ASSERT(a.lmax <= 2, "fe_triple: limb overflow");
    fe_normalize(r);
    ZK_DBG(r.vb = 3 * a.vb;)
    return r;
}

// Montgomery product a*b*2^-261 (product scanning; one 64-bit running column accumulator).
template <class Tag> ZK_HD Fe<Tag> fe_mul(const Fe<Tag> &a, const Fe<Tag> &b) {
    typedef FieldConst<Tag> C;
    uint32_t q[NL];
    Fe<Tag> r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < NL; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)q[i] * C::mod(k - i);
        q[k] = ((uint32_t)acc * C::inv) & LMASK;
        acc += (uint64_t)q[k] * C::mod(0);
        acc >>= LB;
    }
#pragma unroll
    for (int k = NL; k < 2 * NL - 1; k++) {
#pragma unroll
        for (int i = k - NL + 1; i < NL; i++) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
        for (int i = k - NL + 1; i < NL; i++) acc += (uint64_t)q[i] * C::mod(k - i);
        r.l[k - NL] = (uint32_t)acc & LMASK;
        acc >>= LB;
    }
    r.l[NL - 1] = (uint32_t)acc;
    ZK_DBG_ASSERT(a.vb * b.vb < 169.0, "fe_mul: value bounds va*vb >= 169");
    ZK_DBG_ASSERT(9.0 * a.lmax * b.lmax + 9.0 + 1.0 < 64.0, "fe_mul: column accumulator may overflow");  // units of 2^58
    ZK_DBG(r.vb = a.vb * b.vb / 169.0 + 1.0; r.lmax = 1;)
    return r;
}

// a*b - c in the Montgomery domain (c normalised, <= K*m) with no separate subtraction: (K+1)*m - c is added to the
// UPPER half of the double-width product (limb j lands in column NL + j, i.e. times R), so the one reduction delivers
// (a*b + ((K+1)*m - c)*R) / R = a*b/R + (K+1)*m - c.  Nine column additions instead of a subtraction pass plus a carry
// propagation; result normalised, value < va*vb/169 + K + 2 (in units of m).
template <int K, class Tag> ZK_HD Fe<Tag> fe_mul_minus(const Fe<Tag> &a, const Fe<Tag> &b, const Fe<Tag> &c) {
    typedef FieldConst<Tag> C;
    const Fe<Tag> nc = fe_neg_lazy<K>(c);   // limbs < 2^30
    uint32_t q[NL];
    Fe<Tag> r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < NL; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)q[i] * C::mod(k - i);
        q[k] = ((uint32_t)acc * C::inv) & LMASK;
        acc += (uint64_t)q[k] * C::mod(0);
        acc >>= LB;
    }
#pragma unroll
    for (int k = NL; k < 2 * NL - 1; k++) {
#pragma unroll
        for (int i = k - NL + 1; i < NL; i++) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
        for (int i = k - NL + 1; i < NL; i++) acc += (uint64_t)q[i] * C::mod(k - i);
        acc += nc.l[k - NL];
        r.l[k - NL] = (uint32_t)acc & LMASK;
        acc >>= LB;
    }
    r.l[NL - 1] = (uint32_t)acc + nc.l[NL - 1];
    ZK_DBG_ASSERT(a.vb * b.vb < 169.0, "fe_mul_minus: value bounds va*vb >= 169");
    ZK_DBG_ASSERT(9.0 * a.lmax * b.lmax + 9.0 + 1.0 + 1.0 < 64.0, "fe_mul_minus: column accumulator may overflow");
    ZK_DBG(r.vb = a.vb * b.vb / 169.0 + 1.0 + K + 1; r.lmax = 1;)
    return r;
}

// sum_t a[t] * b[t] with ONE Montgomery reduction: the N products are accumulated column-wise before the
// reduction runs, saving (N-1) * (81 + 9) multiply-adds over N separate fe_mul.  Operands: limbs < la, lb times 2^29
// with sum_t 9 la[t] lb[t] + 10 < 64 (normalised operands: N * 9 + 10), values sum_t va[t] * vb[t] < 169;  result < 2m.
template <int N, class Tag> ZK_HD Fe<Tag> fe_dot(const Fe<Tag> *const (&a)[N], const Fe<Tag> *const (&b)[N]) {
    typedef FieldConst<Tag> C;
    static_assert(N * 9 + 10 < 64, "fe_dot: column accumulator would overflow");
    uint32_t q[NL];
    Fe<Tag> r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < NL; k++) {
#pragma unroll
        for (int t = 0; t < N; t++)
#pragma unroll
            for (int i = 0; i <= k; i++) acc += (uint64_t)a[t]->l[i] * b[t]->l[k - i];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)q[i] * C::mod(k - i);
        q[k] = ((uint32_t)acc * C::inv) & LMASK;
        acc += (uint64_t)q[k] * C::mod(0);
        acc >>= LB;
    }
#pragma unroll
    for (int k = NL; k < 2 * NL - 1; k++) {
#pragma unroll
        for (int t = 0; t < N; t++)
#pragma unroll
            for (int i = k - NL + 1; i < NL; i++) acc += (uint64_t)a[t]->l[i] * b[t]->l[k - i];
#pragma unroll
        for (int i = k - NL + 1; i < NL; i++) acc += (uint64_t)q[i] * C::mod(k - i);
        r.l[k - NL] = (uint32_t)acc & LMASK;
        acc >>= LB;
    }
    r.l[NL - 1] = (uint32_t)acc;
#ifdef ZK_FIELD_DEBUG
    double vsum = 0, lsum = 0;
    for (int t = 0; t < N; t++) {
        vsum += a[t]->vb * b[t]->vb;
        lsum += 9.0 * a[t]->lmax * b[t]->lmax;
    }
    ZK_DBG_ASSERT(lsum + 9.0 + 1.0 < 64.0, "fe_dot: column accumulator may overflow");
    ZK_DBG_ASSERT(vsum < 169.0, "fe_dot: sum of value bounds >= 169");
    r.vb = vsum / 169.0 + 1.0;
    r.lmax = 1;
#endif
    return r;
}
// a*b - c*d (c normalised, <= K*m), one reduction;  result < 2m.  a or b may be a lazy difference (fe_sub_lazy);
// values: va*vb + (K+1)*vd < 169.
template <int K, class Tag> ZK_HD Fe<Tag> fe_mulsub(const Fe<Tag> &a, const Fe<Tag> &b, const Fe<Tag> &c, const Fe<Tag> &d) {
    const Fe<Tag> nc = fe_neg_lazy<K>(c);
    const Fe<Tag> *const x[2] = {&a, &nc};
    const Fe<Tag> *const y[2] = {&b, &d};
    return fe_dot<2>(x, y);
}

// fe_dot with an addend riding in the UPPER half of the double-width sum (limb j lands in column NL + j, i.e. times R), as in
// fe_mul_minus: sum_t a[t]*b[t] / R + add out of the one reduction.  add: limbs < 2^30, not necessarily normalised (typically
// fe_neg_lazy<K>(c): the F_p^2 form of "a*b - c", both components of which are two-product sums).  Result normalised,
// value < sum va*vb / 169 + 1 + value(add).
template <int N, class Tag> ZK_HD Fe<Tag> fe_dot_add(const Fe<Tag> *const (&a)[N], const Fe<Tag> *const (&b)[N], const Fe<Tag> &add) {
    typedef FieldConst<Tag> C;
    static_assert(N * 9 + 11 < 64, "fe_dot_add: column accumulator would overflow");
    uint32_t q[NL];
    Fe<Tag> r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < NL; k++) {
#pragma unroll
        for (int t = 0; t < N; t++)
#pragma unroll
            for (int i = 0; i <= k; i++) acc += (uint64_t)a[t]->l[i] * b[t]->l[k - i];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)q[i] * C::mod(k - i);
        q[k] = ((uint32_t)acc * C::inv) & LMASK;
        acc += (uint64_t)q[k] * C::mod(0);
        acc >>= LB;
    }
#pragma unroll
    for (int k = NL; k < 2 * NL - 1; k++) {
#pragma unroll
        for (int t = 0; t < N; t++)
#pragma unroll
            for (int i = k - NL + 1; i < NL; i++) acc += (uint64_t)a[t]->l[i] * b[t]->l[k - i];
#pragma unroll
        for (int i = k - NL + 1; i < NL; i++) acc += (uint64_t)q[i] * C::mod(k - i);
        acc += add.l[k - NL];
        r.l[k - NL] = (uint32_t)acc & LMASK;
        acc >>= LB;
    }
    r.l[NL - 1] = (uint32_t)acc + add.l[NL - 1];
#ifdef ZK_FIELD_DEBUG
    double vsum = 0, lsum = 0;
    for (int t = 0; t < N; t++) {
        vsum += a[t]->vb * b[t]->vb;
        lsum += 9.0 * a[t]->lmax * b[t]->lmax;
    }
    ZK_DBG_ASSERT(lsum + 9.0 + 1.0 + 1.0 < 64.0, "fe_dot_add: column accumulator may overflow");
    ZK_DBG_ASSERT(vsum < 169.0, "fe_dot_add: sum of value bounds >= 169");
    ZK_DBG_ASSERT(add.lmax <= 2 && add.vb <= 16, "fe_dot_add: addend out of range");
    r.vb = vsum / 169.0 + 1.0 + add.vb;
    r.lmax = 1;
#endif
    return r;
}

// Montgomery square: 45 distinct products instead of 81.
template <class Tag> ZK_HD Fe<Tag> fe_sqr(const Fe<Tag> &a) {
    typedef FieldConst<Tag> C;
    uint32_t q[NL], a2[NL];
    Fe<Tag> r;
#pragma unroll
    for (int i = 0; i < NL; i++) a2[i] = a.l[i] << 1;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < NL; k++) {
#pragma unroll
        for (int i = 0; 2 * i < k; i++) acc += (uint64_t)a2[i] * a.l[k - i];
        if ((k & 1) == 0) acc += (uint64_t)a.l[k >> 1] * a.l[k >> 1];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)q[i] * C::mod(k - i);
        q[k] = ((uint32_t)acc * C::inv) & LMASK;
        acc += (uint64_t)q[k] * C::mod(0);
        acc >>= LB;
    }
#pragma unroll
    for (int k = NL; k < 2 * NL - 1; k++) {
#pragma unroll
        for (int i = k - NL + 1; 2 * i < k; i++) acc += (uint64_t)a2[i] * a.l[k - i];
        if ((k & 1) == 0) acc += (uint64_t)a.l[k >> 1] * a.l[k >> 1];
#pragma unroll
        for (int i = k - NL + 1; i < NL; i++) acc += (uint64_t)q[i] * C::mod(k - i);
        r.l[k - NL] = (uint32_t)acc & LMASK;
        acc >>= LB;
    }
    r.l[NL - 1] = (uint32_t)acc;
    ZK_DBG_ASSERT(a.vb * a.vb < 169.0, "fe_sqr: value bound va^2 >= 169");
    ZK_DBG_ASSERT(9.0 * a.lmax * a.lmax + 9.0 + 1.0 < 64.0 && a.lmax <= 4, "fe_sqr: column accumulator may overflow");
    ZK_DBG(r.vb = a.vb * a.vb / 169.0 + 1.0; r.lmax = 1;)
    return r;
}

template <class Tag> ZK_HD Fe<Tag> fe_to_mont(const Fe<Tag> &a) {
    typedef FieldConst<Tag> C;
    Fe<Tag> r2;
#pragma unroll
    for (int i = 0; i < NL; i++) r2.l[i] = C::r2(i);
    ZK_DBG(r2.vb = 1; r2.lmax = 1;)
    return fe_mul(a, r2);
}
template <class Tag> ZK_HD Fe<Tag> fe_from_mont(const Fe<Tag> &a) {
    Fe<Tag> one;
#pragma unroll
    for (int i = 0; i < NL; i++) one.l[i] = (i == 0);
    ZK_DBG(one.vb = 1; one.lmax = 1;)
    return fe_mul(a, one);
}

// Canonical 256-bit little-endian words (the ABI encoding) <-> normalised limbs.
template <class Tag> ZK_HD Fe<Tag> fe_from_words(const uint32_t w[8]) {
    Fe<Tag> r;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        const int bit = LB * i, word = bit >> 5, sh = bit & 31;
        uint32_t v = w[word] >> sh;
        if (sh + LB > 32 && word + 1 < 8) v |= w[word + 1] << (32 - sh);
        r.l[i] = (i < NL - 1) ? (v & LMASK) : v;
    }
    ZK_DBG(r.vb = 5.3; r.lmax = 1;)  // any 256-bit integer is < 5.3 m; canonical inputs are < m
    return r;
}
// Input: normalised limbs, value < 2^256 (canonical after fe_reduce_full; lazy values < 4m also fit).
template <class Tag> ZK_HD void fe_to_words(const Fe<Tag> &a, uint32_t w[8]) {
    ZK_DBG_ASSERT(a.vb <= 5.25 && a.lmax <= 1, "fe_to_words: value may not fit 256 bits");
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int bit = 32 * j, li = bit / LB, sh = bit - li * LB;  // word j starts inside limb li
        uint64_t v = (uint64_t)a.l[li] >> sh;
        if (li + 1 < NL) v |= (uint64_t)a.l[li + 1] << (LB - sh);
        if (li + 2 < NL && 2 * LB - sh < 32) v |= (uint64_t)a.l[li + 2] << (2 * LB - sh);
        w[j] = (uint32_t)v;
    }
}

// a^e for a 256-bit exponent given as 32-bit words (square-and-multiply, MSB first); a < 2m.
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
    Fe<Tag> m;
#pragma unroll
    for (int i = 0; i < NL; i++) m.l[i] = C::mod(i);
    ZK_DBG(m.vb = 1; m.lmax = 1;)
    m.l[0] -= 2;  // low limb of both moduli is > 2
    uint32_t e[8];
    fe_to_words(m, e);
    return fe_pow(a, e);
}

// c ? a : b limb by limb (v_cndmask): the ?: operator on two element lvalues selects between their ADDRESSES, which parks both in scratch memory
template <class Tag> ZK_HD Fe<Tag> fe_select(bool c, const Fe<Tag> &a, const Fe<Tag> &b) {
    Fe<Tag> r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.l[i] = c ? a.l[i] : b.l[i];
    ZK_DBG(r.vb = c ? a.vb : b.vb; r.lmax = c ? a.lmax : b.lmax;)
    return r;
}

typedef Fe<FpTag> Fp;
typedef Fe<FrTag> Fr;

// Generic names used by curve.h (the host epilogue types in host_field.h provide the same set).
// mul outputs are < 2m, which is what the K arguments in curve.h are derived from.
template <int K, class Tag> ZK_HD Fe<Tag> fe_sub(const Fe<Tag> &a, const Fe<Tag> &b) { return fe_sub_k<K>(a, b); }
template <int K, class Tag> ZK_HD Fe<Tag> fe_neg(const Fe<Tag> &a) { return fe_neg_k<K>(a); }
// fe_sub_once<K>(a, b): a - b for a result that is used as ONE multiplication operand and nothing else.
template <int K, class Tag> ZK_HD Fe<Tag> fe_sub_once(const Fe<Tag> &a, const Fe<Tag> &b) { return fe_sub_lazy<K>(a, b); }
// fe_neg_once<K>(a): -a, again only as one multiplication operand (value <= (K+1)*m, limbs < 2^30); fe_tidy makes such a
// value an ordinary normalised element again (same value).
template <int K, class Tag> ZK_HD Fe<Tag> fe_neg_once(const Fe<Tag> &a) { return fe_neg_lazy<K>(a); }
template <class Tag> ZK_HD Fe<Tag> fe_tidy(Fe<Tag> a) {
    fe_normalize(a);
    return a;
}

// ---------------------------------------------------------------------------------------
// F_p^2 = F_p[i]/(i^2+1); element c0 + c1*i  (py_ecc FQ2, coeffs [c0, c1]).
// Every F_p^2 result is weakly reduced (< 2p per component) so that the Karatsuba operand sums
// stay inside fe_mul's contract: (2p + 2p)^2 = 16 p^2 < 169 p^2.
struct Fp2 {
    Fp c0, c1;
    static constexpr int CANON_WORDS = 16;
    static ZK_HD Fp2 zero() { return Fp2{Fp::zero(), Fp::zero()}; }
    static ZK_HD Fp2 one() { return Fp2{Fp::one(), Fp::zero()}; }
    ZK_HD bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
    ZK_HD bool maybe_zero() const { return c0.maybe_zero() && c1.maybe_zero(); }
    ZK_HD bool equals(const Fp2 &b) const { return c0.equals(b.c0) && c1.equals(b.c1); }
};

ZK_HD Fp2 fe_select(bool c, const Fp2 &a, const Fp2 &b) { return Fp2{fe_select(c, a.c0, b.c0), fe_select(c, a.c1, b.c1)}; }
ZK_HD Fp2 fe_add(const Fp2 &a, const Fp2 &b) {
    Fp2 r{fe_add(a.c0, b.c0), fe_add(a.c1, b.c1)};
    fe_wreduce<4>(r.c0);
    fe_wreduce<4>(r.c1);
    return r;
}
template <int K> ZK_HD Fp2 fe_sub(const Fp2 &a, const Fp2 &b) {  // inputs < 2p per component
    Fp2 r{fe_sub_k<2>(a.c0, b.c0), fe_sub_k<2>(a.c1, b.c1)};
    fe_wreduce<4>(r.c0);
    fe_wreduce<4>(r.c1);
    return r;
}
template <int K> ZK_HD Fp2 fe_neg(const Fp2 &a) { return Fp2{fe_neg_k<2>(a.c0), fe_neg_k<2>(a.c1)}; }
template <int K> ZK_HD Fp2 fe_sub_once(const Fp2 &a, const Fp2 &b) { return fe_sub<K>(a, b); }  // F_p^2 keeps components < 2p
template <int K> ZK_HD Fp2 fe_neg_once(const Fp2 &a) { return fe_neg<K>(a); }
ZK_HD Fp2 fe_tidy(const Fp2 &a) { return a; }
ZK_HD Fp2 fe_dbl(const Fp2 &a) {
    Fp2 r{fe_dbl(a.c0), fe_dbl(a.c1)};
    fe_wreduce<4>(r.c0);
    fe_wreduce<4>(r.c1);
    return r;
}
template <int K> ZK_HD Fp2 fe_sub2(const Fp2 &a, const Fp2 &b, const Fp2 &c) {  // a - b - 2c; inputs < 2p per component
    Fp2 r{fe_sub2<6>(a.c0, b.c0, c.c0), fe_sub2<6>(a.c1, b.c1, c.c1)};             // < 8p
    fe_wreduce<8>(r.c0);
    fe_wreduce<8>(r.c1);
    return r;
}
ZK_HD Fp2 fe_triple(const Fp2 &a) {
    Fp2 r{fe_triple(a.c0), fe_triple(a.c1)};
    fe_wreduce<8>(r.c0);
    fe_wreduce<8>(r.c1);
    return r;
}
// Schoolbook with lazy reduction: c0 = a0 b0 + a1 (-b1), c1 = a0 b1 + a1 b0 -- four limb products but only two
// Montgomery reductions (fe_dot<2>), cheaper here than Karatsuba's three full multiplications plus its
// additions.  Components < 2p in, < 2p out (sum of value bounds 8 < 169).
ZK_HD Fp2 fe_mul(const Fp2 &a, const Fp2 &b) {
    const Fp nb1 = fe_neg_lazy<2>(b.c1);   // 3p - b1, limbs < 2 * 2^29: column bound 9 + 18 + 10 < 64
    const Fp *const x[2] = {&a.c0, &a.c1};
    const Fp *const y0[2] = {&b.c0, &nb1};
    const Fp *const y1[2] = {&b.c1, &b.c0};
    return Fp2{fe_dot<2>(x, y0), fe_dot<2>(x, y1)};
}
template <int K> ZK_HD Fp2 fe_mul_minus(const Fp2 &a, const Fp2 &b, const Fp2 &c) { return fe_sub<K>(fe_mul(a, b), c); }
// a*b - c*d in F_p^2 with two reductions (fe_dot<4> per component).  K is ignored: components are < 2p.
template <int K> ZK_HD Fp2 fe_mulsub(const Fp2 &a, const Fp2 &b, const Fp2 &c, const Fp2 &d) {
    // nc0 enters both components next to another lazy operand, so it stays normalised: 9 + 18 + 9 + 9 + 10 < 64
    const Fp nb1 = fe_neg_lazy<2>(b.c1), nc0 = fe_neg_k<2>(c.c0), nc1 = fe_neg_lazy<2>(c.c1);
    const Fp *const x0[4] = {&a.c0, &a.c1, &nc0, &c.c1};
    const Fp *const y0[4] = {&b.c0, &nb1, &d.c0, &d.c1};
    const Fp *const x1[4] = {&a.c0, &a.c1, &nc0, &nc1};
    const Fp *const y1[4] = {&b.c1, &b.c0, &d.c1, &d.c0};
    return Fp2{fe_dot<4>(x0, y0), fe_dot<4>(x1, y1)};
}
// (c0+c1 i)^2 = (c0+c1)(c0-c1) + 2 c0 c1 i : 2 base-field products.
ZK_HD Fp2 fe_sqr(const Fp2 &a) {
    const Fp t = fe_mul(a.c0, a.c1);
    const Fp u = fe_mul(fe_add_lazy(a.c0, a.c1), fe_sub_k<4>(a.c0, a.c1));   // components up to 4p (X of a G2 accumulator): 8 * 8 < 169
    Fp2 r{u, fe_dbl(t)};
    fe_wreduce<4>(r.c1);
    return r;
}
// ---- lazier forms for the G2 mixed addition (curve.h): components may exceed 2p where the comment says so; everything is normalised.
// a*b - c with c riding in the upper columns of both components' sums (no separate subtraction, carry chain or conditional
// subtraction): a, b < 2p (b.c1 normalised), c <= K*p normalised;  components < 10/169 + K + 2 (in units of p).
template <int K> ZK_HD Fp2 fp2_mul_minus_lazy(const Fp2 &a, const Fp2 &b, const Fp2 &c) {
    const Fp nb1 = fe_neg_lazy<2>(b.c1);
    const Fp n0 = fe_neg_lazy<K>(c.c0), n1 = fe_neg_lazy<K>(c.c1);   // (K+1)p - c, limbs < 2^30
    const Fp *const x[2] = {&a.c0, &a.c1};
    const Fp *const y0[2] = {&b.c0, &nb1};
    const Fp *const y1[2] = {&b.c1, &b.c0};
    return Fp2{fe_dot_add<2>(x, y0, n0), fe_dot_add<2>(x, y1, n1)};
}
// a^2 for components <= K*p (K <= 7: (2K)(2K + 1) < 169 needs K <= 6.2, the callers' bounds are 4.06 and 6.06): the doubling of
// 2 c0 c1 is a limb shift of one operand, nothing is conditionally subtracted;  components < 2p.
template <int K> ZK_HD Fp2 fp2_sqr_lazy(const Fp2 &a) {
    const Fp u = fe_mul(fe_add_lazy(a.c0, a.c1), fe_sub_k<K>(a.c0, a.c1));
    const Fp t2 = fe_mul(fe_dbl_lazy(a.c0), a.c1);
    return Fp2{u, t2};
}
// a*b - c*d for a <= ~4p, b <= KB*p, c < 2p, d < 2p (all normalised), two reductions;  components < 2p.
template <int KB> ZK_HD Fp2 fp2_mulsub_lazy(const Fp2 &a, const Fp2 &b, const Fp2 &c, const Fp2 &d) {
    const Fp nb1 = fe_neg_lazy<KB>(b.c1), nc0 = fe_neg_k<2>(c.c0), nc1 = fe_neg_lazy<2>(c.c1);
    const Fp *const x0[4] = {&a.c0, &a.c1, &nc0, &c.c1};
    const Fp *const y0[4] = {&b.c0, &nb1, &d.c0, &d.c1};
    const Fp *const x1[4] = {&a.c0, &a.c1, &nc0, &nc1};
    const Fp *const y1[4] = {&b.c1, &b.c0, &d.c1, &d.c0};
    return Fp2{fe_dot<4>(x0, y0), fe_dot<4>(x1, y1)};
}
ZK_HD Fp2 fe_inv(const Fp2 &a) {
    const Fp d = fe_inv(fe_add(fe_sqr(a.c0), fe_sqr(a.c1)));
    return Fp2{fe_mul(a.c0, d), fe_neg_k<2>(fe_mul(a.c1, d))};
}
ZK_HD Fp2 fe_to_mont(const Fp2 &a) { return Fp2{fe_to_mont(a.c0), fe_to_mont(a.c1)}; }
ZK_HD Fp2 fe_from_mont(const Fp2 &a) { return Fp2{fe_from_mont(a.c0), fe_from_mont(a.c1)}; }
ZK_HD Fp2 fe_reduce_full(const Fp2 &a) { return Fp2{fe_reduce_full(a.c0), fe_reduce_full(a.c1)}; }

// Canonical words <-> element, generic over F_p / F_p^2 (Montgomery conversion NOT included).
ZK_HD Fp fe_load_canonical(const uint32_t *w, Fp *) { return fe_from_words<FpTag>(w); }
ZK_HD Fp2 fe_load_canonical(const uint32_t *w, Fp2 *) { return Fp2{fe_from_words<FpTag>(w), fe_from_words<FpTag>(w + 8)}; }
ZK_HD void fe_store_canonical(uint32_t *w, const Fp &a) { fe_to_words(fe_reduce_full(a), w); }
ZK_HD void fe_store_canonical(uint32_t *w, const Fp2 &a) {
    fe_to_words(fe_reduce_full(a.c0), w);
    fe_to_words(fe_reduce_full(a.c1), w + 8);
}

}  // namespace zk
