// curve.h -- BN254 G1 (over F_p) and G2 (twist over F_p^2) group arithmetic, a = 0 curves,
// in extended Jacobian "XYZZ" coordinates (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2), which give the
// cheapest mixed addition (8M+2S) for Pippenger bucket accumulation.  Infinity: ZZ == 0.
//
// Replaces (as arithmetic) py_ecc.bn128.add/double/multiply/neg used through the aliases at
// zkp/groth16/proving.py:12-15 and the wrappers zkp/plonk/field.py:72-115.  Results are
// converted back to canonical affine, so they equal the reference's affine values exactly.
#pragma once
#include "field.h"

// true when the condition holds in ANY lane of the wavefront (a wave-uniform value: branching on it is a scalar branch)
#if defined(__HIP_DEVICE_COMPILE__)
#define ZK_WAVE_ANY(c) (__builtin_amdgcn_ballot_w64(c) != 0)
#else
#define ZK_WAVE_ANY(c) (c)
#endif
namespace zk {

template <class F> struct Affine {
    F x, y;  // Montgomery form; infinity encoded as x = y = 0 (not on either curve)
    ZK_HD bool is_inf() const { return x.is_zero() && y.is_zero(); }
    static ZK_HD Affine inf() { return Affine{F::zero(), F::zero()}; }
};

template <class F> struct Xyzz {
    F x, y, zz, zzz;
    ZK_HD bool is_inf() const { return zz.is_zero(); }
    static ZK_HD Xyzz inf() { return Xyzz{F::zero(), F::zero(), F::zero(), F::zero()}; }
    static ZK_HD Xyzz from_affine(const Affine<F> &p) {
        if (p.is_inf()) return inf();
        return Xyzz{p.x, p.y, F::one(), F::one()};
    }
};

// Value bounds (multiples of the modulus m) maintained by every routine below; they are what the
// K arguments of fe_sub<K> / fe_neg<K> encode (field.h contracts; mul outputs are < 2m):
//   affine coordinates < 2m;   XYZZ:  X < 8m,  Y < 4m,  ZZ < 2m,  ZZZ < 2m   (G2: X < 4p per component, everything else < 2p).
template <class F> ZK_HD Affine<F> affine_neg(const Affine<F> &p) { return Affine<F>{p.x, fe_neg<2>(p.y)}; }
template <class F> ZK_HD Xyzz<F> xyzz_neg(const Xyzz<F> &p) { return Xyzz<F>{p.x, fe_neg<4>(p.y), p.zz, p.zzz}; }

// 2*P for affine P (mdbl-2008-s-1); y may be up to 3m (a negated table entry).
template <class F> ZK_HD Xyzz<F> xyzz_dbl_affine(const Affine<F> &p) {
    if (p.is_inf() || p.y.is_zero()) return Xyzz<F>::inf();
    F u = fe_dbl(p.y);                       // < 6m
    F v = fe_sqr(u);
    F w = fe_mul(u, v);
    F s = fe_mul(p.x, v);
    F m = fe_triple(fe_sqr(p.x));            // < 6m
    Xyzz<F> r;
    r.x = fe_sub<4>(fe_sqr(m), fe_dbl(s));   // < 6m
    r.y = fe_mulsub<3>(m, fe_sub<6>(s, r.x), p.y, w);  // 6*8 + 4*2 < 169;  < 2m
    r.zz = v;
    r.zzz = w;
    return r;
}

// 2*P (dbl-2008-s-1).
template <class F> ZK_HD Xyzz<F> xyzz_dbl(const Xyzz<F> &p) {
    if (p.is_inf() || p.y.is_zero()) return Xyzz<F>::inf();
    F u = fe_dbl(p.y);                       // < 8m
    F v = fe_sqr(u);
    F w = fe_mul(u, v);
    F s = fe_mul(p.x, v);
    F m = fe_triple(fe_sqr(p.x));            // < 6m
    Xyzz<F> r;
    r.x = fe_sub<4>(fe_sqr(m), fe_dbl(s));   // < 6m
    r.y = fe_mulsub<4>(m, fe_sub<6>(s, r.x), p.y, w);  // 6*8 + 4*2 < 169;  < 2m
    r.zz = fe_mul(v, p.zz);
    r.zzz = fe_mul(w, p.zzz);
    return r;
}

// acc += q, q affine (madd-2008-s), all exceptional cases handled.  q.y may be a single-use negation (fe_neg_once<2>:
// value <= 3m, limbs not normalised): it only enters one product here; the two rare paths that keep it tidy it first.
template <class F> ZK_HD void xyzz_add_affine(Xyzz<F> &acc, const Affine<F> &q) {
    if (q.is_inf()) return;
    if (acc.is_inf()) {
        acc = Xyzz<F>{q.x, fe_tidy(q.y), F::one(), F::one()};   // Y <= 3m < 4m
        return;
    }
    F p = fe_mul_minus<8>(q.x, acc.zz, acc.x);   // U2 - X1;  < 10.1m
    F r = fe_mul_minus<4>(q.y, acc.zzz, acc.y);  // S2 - Y1;  < 6.1m
    if (p.is_zero()) {
        if (r.is_zero())
            acc = xyzz_dbl_affine(Affine<F>{q.x, fe_tidy(q.y)});
        else
            acc = Xyzz<F>::inf();
        return;
    }
    F pp = fe_sqr(p);                        // 10.1^2 < 169
    F ppp = fe_mul(p, pp);
    F qq = fe_mul(acc.x, pp);                // 8*2
    F x3 = fe_sub2<6>(fe_sqr(r), ppp, qq);                           // r^2 - ppp - 2 qq;  < 8m
    F y3 = fe_mulsub<4>(r, fe_sub_once<8>(qq, x3), acc.y, ppp);      // 6.1*11 + 5*2 < 169;  < 2m
    acc.x = x3;
    acc.y = y3;
    acc.zz = fe_mul(acc.zz, pp);
    acc.zzz = fe_mul(acc.zzz, ppp);
}

// The same for G2, with the lazier F_p^2 forms of field.h (the accumulate kernel is issue-bound, and every conditional subtraction
// and carry chain that F_p^2's "everything below 2p" convention costs is an instruction it executes 13-16 million times per MSM):
// bounds per component  X < 4p,  Y < 2p,  ZZ, ZZZ < 2p;  P = U2 - X1 < 6.06p and R = S2 - Y1 < 4.06p come straight out of their
// products (fp2_mul_minus_lazy), their squares take them as they are (fp2_sqr_lazy), X3 needs ONE conditional subtraction (of 4p)
// instead of three, Q - X3 is only normalised.  q.y may be a negated table entry (fe_neg_once<2> of F_p^2: normalised, <= 2p).
ZK_HD void xyzz_add_affine(Xyzz<Fp2> &acc, const Affine<Fp2> &q) {
    if (q.is_inf()) return;
    const Fp2 p = fp2_mul_minus_lazy<4>(q.x, acc.zz, acc.x);    // < 6.06p
    const Fp2 r = fp2_mul_minus_lazy<2>(q.y, acc.zzz, acc.y);   // < 4.06p
    // The exceptional cases are classified HERE, completely, before the main path starts: under SIMT both sides of a divergent
    // branch are laid out one after the other, and whatever the later side needs stays in registers across the earlier one -- with
    // `if (p.is_zero()) { ... dbl(q) ... }` written the usual way that was q (36 registers) and the limbs of p and r for the full
    // zero tests (the kernel spilled).  The full tests sit behind a wave-uniform branch (almost never taken), the doubling works
    // on acc itself (acc == q as points), and only acc == infinity still reads q.
    int special = 0;   // 1: acc is infinity, 2: acc == q, 3: acc == -q
    const bool c_inf = acc.zz.maybe_zero(), c_p = p.maybe_zero();
    if (ZK_WAVE_ANY(c_inf || c_p)) {
        if (c_inf && acc.is_inf())
            special = 1;
        else if (c_p && p.is_zero())
            special = r.is_zero() ? 2 : 3;
    }
    if (special) {
        if (special == 1)
            acc = Xyzz<Fp2>{q.x, q.y, Fp2::one(), Fp2::one()};
        else if (special == 2)
            acc = xyzz_dbl(acc);
        else
            acc = Xyzz<Fp2>::inf();
        return;
    }
    const Fp2 pp = fp2_sqr_lazy<7>(p);       // 12.12 * 13.06 < 169
    const Fp2 ppp = fe_mul(p, pp);           // 6.06 * (2 + 3)
    const Fp2 qq = fe_mul(acc.x, pp);        // 4 * (2 + 3)
    const Fp2 rr = fp2_sqr_lazy<5>(r);       // 8.12 * 9.06
    Fp2 x3{fe_sub2<6>(rr.c0, ppp.c0, qq.c0), fe_sub2<6>(rr.c1, ppp.c1, qq.c1)};   // r^2 - ppp - 2 qq + 6p < 8p
    fe_cond_sub<4>(x3.c0);                   // < 4p
    fe_cond_sub<4>(x3.c1);
    const Fp2 d{fe_sub_k<4>(qq.c0, x3.c0), fe_sub_k<4>(qq.c1, x3.c1)};            // < 6p
    acc.y = fp2_mulsub_lazy<6>(r, d, acc.y, ppp);                                  // 4.06 * (6 + 7) + 2 * 2 + 3 * 2 < 169;  < 2p
    acc.x = x3;
    acc.zz = fe_mul(acc.zz, pp);
    acc.zzz = fe_mul(acc.zzz, ppp);
}

// acc += q (add-2008-s), all exceptional cases handled -- and classified before the main path, as in the G2 mixed addition above:
// an infinite operand is settled first (acc = q is a handful of moves), the products are formed for every lane, and acc == +-q is
// recognised from P = U2 - U1, R = S2 - S1 behind a wave-uniform branch; the doubling works on acc, so nothing of q outlives the
// main path's own use of it.
template <class F> ZK_HD void xyzz_add(Xyzz<F> &acc, const Xyzz<F> &q) {
    int special = 0;   // 1: nothing (left) to add, 2: acc == q, 3: acc == -q
    const bool c_q = q.zz.maybe_zero(), c_a = acc.zz.maybe_zero();
    if (ZK_WAVE_ANY(c_q || c_a)) {
        if (c_q && q.is_inf()) {
            special = 1;
        } else if (c_a && acc.is_inf()) {
            acc = q;
            special = 1;
        }
    }
    // (U1, S1, T1, T2) = (X1 ZZ2, Y1 ZZZ2, ZZ1 ZZ2, ZZZ1 ZZZ2) is acc itself in other coordinates (x = U1 / T1, y = S1 / T2,
    // T1^3 = T2^2), so after these six products neither acc nor q is needed any more -- 108 live registers in G2 instead of 144
    // (the two ZZ pairs), which is what lets the reduction kernels keep two wavefronts per SIMD.
    const F u1 = fe_mul(acc.x, q.zz);        // 8*2
    const F u2 = fe_mul(q.x, acc.zz);
    const F t1 = fe_mul(acc.zz, q.zz);
    const F s1 = fe_mul(acc.y, q.zzz);
    const F s2 = fe_mul(q.y, acc.zzz);
    const F t2 = fe_mul(acc.zzz, q.zzz);
    const F p = fe_sub<2>(u2, u1);           // < 4m
    const F r = fe_sub<2>(s2, s1);           // < 4m
    const bool c_p = !special && p.maybe_zero();
    if (ZK_WAVE_ANY(c_p)) {
        if (c_p && p.is_zero()) special = r.is_zero() ? 2 : 3;
    }
    if (special) {
        if (special == 2)
            acc = xyzz_dbl(Xyzz<F>{u1, s1, t1, t2});
        else if (special == 3)
            acc = Xyzz<F>::inf();
        return;
    }
    const F pp = fe_sqr(p);
    const F ppp = fe_mul(p, pp);
    const F qq = fe_mul(u1, pp);
    acc.zz = fe_mul(t1, pp);
    acc.zzz = fe_mul(t2, ppp);
    const F x3 = fe_sub2<6>(fe_sqr(r), ppp, qq);                     // < 8m
    acc.y = fe_mulsub<2>(r, fe_sub_once<8>(qq, x3), s1, ppp);        // 4*11 + 3*2 < 169;  < 2m
    acc.x = x3;
}

// ------------------------------------------------------------------------------------------------------------------
// acc += q (the same add-2008-s as xyzz_add) by a TEAM of four or of two neighbouring lanes of one wavefront.
//
// Why: the upper levels of the bucket reduction (msm_reduce.h) have fewer additions than the chip has lanes, so what they cost is
// the LATENCY of one addition -- 12 products + 2 squares one after the other in one lane, ~3500 instructions, 6-7 us in G1 and
// ~20 us in G2 -- times the depth of the tree.  The formula's dependency graph is only four products deep:
//     slot A   U1 = X1 ZZ2      U2 = X2 ZZ1      S1 = Y1 ZZZ2     S2 = Y2 ZZZ1                       (roles 0 1 2 3)
//     slot B   P = U2 - U1, PP = P^2    R = S2 - S1, RR = R^2     T1 = ZZ1 ZZ2     T2 = ZZZ1 ZZZ2    (roles 0 1 2 3)
//     slot C   PPP = P PP       --               Q = U1 PP        ZZ3 = T1 PP                        (roles 0 - 2 3)
//     slot D   ZZZ3 = T2 PPP    X3 = RR - PPP - 2Q,  Y3 = R (Q - X3) - S1 PPP                        (roles 0 1)
// so four lanes that each compute ONE product per slot finish in 4.6 product times (~1300 instructions) instead of 14; two
// lanes take seven slots (~1800 instructions: the same lane-instructions per addition as one lane alone, half its latency).
// Operands move between the lanes by DPP quad permutations (v_mov_b32_dpp quad_perm: any lane of the aligned group of four, one
// instruction per 32-bit word, no LDS and no waiting).  Every lane runs the same instruction stream -- a slot is "fetch two
// operands chosen by my role, multiply, keep the result" -- so an idle role costs nothing but its lane.
//
// The exceptional cases are settled between the slots, uniformly for the team (the flags travel like the operands): an infinite
// operand before slot A's product, acc == +-q after P and R are known; the doubling runs in role 0 alone on (U1, S1, T1, T2),
// which is acc in other coordinates (see xyzz_add).  The exchange is a template parameter: QuadDpp on the device;
// tests/hostmath runs the lanes as host threads that meet at a barrier in every fetch, under the contract checker.
template <class F> ZK_HD F *xyzz_field(Xyzz<F> *p, uint32_t k) { return reinterpret_cast<F *>(p) + k; }   // 0: x, 1: y, 2: zz, 3: zzz
template <class F> ZK_HD const F *xyzz_field(const Xyzz<F> *p, uint32_t k) { return reinterpret_cast<const F *>(p) + k; }

// ex.get<P0, P1, P2, P3>(x): the value x of lane P_i of my aligned group of four lanes, for lane i of the group.
template <class F, class Ex> ZK_HD void team4_add(uint32_t role, Xyzz<F> *acc, const Xyzz<F> *q, const Ex &ex) {
    const bool odd = role & 1u, top = role >= 2u, r1 = role == 1u;
    const Xyzz<F> *cacc = acc;
    const F b = *xyzz_field(odd ? cacc : q, 2u + (role >> 1));   // ZZ2, ZZ1, ZZZ2, ZZZ1
    uint32_t fz = (!top && b.maybe_zero() && b.is_zero()) ? 1u : 0u;
    const uint32_t q_inf = ex.template get<0, 0, 0, 0>(fz), a_inf = ex.template get<1, 1, 1, 1>(fz);
    if (q_inf | a_inf) {
        if (!q_inf) *xyzz_field(acc, role) = *xyzz_field(q, role);   // acc is infinity, q is not: acc = q, one coordinate per role
        return;
    }
    const F A = fe_mul(*xyzz_field(odd ? q : cacc, role >> 1), b);   // slot A: U1, U2, S1, S2
    const F d = fe_sub<2>(ex.template get<1, 3, 1, 3>(A), ex.template get<0, 2, 0, 2>(A));   // roles 0, 2: P (< 4m); roles 1, 3: R
    const F bx = ex.template get<1, 1, 1, 3>(b), by = ex.template get<0, 0, 0, 2>(b);
    const F B = fe_mul(fe_select(top, bx, d), fe_select(top, by, d));                  // slot B: PP, RR, T1 = ZZ1 ZZ2, T2 = ZZZ1 ZZZ2
    fz = (!top && d.maybe_zero() && d.is_zero()) ? 1u : 0u;
    const uint32_t p_zero = ex.template get<0, 0, 0, 0>(fz), r_zero = ex.template get<1, 1, 1, 1>(fz);
    const F t1 = ex.template get<2, 2, 2, 2>(B), t2 = ex.template get<3, 3, 3, 3>(B), s1 = ex.template get<2, 2, 2, 2>(A);
    if (p_zero) {
        if (!r_zero)
            *xyzz_field(acc, role) = F::zero();                      // acc == -q
        else if (role == 0)
            *acc = xyzz_dbl(Xyzz<F>{A, s1, t1, t2});                 // acc == q
        return;
    }
    const F pp = ex.template get<0, 0, 0, 0>(B), u1 = ex.template get<0, 0, 0, 0>(A);
    const F C = fe_mul(fe_select(role == 0, d, fe_select(role == 3, t1, u1)), pp);     // slot C: PPP, (U1 PP), Q = U1 PP, ZZ3 = T1 PP
    if (role == 3) acc->zz = C;
    const F ppp = ex.template get<0, 0, 0, 0>(C), qq = ex.template get<2, 2, 2, 2>(C);
    // slot D, role 1: X3 = RR - PPP - 2Q (< 8m), Y3 = R (Q - X3) - S1 PPP (4*11 + 3*2 < 169); the others: T2 PPP - 0 (role 0 stores it)
    const F x3 = fe_sub2<6>(B, ppp, qq);
    const F y3 = fe_mulsub<2>(fe_select(r1, d, t2), fe_select(r1, fe_sub_once<8>(qq, x3), ppp), fe_select(r1, s1, F::zero()), ppp);
    if (r1) {
        acc->x = x3;
        acc->y = y3;
    }
    if (role == 0) acc->zzz = y3;
}
// The same by two lanes (role = lane & 1; the exchange's groups of four hold two teams, so only pair-symmetric patterns).
template <class F, class Ex> ZK_HD void team2_add(uint32_t role, Xyzz<F> *acc, const Xyzz<F> *q, const Ex &ex) {
    const bool odd = role & 1u;
    const Xyzz<F> *cacc = acc;
    const F b = *xyzz_field(odd ? cacc : q, 2u);                      // ZZ2 | ZZ1
    uint32_t fz = (b.maybe_zero() && b.is_zero()) ? 1u : 0u;
    const uint32_t q_inf = ex.template get<0, 0, 2, 2>(fz), a_inf = ex.template get<1, 1, 3, 3>(fz);
    if (q_inf | a_inf) {
        if (!q_inf) {
            *xyzz_field(acc, 2u * role) = *xyzz_field(q, 2u * role);
            *xyzz_field(acc, 2u * role + 1u) = *xyzz_field(q, 2u * role + 1u);
        }
        return;
    }
    const F A = fe_mul(*xyzz_field(odd ? q : cacc, 0u), b);           // U1 = X1 ZZ2 | U2 = X2 ZZ1
    const F b2 = *xyzz_field(odd ? cacc : q, 3u);                     // ZZZ2 | ZZZ1
    const F S = fe_mul(*xyzz_field(odd ? q : cacc, 1u), b2);          // S1 = Y1 ZZZ2 | S2 = Y2 ZZZ1
    const F sA = ex.template get<1, 0, 3, 2>(A), sS = ex.template get<1, 0, 3, 2>(S);
    const F d = fe_sub<2>(fe_select(odd, S, sA), fe_select(odd, sS, A));                // P = U2 - U1 | R = S2 - S1
    const F B = fe_sqr(d);                                            // PP | RR
    fz = (d.maybe_zero() && d.is_zero()) ? 1u : 0u;
    const uint32_t p_zero = ex.template get<0, 0, 2, 2>(fz), r_zero = ex.template get<1, 1, 3, 3>(fz);
    const F sb = ex.template get<1, 0, 3, 2>(b), sb2 = ex.template get<1, 0, 3, 2>(b2);
    const F T = fe_mul(fe_select(odd, b2, b), fe_select(odd, sb2, sb));                 // T1 = ZZ1 ZZ2 | T2 = ZZZ1 ZZZ2
    const F sT = ex.template get<1, 0, 3, 2>(T);
    if (p_zero) {
        if (!r_zero) {
            *xyzz_field(acc, 2u * role) = F::zero();
            *xyzz_field(acc, 2u * role + 1u) = F::zero();
        } else if (!odd) {
            *acc = xyzz_dbl(Xyzz<F>{A, S, T, sT});
        }
        return;
    }
    const F sB = ex.template get<1, 0, 3, 2>(B);
    const F E = fe_mul(fe_select(odd, sA, d), fe_select(odd, sB, B));                   // PPP = P PP | Q = U1 PP
    const F Z = fe_mul(T, B);                                         // ZZ3 = T1 PP | (T2 RR, unused)
    if (!odd) acc->zz = Z;
    const F sE = ex.template get<1, 0, 3, 2>(E);
    const F x3 = fe_sub2<6>(B, fe_select(odd, sE, E), fe_select(odd, E, sE));           // | X3 = RR - PPP - 2Q   (role 0: PP - PPP - 2Q, unused)
    const F y3 = fe_mulsub<2>(fe_select(odd, d, sT), fe_select(odd, fe_sub_once<8>(E, x3), E), fe_select(odd, sS, F::zero()), fe_select(odd, sE, E));   // ZZZ3 = T2 PPP | Y3
    if (odd) {
        acc->x = x3;
        acc->y = y3;
    } else {
        acc->zzz = y3;
    }
}
#if defined(__HIPCC__)
// The exchange on the device.  Written as inline assembly with its own wait states, not as __builtin_amdgcn_mov_dpp: with the
// builtin (ROCm 7.2, gfx950) the four-lane addition came out WRONG on the chip while the same code over ds_bpermute, and the
// builtin with its operands pinned by empty asm statements, were right (tools/reduce_probe.hip `check`, which keeps all the
// variants) -- the compiler leaves two wait states between the instruction that writes a register and the DPP move that reads
// it across lanes, and that is not enough here (a wave64 instruction occupies a SIMD-32 for two cycles, not four).  One
// statement moves a whole element: five wait states, nine moves into registers that are none of the sources (early clobber:
// also no move is folded into a neighbouring instruction), two more before the results are used.
#define ZK_DPP_MOV(o, i) "v_mov_b32_dpp %" #o ", %" #i " quad_perm:[%18,%19,%20,%21] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
template <int P0, int P1, int P2, int P3> __device__ __forceinline__ uint32_t dpp_quad(uint32_t v) {
    uint32_t r;
    asm volatile("s_nop 4\n\tv_mov_b32_dpp %0, %1 quad_perm:[%2,%3,%4,%5] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 1"
                 : "=&v"(r)
                 : "v"(v), "n"(P0), "n"(P1), "n"(P2), "n"(P3));
    return r;
}
template <int P0, int P1, int P2, int P3, class Tag> __device__ __forceinline__ Fe<Tag> dpp_quad(const Fe<Tag> &x) {
    static_assert(NL == 9, "nine moves per element");
    Fe<Tag> r;
    asm volatile("s_nop 4\n\t" ZK_DPP_MOV(0, 9) ZK_DPP_MOV(1, 10) ZK_DPP_MOV(2, 11) ZK_DPP_MOV(3, 12) ZK_DPP_MOV(4, 13) ZK_DPP_MOV(5, 14)
                     ZK_DPP_MOV(6, 15) ZK_DPP_MOV(7, 16) ZK_DPP_MOV(8, 17) "s_nop 1"
                 : "=&v"(r.l[0]), "=&v"(r.l[1]), "=&v"(r.l[2]), "=&v"(r.l[3]), "=&v"(r.l[4]), "=&v"(r.l[5]), "=&v"(r.l[6]), "=&v"(r.l[7]), "=&v"(r.l[8])
                 : "v"(x.l[0]), "v"(x.l[1]), "v"(x.l[2]), "v"(x.l[3]), "v"(x.l[4]), "v"(x.l[5]), "v"(x.l[6]), "v"(x.l[7]), "v"(x.l[8]), "n"(P0), "n"(P1),
                   "n"(P2), "n"(P3));
    return r;
}
template <int P0, int P1, int P2, int P3> __device__ __forceinline__ Fp2 dpp_quad(const Fp2 &x) {
    return Fp2{dpp_quad<P0, P1, P2, P3>(x.c0), dpp_quad<P0, P1, P2, P3>(x.c1)};
}
struct QuadDpp {
    template <int P0, int P1, int P2, int P3, class V> __device__ __forceinline__ V get(const V &x) const { return dpp_quad<P0, P1, P2, P3>(x); }
};
#endif

// Affine x = X/ZZ, y = Y/ZZZ with one inversion: since ZZ^3 = ZZZ^2, 1/ZZ = (ZZ/ZZZ)^2.
template <class F> ZK_HD Affine<F> xyzz_to_affine(const Xyzz<F> &p) {
    if (p.is_inf()) return Affine<F>::inf();
    F izzz = fe_inv(p.zzz);
    F izz = fe_sqr(fe_mul(p.zz, izzz));
    return Affine<F>{fe_mul(p.x, izz), fe_mul(p.y, izzz)};
}

// k*P by MSB-first double-and-add over a 256-bit little-endian limb scalar.
template <class F> ZK_HD Xyzz<F> xyzz_scalar_mul(const Affine<F> &p, const uint32_t k[8]) {
    Xyzz<F> acc = Xyzz<F>::inf();
    for (int i = 7; i >= 0; i--) {
        for (int b = 31; b >= 0; b--) {
            acc = xyzz_dbl(acc);
            if ((k[i] >> b) & 1) xyzz_add_affine(acc, p);
        }
    }
    return acc;
}

// k*P for a small non-negative k (bucket-segment bases in the reduction).
template <class F> ZK_HD Xyzz<F> xyzz_small_mul(const Xyzz<F> &p, uint32_t k) {
    Xyzz<F> acc = Xyzz<F>::inf();
    for (int b = 31; b >= 0; b--) {
        acc = xyzz_dbl(acc);
        if ((k >> b) & 1) xyzz_add(acc, p);
    }
    return acc;
}

typedef Affine<Fp> G1Affine;
typedef Xyzz<Fp> G1Xyzz;
typedef Affine<Fp2> G2Affine;
typedef Xyzz<Fp2> G2Xyzz;

}  // namespace zk
