// tests/hostmath/hostmath.cpp -- TEST-ONLY host build of the device math headers (field.h, curve.h).
// Lets the CPU test-suite check the exact code the HIP kernels run (9x29-bit lazy limb arithmetic,
// Montgomery conversions, XYZZ formulas, exceptional cases) against the oracle without a GPU, and
// the host epilogue types of host_field.h (device <-> host conversions).  Not part of the product.
#include <string.h>
#include <pthread.h>
#include <thread>
#include "curve.h"
#include "host_field.h"
using namespace zk;

static void words_of(const uint64_t *p, uint32_t w[8]) {
    for (int i = 0; i < 4; i++) { w[2 * i] = (uint32_t)p[i]; w[2 * i + 1] = (uint32_t)(p[i] >> 32); }
}
static void words_to(uint64_t *p, const uint32_t w[8]) {
    for (int i = 0; i < 4; i++) p[i] = (uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32);
}
template <class T> static Fe<T> load4(const uint64_t *p) {
    uint32_t w[8];
    words_of(p, w);
    return fe_from_words<T>(w);
}
template <class T> static void store4(uint64_t *p, const Fe<T> &a) {  // a: any lazy representative
    uint32_t w[8];
    fe_to_words(fe_reduce_full(a), w);
    words_to(p, w);
}
template <class T> static void field_op(int op, const uint64_t *a, const uint64_t *b, uint64_t *o) {
    Fe<T> x = fe_to_mont(load4<T>(a)), y = fe_to_mont(load4<T>(b)), r;
    switch (op) {
        case 0: r = fe_add(x, y); break;
        case 1: r = fe_sub_k<2>(x, y); break;
        case 2: r = fe_mul(x, y); break;
        case 3: r = fe_inv(x); break;
        case 4: r = fe_neg_k<2>(x); break;
        case 5: r = fe_sqr(x); break;
        case 6: r = fe_mul(fe_add_lazy(x, y), fe_add_lazy(y, y)); break;          // lazy operands: (x+y)*(2y)
        case 7: {                                                               // worst-case bounds: (x+8m-y)^2 chain
            Fe<T> t = fe_sub_k<8>(x, y);                                        // < 10m
            r = fe_mul(fe_sqr(t), t);
            break;
        }
        case 8: r = fe_dbl(fe_triple(x)); break;                                // 6x
        default: r = x.equals(y) ? Fe<T>::one() : Fe<T>::zero(); break;
    }
    store4(o, fe_from_mont(r));
}
static G1Affine g1_load(const uint64_t *p) { return G1Affine{fe_to_mont(load4<FpTag>(p)), fe_to_mont(load4<FpTag>(p + 4))}; }
static void g1_store(uint64_t *o, const G1Xyzz &p) {
    G1Affine a = xyzz_to_affine(p);
    store4(o, fe_from_mont(a.x)); store4(o + 4, fe_from_mont(a.y));
}
static G2Affine g2_load(const uint64_t *p) {
    return G2Affine{Fp2{fe_to_mont(load4<FpTag>(p)), fe_to_mont(load4<FpTag>(p + 4))},
                    Fp2{fe_to_mont(load4<FpTag>(p + 8)), fe_to_mont(load4<FpTag>(p + 12))}};
}
static void g2_store(uint64_t *o, const G2Xyzz &p) {
    G2Affine a = xyzz_to_affine(p);
    store4(o, fe_from_mont(a.x.c0)); store4(o + 4, fe_from_mont(a.x.c1));
    store4(o + 8, fe_from_mont(a.y.c0)); store4(o + 12, fe_from_mont(a.y.c1));
}
static void k32(const uint64_t *k, uint32_t out[8]) { words_of(k, out); }

// The bucket reduction's team additions (curve.h team4_add / team2_add) with the lanes of one aligned group of four as host
// threads: every fetch publishes the lane's value, meets the other lanes at a barrier and reads the lane the pattern names --
// what v_mov_b32_dpp quad_perm does for the lanes of a wavefront, which run in lockstep.
struct HostQuadShared {
    pthread_barrier_t bar;
    alignas(16) unsigned char slot[4][sizeof(Fp2)];
};
struct HostQuad {
    HostQuadShared *sh;
    int lane;
    template <int P0, int P1, int P2, int P3, class V> V get(const V &x) const {
        static_assert(sizeof(V) <= sizeof(Fp2), "slot too small");
        const int perm[4] = {P0, P1, P2, P3};
        memcpy(sh->slot[lane], &x, sizeof(V));
        pthread_barrier_wait(&sh->bar);
        V r;
        memcpy(&r, sh->slot[perm[lane]], sizeof(V));
        pthread_barrier_wait(&sh->bar);
        return r;
    }
};
// lanes: 4 (one team of four on acc[0] += q[0]) or 2 (two teams of two side by side: acc[0] += q[0] and acc[1] += q[1])
template <class F> static void team_add_threads(int lanes, Xyzz<F> *acc, const Xyzz<F> *q) {
    HostQuadShared sh;
    pthread_barrier_init(&sh.bar, nullptr, 4);
    std::thread th[4];
    for (int lane = 0; lane < 4; lane++)
        th[lane] = std::thread([&, lane] {
            HostQuad ex{&sh, lane};
            if (lanes == 4) team4_add((uint32_t)lane, &acc[0], &q[0], ex);
            else team2_add((uint32_t)(lane & 1), &acc[lane >> 1], &q[lane >> 1], ex);
        });
    for (auto &t : th) t.join();
    pthread_barrier_destroy(&sh.bar);
}

// fe_add_r2 / fe_sub_r2 against the two-chain forms they replace, on RAW normalised 9-limb operands (values < 2m): both results
// as raw limbs, so the test can demand the very same representative.  which: 0 = F_p, 1 = F_r; op: 0 = add, 1 = sub.
template <class T> static void r2_pair(int op, const uint32_t *a9, const uint32_t *b9, uint32_t *fast9, uint32_t *ref9) {
    Fe<T> a, b;
    for (int i = 0; i < NL; i++) { a.l[i] = a9[i]; b.l[i] = b9[i]; }
    ZK_DBG(a.vb = 2; b.vb = 2; a.lmax = 1; b.lmax = 1;)
    Fe<T> f, r;
    if (op == 0) {
        f = fe_add_r2(a, b);
        r = fe_add(a, b);
        fe_wreduce<4>(r);
    } else {
        f = fe_sub_r2(a, b);
        r = fe_sub_k<2>(a, b);
        fe_wreduce<4>(r);
    }
    for (int i = 0; i < NL; i++) { fast9[i] = f.l[i]; ref9[i] = r.l[i]; }
}

extern "C" {
void hm_r2_pair(int which, int op, const uint32_t *a9, const uint32_t *b9, uint32_t *fast9, uint32_t *ref9) {
    if (which) r2_pair<FrTag>(op, a9, b9, fast9, ref9); else r2_pair<FpTag>(op, a9, b9, fast9, ref9);
}
void hm_field_op(int which, int op, const uint64_t *a, const uint64_t *b, uint64_t *o) {
    if (which) field_op<FrTag>(op, a, b, o); else field_op<FpTag>(op, a, b, o);
}
// device element -> host epilogue element -> canonical, and back: exercises HFe::from_dev / to_dev
void hm_host_roundtrip(int which, const uint64_t *a, uint64_t *via_host, uint64_t *via_dev) {
    if (which) {
        Fr x = fe_to_mont(load4<FrTag>(a));
        HFr h = HFr::from_dev(fe_add(x, Fr::zero()));
        HFr c = fe_from_mont(h);
        memcpy(via_host, c.l, 32);
        store4(via_dev, fe_from_mont(h.to_dev()));
    } else {
        Fp x = fe_to_mont(load4<FpTag>(a));
        HFp h = HFp::from_dev(fe_sub_k<8>(x, Fp::zero()));  // a lazy representative (value + 8p)
        HFp c = fe_from_mont(h);
        memcpy(via_host, c.l, 32);
        store4(via_dev, fe_from_mont(h.to_dev()));
    }
}
void hm_g1_mul(const uint64_t *p, const uint64_t *k, uint64_t *o) {
    uint32_t kk[8]; k32(k, kk);
    g1_store(o, xyzz_scalar_mul(g1_load(p), kk));
}
void hm_g2_mul(const uint64_t *p, const uint64_t *k, uint64_t *o) {
    uint32_t kk[8]; k32(k, kk);
    g2_store(o, xyzz_scalar_mul(g2_load(p), kk));
}
// mode 0: mixed add (xyzz(p) + affine q); 1: full add after scaling both by scalar muls (k1*p + k2*q);
// 2: the same full add by the four-lane team addition of the bucket reduction (curve.h team4_add), lanes as host threads;
// 3: mode 2 on accumulators that come out of the mixed addition (the lazy bounds the reduction's first level really sees);
// 4, 5: modes 2, 3 by two teams of two lanes side by side (team2_add; the second team adds q + acc)
void hm_g1_add(int mode, const uint64_t *p, const uint64_t *q, const uint64_t *k1, const uint64_t *k2, uint64_t *o) {
    if (mode == 0) {
        G1Xyzz a = G1Xyzz::from_affine(g1_load(p));
        xyzz_add_affine(a, g1_load(q));
        g1_store(o, a);
    } else {
        uint32_t a8[8], b8[8]; k32(k1, a8); k32(k2, b8);
        G1Xyzz a = xyzz_scalar_mul(g1_load(p), a8), b = xyzz_scalar_mul(g1_load(q), b8);
        if (mode == 3 || mode == 5) {   // (k1*p + q) and (k2*q + p), each ending in a mixed addition
            xyzz_add_affine(a, g1_load(q));
            xyzz_add_affine(b, g1_load(p));
        }
        if (mode >= 4) {
            G1Xyzz acc2[2] = {a, b}, q2[2] = {b, a};
            team_add_threads(2, acc2, q2);
            a = acc2[0];
            G1Affine x = xyzz_to_affine(a), y = xyzz_to_affine(acc2[1]);     // b + a must be the same point
            const bool same = x.is_inf() ? y.is_inf() : (!y.is_inf() && x.x.equals(y.x) && x.y.equals(y.y));
            if (!same) {   // no valid encoding: the caller's comparison fails
                memset(o, 0xff, 8 * sizeof(uint64_t));
                return;
            }
        } else if (mode >= 2) {
            team_add_threads(4, &a, &b);
        } else {
            xyzz_add(a, b);
        }
        g1_store(o, a);
    }
}
void hm_g2_add(int mode, const uint64_t *p, const uint64_t *q, const uint64_t *k1, const uint64_t *k2, uint64_t *o) {
    if (mode == 0) {
        G2Xyzz a = G2Xyzz::from_affine(g2_load(p));
        xyzz_add_affine(a, g2_load(q));
        g2_store(o, a);
    } else {
        uint32_t a8[8], b8[8]; k32(k1, a8); k32(k2, b8);
        G2Xyzz a = xyzz_scalar_mul(g2_load(p), a8), b = xyzz_scalar_mul(g2_load(q), b8);
        if (mode == 3 || mode == 5) {   // (k1*p + q) and (k2*q + p), each ending in a mixed addition
            xyzz_add_affine(a, g2_load(q));
            xyzz_add_affine(b, g2_load(p));
        }
        if (mode >= 4) {
            G2Xyzz acc2[2] = {a, b}, q2[2] = {b, a};
            team_add_threads(2, acc2, q2);
            a = acc2[0];
            G2Affine x = xyzz_to_affine(a), y = xyzz_to_affine(acc2[1]);     // b + a must be the same point
            const bool same = x.is_inf() ? y.is_inf() : (!y.is_inf() && x.x.equals(y.x) && x.y.equals(y.y));
            if (!same) {   // no valid encoding: the caller's comparison fails
                memset(o, 0xff, 16 * sizeof(uint64_t));
                return;
            }
        } else if (mode >= 2) {
            team_add_threads(4, &a, &b);
        } else {
            xyzz_add(a, b);
        }
        g2_store(o, a);
    }
}
void hm_g1_small_mul(const uint64_t *p, uint32_t k, uint64_t *o) {
    g1_store(o, xyzz_small_mul(G1Xyzz::from_affine(g1_load(p)), k));
}
// sum of n affine points through the mixed-add accumulator, then the host epilogue conversion
// (device XYZZ -> host XYZZ -> affine), as the MSM tail does.
void hm_g1_accumulate(const uint64_t *pts, uint32_t n, const uint8_t *negate, uint64_t *o) {
    G1Xyzz acc = G1Xyzz::inf();
    for (uint32_t i = 0; i < n; i++) {
        G1Affine q = g1_load(pts + 8 * i);
        if (negate && negate[i]) q.y = fe_neg_once<2>(q.y);   // as the accumulate kernel negates: a single-use, unnormalised value
        xyzz_add_affine(acc, q);
    }
    Xyzz<HFp> h{HFp::from_dev(acc.x), HFp::from_dev(acc.y), HFp::from_dev(acc.zz), HFp::from_dev(acc.zzz)};
    Affine<HFp> a = xyzz_to_affine(h);
    HFp x = fe_from_mont(a.x), y = fe_from_mont(a.y);
    memcpy(o, x.l, 32); memcpy(o + 4, y.l, 32);
}
// the same for G2 (its mixed addition has bounds of its own: curve.h), result through the device-side conversion
void hm_g2_accumulate(const uint64_t *pts, uint32_t n, const uint8_t *negate, uint64_t *o) {
    G2Xyzz acc = G2Xyzz::inf();
    for (uint32_t i = 0; i < n; i++) {
        G2Affine q = g2_load(pts + 16 * i);
        if (negate && negate[i]) q.y = fe_neg_once<2>(q.y);
        xyzz_add_affine(acc, q);
    }
    // what the bucket reduction does with such an accumulator: a full addition and a doubling take it as an operand
    G2Xyzz twice = acc;
    xyzz_add(twice, acc);
    G2Xyzz dbl = xyzz_dbl(acc);
    G2Affine a = xyzz_to_affine(acc), b = xyzz_to_affine(twice), c = xyzz_to_affine(dbl);
    store4(o, fe_from_mont(a.x.c0)); store4(o + 4, fe_from_mont(a.x.c1));
    store4(o + 8, fe_from_mont(a.y.c0)); store4(o + 12, fe_from_mont(a.y.c1));
    store4(o + 16, fe_from_mont(b.x.c0)); store4(o + 20, fe_from_mont(b.x.c1));
    store4(o + 24, fe_from_mont(b.y.c0)); store4(o + 28, fe_from_mont(b.y.c1));
    store4(o + 32, fe_from_mont(c.x.c0)); store4(o + 36, fe_from_mont(c.x.c1));
    store4(o + 40, fe_from_mont(c.y.c0)); store4(o + 44, fe_from_mont(c.y.c1));
}
}
