// tests/hostmath/hostmath.cpp -- TEST-ONLY host build of the device math headers (field.h, curve.h).
// Lets the CPU test-suite check the exact code the HIP kernels run (limb arithmetic, XYZZ formulas,
// exceptional cases) against the oracle without a GPU.  Not part of the product library.
#include <string.h>
#include "curve.h"
using namespace zk;

template <class T> static Fe<T> load4(const uint64_t *p) {
    Fe<T> r;
    for (int i = 0; i < 4; i++) { r.l[2 * i] = (uint32_t)p[i]; r.l[2 * i + 1] = (uint32_t)(p[i] >> 32); }
    return r;
}
template <class T> static void store4(uint64_t *p, const Fe<T> &a) {
    for (int i = 0; i < 4; i++) p[i] = (uint64_t)a.l[2 * i] | ((uint64_t)a.l[2 * i + 1] << 32);
}
template <class T> static void field_op(int op, const uint64_t *a, const uint64_t *b, uint64_t *o) {
    Fe<T> x = fe_to_mont(load4<T>(a)), y = fe_to_mont(load4<T>(b)), r;
    switch (op) {
        case 0: r = fe_add(x, y); break;
        case 1: r = fe_sub(x, y); break;
        case 2: r = fe_mul(x, y); break;
        case 3: r = fe_inv(x); break;
        case 4: r = fe_neg(x); break;
        default: r = fe_sqr(x); break;
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
static void k32(const uint64_t *k, uint32_t out[8]) {
    for (int i = 0; i < 4; i++) { out[2 * i] = (uint32_t)k[i]; out[2 * i + 1] = (uint32_t)(k[i] >> 32); }
}
extern "C" {
void hm_field_op(int which, int op, const uint64_t *a, const uint64_t *b, uint64_t *o) {
    if (which) field_op<FrTag>(op, a, b, o); else field_op<FpTag>(op, a, b, o);
}
void hm_g1_mul(const uint64_t *p, const uint64_t *k, uint64_t *o) {
    uint32_t kk[8]; k32(k, kk);
    g1_store(o, xyzz_scalar_mul(g1_load(p), kk));
}
void hm_g2_mul(const uint64_t *p, const uint64_t *k, uint64_t *o) {
    uint32_t kk[8]; k32(k, kk);
    g2_store(o, xyzz_scalar_mul(g2_load(p), kk));
}
// mode 0: mixed add (xyzz(p) + affine q); 1: full add after scaling both by scalar muls (k1*p + k2*q)
void hm_g1_add(int mode, const uint64_t *p, const uint64_t *q, const uint64_t *k1, const uint64_t *k2, uint64_t *o) {
    if (mode == 0) {
        G1Xyzz a = G1Xyzz::from_affine(g1_load(p));
        xyzz_add_affine(a, g1_load(q));
        g1_store(o, a);
    } else {
        uint32_t a8[8], b8[8]; k32(k1, a8); k32(k2, b8);
        G1Xyzz a = xyzz_scalar_mul(g1_load(p), a8), b = xyzz_scalar_mul(g1_load(q), b8);
        xyzz_add(a, b);
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
        xyzz_add(a, b);
        g2_store(o, a);
    }
}
void hm_g1_small_mul(const uint64_t *p, uint32_t k, uint64_t *o) {
    g1_store(o, xyzz_small_mul(G1Xyzz::from_affine(g1_load(p)), k));
}
}
