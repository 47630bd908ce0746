// tests/hostmath/sanitize_main.cpp -- TEST-ONLY: one executable built with -fsanitize=address,undefined
// (-fno-sanitize-recover: any report aborts) that links
//   * hostmath.cpp            the device field / curve headers compiled for the host,
//   * csrc/pairing.hip        the product's HOST code (pairing, input validation, host_field.h epilogue types),
//   * oracle/bn254_oracle.c   the C oracle (so the checker itself is sanitizer-clean),
// runs a fixed pseudo-random workload through all three and cross-checks them.  GPU AddressSanitizer is not
// available on the pool; this is the CPU build SURVEY.md section 5 asks for.  Prints "sanitize ok" and exits 0.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

extern "C" {
// hostmath.cpp
void hm_field_op(int which, int op, const uint64_t *a, const uint64_t *b, uint64_t *o);
void hm_host_roundtrip(int which, const uint64_t *a, uint64_t *via_host, uint64_t *via_dev);
void hm_g1_mul(const uint64_t *p, const uint64_t *k, uint64_t *o);
void hm_g2_mul(const uint64_t *p, const uint64_t *k, uint64_t *o);
void hm_g1_add(int mode, const uint64_t *p, const uint64_t *q, const uint64_t *k1, const uint64_t *k2, uint64_t *o);
void hm_g2_add(int mode, const uint64_t *p, const uint64_t *q, const uint64_t *k1, const uint64_t *k2, uint64_t *o);
void hm_g1_accumulate(const uint64_t *pts, uint32_t n, const uint8_t *negate, uint64_t *o);
// oracle/bn254_oracle.c
void orc_field_op(int which, int op, const uint64_t a[4], const uint64_t b[4], uint64_t out[4]);
void orc_g1_mul(const uint64_t p[8], const uint64_t k[4], uint64_t out[8]);
void orc_g1_add(const uint64_t p[8], const uint64_t q[8], uint64_t out[8]);
void orc_g2_mul(const uint64_t p[16], const uint64_t k[4], uint64_t out[16]);
void orc_g2_add(const uint64_t p[16], const uint64_t q[16], uint64_t out[16]);
void orc_g1_msm(const uint64_t *scalars, const uint64_t *points, size_t n, uint64_t out[8]);
int orc_g1_msm_bucket(const uint64_t *scalars, const uint64_t *points, size_t n, unsigned c, uint64_t out[8]);
int orc_g1_msm_bucket_mt(const uint64_t *scalars, const uint64_t *points, size_t n, unsigned c, unsigned threads, uint64_t out[8]);
void orc_g1_fixed_base(const uint64_t p[8], const uint64_t *scalars, size_t n, uint64_t *out);
int orc_ntt(uint64_t *data, unsigned log_n, const uint64_t omega[4], int inverse);
void orc_fr_horner(const uint64_t *coeffs, size_t n, const uint64_t x[4], uint64_t out[4]);
}
namespace zk {  // csrc/pairing.hip, and what it expects from api.hip
void set_last_error(const std::string &) {}
int pairing_product(const uint64_t *g1_points, const uint64_t *g2_points, size_t n, uint64_t *out, int *is_one);
}

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() {
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return rng_state;
}
static void rnd_scalar(uint64_t k[4]) {  // < 2^253 < r < p
    for (int i = 0; i < 4; i++) k[i] = rnd();
    k[3] &= (1ull << 61) - 1;
}
#define CHECK(cond)                                                    \
    do {                                                               \
        if (!(cond)) {                                                 \
            fprintf(stderr, "sanitize_main: check failed: %s (line %d)\n", #cond, __LINE__); \
            return 1;                                                  \
        }                                                              \
    } while (0)

static const uint64_t G1[8] = {1, 0, 0, 0, 2, 0, 0, 0};
static const uint64_t G2[16] = {0x46debd5cd992f6edull, 0x674322d4f75edaddull, 0x426a00665e5c4479ull, 0x1800deef121f1e76ull,
                                0x97e485b7aef312c2ull, 0xf1aa493335a9e712ull, 0x7260bfb731fb5d25ull, 0x198e9393920d483aull,
                                0x4ce6cc0166fa7daaull, 0xe3d1e7690c43d37bull, 0x4aab71808dcb408full, 0x12c85ea5db8c6debull,
                                0x55acdadcd122975bull, 0xbc4b313370b38ef3ull, 0xec9e99ad690c3395ull, 0x090689d0585ff075ull};
// omega_16 = 5^((r-1)/16) mod r is derived below from the field ops instead of being pasted in

int main() {
    // 1. field ops: device arithmetic on the host vs the oracle, both fields, including the lazy-bound cases
    for (int which = 0; which < 2; which++)
        for (int op = 0; op <= 5; op++)
            for (int it = 0; it < 40; it++) {
                uint64_t a[4], b[4], x[4], y[4];
                rnd_scalar(a);
                rnd_scalar(b);
                if (it == 0) memset(a, 0, 32);
                if (it == 1) memset(b, 0, 32);
                hm_field_op(which, op, a, b, x);
                const uint64_t zero[4] = {0, 0, 0, 0};
                if (op <= 3) orc_field_op(which, op, a, b, y);
                else if (op == 4) orc_field_op(which, 1, zero, a, y);  // -a
                else orc_field_op(which, 2, a, a, y);                  // a^2
                CHECK(!memcmp(x, y, 32));
                hm_host_roundtrip(which, a, x, y);
                CHECK(!memcmp(x, a, 32) && !memcmp(y, a, 32));
            }
    // 2. curve ops: scalar multiplications, mixed / full additions incl. P+P, P+(-P), infinity
    uint64_t k1[4], k2[4], p1[8], p2[8], s1[8], s2[8], inf8[8] = {0};
    for (int it = 0; it < 6; it++) {
        rnd_scalar(k1);
        rnd_scalar(k2);
        hm_g1_mul(G1, k1, p1);
        orc_g1_mul(G1, k1, p2);
        CHECK(!memcmp(p1, p2, 64));
        hm_g1_add(0, p1, G1, k1, k2, s1);
        orc_g1_add(p1, G1, s2);
        CHECK(!memcmp(s1, s2, 64));
        hm_g1_add(0, p1, p1, k1, k2, s1);  // doubling through the addition formula
        orc_g1_add(p1, p1, s2);
        CHECK(!memcmp(s1, s2, 64));
        hm_g1_add(0, p1, inf8, k1, k2, s1);
        CHECK(!memcmp(s1, p1, 64));
        hm_g1_add(1, G1, p1, k1, k2, s1);  // k1*G + k2*(k1*G)
        uint64_t t1[8], t2[8];
        orc_g1_mul(G1, k1, t1);
        orc_g1_mul(p1, k2, t2);
        orc_g1_add(t1, t2, s2);
        CHECK(!memcmp(s1, s2, 64));
    }
    uint64_t q1[16], q2[16];
    rnd_scalar(k1);
    hm_g2_mul(G2, k1, q1);
    orc_g2_mul(G2, k1, q2);
    CHECK(!memcmp(q1, q2, 128));
    hm_g2_add(0, q1, G2, k1, k2, q2);
    uint64_t q3[16];
    orc_g2_add(q1, G2, q3);
    CHECK(!memcmp(q2, q3, 128));
    // 3. the oracle's MSM variants against each other and against the accumulate path of the device math
    const size_t n = 700;
    std::vector<uint64_t> sc(4 * n), ks(4 * n), pts(8 * n);
    for (size_t i = 0; i < n; i++) {
        rnd_scalar(&sc[4 * i]);
        rnd_scalar(&ks[4 * i]);
    }
    memset(&sc[0], 0, 32);  // a zero scalar
    orc_g1_fixed_base(G1, ks.data(), n, pts.data());
    memset(&pts[8 * 5], 0, 64);  // an infinity base
    uint64_t m0[8], m1[8], m2[8];
    orc_g1_msm(sc.data(), pts.data(), n, m0);
    CHECK(orc_g1_msm_bucket(sc.data(), pts.data(), n, 9, m1) == 0 && !memcmp(m0, m1, 64));
    CHECK(orc_g1_msm_bucket_mt(sc.data(), pts.data(), n, 8, 3, m2) == 0 && !memcmp(m0, m2, 64));
    std::vector<uint8_t> neg(64);
    for (auto &v : neg) v = (uint8_t)(rnd() & 1);
    uint64_t acc_dev[8], acc_orc[8] = {0};
    hm_g1_accumulate(pts.data() + 8 * 16, 64, neg.data(), acc_dev);
    for (int i = 0; i < 64; i++) {
        uint64_t t[8], u[8];
        memcpy(t, &pts[8 * (16 + i)], 64);
        if (neg[i]) {  // -P = (r - 1) * P
            static const uint64_t RM1[4] = {0x43e1f593f0000000ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
            orc_g1_mul(t, RM1, u);
            memcpy(t, u, 64);
        }
        orc_g1_add(acc_orc, t, u);
        memcpy(acc_orc, u, 64);
    }
    CHECK(!memcmp(acc_dev, acc_orc, 64));
    // 4. NTT round trip of the oracle (omega_16 = 5^((r-1)/16), computed with the oracle's own field ops)
    {
        // (r - 1) / 16
        static const uint64_t E[4] = {0x143e1f593f000000ull, 0xd2833e84879b9709ull, 0x9b85045b68181585ull, 0x030644e72e131a02ull};
        uint64_t w[4] = {1, 0, 0, 0}, base[4] = {5, 0, 0, 0}, t[4];
        for (int bit = 0; bit < 256; bit++) {
            if ((E[bit >> 6] >> (bit & 63)) & 1) {
                orc_field_op(1, 2, w, base, t);
                memcpy(w, t, 32);
            }
            orc_field_op(1, 2, base, base, t);
            memcpy(base, t, 32);
        }
        uint64_t w16[4];
        memcpy(w16, w, 32);
        for (int i = 0; i < 4; i++) {  // w^16 == 1 and w^8 != 1
            orc_field_op(1, 2, w, w, t);
            memcpy(w, t, 32);
            if (i == 2) CHECK(!(w[0] == 1 && !w[1] && !w[2] && !w[3]));
        }
        CHECK(w[0] == 1 && !w[1] && !w[2] && !w[3]);
        std::vector<uint64_t> d(4 * 16), orig;
        for (int i = 0; i < 16; i++) rnd_scalar(&d[4 * i]);
        orig = d;
        CHECK(orc_ntt(d.data(), 4, w16, 0) == 0);
        uint64_t one[4] = {1, 0, 0, 0}, h[4];
        orc_fr_horner(orig.data(), 16, one, h);  // X[0] = p(1)
        CHECK(!memcmp(h, d.data(), 32));
        orc_fr_horner(orig.data(), 16, w16, h);  // X[1] = p(omega)
        CHECK(!memcmp(h, d.data() + 4, 32));
        CHECK(orc_ntt(d.data(), 4, w16, 1) == 0);
        CHECK(d == orig);
    }
    // 5. product host code: pairing bilinearity e(aP, Q) e(-P, aQ) == 1, inputs validated
    {
        uint64_t a[4], aP[8], aQ[16], negP[8];
        rnd_scalar(a);
        orc_g1_mul(G1, a, aP);
        orc_g2_mul(G2, a, aQ);
        static const uint64_t RM1[4] = {0x43e1f593f0000000ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
        orc_g1_mul(G1, RM1, negP);
        uint64_t g1s[16], g2s[32];
        memcpy(g1s, aP, 64);
        memcpy(g1s + 8, negP, 64);
        memcpy(g2s, G2, 128);
        memcpy(g2s + 16, aQ, 128);
        int is_one = -1;
        CHECK(zk::pairing_product(g1s, g2s, 2, nullptr, &is_one) == 0 && is_one == 1);
        memcpy(g1s + 8, G1, 64);
        CHECK(zk::pairing_product(g1s, g2s, 2, nullptr, &is_one) == 0 && is_one == 0);
        g1s[4] ^= 1;  // off the curve
        CHECK(zk::pairing_product(g1s, g2s, 2, nullptr, &is_one) != 0);
        uint64_t out[48];
        CHECK(zk::pairing_product(inf8, G2, 1, out, &is_one) == 0 && is_one == 1 && out[0] == 1);
    }
    printf("sanitize ok\n");
    return 0;
}
