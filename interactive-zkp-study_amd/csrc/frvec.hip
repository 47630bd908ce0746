// frvec.hip -- F_r vector primitives on device buffers of canonical elements (32 B each): linear combinations,
// products, x[i] *= g^i, and inclusive scans under + or * in either direction.  They are what the coefficient /
// evaluation algebra of a prover is made of once its vectors live in HBM: the reference's Polynomial.__add__ / scale /
// __mul__ / evaluate (zkp/plonk/polynomial.py:85-162,189-198), poly_div by a linear factor (polynomial.py:385-436: synthetic
// division = a suffix sum of c_j zeta^j) and the permutation grand product (zkp/plonk/permutation.py:89-140: prefix and
// suffix products).  Used by zkhip/plonk/prover_device.py.
#include <vector>
#include <string.h>
#include "frvec.h"
#include "host_field.h"

namespace zk {

namespace {

__device__ __forceinline__ Fr ldc(const uint32_t *p) {
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
    const uint4 a = q[0], b = q[1];
    const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    return fe_from_words<FrTag>(w);
}
__device__ __forceinline__ void stc(uint32_t *p, const Fr &v) {
    uint32_t w[8];
    fe_to_words(fe_reduce_full(v), w);
    uint4 *q = reinterpret_cast<uint4 *>(p);
    q[0] = make_uint4(w[0], w[1], w[2], w[3]);
    q[1] = make_uint4(w[4], w[5], w[6], w[7]);
}

struct LincombArgs {
    const uint32_t *in[FR_LINCOMB_MAX];
    Fr coef[FR_LINCOMB_MAX];  // Montgomery form: mont_mul(x, c * R) = x * c
    Fr constant;              // plain value (< r), added to every element
    uint32_t k;
};

// out[i] = constant + sum_j coef[j] * in[j][i]
__global__ __launch_bounds__(256) void fr_lincomb_kernel(uint32_t *__restrict__ out, LincombArgs A, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    Fr acc = A.constant;
    for (uint32_t j = 0; j < A.k; j++) {
        acc = fe_add(acc, fe_mul(ldc(A.in[j] + i * 8), A.coef[j]));  // < 2r + 2r
        fe_wreduce<4>(acc);                                           // < 2r
    }
    stc(out + i * 8, acc);
}

// out[i] = a[i] * b[i]
__global__ __launch_bounds__(256) void fr_mul_kernel(uint32_t *__restrict__ out, const uint32_t *__restrict__ a, const uint32_t *__restrict__ b, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    stc(out + i * 8, fe_to_mont(fe_mul(ldc(a + i * 8), ldc(b + i * 8))));  // (ab / R) * R^2 / R = ab
}

// x[i] *= g^i with g^i = A[i & mask] * B[i >> lh] (two-level table, Montgomery form)
__global__ __launch_bounds__(256) void fr_powers_kernel(uint32_t *__restrict__ x, const Fr *__restrict__ A, const Fr *__restrict__ B, uint32_t lh, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const Fr w = fe_mul(A[i & (((size_t)1 << lh) - 1)], B[i >> lh]);
    stc(x + i * 8, fe_mul(ldc(x + i * 8), w));
}

// The two-level table itself, built on the device: out[i] = g^i for i < na, out[na + j] = (g^na)^j for j < nb; every
// thread raises its base to its own index by square-and-multiply (at most ~28 products), so a call needs no host table,
// no upload and no synchronisation.  g, gh = g^na: Montgomery form.
__global__ __launch_bounds__(256) void fr_power_table_kernel(Fr *__restrict__ out, Fr g, Fr gh, uint32_t na, uint32_t nb) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= na + nb) return;
    const Fr base = t < na ? g : gh;
    uint32_t e = t < na ? t : t - na;
    Fr acc = Fr::one(), sq = base;
#pragma unroll 1
    while (e) {
        if (e & 1u) acc = fe_mul(acc, sq);
        sq = fe_sqr(sq);
        e >>= 1;
    }
    out[t] = acc;
}

// ------------------------------------------------------------------------------ PLONK quotient, fused
// t[i] = (gate + alpha * (num - den) + alpha^2 * (z - 1) * L1) / Z_H on the evaluation coset (zkp/plonk/prover/round3.py:114-147
// builds the same numerator by polynomial products and divides by Z_H with poly_div):
//   gate = q_L a + q_R b + q_O c + q_M a b + q_C
//   num  = (a + beta x + gamma)(b + beta K1 x + gamma)(c + beta K2 x + gamma) z(x)          K1 = 2, K2 = 3
//   den  = (a + beta s1 + gamma)(b + beta s2 + gamma)(c + beta s3 + gamma) z(omega x)
// Inputs are plain (canonical) values; a Montgomery product of two plain values carries 1/R, so the terms are brought to the
// common scale 1/R by constants prepared on the host (alpha R^3, alpha^2 R, R^2) and the last product by zh_inv R^2 lands on
// the plain result: 21 multiplications per element, one pass over the 15 input vectors.
struct QuotientArgs {
    const uint32_t *in[15];  // a b c z zw | ql qr qo qm qc | s1 s2 s3 | x l1
    Fr beta_m, gamma, alpha_r3, alpha2_r, zh_inv_r2[8];
    uint32_t period;
};
__global__ __launch_bounds__(256) void plonk_quotient_kernel(uint32_t *__restrict__ out, QuotientArgs A, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    auto ld = [&](int k) { return ldc(A.in[k] + i * 8); };
    auto add2 = [](const Fr &p, const Fr &q) { Fr r = fe_add(p, q); fe_wreduce<4>(r); return r; };   // < 2r
    const Fr a = ld(0), b = ld(1), c = ld(2), z = ld(3), zw = ld(4);
    // gate / R
    Fr gate = add2(fe_mul(ld(5), a), fe_mul(ld(6), b));
    gate = add2(gate, fe_mul(ld(7), c));
    gate = add2(gate, fe_to_mont(fe_mul(fe_mul(a, b), ld(8))));     // (ab qm / R^2) * R^2 / R
    gate = add2(gate, fe_from_mont(ld(9)));                          // q_C / R
    // num / R^3 and den / R^3
    const Fr bx = fe_mul(ld(13), A.beta_m);                          // beta x, plain, < 2r
    const Fr bx2 = add2(bx, bx), bx3 = add2(bx2, bx);
    auto term = [&](const Fr &w, const Fr &bs) { return add2(add2(w, bs), A.gamma); };
    Fr num = fe_mul(fe_mul(fe_mul(term(a, bx), term(b, bx2)), term(c, bx3)), z);
    Fr den = fe_mul(fe_mul(fe_mul(term(a, fe_mul(ld(10), A.beta_m)), term(b, fe_mul(ld(11), A.beta_m))), term(c, fe_mul(ld(12), A.beta_m))), zw);
    Fr diff = fe_sub_k<2>(num, den);                                 // num - den + 2r < 4r
    fe_wreduce<4>(diff);
    Fr tot = add2(gate, fe_mul(diff, A.alpha_r3));                   // alpha (num - den) / R
    Fr one_plain = Fr::zero();
    one_plain.l[0] = 1u;
    Fr zm1 = fe_sub_k<2>(z, one_plain);                              // z - 1 + 2r
    fe_wreduce<4>(zm1);
    tot = add2(tot, fe_mul(fe_mul(zm1, ld(14)), A.alpha2_r));        // alpha^2 (z - 1) L1 / R
    stc(out + i * 8, fe_mul(tot, A.zh_inv_r2[i % A.period]));        // (S / R) * zh_inv R^2 / R
}

// ------------------------------------------------------------------------------ PLONK grand product, the per-row factors fused
// num[i] = (a + beta x + gamma)(b + beta K1 x + gamma)(c + beta K2 x + gamma),   den[i] = (a + beta s1 + gamma)(b + beta s2 + gamma)(c + beta s3 + gamma)
// with x = omega^i, K1 = 2, K2 = 3 (zkp/plonk/permutation.py:89-137 forms them row by row on Python integers; the device prover
// then takes prefix / suffix products of them).  One pass over the seven input vectors instead of six linear combinations and four
// products (ten launches, each streaming two or three vectors of n elements).  Plain values in and out: the first two factors
// of each product are brought to Montgomery form, so that the product of the three carries no power of R.
struct PermFactorArgs {
    const uint32_t *in[7];   // a b c | s1 s2 s3 | x
    Fr beta_m, gamma;
};
__global__ __launch_bounds__(256) void plonk_perm_factors_kernel(uint32_t *__restrict__ num, uint32_t *__restrict__ den, PermFactorArgs A, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    auto ld = [&](int k) { return ldc(A.in[k] + i * 8); };
    auto add2 = [](const Fr &p, const Fr &q) { Fr r = fe_add(p, q); fe_wreduce<4>(r); return r; };   // < 2r
    auto term = [&](const Fr &w, const Fr &bs) { return add2(add2(w, bs), A.gamma); };
    auto prod3 = [](const Fr &t1, const Fr &t2, const Fr &t3) { return fe_mul(fe_mul(fe_to_mont(t1), fe_to_mont(t2)), t3); };   // (t1 R)(t2 R)/R * t3 / R
    const Fr a = ld(0), b = ld(1), c = ld(2);
    const Fr bx = fe_mul(ld(6), A.beta_m);                           // beta x, plain, < 2r
    const Fr bx2 = add2(bx, bx), bx3 = add2(bx2, bx);
    stc(num + i * 8, prod3(term(a, bx), term(b, bx2), term(c, bx3)));
    stc(den + i * 8, prod3(term(a, fe_mul(ld(3), A.beta_m)), term(b, fe_mul(ld(4), A.beta_m)), term(c, fe_mul(ld(5), A.beta_m))));
}

// ------------------------------------------------------------------------------ scans
// Values are kept in the form in which `op` is one field operation: plain for +, Montgomery for * (mont_mul of two
// Montgomery values is the Montgomery product).  Bounds: op results < 2r for both.
template <bool MUL> __device__ __forceinline__ Fr scan_op(const Fr &a, const Fr &b) {
    if (MUL) return fe_mul(a, b);
    Fr s = fe_add(a, b);
    fe_wreduce<4>(s);
    return s;
}
template <bool MUL> __device__ __forceinline__ Fr scan_identity() { return MUL ? Fr::one() : Fr::zero(); }

constexpr int SCAN_NT = 256, SCAN_EPT = 8, SCAN_CHUNK = SCAN_NT * SCAN_EPT;

// Phase 1: every workgroup scans its 2048-element chunk in place (thread-serial over 8 consecutive elements, then a
// Hillis-Steele pass over the 256 thread totals in LDS) and writes the chunk total.  REV scans from the end: logical
// element e of the scan is physical element n - 1 - e.
template <bool MUL, bool REV>
__global__ __launch_bounds__(SCAN_NT) void fr_scan_chunk_kernel(uint32_t *__restrict__ x, uint32_t *__restrict__ totals, size_t n) {
    __shared__ uint32_t sh[2][NL][SCAN_NT];
    const uint32_t t = threadIdx.x;
    const size_t base = (size_t)blockIdx.x * SCAN_CHUNK + (size_t)t * SCAN_EPT;
    Fr v[SCAN_EPT];
    Fr run = scan_identity<MUL>();
#pragma unroll
    for (int k = 0; k < SCAN_EPT; k++) {
        const size_t e = base + k;
        if (e < n) {
            Fr a = ldc(x + (REV ? n - 1 - e : e) * 8);
            if (MUL) a = fe_to_mont(a);
            run = scan_op<MUL>(run, a);
        }
        v[k] = run;
    }
    // inclusive scan of the thread totals
    int cur = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) sh[0][i][t] = run.l[i];
    __syncthreads();
    for (uint32_t d = 1; d < SCAN_NT; d <<= 1) {
        Fr mine, other;
#pragma unroll
        for (int i = 0; i < NL; i++) mine.l[i] = sh[cur][i][t];
        if (t >= d) {
#pragma unroll
            for (int i = 0; i < NL; i++) other.l[i] = sh[cur][i][t - d];
            mine = scan_op<MUL>(other, mine);
        }
#pragma unroll
        for (int i = 0; i < NL; i++) sh[cur ^ 1][i][t] = mine.l[i];
        cur ^= 1;
        __syncthreads();
    }
    Fr before = scan_identity<MUL>();
    if (t > 0) {
#pragma unroll
        for (int i = 0; i < NL; i++) before.l[i] = sh[cur][i][t - 1];
    }
#pragma unroll
    for (int k = 0; k < SCAN_EPT; k++) {
        const size_t e = base + k;
        if (e < n) {
            Fr r = t > 0 ? scan_op<MUL>(before, v[k]) : v[k];
            if (MUL) r = fe_from_mont(r);
            stc(x + (REV ? n - 1 - e : e) * 8, r);
        }
    }
    if (t == SCAN_NT - 1) {
        Fr tot;
#pragma unroll
        for (int i = 0; i < NL; i++) tot.l[i] = sh[cur][i][t];
        if (MUL) tot = fe_from_mont(tot);
        stc(totals + (size_t)blockIdx.x * 8, tot);
    }
}
// Phase 3: element of chunk b (b >= 1) op= inclusive scan of the chunk totals at b - 1.
template <bool MUL, bool REV>
__global__ __launch_bounds__(SCAN_NT) void fr_scan_apply_kernel(uint32_t *__restrict__ x, const uint32_t *__restrict__ totals_scanned, size_t n) {
    const size_t e = (size_t)blockIdx.x * SCAN_NT + threadIdx.x + SCAN_CHUNK;  // chunk 0 needs nothing
    if (e >= n) return;
    const size_t b = e / SCAN_CHUNK;
    Fr pre = ldc(totals_scanned + (b - 1) * 8);
    uint32_t *p = x + (REV ? n - 1 - e : e) * 8;
    if (MUL)
        stc(p, fe_mul(fe_to_mont(pre), ldc(p)));  // (pre R)(v) / R = pre v
    else
        stc(p, scan_op<false>(pre, ldc(p)));
}

template <bool MUL, bool REV> void scan_impl(uint32_t *x, size_t n, hipStream_t st, DevBuf *scratch, size_t level) {
    if (n <= 1) return;
    const size_t chunks = (n + SCAN_CHUNK - 1) / SCAN_CHUNK;
    if (level >= 4) throw std::runtime_error("zk_fr_scan_dev: vector too long");
    if (scratch[level].bytes < chunks * 32) {
        ZK_HIP(hipStreamSynchronize(st));
        scratch[level].alloc(chunks * 32);
    }
    uint32_t *totals = scratch[level].as<uint32_t>();
    hipLaunchKernelGGL((fr_scan_chunk_kernel<MUL, REV>), dim3((unsigned)chunks), dim3(SCAN_NT), 0, st, x, totals, n);
    if (chunks > 1) {
        scan_impl<MUL, false>(totals, chunks, st, scratch, level + 1);  // the totals array is already in scan order
        hipLaunchKernelGGL((fr_scan_apply_kernel<MUL, REV>), dim3((unsigned)((n - SCAN_CHUNK + SCAN_NT - 1) / SCAN_NT)), dim3(SCAN_NT), 0, st, x, totals, n);
    }
    ZK_HIP(hipGetLastError());
}

// ------------------------------------------------------------------------------ p(z) for up to 8 polynomials
// out[j] = sum_i c_j[i] z^i (Polynomial.evaluate, zkp/plonk/polynomial.py:85-106, is Horner's rule: n dependent products).  Thread t of
// block b starts at element b * 256 + t with z^i from the two-level table (one product) and walks on in steps of EVAL_BLOCKS * 256
// elements, multiplying its power by z^stride: two products per element, consecutive lanes on consecutive elements, no table in
// the loop.  The coefficients are plain, the powers Montgomery: mont_mul(c, z^i R) = c z^i.  Block sums through LDS, one partial per
// block and polynomial; the second kernel adds a polynomial's partials.
constexpr int EVAL_NT = 256, EVAL_BLOCKS = 256, EVAL_MAX = 8;
struct EvalArgs {
    const uint32_t *coef[EVAL_MAX];
    uint32_t count[EVAL_MAX];
    Fr zstride;   // z^(EVAL_BLOCKS * EVAL_NT), Montgomery form
};
__device__ __forceinline__ Fr eval_block_sum(Fr acc, uint32_t (*sh)[EVAL_NT]) {
    const uint32_t t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < NL; i++) sh[i][t] = acc.l[i];
    __syncthreads();
    for (uint32_t d = EVAL_NT / 2; d >= 1; d >>= 1) {
        if (t < d) {
            Fr a, b;
#pragma unroll
            for (int i = 0; i < NL; i++) {
                a.l[i] = sh[i][t];
                b.l[i] = sh[i][t + d];
            }
            ZK_DBG(a.vb = 2; a.lmax = 1; b.vb = 2; b.lmax = 1;)
            a = fe_add(a, b);
            fe_wreduce<4>(a);
#pragma unroll
            for (int i = 0; i < NL; i++) sh[i][t] = a.l[i];
        }
        __syncthreads();
    }
    Fr r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.l[i] = sh[i][0];
    return r;
}
__global__ __launch_bounds__(EVAL_NT) void fr_eval_partial_kernel(EvalArgs A, const Fr *__restrict__ tabA, const Fr *__restrict__ tabB,
                                                                 uint32_t *__restrict__ partial) {
    __shared__ uint32_t sh[NL][EVAL_NT];
    const uint32_t j = blockIdx.y, n = A.count[j];
    const uint32_t *__restrict__ c = A.coef[j];
    Fr acc = Fr::zero();
    uint32_t i = blockIdx.x * EVAL_NT + threadIdx.x;
    if (i < n) {
        Fr pw = fe_mul(tabA[threadIdx.x], tabB[blockIdx.x]);          // z^i = z^t * (z^256)^b, Montgomery form, < 2r
        for (;;) {
            acc = fe_add(acc, fe_mul(ldc(c + (size_t)i * 8), pw));     // < 2r + 2r
            fe_wreduce<4>(acc);
            i += EVAL_BLOCKS * EVAL_NT;
            if (i >= n || i < EVAL_BLOCKS * EVAL_NT) break;            // (the second test: 32-bit wrap-around)
            pw = fe_mul(pw, A.zstride);
        }
    }
    const Fr tot = eval_block_sum(acc, sh);
    if (threadIdx.x == 0) stc(partial + ((size_t)j * EVAL_BLOCKS + blockIdx.x) * 8, tot);
}
__global__ __launch_bounds__(EVAL_NT) void fr_eval_final_kernel(const uint32_t *__restrict__ partial, uint32_t *__restrict__ out) {
    __shared__ uint32_t sh[NL][EVAL_NT];
    static_assert(EVAL_BLOCKS == EVAL_NT, "one partial per thread");
    const Fr tot = eval_block_sum(ldc(partial + ((size_t)blockIdx.x * EVAL_BLOCKS + threadIdx.x) * 8), sh);
    if (threadIdx.x == 0) stc(out + (size_t)blockIdx.x * 8, tot);
}

HFr host_fr(const uint64_t v[4]) {
    HFr a;
    memcpy(a.l, v, 32);
    return a;
}

}  // namespace

void fr_lincomb(void *d_out, const void *const *d_in, const uint64_t *coeffs, unsigned k, const uint64_t constant[4], size_t n, hipStream_t st) {
    if (k > FR_LINCOMB_MAX) throw std::runtime_error("zk_fr_lincomb_dev: at most 8 input vectors");
    if (n == 0) return;
    LincombArgs A;
    memset(&A, 0, sizeof(A));
    A.k = k;
    for (unsigned j = 0; j < k; j++) {
        A.in[j] = static_cast<const uint32_t *>(d_in[j]);
        A.coef[j] = fe_to_mont(host_fr(coeffs + 4 * j)).to_dev();
    }
    if (constant) {
        uint32_t w[8];
        memcpy(w, constant, 32);
        A.constant = fe_from_words<FrTag>(w);  // plain value, normalised limbs
    } else {
        A.constant = Fr::zero();
    }
    hipLaunchKernelGGL(fr_lincomb_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, static_cast<uint32_t *>(d_out), A, n);
    ZK_HIP(hipGetLastError());
}

void fr_mul(void *d_out, const void *d_a, const void *d_b, size_t n, hipStream_t st) {
    if (n == 0) return;
    hipLaunchKernelGGL(fr_mul_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, static_cast<uint32_t *>(d_out), static_cast<const uint32_t *>(d_a),
                       static_cast<const uint32_t *>(d_b), n);
    ZK_HIP(hipGetLastError());
}

void plonk_quotient(void *d_out, const void *const *d_in, const uint64_t *zh_inv, unsigned period, const uint64_t alpha[4], const uint64_t beta[4],
                    const uint64_t gamma[4], size_t n, hipStream_t st) {
    if (period == 0 || period > 8 || (period & (period - 1))) throw std::runtime_error("zk_plonk_quotient_dev: period must be 1, 2, 4 or 8");
    if (n == 0) return;
    QuotientArgs A;
    memset(&A, 0, sizeof(A));
    for (int k = 0; k < 15; k++) A.in[k] = static_cast<const uint32_t *>(d_in[k]);
    auto dev_mont = [](const HFr &plain) { return fe_to_mont(plain).to_dev(); };   // x -> x R (device radix)
    const HFr al = host_fr(alpha);
    A.beta_m = dev_mont(host_fr(beta));
    uint32_t w[8];
    memcpy(w, gamma, 32);
    A.gamma = fe_from_words<FrTag>(w);
    A.alpha_r3 = fe_to_mont(fe_to_mont(dev_mont(al)));              // device arithmetic compiled for the host: x R -> x R^2 -> x R^3
    const HFr am = fe_to_mont(al);
    A.alpha2_r = fe_mul(am, am).to_dev();                            // host Montgomery square (alpha R)(alpha R) / R = alpha^2 R, then device radix
    for (unsigned k = 0; k < period; k++) A.zh_inv_r2[k] = fe_to_mont(dev_mont(host_fr(zh_inv + 4 * k)));
    A.period = period;
    hipLaunchKernelGGL(plonk_quotient_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, static_cast<uint32_t *>(d_out), A, n);
    ZK_HIP(hipGetLastError());
}

void plonk_perm_factors(void *d_num, void *d_den, const void *const *d_in, const uint64_t beta[4], const uint64_t gamma[4], size_t n, hipStream_t st) {
    if (n == 0) return;
    PermFactorArgs A;
    memset(&A, 0, sizeof(A));
    for (int k = 0; k < 7; k++) A.in[k] = static_cast<const uint32_t *>(d_in[k]);
    A.beta_m = fe_to_mont(host_fr(beta)).to_dev();
    uint32_t w[8];
    memcpy(w, gamma, 32);
    A.gamma = fe_from_words<FrTag>(w);
    hipLaunchKernelGGL(plonk_perm_factors_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, static_cast<uint32_t *>(d_num), static_cast<uint32_t *>(d_den), A, n);
    ZK_HIP(hipGetLastError());
}

void FrVecScratch::scale_powers(void *d_data, size_t n, const uint64_t base[4], hipStream_t st) {
    if (n == 0) return;
    unsigned L = 0;
    while (((size_t)1 << L) < n) L++;
    const unsigned lh = (L + 1) / 2;
    const size_t na = (size_t)1 << lh, nb = ((n - 1) >> lh) + 1;
    const HFr g = fe_to_mont(host_fr(base));
    HFr gh = g;
    for (unsigned i = 0; i < lh; i++) gh = fe_sqr(gh);  // g^(2^lh)
    // The table is built by a kernel on the caller's stream (stream order protects an earlier call's readers); it is
    // sized once for the largest domain (2^28: 2 * 2^14 entries) so that no call allocates.
    const size_t cap = (size_t)2 << 14;
    if (tables.bytes < std::max(cap, na + nb) * sizeof(Fr)) tables.alloc(std::max(cap, na + nb) * sizeof(Fr));
    hipLaunchKernelGGL(fr_power_table_kernel, dim3((unsigned)((na + nb + 255) / 256)), dim3(256), 0, st, tables.as<Fr>(), g.to_dev(), gh.to_dev(),
                       (uint32_t)na, (uint32_t)nb);
    hipLaunchKernelGGL(fr_powers_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, static_cast<uint32_t *>(d_data), tables.as<Fr>(),
                       tables.as<Fr>() + na, lh, n);
    ZK_HIP(hipGetLastError());
}

void FrVecScratch::eval(const void *const *d_coefs, const size_t *counts, unsigned k, const uint64_t point[4], void *d_out, hipStream_t st) {
    if (k == 0) return;
    if (k > (unsigned)EVAL_MAX) throw std::runtime_error("zk_fr_eval_dev: at most 8 polynomials per call");
    EvalArgs A;
    memset(&A, 0, sizeof(A));
    for (unsigned j = 0; j < k; j++) {
        if (counts[j] >= ((size_t)1 << 31)) throw std::runtime_error("zk_fr_eval_dev: polynomial too long");
        A.coef[j] = static_cast<const uint32_t *>(d_coefs[j]);
        A.count[j] = (uint32_t)counts[j];
    }
    const HFr g = fe_to_mont(host_fr(point));
    HFr gh = g;
    for (int i = 0; i < 8; i++) gh = fe_sqr(gh);     // z^256
    HFr gs = gh;
    for (int i = 0; i < 8; i++) gs = fe_sqr(gs);     // z^65536 = z^(EVAL_BLOCKS * EVAL_NT)
    A.zstride = gs.to_dev();
    // tables: z^t (256) | (z^256)^b (256) | partial sums (8 * 256 canonical elements): inside the buffer scale_powers sizes once
    const size_t cap = (size_t)2 << 14;
    if (tables.bytes < cap * sizeof(Fr)) tables.alloc(cap * sizeof(Fr));
    Fr *tab = tables.as<Fr>();
    uint32_t *partial = reinterpret_cast<uint32_t *>(tab + 2 * EVAL_NT);
    static_assert((2 * EVAL_NT) * sizeof(Fr) + (size_t)EVAL_MAX * EVAL_BLOCKS * 32 <= ((size_t)2 << 14) * sizeof(Fr), "scratch layout");
    hipLaunchKernelGGL(fr_power_table_kernel, dim3(2), dim3(256), 0, st, tab, g.to_dev(), gh.to_dev(), (uint32_t)EVAL_NT, (uint32_t)EVAL_BLOCKS);
    hipLaunchKernelGGL(fr_eval_partial_kernel, dim3(EVAL_BLOCKS, k), dim3(EVAL_NT), 0, st, A, tab, tab + EVAL_NT, partial);
    hipLaunchKernelGGL(fr_eval_final_kernel, dim3(k), dim3(EVAL_NT), 0, st, partial, static_cast<uint32_t *>(d_out));
    ZK_HIP(hipGetLastError());
}

void FrVecScratch::scan(void *d_data, size_t n, bool mul, bool reverse, hipStream_t st) {
    uint32_t *x = static_cast<uint32_t *>(d_data);
    if (mul) {
        if (reverse) scan_impl<true, true>(x, n, st, levels, 0);
        else scan_impl<true, false>(x, n, st, levels, 0);
    } else {
        if (reverse) scan_impl<false, true>(x, n, st, levels, 0);
        else scan_impl<false, false>(x, n, st, levels, 0);
    }
}

}  // namespace zk
