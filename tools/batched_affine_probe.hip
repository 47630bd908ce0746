// batched_affine_probe.hip -- the bandwidth-for-ALU trade of bucket accumulation, built and measured.
//
// Bucket accumulation today: XYZZ += affine, 8M + 2S per addition, one 64-byte gather, no stores (csrc/msm_impl.h,
// msm_accumulate_kernel).  The alternative: affine + affine -> affine, 2M + 1S plus one shared inversion (Montgomery's trick),
// organised as pairwise rounds over the sorted bucket lists.  This probe runs ONE such round with the library's own field
// arithmetic (csrc/field.h, curve.h) on the access pattern of the first round of a 2^20-point MSM -- 2^23 pairs of points gathered
// from a 2^20-entry (64 MB) table, 64-byte affine results written out -- and checks it bit for bit against the XYZZ formulas:
//
//   ba_products   256 threads x 4 pairs: dx_i = x_b - x_a, thread product, workgroup product tree in LDS -> one product per block
//   ba_invert     one Fermat inversion per block product (2^23 / 1024 = 8192 of them, every lane busy)
//   ba_add        the same pairs again: dx_i and the thread's prefix products (recomputed: they cannot survive between two
//                 kernels except through HBM), product tree up, inverse tree down (the inverse of every thread's product from the
//                 block inverse), back-substitution, lambda, x3, y3, store
//   xyzz_pair     the same pairs with the library's mixed addition (what the accumulate kernel does per list entry), for the
//                 same two gathers and a 144-byte XYZZ store: the like-for-like comparison
//
// Build: make -C tools probe        Run (GPU box): tools/build/batched_affine_probe [log2 pairs] [reps]
// Prints one JSON line.  See profiles/r03_batched_affine.md for the numbers and what they mean.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../interactive-zkp-study_amd/csrc/curve.h"

using namespace zk;

#define CK(x)                                                                                  \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) {                                                                \
            fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                           \
        }                                                                                      \
    } while (0)

constexpr int NT = 256, G = 4;

struct alignas(16) Packed {
    uint32_t w[16];  // x (8 words) | y (8 words): lazy Montgomery coordinates < 2p packed to 256 bits each (as PackedAffine<Fp>)
};
__device__ __forceinline__ Fp ld8(const uint32_t *p) {
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
    const uint4 a = q[0], b = q[1];
    const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    return fe_from_words<FpTag>(w);
}
__device__ __forceinline__ void st8(uint32_t *p, const Fp &v) {
    uint32_t w[8];
    fe_to_words(v, w);
    uint4 *q = reinterpret_cast<uint4 *>(p);
    q[0] = make_uint4(w[0], w[1], w[2], w[3]);
    q[1] = make_uint4(w[4], w[5], w[6], w[7]);
}
// limb-major LDS array of field elements (conflict-free for consecutive indices)
__device__ __forceinline__ Fp lds_get(const uint32_t *base, uint32_t stride, uint32_t i) {
    Fp r;
#pragma unroll
    for (int k = 0; k < NL; k++) r.l[k] = base[k * stride + i];
    return r;
}
__device__ __forceinline__ void lds_put(uint32_t *base, uint32_t stride, uint32_t i, const Fp &v) {
#pragma unroll
    for (int k = 0; k < NL; k++) base[k * stride + i] = v.l[k];
}

// pseudo-random table entry: any pair of field elements will do (the addition formulas are identities of rational functions)
__global__ void fill_table(Packed *T, uint32_t n) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    Fp x = Fp::one(), y = Fp::one();
    x.l[0] = (x.l[0] + i * 2654435761u) & LMASK;
    x.l[3] = (x.l[3] ^ (i * 40503u)) & LMASK;
    y.l[1] = (y.l[1] + i * 2246822519u) & LMASK;
    y.l[5] = (y.l[5] ^ (i * 3266489917u)) & LMASK;
    const Fp xx = fe_mul(fe_sqr(x), y), yy = fe_mul(fe_sqr(y), x);   // spread over the field, < 2p
    st8(T[i].w, xx);
    st8(T[i].w + 8, yy);
}
__global__ void fill_pairs(uint32_t *ia, uint32_t *ib, uint32_t pairs, uint32_t mask) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= pairs) return;
    uint32_t a = (i * 2654435761u) ^ (i >> 7), b = (i * 2246822519u + 12345u) ^ (i >> 5);
    a &= mask;
    b &= mask;
    if (a == b) b = (b + 1) & mask;
    ia[i] = a;
    ib[i] = b;
}

// workgroup product tree over the 256 thread products: node[256 + t] = leaf, node[k] = node[2k] * node[2k + 1]
__device__ __forceinline__ void tree_up(uint32_t *node, uint32_t t) {
    for (uint32_t s = NT / 2; s >= 1; s >>= 1) {
        __syncthreads();
        if (t < s) lds_put(node, 2 * NT, s + t, fe_mul(lds_get(node, 2 * NT, 2 * (s + t)), lds_get(node, 2 * NT, 2 * (s + t) + 1)));
    }
    __syncthreads();
}

__global__ __launch_bounds__(NT) void ba_products(const Packed *__restrict__ T, const uint32_t *__restrict__ ia, const uint32_t *__restrict__ ib,
                                                  Fp *__restrict__ blockprod) {
    __shared__ uint32_t node[NL * 2 * NT];
    const uint32_t t = threadIdx.x, p0 = (blockIdx.x * NT + t) * G;
    Fp prod;
#pragma unroll
    for (int i = 0; i < G; i++) {
        const Fp dx = fe_sub_k<2>(ld8(T[ib[p0 + i]].w), ld8(T[ia[p0 + i]].w));   // < 4p
        prod = i == 0 ? dx : fe_mul(prod, dx);
    }
    lds_put(node, 2 * NT, NT + t, prod);
    tree_up(node, t);
    if (t == 0) blockprod[blockIdx.x] = lds_get(node, 2 * NT, 1);
}

__global__ __launch_bounds__(64) void ba_invert(const Fp *__restrict__ blockprod, Fp *__restrict__ blockinv, uint32_t n) {
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i < n) blockinv[i] = fe_inv(blockprod[i]);
}

__global__ __launch_bounds__(NT) void ba_add(const Packed *__restrict__ T, const uint32_t *__restrict__ ia, const uint32_t *__restrict__ ib,
                                             const Fp *__restrict__ blockinv, Packed *__restrict__ out) {
    __shared__ uint32_t node[NL * 2 * NT];
    __shared__ uint32_t inv[NL * 2 * NT];
    const uint32_t t = threadIdx.x, p0 = (blockIdx.x * NT + t) * G;
    Fp c[G];   // c[i] = dx_0 * ... * dx_i
#pragma unroll
    for (int i = 0; i < G; i++) {
        const Fp dx = fe_sub_k<2>(ld8(T[ib[p0 + i]].w), ld8(T[ia[p0 + i]].w));
        c[i] = i == 0 ? dx : fe_mul(c[i - 1], dx);
    }
    lds_put(node, 2 * NT, NT + t, c[G - 1]);
    tree_up(node, t);
    // inverse tree: inv[k] = 1 / node[k];  inv[2k] = inv[k] * node[2k + 1],  inv[2k + 1] = inv[k] * node[2k]
    if (t == 0) lds_put(inv, 2 * NT, 1, blockinv[blockIdx.x]);
    for (uint32_t s = 1; s <= NT / 2; s <<= 1) {   // s parents, 2 s children
        __syncthreads();
        if (t < 2 * s) {
            const uint32_t child = 2 * s + t;
            lds_put(inv, 2 * NT, child, fe_mul(lds_get(inv, 2 * NT, child >> 1), lds_get(node, 2 * NT, child ^ 1u)));
        }
    }
    __syncthreads();
    Fp u = lds_get(inv, 2 * NT, NT + t);   // 1 / (dx_0 ... dx_{G-1})
    auto step = [&](const int i, const Fp *cprev) {   // pair i: cprev = c[i - 1] (nullptr for i == 0)
        const uint32_t a = ia[p0 + i], b = ib[p0 + i];
        const Fp xa = ld8(T[a].w), ya = ld8(T[a].w + 8), xb = ld8(T[b].w), yb = ld8(T[b].w + 8);
        const Fp dinv = cprev ? fe_mul(u, *cprev) : u;                   // 1 / dx_i
        if (cprev) u = fe_mul(u, fe_sub_k<2>(xb, xa));
        const Fp lam = fe_mul(fe_sub_k<2>(yb, ya), dinv);                // < 2p
        Fp x3 = fe_sub_k<2>(fe_sub_k<2>(fe_sqr(lam), xa), xb);           // < 6p
        fe_wreduce<8>(x3);                                               // < 2p
        Fp y3 = fe_mul_minus<2>(lam, fe_sub_lazy<2>(xa, x3), ya);        // lam (xa - x3) - ya;  < 6p
        fe_wreduce<8>(y3);
        st8(out[p0 + i].w, x3);
        st8(out[p0 + i].w + 8, y3);
    };
    static_assert(G == 4, "unrolled by hand");
    step(3, &c[2]);
    step(2, &c[1]);
    step(1, &c[0]);
    step(0, nullptr);
}

// like-for-like: the library's mixed addition on the same pairs (two gathers, one 144-byte XYZZ store)
__global__ __launch_bounds__(64, 4) void xyzz_pair(const Packed *__restrict__ T, const uint32_t *__restrict__ ia, const uint32_t *__restrict__ ib,
                                                   Xyzz<Fp> *__restrict__ out, uint32_t pairs) {
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= pairs) return;
    const uint32_t a = ia[i], b = ib[i];
    Xyzz<Fp> acc{ld8(T[a].w), ld8(T[a].w + 8), Fp::one(), Fp::one()};
    xyzz_add_affine(acc, Affine<Fp>{ld8(T[b].w), ld8(T[b].w + 8)});
    out[i] = acc;
}
// a chain like the real accumulate kernel: `len` gathered points added into one XYZZ accumulator per thread
__global__ __launch_bounds__(64, 4) void xyzz_chain(const Packed *__restrict__ T, const uint32_t *__restrict__ ia, Xyzz<Fp> *__restrict__ out, uint32_t threads,
                                                    uint32_t len) {
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= threads) return;
    Xyzz<Fp> acc = Xyzz<Fp>::inf();
#pragma unroll 1
    for (uint32_t k = 0; k < len; k++) {
        const uint32_t a = ia[(size_t)i * len + k];
        xyzz_add_affine(acc, Affine<Fp>{ld8(T[a].w), ld8(T[a].w + 8)});
    }
    out[i] = acc;
}

// verification: affine result == XYZZ result brought to affine, on the first `count` pairs
__global__ void check_kernel(const Packed *__restrict__ aff, const Xyzz<Fp> *__restrict__ xyzz, uint32_t count, uint32_t *bad) {
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= count) return;
    const Affine<Fp> want = xyzz_to_affine(xyzz[i]);
    if (!want.x.equals(ld8(aff[i].w)) || !want.y.equals(ld8(aff[i].w + 8))) atomicAdd(bad, 1u);
}

int main(int argc, char **argv) {
    const int logp = argc > 1 ? atoi(argv[1]) : 23, reps = argc > 2 ? atoi(argv[2]) : 5;
    const uint32_t pairs = 1u << logp, n = 1u << 20, nblocks = pairs / (NT * G);
    Packed *T, *out;
    uint32_t *ia, *ib, *bad;
    Fp *bprod, *binv;
    Xyzz<Fp> *xo;
    CK(hipMalloc(&T, (size_t)n * sizeof(Packed)));
    CK(hipMalloc(&out, (size_t)pairs * sizeof(Packed)));
    CK(hipMalloc(&ia, (size_t)pairs * 4));
    CK(hipMalloc(&ib, (size_t)pairs * 4));
    CK(hipMalloc(&bprod, (size_t)nblocks * sizeof(Fp)));
    CK(hipMalloc(&binv, (size_t)nblocks * sizeof(Fp)));
    CK(hipMalloc(&xo, (size_t)pairs * sizeof(Xyzz<Fp>)));
    CK(hipMalloc(&bad, 4));
    CK(hipMemset(bad, 0, 4));
    hipLaunchKernelGGL(fill_table, dim3(n / 256), dim3(256), 0, 0, T, n);
    hipLaunchKernelGGL(fill_pairs, dim3(pairs / 256), dim3(256), 0, 0, ia, ib, pairs, n - 1);
    CK(hipDeviceSynchronize());
    hipEvent_t e[6];
    for (auto &x : e) CK(hipEventCreate(&x));
    float t_prod = 0, t_inv = 0, t_add = 0, t_xyzz = 0, t_chain = 0;
    const uint32_t chain_len = 32, chain_threads = pairs / chain_len;
    for (int r = 0; r <= reps; r++) {   // r == 0: warm-up
        CK(hipEventRecord(e[0], 0));
        hipLaunchKernelGGL(ba_products, dim3(nblocks), dim3(NT), 0, 0, T, ia, ib, bprod);
        CK(hipEventRecord(e[1], 0));
        hipLaunchKernelGGL(ba_invert, dim3((nblocks + 63) / 64), dim3(64), 0, 0, bprod, binv, nblocks);
        CK(hipEventRecord(e[2], 0));
        hipLaunchKernelGGL(ba_add, dim3(nblocks), dim3(NT), 0, 0, T, ia, ib, binv, out);
        CK(hipEventRecord(e[3], 0));
        hipLaunchKernelGGL(xyzz_pair, dim3(pairs / 64), dim3(64), 0, 0, T, ia, ib, xo, pairs);
        CK(hipEventRecord(e[4], 0));
        hipLaunchKernelGGL(xyzz_chain, dim3((chain_threads + 63) / 64), dim3(64), 0, 0, T, ia, xo + (pairs / 2), chain_threads, chain_len);
        CK(hipEventRecord(e[5], 0));
        CK(hipEventSynchronize(e[5]));
        CK(hipGetLastError());
        if (r == 0) continue;
        float ms;
        CK(hipEventElapsedTime(&ms, e[0], e[1])); t_prod += ms;
        CK(hipEventElapsedTime(&ms, e[1], e[2])); t_inv += ms;
        CK(hipEventElapsedTime(&ms, e[2], e[3])); t_add += ms;
        CK(hipEventElapsedTime(&ms, e[3], e[4])); t_xyzz += ms;
        CK(hipEventElapsedTime(&ms, e[4], e[5])); t_chain += ms;
    }
    // verification on the first 2^16 pairs (xo[0..] still holds xyzz_pair's output: the chain wrote to the upper half)
    const uint32_t vcount = pairs < (1u << 16) ? pairs / 4 : (1u << 16);
    hipLaunchKernelGGL(check_kernel, dim3((vcount + 63) / 64), dim3(64), 0, 0, out, xo, vcount, bad);
    uint32_t hbad = 0;
    CK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost));
    const double inv = 1.0 / reps;
    const double aff_ms = (t_prod + t_inv + t_add) * inv, xy_ms = t_xyzz * inv, ch_ms = t_chain * inv;
    printf("{\"pairs\": %u, \"table_points\": %u, \"reps\": %d, \"verified_pairs\": %u, \"mismatches\": %u, "
           "\"affine_round_ms\": {\"products\": %.4f, \"invert\": %.4f, \"add\": %.4f, \"total\": %.4f}, \"affine_ps_per_add\": %.1f, "
           "\"xyzz_pair_ms\": %.4f, \"xyzz_pair_ps_per_add\": %.1f, \"xyzz_chain32_ms\": %.4f, \"xyzz_chain_ps_per_add\": %.1f}\n",
           pairs, n, reps, vcount, hbad, t_prod * inv, t_inv * inv, t_add * inv, aff_ms, aff_ms * 1e9 / pairs, xy_ms, xy_ms * 1e9 / pairs, ch_ms,
           ch_ms * 1e9 / pairs);
    return hbad ? 1 : 0;
}
