// tools/reduce_probe.hip -- per-step times of the bucket reduction kernels (csrc/msm_reduce.h) on a synthetic bucket array.
// The block kernel is run with a step limit of 1, 2, ... BL: the differences are the cost of every step.  Not part of libzkhip.so.
// Build: make -C tools reduce_probe     Run: tools/build/reduce_probe [g1|g2] [log2 buckets, default 19] [reps, default 20]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include "msm_reduce.h"
using namespace zk;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

// bucket j = k_j * G for a 20-bit k_j taken from the index (no two neighbouring buckets equal or opposite, none empty)
template <class F> __global__ void fill_kernel(Xyzz<F> *x, uint32_t n, Affine<F> g) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const uint32_t k = ((j * 2654435761u) >> 12) | 1u;
    x[j] = xyzz_small_mul(Xyzz<F>::from_affine(g), k);
}
static Fp fp_from_u64(const uint64_t w[4]) {
    uint32_t v[8];
    for (int i = 0; i < 4; i++) { v[2 * i] = (uint32_t)w[i]; v[2 * i + 1] = (uint32_t)(w[i] >> 32); }
    return fe_from_words<FpTag>(v);
}
template <class F> static Affine<F> generator();
template <> Affine<Fp> generator<Fp>() {
    const uint64_t x[4] = {1, 0, 0, 0}, y[4] = {2, 0, 0, 0};
    return Affine<Fp>{fe_to_mont(fp_from_u64(x)), fe_to_mont(fp_from_u64(y))};
}
template <> Affine<Fp2> generator<Fp2>() {   // the BN254 G2 generator (EIP-197)
    const uint64_t x0[4] = {0x46debd5cd992f6edull, 0x674322d4f75edaddull, 0x426a00665e5c4479ull, 0x1800deef121f1e76ull};
    const uint64_t x1[4] = {0x97e485b7aef312c2ull, 0xf1aa493335a9e712ull, 0x7260bfb731fb5d25ull, 0x198e9393920d483aull};
    const uint64_t y0[4] = {0x4ce6cc0166fa7daaull, 0xe3d1e7690c43d37bull, 0x4aab71808dcb408full, 0x12c85ea5db8c6debull};
    const uint64_t y1[4] = {0x55acdadcd122975bull, 0xbc4b313370b38ef3ull, 0xec9e99ad690c3395ull, 0x090689d0585ff075ull};
    return Affine<Fp2>{Fp2{fe_to_mont(fp_from_u64(x0)), fe_to_mont(fp_from_u64(x1))}, Fp2{fe_to_mont(fp_from_u64(y0)), fe_to_mont(fp_from_u64(y1))}};
}

template <class F> static void run(uint32_t logn, int reps) {
    const uint32_t n = 1u << logn, BL = std::min<uint32_t>(RED_BL, logn);
    Xyzz<F> *x, *out;
    CK(hipMalloc(&x, (size_t)n * sizeof(Xyzz<F>)));
    CK(hipMalloc(&out, 64 * sizeof(Xyzz<F>)));
    hipLaunchKernelGGL((fill_kernel<F>), dim3((n + 255) / 256), dim3(256), 0, 0, x, n, generator<F>());
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto timed = [&](auto launch) {
        std::vector<float> ts;
        for (int r = 0; r < reps + 3; r++) {
            CK(hipEventRecord(e0, 0));
            launch();
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (r >= 3) ts.push_back(ms * 1e3f);
        }
        std::sort(ts.begin(), ts.end());
        return ts[ts.size() / 2];
    };
    // per-step times from wall-clock stamps (100 MHz) thread 0 of every workgroup takes at the step boundaries of ONE launch:
    // median over the workgroups of each step's duration, median over the launches
    const uint32_t nblk = n >> BL;
    uint64_t *stamps;
    CK(hipMalloc(&stamps, (size_t)nblk * (BL + 1) * sizeof(uint64_t)));
    std::vector<uint64_t> hs((size_t)nblk * (BL + 1));
    std::vector<std::vector<float>> per_step(BL + 1);
    std::vector<float> spans;
    const float whole = timed([&] { hipLaunchKernelGGL((msm_reduce_block_kernel<F, RED_BLOCK_NT>), dim3(nblk), dim3(RED_BLOCK_NT), 0, 0, x, BL, (uint64_t *)nullptr); });
    for (int r = 0; r < reps; r++) {
        hipLaunchKernelGGL((msm_reduce_block_kernel<F, RED_BLOCK_NT>), dim3(nblk), dim3(RED_BLOCK_NT), 0, 0, x, BL, stamps);
        CK(hipMemcpy(hs.data(), stamps, hs.size() * sizeof(uint64_t), hipMemcpyDeviceToHost));
        uint64_t first = ~0ull, last = 0;
        for (uint32_t s = 0; s < BL; s++) {
            std::vector<float> d(nblk);
            for (uint32_t b = 0; b < nblk; b++) d[b] = (float)(hs[(size_t)b * (BL + 1) + s + 1] - hs[(size_t)b * (BL + 1) + s]) * 0.01f;
            std::sort(d.begin(), d.end());
            per_step[s].push_back(d[nblk / 2]);
        }
        for (uint32_t b = 0; b < nblk; b++) { first = std::min(first, hs[(size_t)b * (BL + 1)]); last = std::max(last, hs[(size_t)b * (BL + 1) + BL]); }
        spans.push_back((float)(last - first) * 0.01f);
    }
    std::sort(spans.begin(), spans.end());
    printf("block kernel, 2^%u buckets, BL = %u, %d threads, %u workgroups: %.1f us by events (median of %d); first start to last end inside the kernel %.1f us\n",
           logn, BL, RED_BLOCK_NT, nblk, whole, reps, spans[spans.size() / 2]);
    float sum = 0;
    for (uint32_t s = 0; s < BL; s++) {
        std::sort(per_step[s].begin(), per_step[s].end());
        const float us = per_step[s][per_step[s].size() / 2];
        const uint32_t sh = BL - 1 - s, tasks = (s + 1) << sh;
        sum += us;
        printf("  step %2u  %7.1f us   %4u additions per workgroup%s\n", s, us, tasks,
               (1u << sh) >= (uint32_t)RED_BLOCK_NT ? ", one lane each, no barrier" : tasks > RED_BLOCK_NT / 4 ? ", rounds of one lane each + a remainder on teams" : ", four lanes each");
    }
    printf("  sum      %7.1f us\n", sum);
    CK(hipFree(stamps));
    constexpr int WNT = RED_WINDOW_NT;
    if (logn > BL) {
        const uint32_t levels = logn, GL = (levels - BL) / 2;
        const float a = timed([&] { hipLaunchKernelGGL((msm_reduce_window_kernel<F, WNT>), dim3(n >> (BL + GL)), dim3(WNT), 0, 0, x, (Xyzz<F> *)nullptr, 1u << (BL + GL), BL, BL + GL); });
        const float b = timed([&] { hipLaunchKernelGGL((msm_reduce_window_kernel<F, WNT>), dim3(1), dim3(WNT), 0, 0, x, out, n, BL + GL, levels); });
        printf("window kernel (%d threads): %u groups of %u blocks %.1f us, final %.1f us\n", WNT, n >> (BL + GL), 1u << GL, a, b);
        const uint32_t wl = 15;   // sixteen 2^15-bucket windows, as the unbound mode at c = 16
        if (logn >= wl + 1) {
            const float c = timed([&] { hipLaunchKernelGGL((msm_reduce_window_kernel<F, WNT>), dim3(n >> wl), dim3(WNT), 0, 0, x, out, 1u << wl, BL, wl); });
            printf("window kernel, %u windows of 2^%u buckets: %.1f us\n", n >> wl, wl, c);
        }
    }
    CK(hipFree(x));
    CK(hipFree(out));
}

// ---- instruction-level parallelism at one wavefront per SIMD: K independent fe_mul chains per lane, written side by side
template <int K> __global__ __launch_bounds__(64) void ilp_kernel(Fp *out, int iters, uint32_t sink) {
    Fp x[K], y[K];
    for (int k = 0; k < K; k++) {
        x[k] = Fp::one(); y[k] = Fp::one();
        x[k].l[0] = (x[k].l[0] + threadIdx.x + k) & 0x1fffffffu;
        y[k].l[1] = (y[k].l[1] + blockIdx.x + 3 * k) & 0x1fffffffu;
    }
#pragma unroll 1
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < K; k++) x[k] = fe_mul(x[k], y[k]);
#pragma unroll
        for (int k = 0; k < K; k++) y[k] = fe_mul(y[k], x[k]);
    }
    Fp acc = x[0];
    for (int k = 1; k < K; k++) acc = fe_add(acc, x[k]);
    for (int k = 0; k < K; k++) acc = fe_add(acc, y[k]);
    if (blockIdx.x * 64 + threadIdx.x == sink) out[0] = acc;
}
// the same two chains with the columns of both products interleaved by hand (one instruction stream, two accumulators)
__device__ __forceinline__ void fe_mul2(const Fp &a, const Fp &b, const Fp &c, const Fp &d, Fp &r0, Fp &r1) {
    typedef FieldConst<FpTag> C;
    uint32_t q0[NL], q1[NL];
    uint64_t acc0 = 0, acc1 = 0;
    Fp o0, o1;
#pragma unroll
    for (int k = 0; k < NL; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) { acc0 += (uint64_t)a.l[i] * b.l[k - i]; acc1 += (uint64_t)c.l[i] * d.l[k - i]; }
#pragma unroll
        for (int i = 0; i < k; i++) { acc0 += (uint64_t)q0[i] * C::mod(k - i); acc1 += (uint64_t)q1[i] * C::mod(k - i); }
        q0[k] = ((uint32_t)acc0 * C::inv) & LMASK; q1[k] = ((uint32_t)acc1 * C::inv) & LMASK;
        acc0 += (uint64_t)q0[k] * C::mod(0); acc1 += (uint64_t)q1[k] * C::mod(0);
        acc0 >>= LB; acc1 >>= LB;
    }
#pragma unroll
    for (int k = NL; k < 2 * NL - 1; k++) {
#pragma unroll
        for (int i = k - NL + 1; i < NL; i++) { acc0 += (uint64_t)a.l[i] * b.l[k - i]; acc1 += (uint64_t)c.l[i] * d.l[k - i]; }
#pragma unroll
        for (int i = k - NL + 1; i < NL; i++) { acc0 += (uint64_t)q0[i] * C::mod(k - i); acc1 += (uint64_t)q1[i] * C::mod(k - i); }
        o0.l[k - NL] = (uint32_t)acc0 & LMASK; o1.l[k - NL] = (uint32_t)acc1 & LMASK;
        acc0 >>= LB; acc1 >>= LB;
    }
    o0.l[NL - 1] = (uint32_t)acc0; o1.l[NL - 1] = (uint32_t)acc1;
    r0 = o0; r1 = o1;
}
__global__ __launch_bounds__(64) void ilp2_manual_kernel(Fp *out, int iters, uint32_t sink) {
    Fp x0 = Fp::one(), y0 = Fp::one(), x1 = Fp::one(), y1 = Fp::one();
    x0.l[0] = (x0.l[0] + threadIdx.x) & 0x1fffffffu; x1.l[0] = (x1.l[0] + threadIdx.x + 1) & 0x1fffffffu;
    y0.l[1] = (y0.l[1] + blockIdx.x) & 0x1fffffffu; y1.l[1] = (y1.l[1] + blockIdx.x + 3) & 0x1fffffffu;
#pragma unroll 1
    for (int i = 0; i < iters; i++) {
        fe_mul2(x0, y0, x1, y1, x0, x1);
        fe_mul2(y0, x0, y1, x1, y0, y1);
    }
    if (blockIdx.x * 64 + threadIdx.x == sink) out[0] = fe_add(fe_add(x0, x1), fe_add(y0, y1));
}
static void run_ilp() {
    Fp *out;
    CK(hipMalloc(&out, sizeof(Fp)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int iters = 2000;
    auto timed = [&](auto launch) {
        launch();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        launch();
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        return ms * 1e3f;
    };
    for (unsigned wgs : {1024u, 2048u, 4096u}) {   // one, two, four wavefronts per SIMD
        const float t1 = timed([&] { hipLaunchKernelGGL((ilp_kernel<1>), dim3(wgs), dim3(64), 0, 0, out, iters, 0xffffffffu); });
        const float t2 = timed([&] { hipLaunchKernelGGL((ilp_kernel<2>), dim3(wgs), dim3(64), 0, 0, out, iters, 0xffffffffu); });
        const float t3 = timed([&] { hipLaunchKernelGGL((ilp_kernel<3>), dim3(wgs), dim3(64), 0, 0, out, iters, 0xffffffffu); });
        const float tm = timed([&] { hipLaunchKernelGGL(ilp2_manual_kernel, dim3(wgs), dim3(64), 0, 0, out, iters, 0xffffffffu); });
        printf("%u wavefronts per SIMD, ns per fe_mul per wavefront (latency of one dependent step / K):  K=1 %.1f   K=2 %.1f   K=3 %.1f   K=2 by hand %.1f\n",
               wgs / 1024, t1 * 1e3 / (2 * iters), t2 * 1e3 / (4 * iters), t3 * 1e3 / (6 * iters), tm * 1e3 / (4 * iters));
    }
    CK(hipFree(out));
}

// ---- what the quad permutations deliver on this chip, lane by lane (lanes 0..7 of one wavefront)
template <int P0, int P1, int P2, int P3> __device__ void dpp_row(uint32_t *out, int row) {
    const uint32_t v = 100u + threadIdx.x;
    const uint32_t r = QuadDpp{}.get<P0, P1, P2, P3>(v);
    const uint32_t folded = 1000u - QuadDpp{}.get<P0, P1, P2, P3>(v);    // lets the compiler fold the permutation into the subtraction
    if (threadIdx.x < 8) { out[row * 16 + threadIdx.x] = r; out[row * 16 + 8 + threadIdx.x] = folded; }
}
__global__ void dpp_kernel(uint32_t *out) {
    dpp_row<1, 3, 1, 3>(out, 0); dpp_row<0, 2, 0, 2>(out, 1); dpp_row<1, 1, 1, 3>(out, 2); dpp_row<0, 0, 0, 2>(out, 3);
    dpp_row<2, 2, 2, 2>(out, 4); dpp_row<3, 3, 3, 3>(out, 5); dpp_row<1, 0, 3, 2>(out, 6); dpp_row<0, 0, 0, 0>(out, 7);
}
static void run_dpp() {
    uint32_t *out, h[128];
    CK(hipMalloc(&out, sizeof(h)));
    hipLaunchKernelGGL(dpp_kernel, dim3(1), dim3(64), 0, 0, out);
    CK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
    const char *names[8] = {"1,3,1,3", "0,2,0,2", "1,1,1,3", "0,0,0,2", "2,2,2,2", "3,3,3,3", "1,0,3,2", "0,0,0,0"};
    for (int r = 0; r < 8; r++) {
        printf("quad_perm:[%s]  lanes 0..7 read lane:", names[r]);
        for (int i = 0; i < 8; i++) printf(" %d", (int)h[r * 16 + i] - 100);
        printf("   folded:");
        for (int i = 0; i < 8; i++) printf(" %d", 900 - (int)h[r * 16 + 8 + i]);
        printf("\n");
    }
}
// the same exchange through ds_bpermute_b32 (any lane of the wavefront; the LDS crossbar, no LDS memory)
struct QuadBpermute {
    template <int P0, int P1, int P2, int P3> __device__ __forceinline__ uint32_t word(uint32_t v) const {
        const uint32_t lane = threadIdx.x & 63u, r = lane & 3u;
        const uint32_t src = (lane & ~3u) + (r == 0 ? P0 : r == 1 ? P1 : r == 2 ? P2 : P3);
        return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src << 2), (int)v);
    }
    template <int P0, int P1, int P2, int P3> __device__ __forceinline__ uint32_t get(const uint32_t &x) const { return word<P0, P1, P2, P3>(x); }
    template <int P0, int P1, int P2, int P3, class Tag> __device__ __forceinline__ Fe<Tag> get(const Fe<Tag> &x) const {
        Fe<Tag> r;
        for (int i = 0; i < NL; i++) r.l[i] = word<P0, P1, P2, P3>(x.l[i]);
        return r;
    }
    template <int P0, int P1, int P2, int P3> __device__ __forceinline__ Fp2 get(const Fp2 &x) const {
        return Fp2{get<P0, P1, P2, P3>(x.c0), get<P0, P1, P2, P3>(x.c1)};
    }
};
// DPP by inline assembly: destination never the source (early clobber), nothing folded into a neighbour, wait states by hand
template <int CTRL, int NOPS> struct QuadAsm {
    __device__ __forceinline__ uint32_t word(uint32_t v) const {
        uint32_t r;
        if (NOPS)
            asm volatile("s_nop 4\n\tv_mov_b32_dpp %0, %1 quad_perm:[%2,%3,%4,%5] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 1"
                         : "=&v"(r) : "v"(v), "n"(CTRL & 3), "n"((CTRL >> 2) & 3), "n"((CTRL >> 4) & 3), "n"((CTRL >> 6) & 3));
        else
            asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[%2,%3,%4,%5] row_mask:0xf bank_mask:0xf bound_ctrl:1"
                         : "=&v"(r) : "v"(v), "n"(CTRL & 3), "n"((CTRL >> 2) & 3), "n"((CTRL >> 4) & 3), "n"((CTRL >> 6) & 3));
        return r;
    }
};
template <int NOPS> struct QuadDppAsm {
    template <int P0, int P1, int P2, int P3> __device__ __forceinline__ uint32_t get(const uint32_t &x) const {
        return QuadAsm<P0 | (P1 << 2) | (P2 << 4) | (P3 << 6), NOPS>{}.word(x);
    }
    template <int P0, int P1, int P2, int P3, class Tag> __device__ __forceinline__ Fe<Tag> get(const Fe<Tag> &x) const {
        Fe<Tag> r;
        for (int i = 0; i < NL; i++) r.l[i] = get<P0, P1, P2, P3>(x.l[i]);
        return r;
    }
    template <int P0, int P1, int P2, int P3> __device__ __forceinline__ Fp2 get(const Fp2 &x) const {
        return Fp2{get<P0, P1, P2, P3>(x.c0), get<P0, P1, P2, P3>(x.c1)};
    }
};
// the plain builtin: what curve.h's QuadDpp was first written with (wrong results on the chip: kept here as the record of it)
struct QuadDppBuiltin {
    template <int P0, int P1, int P2, int P3> __device__ __forceinline__ uint32_t get(const uint32_t &x) const {
        return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, P0 | (P1 << 2) | (P2 << 4) | (P3 << 6), 0xf, 0xf, true);
    }
    template <int P0, int P1, int P2, int P3, class Tag> __device__ __forceinline__ Fe<Tag> get(const Fe<Tag> &x) const {
        Fe<Tag> r;
        for (int i = 0; i < NL; i++) r.l[i] = get<P0, P1, P2, P3>(x.l[i]);
        return r;
    }
    template <int P0, int P1, int P2, int P3> __device__ __forceinline__ Fp2 get(const Fp2 &x) const {
        return Fp2{get<P0, P1, P2, P3>(x.c0), get<P0, P1, P2, P3>(x.c1)};
    }
};
// the builtin, its operand and result pinned by empty asm statements (nothing can be folded into or out of the move)
struct QuadDppPinned {
    template <int P0, int P1, int P2, int P3> __device__ __forceinline__ uint32_t get(const uint32_t &x) const {
        uint32_t v = x;
        asm volatile("" : "+v"(v));
        uint32_t r = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, P0 | (P1 << 2) | (P2 << 4) | (P3 << 6), 0xf, 0xf, false);
        asm volatile("" : "+v"(r));
        return r;
    }
    template <int P0, int P1, int P2, int P3, class Tag> __device__ __forceinline__ Fe<Tag> get(const Fe<Tag> &x) const {
        Fe<Tag> r;
        for (int i = 0; i < NL; i++) r.l[i] = get<P0, P1, P2, P3>(x.l[i]);
        return r;
    }
    template <int P0, int P1, int P2, int P3> __device__ __forceinline__ Fp2 get(const Fp2 &x) const {
        return Fp2{get<P0, P1, P2, P3>(x.c0), get<P0, P1, P2, P3>(x.c1)};
    }
};
template <class F, class Ex> __global__ void check_ex_kernel(const Xyzz<F> *b, Xyzz<F> *r4, uint32_t n) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if ((t >> 2) < n) team4_add(t & 3u, r4 + (t >> 2), b + (t >> 2), Ex{});
}

// ---- team additions against the one-lane addition, on the device: acc[i] += q[i] three ways, compared as affine points
template <class F> __global__ void check_kernel(const Xyzz<F> *a, const Xyzz<F> *b, Xyzz<F> *r1, Xyzz<F> *r2, Xyzz<F> *r4, uint32_t n) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) {
        Xyzz<F> x = a[t];
        xyzz_add(x, b[t]);
        r1[t] = x;
    }
    if ((t >> 1) < n) team2_add(t & 1u, r2 + (t >> 1), b + (t >> 1), QuadDpp{});
    if ((t >> 2) < n) team4_add(t & 3u, r4 + (t >> 2), b + (t >> 2), QuadDpp{});
}
template <class F> __global__ void compare_kernel(const Xyzz<F> *r1, const Xyzz<F> *r2, const Xyzz<F> *r4, uint32_t n, uint32_t *bad) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const Affine<F> p1 = xyzz_to_affine(r1[t]), p2 = xyzz_to_affine(r2[t]), p4 = xyzz_to_affine(r4[t]);
    auto same = [](const Affine<F> &x, const Affine<F> &y) { return x.is_inf() ? y.is_inf() : (!y.is_inf() && x.x.equals(y.x) && x.y.equals(y.y)); };
    if (!same(p1, p2)) atomicAdd(&bad[0], 1u);
    if (!same(p1, p4)) atomicAdd(&bad[1], 1u);
    if (p1.is_inf()) atomicAdd(&bad[2], 1u);
}
template <class F> static void run_check() {
    const uint32_t n = 4096;
    Xyzz<F> *a, *b, *r1, *r2, *r4;
    uint32_t *bad;
    for (Xyzz<F> **p : {&a, &b, &r1, &r2, &r4}) CK(hipMalloc(p, n * sizeof(Xyzz<F>)));
    CK(hipMalloc(&bad, 16));
    hipLaunchKernelGGL((fill_kernel<F>), dim3(n / 256), dim3(256), 0, 0, a, n, generator<F>());
    hipLaunchKernelGGL((fill_kernel<F>), dim3(n / 256), dim3(256), 0, 0, b, n, generator<F>());
    CK(hipDeviceSynchronize());
    // b: every fourth the same point as a (doubling), every fourth + 1 its negative, + 2 infinity, + 3 a different point; and a few a's infinity
    std::vector<Xyzz<F>> ha(n), hb(n);
    CK(hipMemcpy(ha.data(), a, n * sizeof(Xyzz<F>), hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < n; i++) {
        hb[i] = ha[(i * 7 + 3) % n];
        if (i % 8 == 0) hb[i] = ha[i];
        if (i % 8 == 1) { hb[i] = ha[i]; hb[i] = xyzz_neg(hb[i]); }
        if (i % 8 == 2) hb[i] = Xyzz<F>::inf();
        if (i % 16 == 3 || i % 16 == 2) ha[i] = Xyzz<F>::inf();
    }
    CK(hipMemcpy(a, ha.data(), n * sizeof(Xyzz<F>), hipMemcpyHostToDevice));
    CK(hipMemcpy(b, hb.data(), n * sizeof(Xyzz<F>), hipMemcpyHostToDevice));
    CK(hipMemcpy(r2, a, n * sizeof(Xyzz<F>), hipMemcpyDeviceToDevice));
    CK(hipMemcpy(r4, a, n * sizeof(Xyzz<F>), hipMemcpyDeviceToDevice));
    CK(hipMemset(bad, 0, 16));
    hipLaunchKernelGGL((check_kernel<F>), dim3(4 * n / 256), dim3(256), 0, 0, a, b, r1, r2, r4, n);
    hipLaunchKernelGGL((compare_kernel<F>), dim3(n / 64), dim3(64), 0, 0, r1, r2, r4, n, bad);
    uint32_t h[4];
    CK(hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost));
    auto variant = [&](const char *name, auto launch) {   // four lanes again with another exchange
        uint32_t hb4[4];
        CK(hipMemcpy(r4, a, n * sizeof(Xyzz<F>), hipMemcpyDeviceToDevice));
        CK(hipMemset(bad, 0, 16));
        launch();
        hipLaunchKernelGGL((compare_kernel<F>), dim3(n / 64), dim3(64), 0, 0, r1, r2, r4, n, bad);
        CK(hipMemcpy(hb4, bad, 16, hipMemcpyDeviceToHost));
        printf("four lanes, %s: %u wrong\n", name, hb4[1]);
    };
    variant("ds_bpermute", [&] { hipLaunchKernelGGL((check_ex_kernel<F, QuadBpermute>), dim3(4 * n / 256), dim3(256), 0, 0, b, r4, n); });
    variant("__builtin_amdgcn_mov_dpp", [&] { hipLaunchKernelGGL((check_ex_kernel<F, QuadDppBuiltin>), dim3(4 * n / 256), dim3(256), 0, 0, b, r4, n); });
    variant("asm dpp, no wait states", [&] { hipLaunchKernelGGL((check_ex_kernel<F, QuadDppAsm<0>>), dim3(4 * n / 256), dim3(256), 0, 0, b, r4, n); });
    variant("asm dpp, s_nop 4 before and s_nop 1 after every move", [&] { hipLaunchKernelGGL((check_ex_kernel<F, QuadDppAsm<1>>), dim3(4 * n / 256), dim3(256), 0, 0, b, r4, n); });
    variant("pinned update_dpp", [&] { hipLaunchKernelGGL((check_ex_kernel<F, QuadDppPinned>), dim3(4 * n / 256), dim3(256), 0, 0, b, r4, n); });
    printf("the library's exchange (curve.h QuadDpp) vs one lane, %u pairs (an eighth equal, an eighth opposite, infinities on both sides): two lanes %u wrong, four lanes %u wrong (%u sums are infinity)\n",
           n, h[0], h[1], h[2]);
    if (h[0] || h[1]) exit(1);
}

int main(int argc, char **argv) {
    if (argc > 1 && !strcmp(argv[1], "ilp")) { run_ilp(); return 0; }
    if (argc > 1 && !strcmp(argv[1], "dpp")) { run_dpp(); return 0; }
    if (argc > 1 && !strcmp(argv[1], "check")) { run_check<Fp>(); run_check<Fp2>(); return 0; }
    const bool g2 = argc > 1 && !strcmp(argv[1], "g2");
    const uint32_t logn = argc > 2 ? (uint32_t)atoi(argv[2]) : 19;
    const int reps = argc > 3 ? atoi(argv[3]) : 20;
    if (g2) run<Fp2>(logn, reps); else run<Fp>(logn, reps);
    return 0;
}
