// tools/ubench.hip -- instruction-rate microbenchmarks for the integer/FP64 ops a 254-bit
// modular multiplication can be built from on gfx950, plus the modmul/s ceiling of field.h.
// Build: hipcc -O3 --offload-arch=gfx950 -Iinteractive-zkp-study_amd/csrc tools/ubench.hip -o gpurun_out/ubench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "curve.h"
using namespace zk;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

enum { OP_ADD32, OP_MAD64, OP_MULLO, OP_MULHI, OP_MAD24, OP_MULHI24, OP_ADD64, OP_ADDC, OP_FMA64, OP_FMA32, OP_NOPS };
static const char *NAMES[] = {"v_add_u32", "v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u32_u24", "v_mul_hi_u32_u24",
                              "v_lshl_add_u64", "v_addc_co_u32", "v_fma_f64", "v_fma_f32"};

template <int OP> __global__ __launch_bounds__(256) void k_rate(uint32_t *out, int iters) {
    uint32_t a = threadIdx.x * 2654435761u + 12345u, b = a ^ 0x9e3779b9u;
    uint32_t r0 = a, r1 = b, r2 = a + 7, r3 = b + 9;
    uint64_t q0 = a, q1 = b, q2 = a + 3, q3 = b + 5;
    double d0 = a, d1 = b, d2 = 1.5, d3 = 2.5, dm = 1.0000001, da = 0.5;
    float f0 = a, f1 = b, f2 = 1.5f, f3 = 2.5f, fm = 1.0000001f, fa = 0.5f;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (OP == OP_ADD32) {
                asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4"
                             : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a));
            } else if (OP == OP_MAD64) {
                asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n"
                             "v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3"
                             : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(a), "v"(b) : "vcc");
            } else if (OP == OP_MULLO) {
                asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4"
                             : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a));
            } else if (OP == OP_MULHI) {
                asm volatile("v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_hi_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4"
                             : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a));
            } else if (OP == OP_MAD24) {
                asm volatile("v_mad_u32_u24 %0, %0, %4, %5\n v_mad_u32_u24 %1, %1, %4, %5\n v_mad_u32_u24 %2, %2, %4, %5\n v_mad_u32_u24 %3, %3, %4, %5"
                             : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b));
            } else if (OP == OP_MULHI24) {
                asm volatile("v_mul_hi_u32_u24 %0, %0, %4\n v_mul_hi_u32_u24 %1, %1, %4\n v_mul_hi_u32_u24 %2, %2, %4\n v_mul_hi_u32_u24 %3, %3, %4"
                             : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a));
            } else if (OP == OP_ADD64) {
                asm volatile("v_lshl_add_u64 %0, %0, 0, %4\n v_lshl_add_u64 %1, %1, 0, %4\n v_lshl_add_u64 %2, %2, 0, %4\n v_lshl_add_u64 %3, %3, 0, %4"
                             : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(q0));
            } else if (OP == OP_ADDC) {
                asm volatile("v_add_co_u32 %0, vcc, %0, %4\n v_addc_co_u32 %1, vcc, %1, %4, vcc\n v_addc_co_u32 %2, vcc, %2, %4, vcc\n v_addc_co_u32 %3, vcc, %3, %4, vcc"
                             : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a) : "vcc");
            } else if (OP == OP_FMA64) {
                asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5"
                             : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(dm), "v"(da));
            } else if (OP == OP_FMA32) {
                asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5"
                             : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(fm), "v"(fa));
            }
        }
    }
    uint32_t acc = r0 ^ r1 ^ r2 ^ r3 ^ (uint32_t)(q0 ^ q1 ^ q2 ^ q3) ^ (uint32_t)((q0 ^ q1 ^ q2 ^ q3) >> 32) ^
                   (uint32_t)(d0 + d1 + d2 + d3) ^ (uint32_t)(f0 + f1 + f2 + f3);
    if (acc == 0x12345678u) out[0] = acc;  // keep live
}

template <class T> __global__ __launch_bounds__(256) void k_modmul(const Fe<T> *in, Fe<T> *out, int iters) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    Fe<T> x = in[i & 1023], y = in[(i + 1) & 1023];
    for (int k = 0; k < iters; k++) {
        x = fe_mul(x, y);
        y = fe_mul(y, x);
    }
    out[i & 1023] = fe_add(x, y);
}

__global__ __launch_bounds__(256) void k_madd(const G1Affine *pts, G1Xyzz *out, int cnt, int npts) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    G1Xyzz acc = G1Xyzz::inf();
    for (int k = 0; k < cnt; k++) xyzz_add_affine(acc, pts[(i * 7 + k * 13) % npts]);
    out[i] = acc;
}

// device-vs-host self check of field.h / curve.h on random operands
__global__ void k_selfcheck(const Fp *a, const Fp *b, Fp *mul, Fp *add, Fp *sub, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    mul[i] = fe_mul(a[i], b[i]);
    add[i] = fe_add(a[i], b[i]);
    sub[i] = fe_sub_k<2>(a[i], b[i]);
}
__global__ void k_selfcheck_g1(const G1Affine *g, const uint32_t *k, G1Xyzz *out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = xyzz_scalar_mul(g[0], k + 8 * i);
}

template <class K> static double time_kernel(K launch, int reps) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; r++) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

template <int OP> static void run_rate(uint32_t *d_out, int blocks) {
    const int iters = 4096;
    double ms = time_kernel([&] { hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(256), 0, 0, d_out, iters); }, 5);
    double ops = (double)blocks * 256 * iters * 32;  // lane-ops
    double rate = ops / (ms * 1e-3);
    // cycles per wave-instruction per SIMD at 2.4 GHz: 1024 SIMDs * 2.4e9 / (rate/64)
    double cyc = 1024.0 * 2.4e9 / (rate / 64.0);
    printf("%-18s %8.3f ms  %8.2f Tlane-op/s  ~%5.2f cyc/wave-instr/SIMD (at 2.4 GHz)\n", NAMES[OP], ms, rate / 1e12, cyc);
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s  CUs=%d  clock=%d MHz\n", prop.name, prop.multiProcessorCount, prop.clockRate / 1000);
    uint32_t *d_out;
    CK(hipMalloc(&d_out, 4096));
    int blocks = prop.multiProcessorCount * 8;
    run_rate<OP_ADD32>(d_out, blocks);
    run_rate<OP_MAD64>(d_out, blocks);
    run_rate<OP_MULLO>(d_out, blocks);
    run_rate<OP_MULHI>(d_out, blocks);
    run_rate<OP_MAD24>(d_out, blocks);
    run_rate<OP_MULHI24>(d_out, blocks);
    run_rate<OP_ADD64>(d_out, blocks);
    run_rate<OP_ADDC>(d_out, blocks);
    run_rate<OP_FMA64>(d_out, blocks);
    run_rate<OP_FMA32>(d_out, blocks);

    // ---- self check
    const int N = 4096;
    std::vector<Fp> ha(N), hb(N), hm(N), hs(N), hd(N);
    srand(1);
    for (int i = 0; i < N; i++) {
        for (int j = 0; j < NL; j++) { ha[i].l[j] = (rand() * 65537u + rand()) & LMASK; hb[i].l[j] = (rand() * 65537u + rand()) & LMASK; }
        ha[i].l[NL - 1] &= 0x1fffff; hb[i].l[NL - 1] &= 0x1fffff;  // < p
    }
    Fp *da, *db, *dm, *ds, *dd;
    CK(hipMalloc(&da, N * sizeof(Fp))); CK(hipMalloc(&db, N * sizeof(Fp))); CK(hipMalloc(&dm, N * sizeof(Fp))); CK(hipMalloc(&ds, N * sizeof(Fp))); CK(hipMalloc(&dd, N * sizeof(Fp)));
    CK(hipMemcpy(da, ha.data(), N * sizeof(Fp), hipMemcpyHostToDevice)); CK(hipMemcpy(db, hb.data(), N * sizeof(Fp), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_selfcheck, dim3(N / 256), dim3(256), 0, 0, da, db, dm, ds, dd, N);
    CK(hipMemcpy(hm.data(), dm, N * sizeof(Fp), hipMemcpyDeviceToHost)); CK(hipMemcpy(hs.data(), ds, N * sizeof(Fp), hipMemcpyDeviceToHost)); CK(hipMemcpy(hd.data(), dd, N * sizeof(Fp), hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < N; i++) {
        if (!hm[i].equals(fe_mul(ha[i], hb[i]))) bad++;
        if (!hs[i].equals(fe_add(ha[i], hb[i]))) bad++;
        if (!hd[i].equals(fe_sub_k<2>(ha[i], hb[i]))) bad++;
    }
    printf("selfcheck field (device vs host, %d cases): %s (%d mismatches)\n", 3 * N, bad ? "FAIL" : "ok", bad);
    {
        const int M = 256;
        G1Affine g{Fp{ZK_G1_X_M}, Fp{ZK_G1_Y_M}};
        std::vector<uint32_t> hk(8 * M);
        for (auto &v : hk) v = rand() * 65537u + rand();
        for (int i = 0; i < M; i++) hk[8 * i + 7] &= 0x1fffffff;
        G1Affine *dg; uint32_t *dk; G1Xyzz *dout;
        CK(hipMalloc(&dg, sizeof(g))); CK(hipMalloc(&dk, hk.size() * 4)); CK(hipMalloc(&dout, M * sizeof(G1Xyzz)));
        CK(hipMemcpy(dg, &g, sizeof(g), hipMemcpyHostToDevice)); CK(hipMemcpy(dk, hk.data(), hk.size() * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_selfcheck_g1, dim3(M / 64), dim3(64), 0, 0, dg, dk, dout, M);
        std::vector<G1Xyzz> ho(M);
        CK(hipMemcpy(ho.data(), dout, M * sizeof(G1Xyzz), hipMemcpyDeviceToHost));
        int bad2 = 0;
        for (int i = 0; i < M; i++) {
            G1Xyzz e = xyzz_scalar_mul(g, &hk[8 * i]);
            if (!(e.x.equals(ho[i].x) && e.y.equals(ho[i].y) && e.zz.equals(ho[i].zz) && e.zzz.equals(ho[i].zzz))) bad2++;
        }
        printf("selfcheck G1 scalar-mul (device vs host, %d cases): %s\n", M, bad2 ? "FAIL" : "ok");
    }

    // ---- modmul throughput (field.h as shipped)
    {
        int bl = prop.multiProcessorCount * 8, iters = 512;
        double ms = time_kernel([&] { hipLaunchKernelGGL(k_modmul<FpTag>, dim3(bl), dim3(256), 0, 0, da, dm, iters); }, 5);
        double muls = (double)bl * 256 * iters * 2;
        printf("fe_mul<Fp> chain: %.3f ms, %.2f G modmul/s\n", ms, muls / ms / 1e6);
    }
    // ---- mixed-add throughput
    {
        const int NP = 1 << 16;
        std::vector<G1Affine> hp(NP);
        G1Affine g{Fp{ZK_G1_X_M}, Fp{ZK_G1_Y_M}};
        G1Xyzz acc = G1Xyzz::from_affine(g);
        for (int i = 0; i < NP; i++) {  // cheap distinct points: P, 2P, 3P.. in XYZZ -> affine on host is slow; use doubling chain subset
            if (i < 64) { hp[i] = xyzz_to_affine(acc); xyzz_add_affine(acc, g); } else hp[i] = hp[i & 63];
        }
        G1Affine *dp; G1Xyzz *dout;
        int bl = prop.multiProcessorCount * 8, cnt = 64;
        CK(hipMalloc(&dp, NP * sizeof(G1Affine))); CK(hipMalloc(&dout, (size_t)bl * 256 * sizeof(G1Xyzz)));
        CK(hipMemcpy(dp, hp.data(), NP * sizeof(G1Affine), hipMemcpyHostToDevice));
        double ms = time_kernel([&] { hipLaunchKernelGGL(k_madd, dim3(bl), dim3(256), 0, 0, dp, dout, cnt, NP); }, 3);
        double adds = (double)bl * 256 * cnt;
        printf("xyzz_add_affine chain: %.3f ms, %.2f G madd/s (~%.1f G modmul/s at 10 mul/add)\n", ms, adds / ms / 1e6, adds * 10 / ms / 1e6);
    }
    return 0;
}
