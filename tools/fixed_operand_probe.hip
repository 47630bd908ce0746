// tools/fixed_operand_probe.hip -- round-4 experiment: fixed-operand ("Shoup" / precomputed-quotient Barrett)
// multiplication for the NTT, whose every product has a table operand (twiddle, coset power, n^-1).
//
//   Montgomery (field.h fe_mul):   x * (w R) / R            81 + 81 multiply-adds, 9 v_mul_lo for the quotient digits, which
//                                                           also chain the nine reduction columns one behind the other;
//   fixed operand (here):          x * w - q * r,  q = floor(x * w' / 2^261),  w' = floor(w * 2^261 / r) stored beside w:
//                                  q from the UPPER columns of x * w' (53 multiply-adds incl. two guard columns), then the LOWER
//                                  nine columns of x * w + q * (2^261 - r) (45 + 45): 143 multiply-adds, no quotient-digit chain,
//                                  result in [0, 4r) -- one conditional subtraction brings it under the 2r the butterflies assume.
//
// The probe runs the NTT's butterfly (sum = a + b, dif = (a - b) * w, both < 2r) in a dependent loop with either product, checks that
// both give the same residues, and reports butterflies per second at the pass kernel's launch bounds, plus registers (hipcc -save-temps
// / -Rpass-analysis).  It does not touch libzkhip.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinteractive-zkp-study_amd/csrc tools/fixed_operand_probe.hip -o gpurun_out/fixed_operand_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "field.h"
#include "host_field.h"
using namespace zk;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

// limb i of 2^261 - r
__host__ __device__ constexpr uint32_t rbar_limb(int i) {
    constexpr uint32_t m[NL] = ZK_FR_MOD;
    uint32_t carry = 1, out = 0;            // two's complement over 9 x 29 bits: ~m + 1
    for (int k = 0; k <= i; k++) {
        const uint32_t t = ((~m[k]) & LMASK) + carry;
        out = t & LMASK;
        carry = t >> LB;
    }
    return out;
}

// x: limbs < 2^31 (a lazy difference is fine), value < 2^260;  w: canonical plain;  wq = floor(w * 2^261 / r).  -> x * w mod r, in [0, 4r)
__device__ __forceinline__ Fr mul_fixed(const Fr &x, const Fr &w, const Fr &wq) {
    uint32_t q[NL];
    uint64_t acc = 0;
#pragma unroll
    for (int i = 0; i <= 7; i++) acc += (uint64_t)x.l[i] * wq.l[7 - i];          // guard column 7
    acc >>= LB;
#pragma unroll
    for (int i = 0; i <= 8; i++) acc += (uint64_t)x.l[i] * wq.l[8 - i];          // guard column 8
    acc >>= LB;
#pragma unroll
    for (int c = NL; c < 2 * NL - 1; c++) {
#pragma unroll
        for (int i = c - NL + 1; i < NL; i++) acc += (uint64_t)x.l[i] * wq.l[c - i];
        q[c - NL] = (uint32_t)acc & LMASK;
        acc >>= LB;
    }
    q[NL - 1] = (uint32_t)acc;
    Fr r;
    acc = 0;
#pragma unroll
    for (int c = 0; c < NL; c++) {
#pragma unroll
        for (int i = 0; i <= c; i++) acc += (uint64_t)x.l[i] * w.l[c - i];
#pragma unroll
        for (int i = 0; i <= c; i++) acc += (uint64_t)q[i] * rbar_limb(c - i);
        r.l[c] = (uint32_t)acc & LMASK;                                           // mod 2^261: the carry out of column 8 is dropped
        acc >>= LB;
    }
    return r;
}

// MODE 0: Montgomery, 1: fixed operand, 2: fixed operand WITHOUT the conditional subtraction (values leave the < 2r contract: the
// residues are still right mod r for a few steps, but this mode is for timing only -- an upper bound on what a lazier butterfly
// contract could get out of the product)
template <int MODE>
__global__ __launch_bounds__(256) void k_butterflies(const Fr *__restrict__ w_plain, const Fr *__restrict__ w_quot, const Fr *__restrict__ w_mont,
                                                     const Fr *__restrict__ seed, Fr *__restrict__ out, int iters) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    Fr a = seed[t & 1023], b = seed[(t * 7 + 3) & 1023];
#pragma unroll 1
    for (int k = 0; k < iters; k++) {
        const int idx = (k * 37 + t) & 255;
        const Fr sum = fe_add_r2(a, b);
        Fr dif;
        if (MODE) {
            dif = mul_fixed(fe_sub_lazy<2>(a, b), w_plain[idx], w_quot[idx]);    // < 4r
            if (MODE == 1) fe_cond_sub<2>(dif);                                   // < 2r
        } else {
            dif = fe_mul(fe_sub_lazy<2>(a, b), w_mont[idx]);                      // < 2r
        }
        a = sum;
        b = dif;
    }
    out[2 * (size_t)t] = fe_reduce_full(a);
    out[2 * (size_t)t + 1] = fe_reduce_full(b);
}

// ---- host: w' = floor(w * 2^261 / r) by shift-and-subtract on 64-bit words
struct Big { uint64_t v[10]; };
static bool geq(const Big &a, const Big &b) { for (int i = 9; i >= 0; i--) { if (a.v[i] != b.v[i]) return a.v[i] > b.v[i]; } return true; }
static void sub(Big &a, const Big &b) { unsigned __int128 br = 0; for (int i = 0; i < 10; i++) { unsigned __int128 d = (unsigned __int128)a.v[i] - b.v[i] - br; a.v[i] = (uint64_t)d; br = (d >> 64) & 1; } }
static void shl1(Big &a, int bit) { for (int i = 9; i > 0; i--) a.v[i] = (a.v[i] << 1) | (a.v[i - 1] >> 63); a.v[0] = (a.v[0] << 1) | (uint64_t)bit; }
static void quotient_261(const uint64_t w[4], const uint64_t r[4], uint64_t q_out[5]) {
    Big rem{}, rr{};
    for (int i = 0; i < 4; i++) rr.v[i] = r[i];
    uint64_t q[9] = {0};
    // numerator = w * 2^261: bits of w (254) followed by 261 zero bits, MSB first
    for (int bit = 253 + 261; bit >= 0; bit--) {
        const int wb = bit - 261;
        const int in = wb >= 0 ? (int)((w[wb >> 6] >> (wb & 63)) & 1) : 0;
        shl1(rem, in);
        const int qb = geq(rem, rr) ? 1 : 0;
        if (qb) sub(rem, rr);
        if (bit < 9 * 64) q[bit >> 6] |= (uint64_t)qb << (bit & 63);
    }
    for (int i = 0; i < 5; i++) q_out[i] = q[i];
}
static Fr limbs_from_words64(const uint64_t *w, int nwords) {   // little-endian 64-bit words -> 9 x 29-bit limbs (value < 2^261)
    Fr r;
    for (int i = 0; i < NL; i++) {
        const int bit = LB * i, word = bit >> 6, sh = bit & 63;
        uint64_t v = word < nwords ? w[word] >> sh : 0;
        if (sh + LB > 64 && word + 1 < nwords) v |= w[word + 1] << (64 - sh);
        r.l[i] = (uint32_t)v & LMASK;
    }
    return r;
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s  CUs=%d\n", prop.name, prop.multiProcessorCount);
    const uint32_t mod32[8] = ZK_FR_MOD32;
    uint64_t r64[4];
    for (int i = 0; i < 4; i++) r64[i] = (uint64_t)mod32[2 * i] | ((uint64_t)mod32[2 * i + 1] << 32);
    srand(7);
    std::vector<Fr> wp(256), wq(256), wm(256), seed(1024);
    for (int i = 0; i < 256; i++) {
        uint64_t w[4];
        for (int j = 0; j < 4; j++) w[j] = ((uint64_t)rand() << 42) ^ ((uint64_t)rand() << 21) ^ (uint64_t)rand();
        w[3] &= (1ull << 60) - 1;                                   // < 2^252 < r
        if (i == 0) { w[0] = 1; w[1] = w[2] = w[3] = 0; }           // w = 1
        if (i == 1) { for (int j = 0; j < 4; j++) w[j] = r64[j]; w[0] -= 1; }   // w = r - 1
        uint64_t q[5];
        quotient_261(w, r64, q);
        wp[i] = limbs_from_words64(w, 4);
        wq[i] = limbs_from_words64(q, 5);
        uint32_t w32[8];
        for (int j = 0; j < 4; j++) { w32[2 * j] = (uint32_t)w[j]; w32[2 * j + 1] = (uint32_t)(w[j] >> 32); }
        wm[i] = fe_mul(HFr::from_words(w32), HFr::r2()).to_dev();   // w -> host Montgomery form -> device Montgomery form w * 2^261
    }
    for (int i = 0; i < 1024; i++) {
        for (int j = 0; j < NL; j++) seed[i].l[j] = (rand() * 65537u + rand()) & LMASK;
        seed[i].l[NL - 1] &= 0x3fffff;                               // < 2^254 < 2r
    }
    Fr *d_wp, *d_wq, *d_wm, *d_seed, *d_out0, *d_out1;
    const int blocks = prop.multiProcessorCount * 12, threads = blocks * 256;
    CK(hipMalloc(&d_wp, 256 * sizeof(Fr))); CK(hipMalloc(&d_wq, 256 * sizeof(Fr))); CK(hipMalloc(&d_wm, 256 * sizeof(Fr)));
    CK(hipMalloc(&d_seed, 1024 * sizeof(Fr))); CK(hipMalloc(&d_out0, 2 * (size_t)threads * sizeof(Fr))); CK(hipMalloc(&d_out1, 2 * (size_t)threads * sizeof(Fr)));
    CK(hipMemcpy(d_wp, wp.data(), 256 * sizeof(Fr), hipMemcpyHostToDevice)); CK(hipMemcpy(d_wq, wq.data(), 256 * sizeof(Fr), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_wm, wm.data(), 256 * sizeof(Fr), hipMemcpyHostToDevice)); CK(hipMemcpy(d_seed, seed.data(), 1024 * sizeof(Fr), hipMemcpyHostToDevice));
    // exactness: the same butterflies with either product give the same residues
    for (int iters : {1, 2, 33}) {
        hipLaunchKernelGGL(k_butterflies<0>, dim3(blocks), dim3(256), 0, 0, d_wp, d_wq, d_wm, d_seed, d_out0, iters);
        hipLaunchKernelGGL(k_butterflies<1>, dim3(blocks), dim3(256), 0, 0, d_wp, d_wq, d_wm, d_seed, d_out1, iters);
        std::vector<Fr> h0(2 * (size_t)threads), h1(2 * (size_t)threads);
        CK(hipMemcpy(h0.data(), d_out0, h0.size() * sizeof(Fr), hipMemcpyDeviceToHost)); CK(hipMemcpy(h1.data(), d_out1, h1.size() * sizeof(Fr), hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t i = 0; i < h0.size(); i++) bad += memcmp(h0[i].l, h1[i].l, sizeof(h0[i].l)) != 0;
        printf("iters %2d: fixed-operand vs Montgomery butterflies, %zu residues compared, %zu differ\n", iters, h0.size(), bad);
    }
    for (int rep = 0; rep < 2; rep++) {
        for (int fixed = 0; fixed < 3; fixed++) {
            const int iters = 2048;
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            auto launch = [&] {
                if (fixed == 2) hipLaunchKernelGGL(k_butterflies<2>, dim3(blocks), dim3(256), 0, 0, d_wp, d_wq, d_wm, d_seed, d_out1, iters);
                else if (fixed) hipLaunchKernelGGL(k_butterflies<1>, dim3(blocks), dim3(256), 0, 0, d_wp, d_wq, d_wm, d_seed, d_out1, iters);
                else hipLaunchKernelGGL(k_butterflies<0>, dim3(blocks), dim3(256), 0, 0, d_wp, d_wq, d_wm, d_seed, d_out0, iters);
            };
            launch();
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int r = 0; r < 5; r++) launch();
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%-34s %8.3f ms per launch, %7.2f G butterflies/s\n", fixed == 2 ? "fixed operand, no cond. subtraction" : fixed ? "fixed operand" : "Montgomery", ms / 5, (double)threads * iters / (ms / 5 * 1e-3) / 1e9);
        }
    }
    return 0;
}
