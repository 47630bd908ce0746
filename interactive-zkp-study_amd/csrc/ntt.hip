// ntt.hip -- radix-2 NTT over the BN254 scalar field F_r for gfx950, natural order in and out.
//
// Replaces fft / ifft (zkp/plonk/polynomial.py:292-378) and coset_fft / coset_ifft
// (zkp/plonk/utils.py:145-205).  n = 2^L is factored into D = ceil(L/8) digits n_1..n_D
// (multi-step / Stockham-style autosort): pass p transforms digit p of every element inside an
// LDS tile of 2^(l_p) x 2^g = 2^10 elements (2^g >= 4 adjacent elements per digit value keep every HBM
// access a >= 128-byte run), multiplies by the inter-pass twiddle w_n^(k_p * rem) (two-level
// table) and writes back; the last pass transforms the contiguous digit and writes to the
// digit-reversed position, which makes the output natural-order with no separate bit-reversal
// pass.  Inside a tile the decimation-in-frequency butterflies run two stages at a time on 4
// elements held in registers (one LDS round trip per two stages; three workgroups per CU).  Element VALUES are never
// converted to Montgomery form: only the twiddles are (mont_mul(x, w*R) = x*w); elements are
// kept as lazy 9x29-bit limbs (< 2r) in LDS; in the scratch buffer between passes they travel as 8 words (a value below 2r fits
// 255 bits), so that four adjacent elements are exactly one aligned 128-byte line (round 4: the 36-byte limb image made 144-byte
// runs that straddle lines -- 2^22: 0.510 -> 0.495 ms, 2^24: 2.17 -> 2.12 ms in same-session A/B runs, although the pass kernels are
// issue-bound and the repacking adds instructions); canonical (< r) only at the first load and the last store.
// HBM traffic: 64 bytes per element per pass; the kernel is bound by the vector-ALU issue rate (~10.5 modular
// products per element: butterflies + one per pass boundary, and 22 modular additions / subtractions), not by bandwidth.
#include <stdlib.h>
#include <string.h>
#include "common.h"
#include "ntt.h"

namespace zk {

constexpr int NTT_NT = 256;  // threads per workgroup
constexpr int NTT_G = 2;     // log2 adjacent elements per digit value

__device__ __forceinline__ Fr lds_ld(const uint32_t *base, uint32_t stride, uint32_t slot) {
    Fr r;
#pragma unroll
    for (int i = 0; i < NL; i++) r.l[i] = base[i * stride + slot];
    return r;
}
__device__ __forceinline__ void lds_st(uint32_t *base, uint32_t stride, uint32_t slot, const Fr &v) {
#pragma unroll
    for (int i = 0; i < NL; i++) base[i * stride + slot] = v.l[i];
}
// Bank swizzle of a tile slot (dword index inside one limb plane).  ds_read_b32 / ds_write_b32 see 32 banks and resolve conflicts
// per 32-lane half (MI355X_MICROARCH.md, LDS); the butterfly rounds address slots  hi << (s + 2 + g) | k << (s + g) | low << g | c,
// the last pass stores its tile transposed (slot = j << g | c with j along the lanes), and with plain slots the late rounds are 2- to
// 4-way and the transposed store 16-way conflicts (SQ_LDS_BANK_CONFLICT: 57-65 % of the LDS cycles of a pass).  XOR-ing slot bits
// 5..8 into the bank bits -- b5 -> B2, b6 -> B3 and B4, b7 -> B0, b8 -> B1 -- makes every one of those patterns (and every aligned
// run of 32 slots) conflict-free.  The map is linear over XOR: swz(a ^ b) = swz(a) ^ swz(b).
__device__ __forceinline__ uint32_t swz(uint32_t slot) {
    const uint32_t u = slot >> 5;
    return slot ^ (((u & 3u) << 2) | ((u & 2u) << 3) | ((u >> 2) & 3u));
}
// canonical element (8 words, 32 B) <-> lazy limbs
__device__ __forceinline__ Fr ld_canon(const uint32_t *p) {
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
    const uint4 a = q[0], b = q[1];
    const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    return fe_from_words<FrTag>(w);
}
// the same for a normalised value already known to be < 2r (every product, every reduced butterfly sum): one conditional
// subtraction instead of the four of fe_reduce_full
__device__ __forceinline__ void st_canon_2r(uint32_t *p, Fr v) {
    uint32_t w[8];
    ZK_DBG_ASSERT(v.vb <= 2 && v.lmax <= 1, "st_canon_2r: value must be normalised and < 2r");
    fe_cond_sub<1>(v);
    fe_to_words(v, w);
    uint4 *q = reinterpret_cast<uint4 *>(p);
    q[0] = make_uint4(w[0], w[1], w[2], w[3]);
    q[1] = make_uint4(w[4], w[5], w[6], w[7]);
}
__device__ __forceinline__ void st_canon(uint32_t *p, const Fr &v) {
    uint32_t w[8];
    fe_to_words(fe_reduce_full(v), w);
    uint4 *q = reinterpret_cast<uint4 *>(p);
    q[0] = make_uint4(w[0], w[1], w[2], w[3]);
    q[1] = make_uint4(w[4], w[5], w[6], w[7]);
}

// R butterfly stages (s_hi .. s_hi-R+1) on 2^R elements held in registers: LDS is read and written
// once per element per round instead of once per stage, and a stage's twiddle is fetched once per
// 2^(R-1-b) butterflies.  Values stay < 2r (see the pass kernel).  A difference that goes into its twiddle
// multiplication and nowhere else is formed without carry propagation (fe_sub_lazy: a - b + 3r, limbs < 3 * 2^29).
// LAST: the round that ends at stage 0 (s_lo == 0): there `low` is 0, so every q == 0 twiddle is w^0 = 1
// and those butterflies skip the multiplication (all of stage 0, half of stage 1, a quarter of stage 2).
// (Leaving the sums unreduced until the end of a round -- 7 conditional subtractions per 8 elements instead of 12 --
// was tried: it needs 288 registers, and at one wavefront per SIMD the pass is 30 % slower; capped at 256 it spills.)
// The pass kernel uses R = 2 (and R = 1 to finish an odd digit): three stages on 8 elements need 255 registers and a
// 2^11-element tile to keep 256 threads busy -- two workgroups per CU, two wavefronts per SIMD, 72 % of the vector-ALU issue
// rate; two stages on 4 elements need 144 registers and a 2^10-element tile: three wavefronts per SIMD, 79 % of the issue
// rate (profiles/r02_pmc_sq_summary.csv), and although LDS is crossed four times per 8-bit digit instead of three a 2^22-point
// transform takes 0.51 ms instead of 0.545 (both libraries on one box, a hundred round trips; 2^20: 0.147 instead of 0.155).
// GTW (the T1K kernels): the LDS table holds the twiddles of stages 0..5 only (63 entries, 2.3 KB instead of 9: with the 36 KB tile
// that is 38.25 KB per workgroup, FOUR workgroups per CU instead of three -- the T1K kernels need 127 registers, so the fourth
// wavefront per SIMD fits); stages 6 and 7 (the first round of a pass over a 7- or 8-bit digit) read theirs from the global table,
// consecutive lanes consecutive entries (L1 / L2 hits: the table has 128 entries).
// Wave priority (s_setprio).  1: a workgroup's first phase -- global loads, conversion, LDS stores up to the first barrier -- runs at
// priority 3: the SIMD arbiter otherwise serves its oldest wavefronts first, a newly placed workgroup issues its loads only in the
// gaps the older ones leave, and its memory latency starts late (tools/ntt_phase_probe.py: a quarter of a workgroup's life went
// into that phase).  2^22: 0.469-0.475 -> 0.462-0.464 ms, 2^24: 2.00-2.02 -> 1.97, 2^20 unchanged (three alternating pairs).  2: the
// epilogue too; 3 / 4: the LDS loads / stores of every round as well -- no further gain (profiles/r05_experiments.md section 4).
#ifndef ZK_NTT_PRIO
#define ZK_NTT_PRIO 1
#endif
#ifndef ZK_NTT_TW_LDS_STAGES
#define ZK_NTT_TW_LDS_STAGES 6   // 8: every stage's twiddles in LDS (9 KB table, three workgroups per CU: the round-5 A/B)
#endif
constexpr int NTT_TW_LDS_STAGES = ZK_NTT_TW_LDS_STAGES;
template <int R, bool LAST, bool GTW>
__device__ __forceinline__ void ntt_round(uint32_t *data, const uint32_t *tw, uint32_t tile, uint32_t ntw, uint32_t lp, uint32_t g,
                                          int s_hi, const Fr *__restrict__ gtw, uint32_t tw_shift) {
    const int s_lo = LAST ? 0 : s_hi - R + 1;
    const uint32_t ngroups = tile >> R;
    uint32_t kx[1 << R];   // swizzle of the element bits k << (s_lo + g): uniform over the wavefront
#pragma unroll
    for (int k = 0; k < (1 << R); k++) kx[k] = swz((uint32_t)k << (s_lo + g));
    for (uint32_t gi = threadIdx.x; gi < ngroups; gi += NTT_NT) {
        const uint32_t c = gi & ((1u << g) - 1u), rest = gi >> g;
        const uint32_t low = rest & ((1u << s_lo) - 1u);
        const uint32_t jb = ((rest >> s_lo) << (s_hi + 1)) | low;
        const uint32_t sb = swz((jb << g) | c);
        Fr x[1 << R];
#if ZK_NTT_PRIO >= 3
        __builtin_amdgcn_s_setprio(2);
#endif
#pragma unroll
        for (int k = 0; k < (1 << R); k++) x[k] = lds_ld(data, tile, sb ^ kx[k]);
#if ZK_NTT_PRIO >= 3
        __builtin_amdgcn_s_setprio(0);
#endif
#pragma unroll
        for (int b = R - 1; b >= 0; b--) {
            const int s = s_lo + b;
#pragma unroll
            for (int q = 0; q < (1 << b); q++) {
                const bool unit = LAST && q == 0;
                Fr w;
                if (!unit) {
                    const uint32_t j = low | ((uint32_t)q << s_lo);
                    if (GTW && s >= NTT_TW_LDS_STAGES) w = gtw[(size_t)(j << (lp - 1 - s)) << tw_shift];
                    else w = lds_ld(tw, ntw, (1u << s) + j);   // stage s: w^(j << (lp - 1 - s)) at 2^s + j
                }
#pragma unroll
                for (int hi = 0; hi < (1 << (R - 1 - b)); hi++) {
                    const int k0 = (hi << (b + 1)) | q, k1 = k0 | (1 << b);
                    const Fr sum = fe_add_r2(x[k0], x[k1]);   // < 2r, one carry chain (field.h)
                    Fr dif;
                    if (unit) {
                        dif = fe_sub_r2(x[k0], x[k1]);          // < 2r
                    } else {
                        dif = fe_mul(fe_sub_lazy<2>(x[k0], x[k1]), w);  // (a - b + 3r) < 5r, w < r  ->  < 2r
                    }
                    x[k1] = dif;
                    x[k0] = sum;
                }
            }
        }
#if ZK_NTT_PRIO >= 4
        __builtin_amdgcn_s_setprio(2);
#endif
#pragma unroll
        for (int k = 0; k < (1 << R); k++) lds_st(data, tile, sb ^ kx[k], x[k]);
#if ZK_NTT_PRIO >= 4
        __builtin_amdgcn_s_setprio(0);
#endif
    }
}

// Where element i of transform b sits in a caller buffer of layout LAYOUT (ntt.h), in elements.
template <int LAYOUT> __device__ __forceinline__ size_t io_addr(uint32_t L, const NttIoArgs &io, uint32_t b, uint32_t i) {
    if (LAYOUT == NTT_PLAIN) return ((size_t)b << L) + i;
    if (LAYOUT == NTT_BLOCKED_TW) return ((((size_t)(i >> io.kbits)) * io.batch + b) << io.kbits) | (i & ((1u << io.kbits) - 1u));
    return (size_t)i * io.batch + b;
}
// w_N^((row0 + b) * i) from the large transform's two-level table (the four-step twiddle; < 2r, Montgomery form)
__device__ __forceinline__ Fr io_twiddle(const NttIoArgs &io, uint32_t b, uint32_t i) {
    const uint64_t e = (io.row0 + b) * (uint64_t)i;
    return fe_mul(io.twA[e & (((uint64_t)1 << io.lh) - 1)], io.twB[e >> io.lh]);
}

// k^(+-i) from the plan's two-level coset table (< 2r, Montgomery form)
__device__ __forceinline__ Fr io_coset(const NttIoArgs &io, uint32_t i) {
    return fe_mul(io.cosA[i & ((1u << io.cos_lh) - 1u)], io.cosB[i >> io.cos_lh]);
}

// Measurement builds only (-DZK_NTT_STAMPS, tools/ntt_phase_probe.py): thread 0 of every workgroup of a batch's first transform writes
// the wall clock (100 MHz) at its phase boundaries -- entry, loads issued and converted, first barrier, every round's barrier, stores
// issued, stores done -- and its hardware id to zk_ntt_stamp_buf[pass][workgroup][16].  Compiled out of the library.
#ifdef ZK_NTT_STAMPS
__device__ unsigned long long *zk_ntt_stamp_buf = nullptr;
#define NTT_STAMP(pass, k)                                                                                             \
    do {                                                                                                               \
        if (threadIdx.x == 0 && blockIdx.y == 0 && zk_ntt_stamp_buf != nullptr && blockIdx.x < 4096)                   \
            zk_ntt_stamp_buf[(((size_t)(pass)) * 4096 + blockIdx.x) * 16 + (k)] = wall_clock64();                      \
    } while (0)
#define NTT_STAMP_HWID(pass)                                                                                           \
    do {                                                                                                               \
        if (threadIdx.x == 0 && blockIdx.y == 0 && zk_ntt_stamp_buf != nullptr && blockIdx.x < 4096) {                 \
            uint32_t hw, xcc;                                                                                          \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));                                           \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));                                         \
            zk_ntt_stamp_buf[(((size_t)(pass)) * 4096 + blockIdx.x) * 16 + 15] = ((unsigned long long)xcc << 32) | hw; \
        }                                                                                                              \
    } while (0)
extern "C" int zk_ntt_set_stamp_buffer(void *d_buf) {
    unsigned long long *p = static_cast<unsigned long long *>(d_buf);
    return hipMemcpyToSymbol(HIP_SYMBOL(zk_ntt_stamp_buf), &p, sizeof(p)) == hipSuccess ? 0 : -1;
}
#else
#define NTT_STAMP(pass, k) do { } while (0)
#define NTT_STAMP_HWID(pass) do { } while (0)
#endif

// One pass over one digit.  Element values stay < 2r in LDS and in the scratch buffer between
// passes (9-limb form in LDS, 8 words = 32 B in the scratch); only the first load and the last store use the canonical
// 32-byte encoding.  LDS: data[9][tile] (limb-major, slots swizzled: swz) | tw[9][2^lp]: the twiddles of stage s,
// w^(j << (lp - 1 - s)) for j < 2^s, sit at 2^s + j, so that the lanes of a wavefront read neighbouring words at every stage
// (one table of w^i indexed i = j << (lp - 1 - s) puts the late stages' few distinct twiddles all on one bank).
//   IN_CANON : `in` holds canonical elements (first pass), else lazy Fr elements (scratch)
//   FINAL    : last pass: contiguous digit, digit-reversed (natural-order) canonical store
//   IN_L / OUT_L : layout of the caller's buffers (ntt.h: NttLayout), seen by the first pass's loads (IN_CANON) and the last pass's
//              stores (FINAL); the scratch between passes is always plain.  The tile carries 2^g "bystander" columns next to the
//              digit: 2^(g - gb) adjacent ELEMENTS and 2^gb adjacent TRANSFORMS of the batch (gb > 0 only with a transposed
//              buffer, whose memory runs along the batch index), so that both sides of a transposing pass move >= 128-byte runs.
//   T1K      : the tile holds 2^10 elements (the usual case): the limb planes of the tile then lie 4096 bytes apart and those of the
//              twiddle table 256 (the table is given 64 columns whatever the digit) -- constants the compiler folds into the
//              LDS instructions' offsets, pairing the limb accesses of an element into ds_read2st64_b32 / ds_write2st64_b32
//              (five LDS instructions per element instead of nine)
template <bool FINAL, bool IN_CANON, int IN_L, int OUT_L, bool T1K = false>
__global__ __launch_bounds__(NTT_NT) void ntt_pass_kernel(const void *__restrict__ in_v, void *__restrict__ out_v,
                                                          const Fr *__restrict__ tile_tw, const Fr *__restrict__ twA,
                                                          const Fr *__restrict__ twB, Fr scale, NttPassParams P, NttIoArgs io) {
    extern __shared__ uint32_t lds[];
    constexpr int STAMP_PASS = FINAL ? 2 : (IN_CANON ? 0 : 1);   // measurement builds: which row of the stamp buffer
    (void)STAMP_PASS;
    NTT_STAMP(STAMP_PASS, 0);
    NTT_STAMP_HWID(STAMP_PASS);
#if ZK_NTT_PRIO >= 1
    __builtin_amdgcn_s_setprio(3);
#endif
    const uint32_t t = threadIdx.x;
    const uint32_t lp = P.lp, g = P.g, G = 1u << g;
    constexpr bool HAS_GB = (IN_L == NTT_TRANSPOSED || OUT_L == NTT_TRANSPOSED);
    const uint32_t gb = HAS_GB ? P.gb : 0u, ga = g - gb, Gb1 = (1u << gb) - 1u, Ga1 = (1u << ga) - 1u;
    const uint32_t tile = T1K ? 1024u : 1u << (lp + g);
    const uint32_t ntw = 1u << lp;                     // entries of the twiddle table,
    const uint32_t tws = T1K ? (1u << NTT_TW_LDS_STAGES) : ntw;   // and the distance of its limb planes (T1K: stages 0..5 only, see ntt_round)
    uint32_t *data = lds;
    uint32_t *tw = lds + NL * tile;
    // blockIdx.y = index of the transform (of the group of 2^gb transforms) inside a batch of independent transforms
    const uint32_t bbase = blockIdx.y << gb;
    const uint32_t *in_c = static_cast<const uint32_t *>(in_v);   // canonical input or the 8-word scratch: both 32 bytes per element

    for (uint32_t i = t + 1; i < (T1K ? min(ntw, 1u << NTT_TW_LDS_STAGES) : ntw); i += NTT_NT) {
        const uint32_t st = 31u - (uint32_t)__clz((int)i), j = i - (1u << st);
        lds_st(tw, tws, i, tile_tw[(size_t)(j << (lp - 1 - st)) << P.tw_shift]);
    }

    const uint32_t tile_id = blockIdx.x;
    uint32_t base_addr = 0, mid = 0, k1_base = 0, mid_in = 0;
    if (!FINAL) {
        mid = tile_id & ((1u << (P.sp - ga)) - 1u);
        const uint32_t hi = tile_id >> (P.sp - ga);
        base_addr = (hi << (P.sp + lp)) + (mid << ga);
        for (uint32_t e = t; e < tile; e += NTT_NT) {
            const uint32_t j = e >> g, c = e & (G - 1u), cb = c & Gb1, ca = c >> gb;   // the batch bits run fastest
            const uint32_t i = base_addr + (j << P.sp) + ca, b = bbase + cb;
            Fr v;
            if (IN_CANON) {
                if (IN_L == NTT_PLAIN && i >= io.in_len) {
                    v = Fr::zero();                                                     // zero-padded input: not read
                } else {
                    if (IN_L == NTT_PLAIN && io.multi) v = ld_canon(static_cast<const uint32_t *>(io.in_multi[b]) + (size_t)i * 8);
                    else v = ld_canon(in_c + io_addr<IN_L>(P.L, io, b, i) * 8);
                    if (IN_L == NTT_BLOCKED_TW) v = fe_mul(v, io_twiddle(io, b, i));
                    if (IN_L == NTT_PLAIN && io.cos_in) v = fe_mul(v, io_coset(io, i));   // canonical < r times < 2r  ->  < 2r
                }
            } else {
                v = ld_canon(in_c + (((size_t)b << P.L) + i) * 8);   // 32-byte scratch: 8 words, value < 2r
            }
            lds_st(data, tile, swz(e), v);
        }
    } else {
        const uint32_t lmid_tot = (P.nmid > 0 ? P.lmid[0] : 0) + (P.nmid > 1 ? P.lmid[1] : 0);
        mid_in = tile_id & ((1u << lmid_tot) - 1u);
        k1_base = (tile_id >> lmid_tot) << ga;
        for (uint32_t e = t; e < tile; e += NTT_NT) {
            uint32_t c, j;
            if (IN_CANON && IN_L == NTT_TRANSPOSED) {   // memory runs along the batch index: those bits fastest
                c = e & (G - 1u);
                j = e >> g;
            } else {
                c = e >> lp;
                j = e & ((1u << lp) - 1u);
            }
            const uint32_t cb = c & Gb1, ca = c >> gb;
            const uint32_t i = ((k1_base + ca) << (P.L - P.l1)) + (mid_in << lp) + j, b = bbase + cb;
            Fr v;
            if (IN_CANON) {
                if (IN_L == NTT_PLAIN && i >= io.in_len) {
                    v = Fr::zero();                                                     // zero-padded input: not read
                } else {
                    if (IN_L == NTT_PLAIN && io.multi) v = ld_canon(static_cast<const uint32_t *>(io.in_multi[b]) + (size_t)i * 8);
                    else v = ld_canon(in_c + io_addr<IN_L>(P.L, io, b, i) * 8);
                    if (IN_L == NTT_BLOCKED_TW) v = fe_mul(v, io_twiddle(io, b, i));
                    if (IN_L == NTT_PLAIN && io.cos_in) v = fe_mul(v, io_coset(io, i));   // canonical < r times < 2r  ->  < 2r
                }
            } else {
                v = ld_canon(in_c + (((size_t)b << P.L) + i) * 8);   // 32-byte scratch: 8 words, value < 2r
            }
            lds_st(data, tile, swz((j << g) | c), v);
        }
    }
    NTT_STAMP(STAMP_PASS, 1);
#if ZK_NTT_PRIO >= 1
    __builtin_amdgcn_s_setprio(0);
#endif
    __syncthreads();
    NTT_STAMP(STAMP_PASS, 2);
    int stamp_k = 3;
    (void)stamp_k;

    // decimation-in-frequency butterflies over the digit index j (slot = j * G + c), two stages per LDS round trip (radix-4 in
    // registers: four elements per thread and round, see the plan's tile size)
    for (int sh = (int)lp - 1; sh >= 0;) {
        const int R = sh == 0 ? 1 : 2;
        const bool last = (sh - R + 1 == 0);
        if (R == 2) {
            if (last) ntt_round<2, true, T1K>(data, tw, tile, tws, lp, g, sh, tile_tw, P.tw_shift);
            else ntt_round<2, false, T1K>(data, tw, tile, tws, lp, g, sh, tile_tw, P.tw_shift);
        } else {
            ntt_round<1, true, T1K>(data, tw, tile, tws, lp, g, sh, tile_tw, P.tw_shift);  // a single stage is only ever the last one
        }
        sh -= R;
        __syncthreads();
        NTT_STAMP(STAMP_PASS, stamp_k++);
    }

#if ZK_NTT_PRIO >= 2
    __builtin_amdgcn_s_setprio(2);
#endif
    if (!FINAL) {
        const uint32_t sh = P.L - lp - P.sp;
        for (uint32_t e = t; e < tile; e += NTT_NT) {   // in LDS order: the digit comes out bit-reversed, k = brev(position)
            // the scratch runs along the element index: those bits fastest (with gb == 0 this is the LDS order itself)
            const uint32_t jpos = e >> g, r = e & (G - 1u), ca = r & Ga1, cb = r >> ga, c = (ca << gb) | cb;
            const uint32_t k = lp ? (__brev(jpos) >> (32 - lp)) : 0u;
            Fr x = lds_ld(data, tile, swz((jpos << g) | c));
            const uint32_t rem = (mid << ga) + ca;
            const uint32_t ex = (k * rem) << sh;
            // boundaries up to 2^24 entries keep the ready-made twiddle per (k, rem); larger ones build it from the
            // two-level table with one extra multiplication
            const Fr w = P.direct_tw ? twA[((size_t)k << P.sp) + rem] : fe_mul(twA[ex & ((1u << P.lh) - 1u)], twB[ex >> P.lh]);  // < 2r
            x = fe_mul(x, w);                                                      // 2 * 2 < 169  ->  < 2r
            {
                uint32_t w8[8];
                fe_to_words(x, w8);
                uint4 *q4 = reinterpret_cast<uint4 *>(static_cast<uint32_t *>(out_v) + (((size_t)(bbase + cb) << P.L) + base_addr + ((size_t)k << P.sp) + ca) * 8);
                q4[0] = make_uint4(w8[0], w8[1], w8[2], w8[3]);
                q4[1] = make_uint4(w8[4], w8[5], w8[6], w8[7]);
            }
        }
    } else {
        uint32_t *out_c = static_cast<uint32_t *>(out_v);
        uint32_t kmid = mid_in, lmid_tot = 0;
        if (P.nmid == 1) {
            lmid_tot = P.lmid[0];
        } else if (P.nmid == 2) {
            const uint32_t k3 = mid_in & ((1u << P.lmid[1]) - 1u), k2 = mid_in >> P.lmid[1];
            kmid = k2 | (k3 << P.lmid[0]);
            lmid_tot = P.lmid[0] + P.lmid[1];
        }
        for (uint32_t e = t; e < tile; e += NTT_NT) {
            const uint32_t jpos = e >> g, c = e & (G - 1u), cb = c & Gb1, ca = c >> gb;
            const uint32_t k = lp ? (__brev(jpos) >> (32 - lp)) : 0u;
            Fr x = lds_ld(data, tile, swz(e));
            if (P.apply_scale) x = fe_mul(x, scale);
            const uint32_t oidx = (k1_base + ca) + ((kmid + (k << lmid_tot)) << P.l1), b = bbase + cb;
            if (OUT_L == NTT_BLOCKED_TW) x = fe_mul(x, io_twiddle(io, b, oidx));   // 2 * 2 < 169  ->  < 2r
            if (OUT_L == NTT_PLAIN && io.cos_out) x = fe_mul(x, io_coset(io, oidx));
            if (OUT_L == NTT_PLAIN && io.multi) st_canon_2r(static_cast<uint32_t *>(io.out_multi[b]) + (size_t)oidx * 8, x);
            else st_canon_2r(out_c + io_addr<OUT_L>(P.L, io, b, oidx) * 8, x);
        }
    }
#ifdef ZK_NTT_STAMPS
    NTT_STAMP(STAMP_PASS, 12);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    NTT_STAMP(STAMP_PASS, 13);
#endif
}

// x[j] *= A[j & mask] * B[j >> lh]   (two-level table of powers g^j, Montgomery form); canonical in/out
__global__ __launch_bounds__(256) void fr_scale_powers_kernel(uint32_t *__restrict__ x, const Fr *__restrict__ A, const Fr *__restrict__ B,
                                                              uint32_t lh, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const Fr w = fe_mul(A[i & ((1u << lh) - 1u)], B[i >> lh]);
    st_canon_2r(x + i * 8, fe_mul(ld_canon(x + i * 8), w));
}

// x[b * cols + k] *= w^((row0 + b) * k) for b < rows, k < cols, with w^e = A[e & mask] * B[e >> lh] (two-level table of the
// size-n plan): the twiddle between the two dimensions of a four-step transform of n = (rows of all ranks) * cols points.
__global__ __launch_bounds__(256) void fr_twiddle_2d_kernel(uint32_t *__restrict__ x, const Fr *__restrict__ A, const Fr *__restrict__ B, uint32_t lh,
                                                            uint32_t log_cols, uint64_t row0, size_t total) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const uint64_t b = i >> log_cols, k = i & (((uint64_t)1 << log_cols) - 1);
    const uint64_t e = (row0 + b) * k;
    if (e == 0) return;
    const Fr w = fe_mul(A[e & (((uint64_t)1 << lh) - 1)], B[e >> lh]);
    st_canon(x + i * 8, fe_mul(ld_canon(x + i * 8), w));
}

// out[(k << sp) + rem] = w^((k * rem) << sh) from the two-level table: the ready-made inter-pass twiddles of one pass.
__global__ __launch_bounds__(256) void ntt_direct_table_kernel(Fr *__restrict__ out, const Fr *__restrict__ A, const Fr *__restrict__ B, uint32_t lh,
                                                               uint32_t sp, uint32_t sh, uint32_t total) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const uint32_t k = i >> sp, rem = i & ((1u << sp) - 1u);
    const uint32_t ex = (k * rem) << sh;
    out[i] = fe_mul(A[ex & ((1u << lh) - 1u)], B[ex >> lh]);
}

__global__ __launch_bounds__(256) void fr_quotient_kernel(uint32_t *__restrict__ out, const uint32_t *__restrict__ a, const uint32_t *__restrict__ b,
                                                          const uint32_t *__restrict__ c, Fr zr, Fr zr2, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    // canonical in/out: mont_mul(a,b) = ab/R; times z*R^2 -> ab*z;  mont_mul(c, z*R) = c*z
    const Fr ab = fe_mul(fe_mul(ld_canon(a + i * 8), ld_canon(b + i * 8)), zr2);
    st_canon(out + i * 8, fe_sub_k<2>(ab, fe_mul(ld_canon(c + i * 8), zr)));
}

// y[i] = sum_{j in row i} vals[j] * x[col[j]]  (CSR, canonical elements): the F_r mat-vec that collapses the
// witness into the per-constraint values A.w, B.w, C.w (zkp/groth16/proving.py:27-31 does this product in the
// group, W*G scalar multiplications).  R1CS rows hold a handful of entries: one thread per row.
__global__ __launch_bounds__(256) void fr_spmv_kernel(const uint32_t *__restrict__ row_ptr, const uint32_t *__restrict__ col, const uint32_t *__restrict__ vals,
                                                      const uint32_t *__restrict__ x, uint32_t *__restrict__ y, size_t rows) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows) return;
    Fr acc = Fr::zero();
    for (uint32_t j = row_ptr[i], e = row_ptr[i + 1]; j < e; j++) {
        const Fr v = fe_to_mont(ld_canon(vals + (size_t)j * 8));              // v * R
        acc = fe_add(acc, fe_mul(v, ld_canon(x + (size_t)col[j] * 8)));        // v * x  (< 2r), running sum < 4r
        fe_wreduce<4>(acc);                                                    // < 2r
    }
    st_canon(y + i * 8, acc);
}

// ------------------------------------------------------------------------------ host side
static HFr hfr_pow_u64(HFr a, uint64_t e) {
    uint64_t ee[4] = {e, 0, 0, 0};
    return fe_pow(a, ee);
}
static HFr root_of_unity(unsigned L) {  // w_n = 5^((r-1)/2^L), host Montgomery form
    const uint32_t w28[8] = ZK_FR_ROOT28_H;
    HFr w = HFr::from_words(w28);
    for (unsigned i = L; i < ZK_FR_TWO_ADICITY; i++) w = fe_sqr(w);
    return w;
}
static void upload_powers(DevBuf &buf, HFr base, size_t count, const HFr *scale = nullptr) {
    std::vector<Fr> h(count);
    HFr cur = scale ? *scale : HFr::one();
    for (size_t i = 0; i < count; i++) {
        h[i] = cur.to_dev();
        cur = fe_mul(cur, base);
    }
    buf.alloc(count * sizeof(Fr));
    ZK_HIP(hipMemcpy(buf.p, h.data(), count * sizeof(Fr), hipMemcpyHostToDevice));
}

static std::vector<const void *> pass_kernel_functions();

NttPlan::NttPlan(unsigned log_n) : L_(log_n) {
    if (L_ <= 10) {
        digits_.push_back(L_);
    } else {
        const unsigned D = (L_ + 7) / 8;
        for (unsigned p = 0; p < D; p++) digits_.push_back(L_ / D + (p < L_ % D ? 1 : 0));
        // the butterflies run two stages per LDS round trip: an odd digit ends on a single-stage round, so odd digits are paired off
        // (2^22: 8 + 7 + 7 -> 8 + 8 + 6, eleven rounds instead of twelve)
        for (unsigned i = 0; i < D; i++)
            for (unsigned j = D; j-- > i + 1;)
                if ((digits_[i] & 1) && digits_[i] < 8 && (digits_[j] & 1) && digits_[j] > 1) {
                    digits_[i]++;
                    digits_[j]--;
                }
    }
    lmax_ = 0;
    for (uint32_t d : digits_) lmax_ = std::max(lmax_, d);
    lh_ = (L_ + 1) / 2;
    build_tables();
    // the inter-pass scratch (n * 32 bytes per transform in flight) is allocated by the first run()
    // per device: the opt-in to more than 64 KiB of dynamic LDS belongs to the function ON the current device
    static bool attr_done_dev[64] = {};
    ZK_HIP(hipGetDevice(&device_));
    bool &attr_done = attr_done_dev[device_ & 63];
    if (!attr_done) {
        for (const void *f : pass_kernel_functions()) ZK_HIP(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        attr_done = true;
    }
}

void NttPlan::build_tables() {
    const HFr w = root_of_unity(L_);
    const HFr wi = fe_inv(w);
    const HFr ninv = fe_inv(hfe_from_u64<FrTag>((uint64_t)1 << L_));
    scale_inv_ = ninv.to_dev();
    for (int dir = 0; dir < 2; dir++) {
        const HFr base = dir ? wi : w;
        // in-tile table: w_{2^lmax}^t, t < 2^(lmax-1)
        const HFr wt = hfr_pow_u64(base, (uint64_t)1 << (L_ - lmax_));
        upload_powers(tile_tw_[dir], wt, lmax_ ? ((size_t)1 << (lmax_ - 1)) : 1);
        // two-level inter-pass tables
        upload_powers(twA_[dir], base, (size_t)1 << lh_);
        const HFr bh = hfr_pow_u64(base, (uint64_t)1 << lh_);
        upload_powers(twB_[dir], bh, (size_t)1 << (L_ - lh_));
        if (dir) upload_powers(twB_scaled_inv_, bh, (size_t)1 << (L_ - lh_), &ninv);
    }
    // ready-made twiddles for the pass boundaries whose table has <= 2^24 entries (576 MB; beyond, the two-level table)
    for (int dir = 0; dir < 2; dir++) {
        uint32_t sp = L_;
        for (size_t p = 0; p + 1 < digits_.size() && p < 3; p++) {
            const uint32_t lp = digits_[p];
            sp -= lp;
            if (lp + sp > 24) continue;
            const uint32_t total = 1u << (lp + sp), sh = L_ - lp - sp;
            tw_direct_[dir][p].alloc((size_t)total * sizeof(Fr));
            const Fr *B = (dir == 1 && p == 0) ? twB_scaled_inv_.as<Fr>() : twB_[dir].as<Fr>();
            hipLaunchKernelGGL(ntt_direct_table_kernel, dim3((total + 255) / 256), dim3(256), 0, 0, tw_direct_[dir][p].as<Fr>(), twA_[dir].as<Fr>(), B, lh_,
                               sp, sh, total);
        }
    }
    ZK_HIP(hipDeviceSynchronize());
}

void NttPlan::coset_tables(const uint64_t k[4], bool inverse) {
    const int d = inverse ? 1 : 0;
    if (cos_valid_[d] && !memcmp(cos_k_[d], k, 32)) return;
    HFr kk;
    memcpy(kk.l, k, 32);
    kk = fe_to_mont(kk);
    if (inverse) kk = fe_inv(kk);
    ZK_HIP(hipDeviceSynchronize());  // a transform still in flight may be reading the tables this replaces
    upload_powers(cosA_[d], kk, (size_t)1 << lh_);
    upload_powers(cosB_[d], hfr_pow_u64(kk, (uint64_t)1 << lh_), (size_t)1 << (L_ - lh_));
    memcpy(cos_k_[d], k, 32);
    cos_valid_[d] = true;
}

void NttPlan::run(void *d_data, bool inverse, const uint64_t coset_shift[4], hipStream_t st, unsigned batch) {
    const size_t n = (size_t)1 << L_;
    if (batch == 0) return;
    if (coset_shift && batch != 1) throw std::runtime_error("zk_ntt: coset shifts are not available for batched transforms");
    uint32_t *data = static_cast<uint32_t *>(d_data);
    NttIoArgs io;
    if (coset_shift) {
        // the scaling by k^i (before a forward transform) / k^-i (after an inverse one) rides in the first pass's loads / the last
        // pass's stores; a transform with no pass at all (n = 1) keeps the separate kernel
        const int d = inverse ? 1 : 0;
        coset_tables(coset_shift, inverse);
        if (L_ == 0) {
            hipLaunchKernelGGL(fr_scale_powers_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, data, cosA_[d].as<Fr>(), cosB_[d].as<Fr>(), lh_, n);
        } else {
            io.cosA = cosA_[d].as<Fr>();
            io.cosB = cosB_[d].as<Fr>();
            io.cos_lh = lh_;
            io.cos_in = inverse ? 0u : 1u;
            io.cos_out = inverse ? 1u : 0u;
        }
    }
    launch_passes(d_data, d_data, inverse, batch, NTT_PLAIN, NTT_PLAIN, io, st);
    ZK_HIP(hipGetLastError());
}

void NttPlan::run_padded(const void *d_in, void *d_out, size_t in_len, bool inverse, const uint64_t coset_shift[4], hipStream_t st) {
    const size_t n = (size_t)1 << L_;
    NttIoArgs io;
    io.in_len = (uint32_t)std::min<size_t>(in_len, n);
    if (L_ == 0) {   // no pass to ride in: copy (or zero) the one element, then the in-place path
        if (io.in_len == 0) ZK_HIP(hipMemsetAsync(d_out, 0, 32, st));
        else if (d_in != d_out) ZK_HIP(hipMemcpyAsync(d_out, d_in, 32, hipMemcpyDeviceToDevice, st));
        run(d_out, inverse, coset_shift, st);
        return;
    }
    if (coset_shift) {
        const int d = inverse ? 1 : 0;
        coset_tables(coset_shift, inverse);
        io.cosA = cosA_[d].as<Fr>();
        io.cosB = cosB_[d].as<Fr>();
        io.cos_lh = lh_;
        io.cos_in = inverse ? 0u : 1u;
        io.cos_out = inverse ? 1u : 0u;
    }
    launch_passes(d_in, d_out, inverse, 1, NTT_PLAIN, NTT_PLAIN, io, st);
    ZK_HIP(hipGetLastError());
}

void NttPlan::run_multi(const void *const *d_in, void *const *d_out, unsigned jobs, size_t in_len, bool inverse, const uint64_t coset_shift[4], hipStream_t st) {
    if (jobs == 0) return;
    if (jobs > NTT_MULTI_MAX) throw std::runtime_error("zk_ntt_dev_multi: at most 4 transforms per call");
    if (jobs == 1 || L_ == 0) {   // nothing to share (L = 0: no pass at all)
        for (unsigned b = 0; b < jobs; b++) run_padded(d_in[b], d_out[b], in_len, inverse, coset_shift, st);
        return;
    }
    const size_t n = (size_t)1 << L_;
    NttIoArgs io;
    io.in_len = (uint32_t)std::min<size_t>(in_len, n);
    io.multi = 1;
    io.batch = jobs;
    for (unsigned b = 0; b < jobs; b++) {
        io.in_multi[b] = d_in[b];
        io.out_multi[b] = d_out[b];
    }
    if (coset_shift) {
        const int d = inverse ? 1 : 0;
        coset_tables(coset_shift, inverse);
        io.cosA = cosA_[d].as<Fr>();
        io.cosB = cosB_[d].as<Fr>();
        io.cos_lh = lh_;
        io.cos_in = inverse ? 0u : 1u;
        io.cos_out = inverse ? 1u : 0u;
    }
    launch_passes(d_in[0], d_out[0], inverse, jobs, NTT_PLAIN, NTT_PLAIN, io, st);
    ZK_HIP(hipGetLastError());
}

void NttPlan::run_io(const void *d_in, void *d_out, bool inverse, unsigned batch, int in_layout, int out_layout, unsigned kbits, uint64_t row0,
                     const NttPlan *big, bool tw_inverse, hipStream_t st) {
    if (batch == 0) return;
    for (int lay : {in_layout, out_layout})
        if (lay != NTT_PLAIN && lay != NTT_BLOCKED_TW && lay != NTT_TRANSPOSED) throw std::runtime_error("zk_ntt: unknown buffer layout");
    if (d_in == d_out && (in_layout != NTT_PLAIN || out_layout != NTT_PLAIN))
        throw std::runtime_error("zk_ntt: a transform that changes the layout cannot run in place");
    NttIoArgs io;
    io.batch = batch;
    io.kbits = kbits;
    io.row0 = row0;
    if (in_layout == NTT_BLOCKED_TW || out_layout == NTT_BLOCKED_TW) {
        if (in_layout == out_layout) throw std::runtime_error("zk_ntt: the four-step twiddle belongs to one side of a transform only");
        if (!big) throw std::runtime_error("zk_ntt: the blocked layout needs the plan of the large transform for its twiddles");
        if (kbits > L_) throw std::runtime_error("zk_ntt: block length exceeds the transform");
        if (((row0 + batch) << L_) > ((uint64_t)1 << big->L_)) throw std::runtime_error("zk_ntt: rows exceed the large transform");
        const int d = tw_inverse ? 1 : 0;
        io.twA = big->twA_[d].as<Fr>();
        io.twB = big->twB_[d].as<Fr>();
        io.lh = big->lh_;
    }
    launch_passes(d_in, d_out, inverse, batch, in_layout, out_layout, io, st);
    ZK_HIP(hipGetLastError());
}

// The pass-kernel instantiations in use: <FINAL, IN_CANON, IN_L, OUT_L>; buffers other than the first pass's input and the last
// pass's output are the plan's scratch (plain).
#define ZK_NTT_PASS_CASES(X)                                                                                                   \
    X(false, true, NTT_PLAIN, NTT_PLAIN) X(false, true, NTT_BLOCKED_TW, NTT_PLAIN) X(false, true, NTT_TRANSPOSED, NTT_PLAIN)  \
    X(false, false, NTT_PLAIN, NTT_PLAIN)                                                                                      \
    X(true, false, NTT_PLAIN, NTT_PLAIN) X(true, false, NTT_PLAIN, NTT_BLOCKED_TW) X(true, false, NTT_PLAIN, NTT_TRANSPOSED)   \
    X(true, true, NTT_PLAIN, NTT_PLAIN) X(true, true, NTT_PLAIN, NTT_BLOCKED_TW) X(true, true, NTT_PLAIN, NTT_TRANSPOSED)      \
    X(true, true, NTT_BLOCKED_TW, NTT_PLAIN) X(true, true, NTT_TRANSPOSED, NTT_PLAIN) X(true, true, NTT_TRANSPOSED, NTT_TRANSPOSED) \
    X(true, true, NTT_BLOCKED_TW, NTT_TRANSPOSED) X(true, true, NTT_TRANSPOSED, NTT_BLOCKED_TW)
// the plain passes of a plain transform again with T1K (2^10-element tiles)
#define ZK_NTT_PASS_CASES_1K(X) X(false, true, NTT_PLAIN, NTT_PLAIN) X(false, false, NTT_PLAIN, NTT_PLAIN) X(true, false, NTT_PLAIN, NTT_PLAIN)
static std::vector<const void *> pass_kernel_functions() {
    std::vector<const void *> v;
#define X(F, C, I, O) v.push_back(reinterpret_cast<const void *>(&ntt_pass_kernel<F, C, I, O>));
    ZK_NTT_PASS_CASES(X)
#undef X
#define X(F, C, I, O) v.push_back(reinterpret_cast<const void *>(&ntt_pass_kernel<F, C, I, O, true>));
    ZK_NTT_PASS_CASES_1K(X)
#undef X
    return v;
}
static void launch_pass(bool fin, bool canon, int in_l, int out_l, dim3 grid, size_t lds, hipStream_t st, const void *src, void *dst, const Fr *tile_tw,
                        const Fr *A, const Fr *B, Fr scale, const NttPassParams &P, const NttIoArgs &io, bool t1k) {
    if (t1k) {
#define X(F, C, I, O)                                                                                                                \
    if (fin == F && canon == C && in_l == I && out_l == O) {                                                                         \
        hipLaunchKernelGGL((ntt_pass_kernel<F, C, I, O, true>), grid, dim3(NTT_NT), lds, st, src, dst, tile_tw, A, B, scale, P, io); \
        return;                                                                                                                      \
    }
        ZK_NTT_PASS_CASES_1K(X)
#undef X
    }
#define X(F, C, I, O)                                                                                                          \
    if (fin == F && canon == C && in_l == I && out_l == O) {                                                                   \
        hipLaunchKernelGGL((ntt_pass_kernel<F, C, I, O>), grid, dim3(NTT_NT), lds, st, src, dst, tile_tw, A, B, scale, P, io); \
        return;                                                                                                                \
    }
    ZK_NTT_PASS_CASES(X)
#undef X
    throw std::runtime_error("zk_ntt: this combination of buffer layouts is not available");
}

void NttPlan::launch_passes(const void *d_in, void *d_out, bool inverse, unsigned batch, int in_layout, int out_layout, const NttIoArgs &io,
                            hipStream_t st) {
    const size_t n = (size_t)1 << L_;
    if (batch > 65535) throw std::runtime_error("zk_ntt: batch must be <= 65535");
    if (digits_.size() > 1 && tmp_.bytes < (size_t)batch * n * 32) {
        ZK_HIP(hipStreamSynchronize(st));  // the old scratch may still be in use
        tmp_.alloc((size_t)batch * n * 32);
    }
    const int dir = inverse ? 1 : 0;
    const unsigned D = (unsigned)digits_.size();
    const bool plain = in_layout == NTT_PLAIN && out_layout == NTT_PLAIN;
    // transposed buffers run along the batch index: up to four transforms side by side in a tile
    uint32_t gb_max = 0;
    while (gb_max < 2 && batch % (2u << gb_max) == 0) gb_max++;
    if (!(L_ > 0 || inverse || !plain)) return;
    uint32_t sp = L_;
    for (unsigned p = 0; p < D; p++) {
        const uint32_t lp = digits_[p];
        sp -= lp;
        NttPassParams P;
        memset(&P, 0, sizeof(P));
        P.L = L_; P.lp = lp; P.sp = sp; P.lh = lh_;
        P.tw_shift = lmax_ - lp;
        const bool first_pass = (p == 0), final_pass = (p == D - 1);
        const int il = first_pass ? in_layout : NTT_PLAIN, ol = final_pass ? out_layout : NTT_PLAIN;
        const bool transposing = il == NTT_TRANSPOSED || ol == NTT_TRANSPOSED;
        // tiles of up to 2^10 elements (every thread owns 4; 41 KB of LDS: three workgroups per CU), but never fewer than ~512 tiles per pass
        // (a batch in one of the four-step layouts has its tiles from all the transforms together)
        const int room = plain ? (int)L_ - (int)lp - 9 : 10 - (int)lp;
        P.g = (D == 1) ? 0 : (uint32_t)std::max<int>(NTT_G, std::min<int>(10 - (int)lp, room));
        if (transposing) {
            P.gb = std::min<uint32_t>(gb_max, 10 - lp);
            if (D == 1) P.g = P.gb;                                   // every carried column is a transform of its own
            else P.g = std::max(P.g, P.gb);
            if (!final_pass) P.g = std::min<uint32_t>(P.g, sp + P.gb);  // the element columns come out of the sp low bits
        }
        const void *src = first_pass ? d_in : static_cast<const void *>(tmp_.p);
        void *dst = final_pass ? d_out : tmp_.p;
        const uint32_t tile = 1u << (lp + P.g);
        static const bool no_1k = getenv("ZK_NTT_NO_T1K") != nullptr;    // A/B runs of the round-5 experiment (tools/ab_ntt.py)
        const bool t1k = !no_1k && tile == 1024 && il == NTT_PLAIN && ol == NTT_PLAIN && lp <= 8 && !(final_pass && first_pass);
        const size_t lds = ((size_t)NL * tile + (size_t)NL * (t1k ? (1u << NTT_TW_LDS_STAGES) : (1u << lp))) * sizeof(uint32_t);
        const unsigned blocks = (unsigned)(n >> (lp + P.g - P.gb));
        const dim3 grid(blocks, batch >> P.gb);
        const Fr *A = twA_[dir].as<Fr>(), *B = twB_[dir].as<Fr>();
        if (!final_pass) {
            if (inverse && p == 0) B = twB_scaled_inv_.as<Fr>();
            if (p < 3 && tw_direct_[dir][p].p) {
                P.direct_tw = 1;
                A = tw_direct_[dir][p].as<Fr>();
            }
        } else {
            P.l1 = (D == 1) ? 0 : digits_[0];
            P.nmid = D > 2 ? D - 2 : 0;
            for (unsigned q = 0; q < P.nmid; q++) P.lmid[q] = digits_[1 + q];
            P.apply_scale = (inverse && D == 1) ? 1 : 0;
        }
        launch_pass(final_pass, first_pass, il, ol, grid, lds, st, src, dst, tile_tw_[dir].as<Fr>(), A, B, scale_inv_, P, io, t1k);
    }
}

void NttPlan::twiddle_2d(void *d_data, unsigned log_cols, uint64_t rows, uint64_t row0, bool inverse, hipStream_t st) {
    const size_t total = (size_t)rows << log_cols;
    if (log_cols > L_ || ((row0 + rows) << log_cols) > ((uint64_t)1 << L_)) throw std::runtime_error("zk_ntt_twiddle: block exceeds the transform");
    if (total == 0) return;
    const int dir = inverse ? 1 : 0;
    hipLaunchKernelGGL(fr_twiddle_2d_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, static_cast<uint32_t *>(d_data), twA_[dir].as<Fr>(),
                       twB_[dir].as<Fr>(), lh_, log_cols, row0, total);
    ZK_HIP(hipGetLastError());
}

void fr_spmv(const void *d_row_ptr, const void *d_col, const void *d_vals, const void *d_x, void *d_y, size_t rows, hipStream_t st) {
    if (rows == 0) return;
    hipLaunchKernelGGL(fr_spmv_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, static_cast<const uint32_t *>(d_row_ptr),
                       static_cast<const uint32_t *>(d_col), static_cast<const uint32_t *>(d_vals), static_cast<const uint32_t *>(d_x),
                       static_cast<uint32_t *>(d_y), rows);
    ZK_HIP(hipGetLastError());
}

void fr_quotient(void *d_out, const void *d_a, const void *d_b, const void *d_c, const uint64_t zinv[4], size_t n, hipStream_t st) {
    HFr z;
    memcpy(z.l, zinv, 32);
    // device Montgomery radix is 2^261: zr -> z*2^261, zr2 -> z*2^522 (host radix 2^256 differs, so the
    // second factor 2^261 is multiplied in explicitly before the host->device conversion)
    const HFr zr = fe_to_mont(z);
    const HFr zr2 = fe_mul(zr, fe_to_mont(HFr::from_words(HostConst<FrTag>::to_dev())));
    if (n == 0) return;
    hipLaunchKernelGGL(fr_quotient_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, static_cast<uint32_t *>(d_out),
                       static_cast<const uint32_t *>(d_a), static_cast<const uint32_t *>(d_b), static_cast<const uint32_t *>(d_c), zr.to_dev(),
                       zr2.to_dev(), n);
    ZK_HIP(hipGetLastError());
}

}  // namespace zk
