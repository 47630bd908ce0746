// msm_fixed.h -- bound-bases mode of the MSM plan: the points of a prover's queries never change (CRS / SRS), so
// they are expanded ONCE into a table  T[w][i] = 2^(20 w) * P_i  (13 rows; 288 GB of HBM make 832 B per point cheap).
// Every signed 20-bit digit d_{i,w} of a scalar then selects the table entry (w, i) for ONE window of 2^19 buckets:
//   sum_i k_i P_i = sum_{i,w} d_{i,w} T[w][i]
// -- 13 n bucket additions instead of the 16 n of sixteen 16-bit windows (-19 %), the same 2^19 buckets to reduce, and
// a single window for the host fold.  The kernels after the partition (cell sort, rank, accumulate, heavy buckets,
// reduce) are the generic ones of msm_impl.h run with W = 1, nb = 2^19; this header adds the table builder, the digit
// kernel and a partition kernel for 2048 cells per window.
#pragma once
#include <stdlib.h>
#include "msm_impl.h"

namespace zk {

// The window width is a property of the group: G1 takes 20-bit digits (13 rows, 2^19 buckets).  A G2 addition costs three
// G1 additions in the accumulate kernel but a G2 bucket costs about the same three in the reduction, whose work grows with the
// bucket count and not with n: ZK_FIX_C_G2 bits (see profiles/r04_experiments.md) balance the two for the 2^20-point query.
#ifndef ZK_FIX_C_G2
#define ZK_FIX_C_G2 20
#endif
template <int C_>
struct FixCfg {
    static constexpr int C = C_;
    static constexpr int W = (255 + C - 1) / C;                    // rows of the table / digits per scalar (20: 13, 19: 14, 18: 15)
    static constexpr uint32_t NB = 1u << (C - 1);                  // buckets of the one window
    static constexpr uint32_t G = NB / SEG_BUCKETS;                // its cells
    static_assert(G <= MAX_CELLS, "the one window's cells must fit the scan kernel");
    static_assert(G == PREP_NT / 2 || G == PREP_NT || G == 2 * PREP_NT, "msm_fixed_partition_kernel holds G / 1024 cells per thread");
};
template <class F>
using FixOf = FixCfg<F::CANON_WORDS == 8 ? 20 : ZK_FIX_C_G2>;

// table[w * stride + i] = 2^(20 w) * P_i as a packed Montgomery affine point (infinity stays the all-zero encoding).
// The 12 multiples are reached by doubling on in XYZZ coordinates and brought back to affine together: one inversion per point
// (of the product of the twelve ZZZ, Montgomery's trick) instead of one per row -- the inversions were 70 % of the kernel
// (binding 2^20 G1 points: 40.6 -> 18.8 ms).
template <class F>
__global__ __launch_bounds__(64) void msm_fixed_table_kernel(const uint32_t *__restrict__ points, PackedAffine<F> *__restrict__ table, uint32_t n,
                                                             size_t stride) {
    constexpr int PW = F::CANON_WORDS;
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const F x = ld_canonical<F>(points + (size_t)i * 2 * PW), y = ld_canonical<F>(points + (size_t)i * 2 * PW + PW);
    const Affine<F> a = (x.is_zero() && y.is_zero()) ? Affine<F>::inf() : Affine<F>{fe_to_mont(x), fe_to_mont(y)};
    table[i] = pack_affine(Affine<F>{fe_reduce_full(a.x), fe_reduce_full(a.y)});
    constexpr int FIX_W = FixOf<F>::W, FIX_C = FixOf<F>::C;
    Xyzz<F> rows[FIX_W - 1];
    F before[FIX_W - 1];          // product of the ZZZ of the rows before this one
    Xyzz<F> q = Xyzz<F>::from_affine(a);
    F run = F::one();
#pragma unroll 1
    for (int w = 0; w < FIX_W - 1; w++) {
#pragma unroll 1
        for (int k = 0; k < FIX_C; k++) q = xyzz_dbl(q);
        rows[w] = q;
        before[w] = run;
        if (!q.is_inf()) run = fe_mul(run, q.zzz);
    }
    F inv = fe_inv(run);          // run is a product of non-zero values (1 when every row is infinity)
#pragma unroll 1
    for (int w = FIX_W - 2; w >= 0; w--) {
        const Xyzz<F> r = rows[w];
        Affine<F> out = Affine<F>::inf();
        if (!r.is_inf()) {
            const F izzz = fe_mul(inv, before[w]);
            inv = fe_mul(inv, r.zzz);
            const F izz = fe_sqr(fe_mul(r.zz, izzz));       // 1/ZZ = (ZZ/ZZZ)^2
            out = Affine<F>{fe_mul(r.x, izz), fe_mul(r.y, izzz)};
        }
        table[(size_t)(w + 1) * stride + i] = pack_affine(Affine<F>{fe_reduce_full(out.x), fe_reduce_full(out.y)});
    }
}

// digits[w * n_pad + i] = signed 20-bit digit w of scalar i (int32); cell_total[g] += digits whose bucket lies in cell g.
//
// spread (G1 only): a canonical scalar has 254 bits, so the top row's digits (bits 240..253) would fall on the lowest 2^14
// of the 2^19 buckets -- 64 extra entries on each, lists of ~90 where the mean is 26, and the threads that own them finish
// long after the rest of the grid (accumulate 1.31 ms where 13 n additions need 1.02).  Every point of G1 has order r
// (cofactor 1), so k P = (k + m r) P: scalar i is replaced by k + m_i r with m_i in [0, 40] taken from the index, which
// makes the top digit uniform over [0, 0.96 * 2^19) like every other row's.  Scalars below 2^240 (their top digit is zero:
// small / witness-like values keep their few non-zero digits) and non-canonical ones >= 2^254 are left alone.  Not for
// G2: the twist has a cofactor, and an input outside the order-r subgroup must still give the reference's result.
template <int C>
__global__ __launch_bounds__(PREP_NT) void msm_fixed_prepare_kernel(const uint32_t *__restrict__ scalars, int32_t *__restrict__ digits,
                                                                   uint32_t *__restrict__ cell_total, uint32_t n, uint32_t n_pad, bool spread) {
    constexpr int W = FixCfg<C>::W;
    constexpr uint32_t FIX_G = FixCfg<C>::G;
    __shared__ uint32_t hist[FIX_G];
    const uint32_t t = threadIdx.x;
    for (uint32_t k = t; k < FIX_G; k += PREP_NT) hist[k] = 0;
    __syncthreads();
#pragma unroll 1
    for (int rep = 0; rep < PREP_PPT; rep++) {
        const uint32_t i = (blockIdx.x * PREP_PPT + rep) * PREP_NT + t;
        if (i >= n) {
#pragma unroll
            for (int w = 0; w < W; w++) digits[(size_t)w * n_pad + i] = 0;
            continue;
        }
        uint32_t s[9];
        ld_words<8>(scalars + (size_t)i * 8, s);
        s[8] = 0;
        if (C == 20 && spread && (s[7] >> 16) != 0 && (s[7] >> 30) == 0) {
            // r as 32-bit words, least significant first
            constexpr uint32_t RW[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
            const uint32_t m = ((i * 0x9E3779B1u) >> 16) % 41u;
            uint64_t acc = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                acc += (uint64_t)m * RW[k] + s[k];
                s[k] = (uint32_t)acc;
                acc >>= 32;
            }
            s[8] = (uint32_t)acc;   // k + m r < 2^254 + 40 r < 2^259 - 2^240: the signed top digit stays below 2^19
        }
        uint32_t carry = 0;
#pragma unroll
        for (int w = 0; w < W; w++) {
            constexpr uint32_t mask = (1u << C) - 1u;
            const int off = w * C, word = off >> 5, sh = off & 31;
            uint32_t raw = 0;
            if (word < 9) {
                raw = s[word] >> sh;
                if (sh + C > 32 && word + 1 < 9) raw |= s[word + 1] << (32 - sh);
            }
            raw &= mask;
            const uint32_t v = raw + carry;
            int d;
            if (v >= (1u << (C - 1))) {
                d = (int)v - (1 << C);
                carry = 1;
            } else {
                d = (int)v;
                carry = 0;
            }
            digits[(size_t)w * n_pad + i] = d;
            if (d != 0) atomicAdd(&hist[((uint32_t)(d < 0 ? -d : d) - 1u) >> SEG_LOG], 1u);
        }
    }
    __syncthreads();
    for (uint32_t k = t; k < FIX_G; k += PREP_NT) {
        const uint32_t h = hist[k];
        if (h) atomicAdd(&cell_total[k], h);
    }
}

// Measurement aid (profiles/r05_experiments.md): with ZK_MSM_ALIAS_TABLE_LOG=k in the environment every table index is folded onto the
// first 2^k entries of the table, so that the accumulate kernel's gathers stay inside a cache-sized window -- the RESULT is then
// garbage, the kernel's time says what the gathers from the full table cost.  Unset (always, outside that experiment): no effect.
inline uint32_t fixed_alias_mask() {
    static const uint32_t mask = [] {
        const char *e = getenv("ZK_MSM_ALIAS_TABLE_LOG");
        const int k = e ? atoi(e) : 0;
        return (k > 0 && k < 31) ? ((1u << k) - 1u) : 0xffffffffu;
    }();
    return mask;
}

// One workgroup = 1024 * PPT consecutive entries of the flat (window-major) digit array.  Ranks them by cell in LDS, reserves
// the cells' spans and writes the entries out cell by cell; the stored index is the TABLE row  w * stride + first + i
// (| sign << 31), so the generic accumulate kernel gathers straight from the table.  With 2048 cells a workgroup's share
// of a cell is only entries / 2048 long: PPT = 8 (8192 entries) gives 16-byte runs and half the cursor atomics of PPT = 4.
template <int PPT, int C>
__global__ __launch_bounds__(PREP_NT) void msm_fixed_partition_kernel(const int32_t *__restrict__ digits, SortBufs B, uint32_t n_pad, size_t stride,
                                                                     uint32_t first, uint32_t total, uint32_t alias_mask /* 0xffffffff; see fixed_alias_mask() */) {
    constexpr int NE = PREP_NT * PPT;
    constexpr uint32_t FIX_G = FixCfg<C>::G;
    constexpr int CPT = FIX_G > PREP_NT ? (int)(FIX_G / PREP_NT) : 1;   // cells per thread (threads past the last cell hold none)
    __shared__ uint32_t hist[FIX_G];   // counts, then exclusive offsets
    __shared__ uint32_t gpos[FIX_G];
    __shared__ uint32_t wave_tot[PREP_NT / 64 + 1];
    __shared__ uint32_t stage_idx[NE];
    __shared__ uint16_t stage_cell[NE];
    __shared__ uint8_t stage_loc[NE];
    const uint32_t t = threadIdx.x;
    const uint32_t v0 = blockIdx.x * NE;
    const bool owner = t * CPT < FIX_G;
    // span starts of this thread's cells: every workgroup scans the cell totals of the prepare kernel itself (no scan launch
    // in between; workgroup 0 publishes them for the cell sort, msm_cellsort_kernel zeroes the counters for the next run)
    uint32_t cb[CPT];
    {
        uint32_t cnt[CPT], sum = 0;
#pragma unroll
        for (int k = 0; k < CPT; k++) {
            cnt[k] = owner ? B.cell_total[CPT * t + k] : 0u;
            cb[k] = sum;
            sum += (cnt[k] + 15u) & ~15u;   // 16-entry aligned spans (the cell sort fetches 16 entries per load)
        }
        uint32_t all;
        const uint32_t base = block_exclusive_scan<PREP_NT>(sum, wave_tot, &all);
#pragma unroll
        for (int k = 0; k < CPT; k++) {
            cb[k] += base;
            if (blockIdx.x == 0 && owner) {
                B.cell_base[CPT * t + k] = cb[k];
                B.cell_cnt[CPT * t + k] = cnt[k];
            }
        }
    }
    for (uint32_t k = t; k < FIX_G; k += PREP_NT) hist[k] = 0;
    __syncthreads();
    uint32_t rk[PPT], jj[PPT];
    int dd[PPT];
#pragma unroll
    for (int rep = 0; rep < PPT; rep++) {
        const uint32_t v = v0 + t + rep * PREP_NT;
        dd[rep] = v < total ? digits[v] : 0;
        jj[rep] = (uint32_t)(dd[rep] < 0 ? -dd[rep] : dd[rep]) - 1u;
        rk[rep] = dd[rep] != 0 ? atomicAdd(&hist[jj[rep] >> SEG_LOG], 1u) : 0u;
    }
    __syncthreads();
    // exclusive scan of the cell counts, CPT per thread
    uint32_t h[CPT], hsum = 0;
#pragma unroll
    for (int k = 0; k < CPT; k++) {
        h[k] = owner ? hist[CPT * t + k] : 0u;
        hsum += h[k];
    }
    uint32_t total_here;
    uint32_t ex = block_exclusive_scan<PREP_NT>(hsum, wave_tot, &total_here);
#pragma unroll
    for (int k = 0; k < CPT; k++) {
        if (owner) {
            hist[CPT * t + k] = ex;
            gpos[CPT * t + k] = h[k] ? cb[k] + atomicAdd(&B.cell_cursor[CPT * t + k], h[k]) : 0u;
        }
        ex += h[k];
    }
    __syncthreads();
#pragma unroll
    for (int rep = 0; rep < PPT; rep++) {
        if (dd[rep] != 0) {
            const uint32_t cellg = jj[rep] >> SEG_LOG;
            const uint32_t p = hist[cellg] + rk[rep];
            const uint32_t v = v0 + t + rep * PREP_NT, w = v / n_pad, i = v - w * n_pad;   // a workgroup may straddle two rows
            const size_t row = (size_t)w * stride + first + i;
            stage_idx[p] = ((uint32_t)row & alias_mask) | (dd[rep] < 0 ? 0x80000000u : 0u);
            stage_loc[p] = (uint8_t)(jj[rep] & (SEG_BUCKETS - 1));
            stage_cell[p] = (uint16_t)cellg;
        }
    }
    __syncthreads();
    for (uint32_t p = t; p < total_here; p += PREP_NT) {
        const uint32_t cellg = stage_cell[p];
        const uint32_t dst = gpos[cellg] + (p - hist[cellg]);
        B.e_idx[dst] = stage_idx[p];
        B.e_loc[dst] = stage_loc[p];
    }
}

}  // namespace zk
