// msm_impl.h -- Pippenger multi-scalar multiplication on BN254 G1 / G2 for gfx950.
//
// Replaces the reference's scalar-mul-and-add loops (zkp/plonk/kzg.py:59-65,
// zkp/groth16/proving.py:23-75).  One MSM = one pass through these kernels on the stream of the plan
// lane it was submitted to, with no host synchronisation until the 36 KiB read-back:
//
//   prepare      one thread per point: canonical affine -> Montgomery affine packed to one 64-byte line (resident
//                for all windows), scalar -> W signed c-bit digits (int16, window-major), per-cell histogram.
//   bucket sort  partition (by cell = 256 buckets of one window, LDS-staged; scans the cell totals itself) -> cell sort (one
//                workgroup per cell, single pass through LDS; cells beyond 10240 entries go to the multi-workgroup count /
//                scatter pair) -> scan + rank (buckets by list length).
//   accumulate   one thread per bucket, one wavefront per workgroup, lists of equal length side by side; XYZZ mixed
//                additions (8M+2S).  Lists longer than heavy_th are summed by the first workgroups of the same grid (wavefront
//                tasks of 64 segments, __shfl trees, per-bucket last-arriver combine), beside the ordinary lists.
//   reduce       sum_j (j+1)*B_j per window without any serial running sum, in place (msm_reduce.h): the index bits are folded
//                from the top down, every fold leaving behind the plain sum of one bit's buckets; the lower levels one lane per
//                addition, the upper ones on teams of two and four lanes (DPP exchange) that shorten the dependent chain.
//   fold (host)  the W*c window/level sums are read back and combined by one 254-doubling Horner pass on the host
//                (a single GPU thread would be latency-bound); it hides behind the next MSM's kernels.
//
// A plan holds three lanes (workspace + stream each): consecutive submissions overlap on the GPU, and MSMs of more
// than 2^22 points run as 2^22-point chunks through the same lanes.
#pragma once
#include "common.h"
#include "curve.h"
#include "host_field.h"
#include "msm.h"
#include "msm_reduce.h"

namespace zk {

// ------------------------------------------------------------------------------ device helpers
// NW canonical 32-bit words (NW = 8 or 16) with 16-byte loads.
template <int NW> __device__ __forceinline__ void ld_words(const uint32_t *p, uint32_t w[NW]) {
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
#pragma unroll
    for (int i = 0; i < NW / 4; i++) {
        const uint4 v = q[i];
        w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w;
    }
}
// Montgomery-form affine point as stored in the workspace: the lazy coordinates (< 2m < 2^255) are
// packed back to 32-bit words, so a G1 point is exactly one aligned 64-byte line (one HBM/L2
// transaction per gather instead of the two a 72-byte limb image straddles).
template <class F> struct alignas(16) PackedAffine {
    uint32_t w[2 * F::CANON_WORDS];
};
__device__ __forceinline__ void pack_fe(uint32_t *w, const Fp &a) { fe_to_words(a, w); }
__device__ __forceinline__ void pack_fe(uint32_t *w, const Fp2 &a) {
    fe_to_words(a.c0, w);
    fe_to_words(a.c1, w + 8);
}
template <class F> __device__ __forceinline__ PackedAffine<F> pack_affine(const Affine<F> &p) {
    PackedAffine<F> r;
    pack_fe(r.w, p.x);
    pack_fe(r.w + F::CANON_WORDS, p.y);
    return r;
}
template <class F> __device__ __forceinline__ Affine<F> unpack_affine(const PackedAffine<F> &p) {
    return Affine<F>{fe_load_canonical(p.w, (F *)nullptr), fe_load_canonical(p.w + F::CANON_WORDS, (F *)nullptr)};
}

template <class F> __device__ __forceinline__ F ld_canonical(const uint32_t *p) {
    uint32_t w[F::CANON_WORDS];
    ld_words<F::CANON_WORDS>(p, w);
    return fe_load_canonical(w, (F *)nullptr);
}

// ------------------------------------------------------------------------------ block scan
// Exclusive scan of one value per thread over a block of NT threads (NT multiple of 64).
// Returns the exclusive prefix; *total receives the block total (same in every thread).
template <int NT> __device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t *wave_tot /* NT/64+1 in LDS */, uint32_t *total) {
    const uint32_t lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t o = __shfl_up(inc, d, 64);
        if (lane >= (uint32_t)d) inc += o;
    }
    if (lane == 63) wave_tot[wid] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < NT / 64; k++) {
        uint32_t t = wave_tot[k];
        if ((uint32_t)k < wid) base += t;
        tot += t;
    }
    *total = tot;
    __syncthreads();
    return base + inc - v;
}

// ------------------------------------------------------------------------------ bucket sort
// Two-level LDS-staged counting sort of the (window, bucket) -> point-index lists.
//
//   cells      a cell = SEG_BUCKETS (256) consecutive buckets of one window; <= 2048 cells.
//   prepare    (above all) also histograms the digits per cell: LDS histogram per 4096-point
//              workgroup, one global atomic per non-empty cell.
//   partition  coarse radix pass: every (window, point) entry is written to its cell's segment
//              (index | sign << 31 in e_idx, bucket-in-cell byte in e_loc); a workgroup reserves one
//              contiguous span per cell, so writes are short runs, never single scattered words.
//   cellsort   one workgroup sorts a whole cell (<= CS_MAX entries) in one pass: entries in registers, LDS-atomic ranks,
//              scan of the 256 counters (-> counts[], bucket_off[]), LDS image of the sorted cell, coalesced write.
//   segcount / segscatter  (cells beyond CS_MAX entries only, from the list msm_cellsort_kernel writes)
//              up to SEG_Z workgroups per cell take its 4096-entry chunks round-robin: count the 256 buckets
//              (-> counts[], bucket_off[]), then sort every chunk entirely in LDS (LDS-atomic ranks, block
//              scan, staging buffer) and write it out as per-bucket runs -- coalesced 16-byte reads,
//              run-coalesced writes, no global atomics.
//   rank       counting sort of the bucket ids by list length, longest first (perm[]), so that the
//              accumulate kernel's wavefronts own equal-length lists; also registers heavy buckets.
constexpr int SEG_BUCKETS = 256;
constexpr int SEG_LOG = 8;
constexpr int MAX_CELLS = 2048;
constexpr int PREP_NT = 1024;
constexpr int PREP_PPT = 4;  // points per thread in prepare / partition
constexpr int SEG_NT = 256;
constexpr int SEG_EPT = 16;  // entries per thread per chunk
constexpr int SEG_CHUNK = SEG_NT * SEG_EPT;
constexpr int SIZE_BINS = 1024;  // list lengths >= SIZE_BINS-1 share the top bin
constexpr int SEG_Z = 8;         // workgroups sharing one cell (they split its chunks round-robin)
constexpr int CS_NT = 512;       // single-workgroup cell sort: threads,
constexpr int CS_EPT = 20;       //   entries per thread (five 16-byte loads),
constexpr int CS_MAX = CS_NT * CS_EPT;   // and the largest cell it takes (10240 entries; uniform 2^20-point MSMs have 8192 +- 90 per cell)
constexpr int HEAVY_SEG = 32;    // entries per heavy-bucket segment (one thread each)
constexpr int HEAVY_WAVE = 64 * HEAVY_SEG;  // entries per wavefront task
constexpr uint32_t HEAVY_BLOCKS = 1024;     // workgroups at the head of the accumulate grid that take the heavy-bucket tasks

struct SortBufs {
    uint32_t *counts;       // [W*nb]   list length of every bucket
    uint32_t *bucket_off;   // [W*nb]   start of every bucket's list inside sorted[]
    uint32_t *cell_total;   // [cells]  entries per cell; accumulated by prepare, zeroed by msm_cellsort_kernel
    uint32_t *cell_base;    // [cells]  exclusive scan of the padded cell_total (published by partition workgroup 0)
    uint32_t *cell_cnt;     // [cells]  copy of cell_total for the later passes
    uint32_t *cell_cursor;  // [cells]  partition write cursors (zeroed by msm_cellsort_kernel)
    uint32_t *e_idx;        // [W*n]    partitioned entries: point index | sign << 31
    uint8_t *e_loc;         // [W*n]    partitioned entries: bucket index inside the cell
    uint32_t *sorted;       // [W*n]    entries grouped by bucket
    uint32_t *zcount;       // [cells][SEG_Z][SEG_BUCKETS] per-workgroup bucket counts of the two-kernel cell sort
    uint32_t *size_hist;    // [SIZE_BINS] buckets per list length (zeroed by the scan kernel after use)
    uint32_t *size_base;    // [SIZE_BINS] first rank of each length, longest first
    uint32_t *size_cursor;  // [SIZE_BINS] running reservation (zeroed by the scan kernel)
    uint32_t *perm;         // [W*nb]   bucket ids ordered by decreasing list length
    // Heavy buckets (list longer than heavy_th: skewed / witness-like scalars, degenerate top window):
    // their lists are cut into segments of HEAVY_SEG entries, one thread per segment, partials combined
    // by one workgroup per bucket.  They take rank "length 0" in perm[] so the main kernel skips them.
    uint32_t heavy_th;
    uint32_t heavy_cap;     // capacity of heavy_tasks / heavy_buckets (entries)
    uint32_t *heavy_ctr;    // [2] number of heavy tasks, number of heavy buckets (zeroed by the scan kernel)
    uint32_t *big_ctr;      // [1] cells too large for msm_cellsort_kernel (zeroed by the scan kernel)
    uint32_t *big_cells;    // [cells] their ids: the work list of msm_segcount_kernel / msm_segscatter_kernel
    uint2 *heavy_tasks;     // [heavy_cap] (heavy-bucket slot, 64-segment group of its list)
    uint4 *heavy_buckets;   // [heavy_cap] (bucket id, first task, segments, -)
    // Device-side error counter of the lane (never reset: the host compares it with the value it saw last).  Bumped when an
    // input breaks a precondition the host cannot check on device buffers (a scalar >= 2^255: the signed-digit carry leaves
    // the top window) or when a heavy bucket does not fit the task arrays; the submission then fails at collect.
    uint32_t *err;
};
__device__ __forceinline__ uint32_t size_bin(uint32_t c, uint32_t heavy_th) {
    return c > heavy_th ? 0u : min(c, (uint32_t)SIZE_BINS - 1u);
}

// ------------------------------------------------------------------------------ prepare
// digits[w * n_pad + i] = signed digit d in [-2^(C-1), 2^(C-1)-1] of scalar i for window w;
// cell_total[w * G + g] += number of non-zero digits of window w whose bucket |d|-1 lies in cell g.
template <class F, int C>
__global__ __launch_bounds__(PREP_NT) void msm_prepare_kernel(const uint32_t *__restrict__ scalars,
                                                              const uint32_t *__restrict__ points,
                                                              PackedAffine<F> *__restrict__ pts_m,
                                                              int16_t *__restrict__ digits, uint32_t *__restrict__ cell_total,
                                                              uint32_t *__restrict__ err, uint32_t n, uint32_t n_pad) {
    constexpr int W = (255 + C - 1) / C;
    constexpr int PW = F::CANON_WORDS;
    constexpr int G = ((1 << (C - 1)) + SEG_BUCKETS - 1) / SEG_BUCKETS;
    static_assert(W * G <= MAX_CELLS, "too many cells");
    __shared__ uint32_t hist[W * G];
    const uint32_t t = threadIdx.x;
    for (uint32_t k = t; k < W * G; k += PREP_NT) hist[k] = 0;
    __syncthreads();
#pragma unroll 1
    for (int rep = 0; rep < PREP_PPT; rep++) {
        const uint32_t i = (blockIdx.x * PREP_PPT + rep) * PREP_NT + t;
        if (i >= n) {  // padding (n_pad is a multiple of the workgroup's 4096 points)
#pragma unroll
            for (int w = 0; w < W; w++) digits[(size_t)w * n_pad + i] = 0;
            continue;
        }
        const F x = ld_canonical<F>(points + (size_t)i * 2 * PW);
        const F y = ld_canonical<F>(points + (size_t)i * 2 * PW + PW);
        const bool inf = x.is_zero() && y.is_zero();  // canonical inputs: infinity is the all-zero encoding
        pts_m[i] = pack_affine(Affine<F>{fe_to_mont(x), fe_to_mont(y)});

        uint32_t s[8];
        ld_words<8>(scalars + (size_t)i * 8, s);
        if (C == 16 && PW == 8 && (i & 1u) && (s[7] >> 16) != 0 && (s[7] >> 30) == 0) {
            // G1, 16-bit windows: a canonical scalar leaves the top window (bits 240..253) only r >> 240 = 12388 of its 2^15
            // buckets, whose lists are then 2.6 times the mean length.  Every point of G1 has order r, so odd-numbered
            // scalars are run as k + r (< 2^255: the signed top digit still fits): twice the buckets, lists of 42.  Scalars
            // below 2^240 (small / witness-like values) keep their few non-zero digits.
            constexpr uint32_t RW[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
            uint64_t acc = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                acc += (uint64_t)RW[k] + s[k];
                s[k] = (uint32_t)acc;
                acc >>= 32;
            }
        }
        uint32_t carry = 0;
#pragma unroll
        for (int w = 0; w < W; w++) {
            constexpr uint32_t mask = (1u << C) - 1u;
            const int off = w * C, word = off >> 5, sh = off & 31;
            uint32_t raw = 0;
            if (word < 8) {
                raw = s[word] >> sh;
                if (sh + C > 32 && word + 1 < 8) raw |= s[word + 1] << (32 - sh);
            }
            raw &= mask;
            uint32_t v = raw + carry;
            int d;
            if (v >= (1u << (C - 1))) {
                d = (int)v - (1 << C);
                carry = 1;
            } else {
                d = (int)v;
                carry = 0;
            }
            if (inf) d = 0;
            digits[(size_t)w * n_pad + i] = (int16_t)d;
            if (d != 0) atomicAdd(&hist[w * G + (((uint32_t)(d < 0 ? -d : d) - 1u) >> SEG_LOG)], 1u);
        }
        // a canonical scalar (< r < 2^254, or k + r < 2^255 above) never carries out of the top window; one that does
        // (>= 2^255 for c = 15, 16) would silently lose 2^(W*C) * P
        if ((carry || (W * C < 256 && (s[7] >> (W * C - 224)) != 0)) && !inf) atomicAdd(err, 1u);
    }
    __syncthreads();
    for (uint32_t k = t; k < W * G; k += PREP_NT) {
        const uint32_t h = hist[k];
        if (h) atomicAdd(&cell_total[k], h);
    }
}

// One workgroup, after the cell sort: exclusive scan of the list-length histogram in DEcreasing length order (rank 0 = longest
// lists).  Leaves the counters it consumed zeroed for the next run (they start zeroed at plan creation).  (The scan of the cell
// totals between prepare and partition was a launch of its own until round 3; the partition kernels do it themselves now.)
template <bool SIZES>
__global__ __launch_bounds__(1024) void msm_scan_kernel(SortBufs B, uint32_t ncells) {
    static_assert(SIZES, "only the list-length scan is a kernel of its own");
    __shared__ uint32_t wave_tot[1024 / 64 + 1];
    const uint32_t t = threadIdx.x;
    uint32_t total;
    const uint32_t bin = SIZE_BINS - 1u - t;  // thread 0 takes the longest lists
    const uint32_t hv = B.size_hist[bin];
    const uint32_t hx = block_exclusive_scan<1024>(hv, wave_tot, &total);
    B.size_base[bin] = hx;
    B.size_hist[bin] = 0;
    B.size_cursor[bin] = 0;
    if (t < 2) B.heavy_ctr[t] = 0;
    if (t == 2) *B.big_ctr = 0;
}

// Coarse radix pass, grid = (n_pad / 8192, W): one workgroup takes 8192 consecutive digits of ONE window (16 KB, read with 16-byte
// loads), ranks them by cell in LDS (LDS-atomic ranks + block scan), reserves the cells' global spans (one atomic per non-empty cell),
// stages the entries in cell order and writes them out with consecutive lanes on consecutive addresses: runs of 8192 / G = 64 entries
// per cell.  (Until round 3 a workgroup walked all W windows of 4096 points, seven barriers of a 1024-thread workgroup per window:
// 81 us for 2^20 points where the traffic needs 40.)  The start of every cell's span is the exclusive scan of the 16-entry padded
// cell totals of the prepare kernel, which every workgroup computes for its window itself (no scan launch between the two kernels);
// the first workgroup of a window publishes it for the cell sort.
constexpr int PART_PTS = 8192;
template <int DUMMY>
__global__ __launch_bounds__(PREP_NT) void msm_partition_kernel(const int16_t *__restrict__ digits, SortBufs B, uint32_t n_pad, uint32_t W,
                                                                uint32_t G) {
    constexpr int EPT = PART_PTS / PREP_NT;  // 8 entries per thread
    __shared__ uint32_t hist[128];           // G <= 128 cells per window: counts, then exclusive offsets
    __shared__ uint32_t gpos[128];           // global position of this workgroup's span in every cell
    __shared__ uint32_t wave_tot[PREP_NT / 64 + 1];
    __shared__ uint32_t stage_idx[PART_PTS];
    __shared__ uint8_t stage_loc[PART_PTS];
    __shared__ uint8_t stage_cell[PART_PTS];
    const uint32_t t = threadIdx.x, w = blockIdx.y;
    const uint32_t i0 = blockIdx.x * PART_PTS + EPT * t;   // this thread's 8 consecutive points
    uint4 dv = make_uint4(0u, 0u, 0u, 0u);
    if (i0 < n_pad) dv = *reinterpret_cast<const uint4 *>(digits + (size_t)w * n_pad + i0);   // n_pad is a multiple of 4096
    // span starts of this window's cells
    uint32_t cb, cell_n = 0;
    {
        const uint32_t ncells = W * G, first = w * G;
        const uint32_t v0 = 2 * t < ncells ? B.cell_total[2 * t] : 0u, v1 = 2 * t + 1 < ncells ? B.cell_total[2 * t + 1] : 0u;
        const uint32_t before = (2 * t < first ? ((v0 + 15u) & ~15u) : 0u) + (2 * t + 1 < first ? ((v1 + 15u) & ~15u) : 0u);
        uint32_t base_w, win_tot;
        (void)block_exclusive_scan<PREP_NT>(before, wave_tot, &base_w);          // padded entries of all earlier windows
        if (t < G) cell_n = B.cell_total[first + t];
        const uint32_t ex = block_exclusive_scan<PREP_NT>(t < G ? ((cell_n + 15u) & ~15u) : 0u, wave_tot, &win_tot);
        cb = base_w + ex;
        if (blockIdx.x == 0 && t < G) {
            B.cell_base[first + t] = cb;
            B.cell_cnt[first + t] = cell_n;
        }
    }
    if (t < 128) hist[t] = 0;
    __syncthreads();
    const uint32_t dw[4] = {dv.x, dv.y, dv.z, dv.w};
    uint32_t rk[EPT], jj[EPT];
    int dd[EPT];
#pragma unroll
    for (int k = 0; k < EPT; k++) {
        dd[k] = (int)(int16_t)(dw[k >> 1] >> (16 * (k & 1)));
        jj[k] = (uint32_t)(dd[k] < 0 ? -dd[k] : dd[k]) - 1u;
        rk[k] = dd[k] != 0 ? atomicAdd(&hist[jj[k] >> SEG_LOG], 1u) : 0u;
    }
    __syncthreads();
    const uint32_t h = t < 128 ? hist[t] : 0u;
    uint32_t total;
    const uint32_t ex = block_exclusive_scan<PREP_NT>(h, wave_tot, &total);
    if (t < 128) {
        hist[t] = ex;
        gpos[t] = (h && t < G) ? cb + atomicAdd(&B.cell_cursor[w * G + t], h) : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < EPT; k++) {
        if (dd[k] != 0) {
            const uint32_t cellg = jj[k] >> SEG_LOG;
            const uint32_t p = hist[cellg] + rk[k];
            stage_idx[p] = (i0 + k) | (dd[k] < 0 ? 0x80000000u : 0u);
            stage_loc[p] = (uint8_t)(jj[k] & (SEG_BUCKETS - 1));
            stage_cell[p] = (uint8_t)cellg;
        }
    }
    __syncthreads();
    for (uint32_t p = t; p < total; p += PREP_NT) {
        const uint32_t cellg = stage_cell[p];
        const uint32_t dst = gpos[cellg] + (p - hist[cellg]);
        B.e_idx[dst] = stage_idx[p];
        B.e_loc[dst] = stage_loc[p];
    }
}

// Cell sort, common case, grid = (G, W): ONE workgroup sorts a whole cell of up to CS_MAX entries in a single pass -- every thread
// keeps its 20 entries in registers (16-byte loads: cell spans are 16-entry aligned), ranks them with LDS atomics on the 256 bucket
// counters, the counters are scanned (-> counts, bucket_off, list-length histogram), the entries are placed in an LDS image of the
// sorted cell and that image is written out in order: one read and one fully coalesced write of the cell, no per-workgroup count
// table, no second kernel.  Larger cells (MSMs beyond ~1.2 M entries per window, hot digits of skewed scalars) are left to the
// multi-workgroup pair below, which skips the cells done here.  Also zeroes the cell's partition counters for the next run (every
// partition workgroup is done with them: stream order).
template <int DUMMY>
__global__ __launch_bounds__(CS_NT) void msm_cellsort_kernel(SortBufs B, uint32_t nb) {
    __shared__ uint32_t cnt[SEG_BUCKETS];   // entries per bucket, then exclusive offsets
    __shared__ uint32_t hist[SIZE_BINS];
    __shared__ uint32_t wave_tot[CS_NT / 64 + 1];
    __shared__ uint32_t img[CS_MAX];        // the sorted cell
    const uint32_t t = threadIdx.x, g = blockIdx.x, w = blockIdx.y, G = gridDim.x;
    const uint32_t cellid = w * G + g;
    const uint32_t seg0 = B.cell_base[cellid], seg_n = B.cell_cnt[cellid];
    if (t == 0) {
        B.cell_total[cellid] = 0;
        B.cell_cursor[cellid] = 0;
    }
    if (seg_n > CS_MAX) {   // left to the multi-workgroup pair, which walks this list
        if (t == 0) B.big_cells[atomicAdd(B.big_ctr, 1u)] = cellid;
        return;
    }
    const uint32_t base = g * SEG_BUCKETS;
    const uint32_t nloc = min((uint32_t)SEG_BUCKETS, nb - base);
    const size_t flat0 = (size_t)w * nb + base;
    if (t < SEG_BUCKETS) cnt[t] = 0;
    for (uint32_t k = t; k < SIZE_BINS; k += CS_NT) hist[k] = 0;
    __syncthreads();
    // slab s holds entries s * 4 * CS_NT + 4 t .. + 3 of thread t: consecutive threads, consecutive 16-byte pieces
    constexpr int SLABS = CS_EPT / 4;
    uint32_t e_i[CS_EPT], e_r[CS_EPT];
    uint32_t e_l[SLABS];                     // four bucket bytes per slab
    const uint8_t *__restrict__ loc = B.e_loc + seg0;
    const uint32_t *__restrict__ idx = B.e_idx + seg0;
#pragma unroll
    for (int sl = 0; sl < SLABS; sl++) {
        const uint32_t e0 = (uint32_t)sl * 4 * CS_NT + 4 * t;
        e_l[sl] = 0;
        if (e0 < seg_n) {
            const uint4 iv = *reinterpret_cast<const uint4 *>(idx + e0);
            e_l[sl] = *reinterpret_cast<const uint32_t *>(loc + e0);
            e_i[4 * sl] = iv.x; e_i[4 * sl + 1] = iv.y; e_i[4 * sl + 2] = iv.z; e_i[4 * sl + 3] = iv.w;
        }
    }
#pragma unroll
    for (int k = 0; k < CS_EPT; k++) {
        const uint32_t e = (uint32_t)(k >> 2) * 4 * CS_NT + 4 * t + (k & 3);
        if (e < seg_n) e_r[k] = atomicAdd(&cnt[(e_l[k >> 2] >> (8 * (k & 3))) & 0xffu], 1u);
    }
    __syncthreads();
    const uint32_t c = t < SEG_BUCKETS ? cnt[t] : 0u;
    uint32_t total;
    const uint32_t ex = block_exclusive_scan<CS_NT>(c, wave_tot, &total);
    if (t < SEG_BUCKETS) cnt[t] = ex;
    if (t < nloc) {
        B.counts[flat0 + t] = c;
        B.bucket_off[flat0 + t] = seg0 + ex;
        atomicAdd(&hist[size_bin(c, B.heavy_th)], 1u);
    }
    __syncthreads();
    for (uint32_t k = t; k < SIZE_BINS; k += CS_NT) {
        const uint32_t h = hist[k];
        if (h) atomicAdd(&B.size_hist[k], h);
    }
#pragma unroll
    for (int k = 0; k < CS_EPT; k++) {
        const uint32_t e = (uint32_t)(k >> 2) * 4 * CS_NT + 4 * t + (k & 3);
        if (e < seg_n) img[cnt[(e_l[k >> 2] >> (8 * (k & 3))) & 0xffu] + e_r[k]] = e_i[k];
    }
    __syncthreads();
    uint32_t *__restrict__ sorted = B.sorted + seg0;   // 16-entry aligned span; the padding behind seg_n is never read
    for (uint32_t e0 = 4 * t; e0 < seg_n; e0 += 4 * CS_NT)
        *reinterpret_cast<uint4 *>(sorted + e0) = make_uint4(img[e0], img[e0 + 1], img[e0 + 2], img[e0 + 3]);
}

// Cell sort, large cells, grid = (SEG_Z, SEG_LIST): the cells msm_cellsort_kernel registered as too large (normally none: both
// kernels then return at once) are taken from its list; up to SEG_Z workgroups share a cell and take its 4096-entry chunks
// round-robin, so a cell swollen by skewed scalars (a hot digit) is still sorted by several CUs.
// Kernel 1 counts: zcount[cell][z][b] = entries of bucket b in the chunks of workgroup z.
constexpr int SEG_LIST = 256;    // grid.y of the pair: list entries are taken round-robin
template <int DUMMY>
__global__ __launch_bounds__(SEG_NT) void msm_segcount_kernel(SortBufs B) {
    __shared__ uint32_t hist[SEG_BUCKETS];
    const uint32_t t = threadIdx.x, z = blockIdx.x, nbig = *B.big_ctr;
    for (uint32_t ci = blockIdx.y; ci < nbig; ci += gridDim.y) {
        const uint32_t cellid = B.big_cells[ci];
        const uint32_t seg0 = B.cell_base[cellid], seg_n = B.cell_cnt[cellid];
        const uint32_t nchunks = (seg_n + SEG_CHUNK - 1) / SEG_CHUNK;
        if (z < nchunks) {
            const uint8_t *__restrict__ loc = B.e_loc + seg0;
            hist[t] = 0;
            __syncthreads();
            for (uint32_t ch = z; ch < nchunks; ch += SEG_Z) {
                const uint32_t c0 = ch * SEG_CHUNK, cn = min((uint32_t)SEG_CHUNK, seg_n - c0);
                if (SEG_EPT * t < cn) {
                    const uint4 lv = *reinterpret_cast<const uint4 *>(loc + c0 + SEG_EPT * t);  // thread t owns entries 16t .. 16t+15
                    const uint32_t lw[4] = {lv.x, lv.y, lv.z, lv.w};
#pragma unroll
                    for (int k = 0; k < SEG_EPT; k++)
                        if (SEG_EPT * t + k < cn) atomicAdd(&hist[(lw[k >> 2] >> (8 * (k & 3))) & 0xffu], 1u);
                }
            }
            __syncthreads();
            B.zcount[((size_t)cellid * SEG_Z + z) * SEG_BUCKETS + t] = hist[t];
        }
        __syncthreads();
    }
}

// Kernel 2 scatters: bucket starts from the summed counts (workgroup 0 of the cell also publishes
// counts / bucket_off / the list-length histogram), then every workgroup sorts its chunks in LDS
// (LDS-atomic ranks, block scan, staging buffer) and writes them out as per-bucket runs behind the
// runs of the workgroups before it.  The order inside a bucket is irrelevant to the sum.
template <int DUMMY>
__global__ __launch_bounds__(SEG_NT) void msm_segscatter_kernel(SortBufs B, uint32_t nb, uint32_t G) {
    __shared__ uint32_t cur[SEG_BUCKETS];      // running global write position of every bucket
    __shared__ uint32_t ch_hist[SEG_BUCKETS];  // per-chunk: entries per bucket, then exclusive offsets
    __shared__ uint32_t hist[SIZE_BINS];
    __shared__ uint32_t wave_tot[SEG_NT / 64 + 1];
    __shared__ uint32_t stage_idx[SEG_CHUNK];
    __shared__ uint8_t stage_loc[SEG_CHUNK];
    const uint32_t t = threadIdx.x, z = blockIdx.x, nbig = *B.big_ctr;
    for (uint32_t ci = blockIdx.y; ci < nbig; ci += gridDim.y) {
    const uint32_t cellid = B.big_cells[ci], g = cellid % G, w = cellid / G;
    const uint32_t base = g * SEG_BUCKETS;
    const uint32_t nloc = min((uint32_t)SEG_BUCKETS, nb - base);
    const size_t flat0 = (size_t)w * nb + base;
    const uint32_t seg0 = B.cell_base[cellid], seg_n = B.cell_cnt[cellid];
    const uint32_t nchunks = (seg_n + SEG_CHUNK - 1) / SEG_CHUNK;
    const uint32_t active = min((uint32_t)SEG_Z, nchunks);
    if (z < active) {   // (uniform over the workgroup)
    const uint8_t *__restrict__ loc = B.e_loc + seg0;
    const uint32_t *__restrict__ idx = B.e_idx + seg0;
    uint32_t *__restrict__ sorted = B.sorted;

    uint32_t c = 0, before = 0;
    for (uint32_t zz = 0; zz < active; zz++) {
        const uint32_t v = B.zcount[((size_t)cellid * SEG_Z + zz) * SEG_BUCKETS + t];
        if (zz < z) before += v;
        c += v;
    }
    uint32_t total;
    const uint32_t ex = block_exclusive_scan<SEG_NT>(c, wave_tot, &total);
    cur[t] = seg0 + ex + before;
    if (z == 0) {
        for (uint32_t k = t; k < SIZE_BINS; k += SEG_NT) hist[k] = 0;
        __syncthreads();
        if (t < nloc) {
            B.counts[flat0 + t] = c;
            B.bucket_off[flat0 + t] = seg0 + ex;
            atomicAdd(&hist[size_bin(c, B.heavy_th)], 1u);
        }
        __syncthreads();
        for (uint32_t k = t; k < SIZE_BINS; k += SEG_NT) {
            const uint32_t h = hist[k];
            if (h) atomicAdd(&B.size_hist[k], h);
        }
    }
    __syncthreads();

    for (uint32_t ch = z; ch < nchunks; ch += SEG_Z) {
        const uint32_t c0 = ch * SEG_CHUNK, cn = min((uint32_t)SEG_CHUNK, seg_n - c0);
        ch_hist[t] = 0;
        __syncthreads();
        // thread t owns entries 16t .. 16t+15 of the chunk: five 16-byte loads in flight, then the LDS ranks
        uint32_t e_i[SEG_EPT], e_r[SEG_EPT];
        uint8_t e_l[SEG_EPT];
        if (SEG_EPT * t < cn) {
            const uint4 lv = *reinterpret_cast<const uint4 *>(loc + c0 + SEG_EPT * t);
            const uint4 *ip = reinterpret_cast<const uint4 *>(idx + c0 + SEG_EPT * t);
            const uint4 i0 = ip[0], i1 = ip[1], i2 = ip[2], i3 = ip[3];
            const uint32_t lw[4] = {lv.x, lv.y, lv.z, lv.w};
            const uint32_t iw[SEG_EPT] = {i0.x, i0.y, i0.z, i0.w, i1.x, i1.y, i1.z, i1.w, i2.x, i2.y, i2.z, i2.w, i3.x, i3.y, i3.z, i3.w};
#pragma unroll
            for (int k = 0; k < SEG_EPT; k++) {
                e_l[k] = (uint8_t)((lw[k >> 2] >> (8 * (k & 3))) & 0xffu);
                e_i[k] = iw[k];
            }
        }
#pragma unroll
        for (int k = 0; k < SEG_EPT; k++)
            if (SEG_EPT * t + k < cn) e_r[k] = atomicAdd(&ch_hist[e_l[k]], 1u);
        __syncthreads();
        const uint32_t hc = ch_hist[t];
        const uint32_t hx = block_exclusive_scan<SEG_NT>(hc, wave_tot, &total);
        ch_hist[t] = hx;  // exclusive offsets inside the chunk
        __syncthreads();
#pragma unroll
        for (int k = 0; k < SEG_EPT; k++) {
            if (SEG_EPT * t + k < cn) {
                const uint32_t p = ch_hist[e_l[k]] + e_r[k];
                stage_idx[p] = e_i[k];
                stage_loc[p] = e_l[k];
            }
        }
        __syncthreads();
        for (uint32_t p = t; p < cn; p += SEG_NT) {
            const uint32_t l = stage_loc[p];
            sorted[cur[l] + (p - ch_hist[l])] = stage_idx[p];
        }
        __syncthreads();
        cur[t] += hc;
        __syncthreads();
    }
    }   // z < active
    __syncthreads();
    }   // list of large cells
}

// grid covers the flattened bucket array, 2 buckets per thread: ranks the buckets by list length
// (workgroup-local LDS histogram, one global reservation per non-empty length bin) and registers
// heavy buckets together with their wavefront tasks (summed by the heavy workgroups of msm_accumulate_kernel).
template <int DUMMY>
__global__ __launch_bounds__(1024) void msm_rank_kernel(SortBufs B, uint32_t nbuckets) {
    __shared__ uint32_t hist[SIZE_BINS];
    const uint32_t t = threadIdx.x;
    const uint32_t b0i = (blockIdx.x * 1024 + t) * 2;
    hist[t] = 0;
    __syncthreads();
    uint32_t cc[2], bin[2], r[2] = {0, 0};
#pragma unroll
    for (int q = 0; q < 2; q++) {
        cc[q] = (b0i + q < nbuckets) ? B.counts[b0i + q] : 0u;
        bin[q] = size_bin(cc[q], B.heavy_th);
        if (b0i + q < nbuckets) {
            r[q] = atomicAdd(&hist[bin[q]], 1u);
            if (cc[q] > B.heavy_th) {
                const uint32_t nseg = (cc[q] + HEAVY_WAVE - 1) / HEAVY_WAVE;  // wavefront tasks of 64 segments
                const uint32_t tpos = atomicAdd(&B.heavy_ctr[0], nseg);
                const uint32_t hb = atomicAdd(&B.heavy_ctr[1], 1u);
                // capacity is sized so this always holds; if it ever does not, the submission fails at collect
                if (tpos + nseg <= B.heavy_cap && hb < B.heavy_cap) {
                    B.heavy_buckets[hb] = make_uint4(b0i + q, tpos, nseg, 0u);   // .w: wavefront tasks of this bucket that have finished
                    for (uint32_t k = 0; k < nseg; k++) B.heavy_tasks[tpos + k] = make_uint2(hb, k);   // (heavy-bucket slot, 64-segment group)
                } else {
                    atomicAdd(B.err, 1u);
                }
            }
        }
    }
    __syncthreads();
    const uint32_t h = hist[t];
    __syncthreads();
    hist[t] = h ? B.size_base[t] + atomicAdd(&B.size_cursor[t], h) : 0u;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 2; q++)
        if (b0i + q < nbuckets) B.perm[hist[bin[q]] + r[q]] = b0i + q;
}

// ------------------------------------------------------------------------------ accumulate
// One thread per bucket, one wavefront per workgroup, buckets taken in order of decreasing list
// length (perm[]): the 64 lanes of a wavefront own lists of (nearly) equal length, wavefronts retire
// independently and the longest lists start first.  Each thread adds its points in XYZZ mixed
// coordinates (8M+2S per point).
// [accumulate-kernel-begin]  (bench.py hashes field.h, curve.h and the text between these two markers: arithmetic_source_hash)
template <class F>
__device__ __forceinline__ Xyzz<F> sum_list(const PackedAffine<F> *__restrict__ pts, const uint32_t *__restrict__ lst, uint32_t len) {
    // No software prefetch: holding the next point would cost the registers that let four (G1) / two (G2) wavefronts
    // share a SIMD, and those wavefronts hide the gather latency better than a prefetch under three / one did
    // (G1 accumulate 1.30 -> 1.27 ms, G2 4.40 -> 3.63 ms).
    Xyzz<F> acc = Xyzz<F>::inf();
    // G1 only: the NEXT list entry is fetched one addition ahead (one register), so the gather's address is ready when this
    // addition ends (accumulate -2 %); the G2 kernel has no register to spare and measured no gain.
    constexpr bool AHEAD = F::CANON_WORDS == 8;
    uint32_t e_next = (AHEAD && len) ? lst[0] : 0u;
#pragma unroll 1
    for (uint32_t k = 0; k < len; k++) {
        const uint32_t e = AHEAD ? e_next : lst[k];
        if (AHEAD && k + 1 < len) e_next = lst[k + 1];
        Affine<F> p = unpack_affine(pts[e & 0x7fffffffu]);
        if (e >> 31) p.y = fe_neg_once<2>(p.y);   // enters one product (or is tidied on the rare paths)
        xyzz_add_affine(acc, p);
    }
    return acc;
}

// Sum of the 64 points a wavefront holds one per lane, left in tree[0]: the points go to LDS and are folded in halves (tree[i] +=
// tree[i + d], d = 32 .. 1) by teams of four lanes (curve.h team4_add: ~1300 instructions per level instead of the ~3400 of a
// one-lane general addition; seven rounds, the first level taking two).  Round 5: until then a __shfl_down tree of one-lane
// additions -- the general addition next to the list loop cost the G1 kernel 7 spilled registers, and these trees are a third of
// the dependent chain a heavy bucket puts on the critical path of a blocking MSM.  One wavefront per workgroup: the lanes run in
// lockstep, an LDS write is visible to the wavefront's later reads (s_waitcnt, no barrier).
template <class F> __device__ __forceinline__ void wave_tree_sum(const Xyzz<F> &mine, Xyzz<F> *tree) {
    const uint32_t lane = threadIdx.x;
    tree[lane] = mine;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll 1
    for (uint32_t d = 32; d >= 1; d >>= 1) {
#pragma unroll 1
        for (uint32_t q = lane >> 2; q < d; q += 16) team4_add(lane & 3u, tree + q, tree + q + d, QuadDpp{});
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}
// Heavy buckets (lists longer than heavy_th: skewed / witness-like scalars): wavefront tasks listed by msm_rank_kernel and taken by the
// first HEAVY_BLOCKS workgroups of the accumulate grid (round 4; a kernel of its own behind the accumulate kernel before -- its ~46
// dependent additions then sat on the critical path of a blocking MSM: 0.3 ms in G1, 1 ms in G2 with a quarter of the scalars equal
// to one).  A task = 64 consecutive HEAVY_SEG-entry segments of one bucket's list: every lane sums its segment, wave_tree_sum
// leaves the task's partial sum in LDS, lane 0 stores it and counts the task as finished on its bucket; the wavefront that
// finishes a bucket's LAST task then sums that bucket's partials (64 at a time through the same tree) into the bucket array.  With
// uniform scalars there are no tasks and those workgroups return at once.
// Hand-off of the partials between wavefronts: plain stores, s_waitcnt, agent-scope release, relaxed agent-scope counter; the last
// arriver acquires at agent scope before it loads (MI355X_MICROARCH.md, inter-workgroup visibility).  Nothing ever waits for another
// wavefront, so the order in which the hardware places the workgroups cannot deadlock it.
template <class F>
__device__ __forceinline__ void heavy_tasks(const PackedAffine<F> *__restrict__ pts, const SortBufs &B, Xyzz<F> *partial, Xyzz<F> *__restrict__ buckets,
                                            uint32_t first, uint32_t stride, Xyzz<F> *tree) {
    const uint32_t ntasks = __builtin_amdgcn_readfirstlane(min(B.heavy_ctr[0], B.heavy_cap));
    for (uint32_t wt_i = first; wt_i < ntasks; wt_i += stride) {
        // Nothing but the task number (a scalar register) lives across the list loop, which leaves no vector register free (four
        // wavefronts x 128): the task's descriptors are read again behind it (the asm makes the number opaque, so the compiler
        // cannot keep them) and the lane number is taken from mbcnt again.
        uint32_t wt = __builtin_amdgcn_readfirstlane(wt_i);
        {
            const uint32_t lane = threadIdx.x;
            const uint32_t tx = __builtin_amdgcn_readfirstlane(B.heavy_tasks[wt].x), ty = __builtin_amdgcn_readfirstlane(B.heavy_tasks[wt].y);
            const uint32_t bucket = __builtin_amdgcn_readfirstlane(B.heavy_buckets[tx].x);
            const uint32_t len = __builtin_amdgcn_readfirstlane(B.counts[bucket]), lo = (ty * 64 + lane) * HEAVY_SEG;
            wave_tree_sum(sum_list(pts, B.sorted + B.bucket_off[bucket] + min(lo, len), lo < len ? min((uint32_t)HEAVY_SEG, len - lo) : 0u), tree);
        }
        asm volatile("" : "+s"(wt));
        const uint32_t lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));   // one wavefront per workgroup
        uint2 task;
        task.x = __builtin_amdgcn_readfirstlane(B.heavy_tasks[wt].x);
        const uint4 hb_v = B.heavy_buckets[task.x];   // (bucket id, first task, tasks, -)
        const uint4 hb = make_uint4(__builtin_amdgcn_readfirstlane(hb_v.x), __builtin_amdgcn_readfirstlane(hb_v.y), __builtin_amdgcn_readfirstlane(hb_v.z), 0u);
        uint32_t done = 0;
        if (lane == 0) {
            partial[wt] = tree[0];
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the compiler may drop the fence's own wait (MI355X_MICROARCH.md, compiler hazard)
            done = __hip_atomic_fetch_add(&B.heavy_buckets[task.x].w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
        }
        done = __shfl(done, 0, 64);
        if (done != hb.z) continue;                 // other tasks of this bucket are still running: their last one combines
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // the bucket's partials, 64 at a time: chunk 0 is the running sum, every further chunk's tree sum is added to it by one team
        for (uint32_t k0 = 0; k0 < hb.z; k0 += 64) {
            Xyzz<F> v = Xyzz<F>::inf();
            if (k0 + lane < hb.z) v = partial[hb.y + k0 + lane];
            if (k0 == 0) {
                wave_tree_sum(v, tree);
                if (lane == 0) buckets[hb.x] = tree[0];
            } else {
                wave_tree_sum(v, tree);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane < 4) team4_add(lane, buckets + hb.x, tree, QuadDpp{});   // buckets[hb.x] += this chunk's sum
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        }
    }
}

template <class F>
__global__ __launch_bounds__(64, (F::CANON_WORDS == 8 ? 4 : 2)) void msm_accumulate_kernel(const PackedAffine<F> *__restrict__ pts, const uint32_t *__restrict__ sorted,
                                                             const uint32_t *__restrict__ counts,
                                                             const uint32_t *__restrict__ bucket_off,
                                                             const uint32_t *__restrict__ perm, Xyzz<F> *__restrict__ buckets,
                                                             uint32_t nbuckets, uint32_t heavy_th, SortBufs B, Xyzz<F> *partial, uint32_t heavy_blocks) {
    if (blockIdx.x < heavy_blocks) {
        __shared__ Xyzz<F> tree[64];   // 9 / 18 KB: sixteen (G1) / eight (G2) workgroups per CU still fit the 160 KB
        heavy_tasks(pts, B, partial, buckets, blockIdx.x, heavy_blocks, tree);
        return;
    }
    const uint32_t r = (blockIdx.x - heavy_blocks) * 64 + threadIdx.x;
    if (r >= nbuckets) return;
    const uint32_t b = perm[r];
    const uint32_t len = counts[b];
    if (len > heavy_th) return;  // summed by the heavy blocks
    buckets[b] = sum_list(pts, sorted + bucket_off[b], len);
}
// [accumulate-kernel-end]

// ------------------------------------------------------------------------------ host side
static int pick_window_bits(size_t n) {
    // Measured on MI355X (blocking and pipelined, n = 2^2 .. 2^19): below ~2^9 points everything is launch latency and the
    // 8-bit windows' short reduction tree wins; from there to 2^17 c = 15 beats every narrower width -- most of its 17 x 2^14
    // buckets stay empty and cost the reduction next to nothing, while the accumulate kernel runs three windows fewer
    // than with c = 13 (2^12 points: 0.97 -> 0.63 ms); 16 from 2^18.  All three keep a non-degenerate top window
    // (255 = W*c - slack leaves it 7 / 15 / 15 scalar bits; c = 9, 11, 12, 14 would leave 1-3 giant buckets).
    if (n <= (1u << 9)) return 8;
    if (n <= (1u << 17)) return 15;
    return 16;
}

}  // namespace zk
#define ZK_MSM_IMPL_KERNELS_DONE
#include "msm_fixed.h"
namespace zk {

template <class F> struct MsmPlanImpl : MsmPlanBase {
    typedef typename HostOf<F>::type HF;
    size_t max_n;
    size_t dig_bytes = 0, arena_bytes = 0, out_bytes_max = 0;
    uint32_t heavy_cap = 0;

    // A lane is everything one in-flight MSM owns: its workspace, a stream of its own and the read-back buffer.  Submissions rotate through the lanes, so the
    // kernels of consecutive MSMs overlap on the GPU: the sort and reduce phases and the accumulate kernel's
    // tail leave CUs idle that the neighbouring MSM's accumulate kernel fills (2^20 points: 1.89 -> 1.73 ms per
    // MSM with three lanes).  Lanes beyond the first are allocated on first use.
    struct Lane {
        DevBuf digits32;  // bound-bases mode: 13 signed 20-bit digits per scalar as int32, allocated on first use
        DevBuf pts_m, digits, sorted, e_idx, e_loc, counts, bucket_off, cells, zcount, size_bins, perm, arena, out, heavy_tasks, heavy_buckets,
            heavy_partial;
        PinnedBuf h_out;
        hipStream_t stream = nullptr;
        hipEvent_t ev_in = nullptr, ev_consumed = nullptr, done = nullptr;
        hipEvent_t ev[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // stage boundaries when profiling; [5], [6] bracket the accumulate kernel
        bool ready = false, busy = false, empty = false, profiled = false, single_window = false;
        int c = 0;
        uint32_t err_seen = 0;  // value of the lane's device error counter at the last collect
    };
    // The read-back buffer of a lane: a 16-byte header (word 0 = the lane's device error counter, see SortBufs::err) followed by
    // the window / level sums, so that one copy brings both back.
    static constexpr size_t OUT_HDR = 16;
    static uint32_t *err_dev(Lane &L) { return L.out.template as<uint32_t>(); }
    static Xyzz<F> *out_dev(Lane &L) { return reinterpret_cast<Xyzz<F> *>(static_cast<char *>(L.out.p) + OUT_HDR); }
    static const Xyzz<F> *out_host(const Lane &L) { return reinterpret_cast<const Xyzz<F> *>(static_cast<const char *>(L.h_out.p) + OUT_HDR); }
    // A launch that throws half-way through a submission must not leave the lane marked busy, and whatever the kernels that did
    // start added to the lane's device error counter must not be charged to the next (valid) submission: wait for the lane and
    // re-read the counter.
    // The sort counters (cell totals / cursors, list-length histogram, heavy-task and big-cell counters) are zeroed by the kernels
    // that consume them, i.e. only by a launch sequence that ran to its end: after a failure they are reset here, or the lane's
    // next MSM would be partitioned from stale totals.
    struct LaneGuard {
        Lane *lane;
        ~LaneGuard() {
            if (!lane) return;
            lane->busy = false;
            uint32_t now = 0;
            if (lane->stream && hipStreamSynchronize(lane->stream) == hipSuccess) {
                (void)hipMemset(lane->cells.p, 0, lane->cells.bytes);
                (void)hipMemset(lane->size_bins.p, 0, lane->size_bins.bytes);
                (void)hipDeviceSynchronize();
                if (hipMemcpy(&now, lane->out.p, sizeof(now), hipMemcpyDeviceToHost) == hipSuccess) lane->err_seen = now;
            }
        }
    };
    static constexpr int MAX_LANES = 3;
    Lane lanes[MAX_LANES];
    // bound bases (msm_fixed.h): table[w * fix_n + i] = 2^(20 w) * P_i, shared by the lanes
    DevBuf fix_table;
    size_t fix_n = 0;
    int nlanes = 1, next_lane = 0;

    // MSMs beyond 2^22 points are run as consecutive chunks of 2^22 (each in its own lane, partial sums added on the
    // host): the packed bases of a chunk (256 MB) stay within reach of the Infinity Cache and the TLB -- at 2^26 in one
    // piece the gathers of the accumulate kernel run 40 % slower -- and the chunks overlap like any other submissions
    // (2^24: 26.3 -> 24.0 ms, 2^26: 138 -> 93 ms).  The workspace never exceeds that of a 2^22-point MSM.
    // 2^22 unless the plan was created with another chunk size (zk_msm_plan_create_ex: chunking at sizes the oracle can check)
    static size_t chunk_points(int chunk_log) {
        const int l = chunk_log ? chunk_log : 22;
        return (size_t)1 << (l < 12 ? 12 : l > 24 ? 24 : l);
    }
    static constexpr int BIG_TICKET = 64;
    size_t cap_n = 0;
    struct {
        bool active = false;
        Xyzz<HF> acc;
        int pend[8];
        int npend = 0;
    } big;

    explicit MsmPlanImpl(size_t max_n_, bool all_lanes = false, int chunk_log = 0) : max_n(max_n_) {
        group = sizeof(F) == sizeof(Fp) ? ZK_GROUP_G1 : ZK_GROUP_G2;
        ZK_HIP(hipGetDevice(&device));
        cap_n = std::min(max_n, chunk_points(chunk_log));
        const size_t n_pad = pad_n(cap_n);
        // worst case over the window choices available to n <= max_n
        const int cs[3] = {8, 15, 16};
        for (int c : cs) {
            if (c > pick_window_bits(cap_n)) continue;
            size_t W = (255 + c - 1) / c, nb = (size_t)1 << (c - 1);
            dig_bytes = std::max(dig_bytes, W * n_pad * sizeof(int16_t));
            arena_bytes = std::max(arena_bytes, W * nb * sizeof(Xyzz<F>));
            out_bytes_max = std::max(out_bytes_max, W * (size_t)c * sizeof(Xyzz<F>));
        }
        // heavy-bucket scratch: a heavy bucket holds > heavy_th >= HEAVY_SEG entries, so there are < entries / HEAVY_SEG of
        // them and sum ceil(len / HEAVY_WAVE) <= entries / HEAVY_WAVE + (heavy buckets) wavefront tasks; entries <= W * n_pad
        heavy_cap = (uint32_t)(dig_bytes / sizeof(int16_t) / HEAVY_SEG + dig_bytes / sizeof(int16_t) / HEAVY_WAVE + 64);
        nlanes = 3;
        for (int i = 0; i < (all_lanes ? nlanes : 1); i++) prepare_lane(lanes[i]);
        if (all_lanes) {
            // Prime every lane with a one-point MSM: the first launches on a new stream pay for the hardware queue and the
            // code objects (milliseconds), which would otherwise land on the caller's first three submissions.
            ZK_HIP(hipMemset(lanes[0].arena.p, 0, 128));
            ZK_HIP(hipStreamSynchronize(0));  // null-stream memset vs the lanes' non-blocking streams
            const bool prof = profile;
            profile = true;  // creates the profiling events as well
            for (int i = 0; i < nlanes; i++) (void)collect_lane(submit_lane(lanes[0].arena.p, lanes[0].arena.p, 1, nullptr));
            profile = prof;
        }
    }
    void prepare_lane(Lane &L) {
        if (L.ready) return;
        const size_t dig = dig_bytes, nbk = arena_bytes / sizeof(Xyzz<F>);
        L.pts_m.alloc(cap_n * sizeof(PackedAffine<F>));
        L.digits.alloc(dig);
        const size_t span_slack = (size_t)MAX_CELLS * 16 + 64;  // cell spans are padded to 16 entries
        L.sorted.alloc(dig * 2 + span_slack * 4);               // one 4-byte entry per (window, point)
        L.counts.alloc(nbk * sizeof(uint32_t));
        L.bucket_off.alloc(nbk * sizeof(uint32_t));
        L.e_idx.alloc(dig * 2 + span_slack * 4);
        L.e_loc.alloc(dig / 2 + span_slack);
        L.cells.alloc(4 * MAX_CELLS * sizeof(uint32_t));  // cell_total | cell_base | cell_cnt | cell_cursor
        ZK_HIP(hipMemset(L.cells.p, 0, 4 * MAX_CELLS * sizeof(uint32_t)));
        L.size_bins.alloc((3 * SIZE_BINS + 4 + MAX_CELLS) * sizeof(uint32_t));  // size_hist | size_base | size_cursor | heavy_ctr[2] | big_ctr, pad | big_cells
        ZK_HIP(hipMemset(L.size_bins.p, 0, (3 * SIZE_BINS + 4 + MAX_CELLS) * sizeof(uint32_t)));
        L.zcount.alloc((size_t)MAX_CELLS * SEG_Z * SEG_BUCKETS * sizeof(uint32_t));
        L.heavy_tasks.alloc((size_t)heavy_cap * sizeof(uint2));
        L.heavy_buckets.alloc((size_t)heavy_cap * sizeof(uint4));
        L.heavy_partial.alloc((size_t)heavy_cap * sizeof(Xyzz<F>));
        L.perm.alloc(nbk * sizeof(uint32_t));
        L.arena.alloc(arena_bytes);
        L.out.alloc(out_bytes_max + OUT_HDR);
        ZK_HIP(hipMemset(L.out.p, 0, OUT_HDR));
        L.h_out.alloc(out_bytes_max + OUT_HDR);
        // the memsets above were issued on the null stream, which the lane's non-blocking stream does not wait for: make sure they
        // have landed before the lane's first kernel can run (it bumps the error counter in the header zeroed here)
        ZK_HIP(hipStreamSynchronize(0));
        for (hipEvent_t *e : {&L.ev_in, &L.ev_consumed, &L.done}) ZK_HIP(hipEventCreateWithFlags(e, hipEventDisableTiming));
        L.ready = true;
    }
    ~MsmPlanImpl() override {
        for (auto &L : lanes) {
            if (L.stream) (void)hipStreamSynchronize(L.stream);   // (the stream belongs to the process-wide pool: msm.h lane_stream_next)
            for (hipEvent_t e : {L.ev_in, L.ev_consumed, L.done})
                if (e) (void)hipEventDestroy(e);
            for (auto &e : L.ev)
                if (e) (void)hipEventDestroy(e);
        }
    }
    static size_t pad_n(size_t n) { return ((n + 4095) / 4096) * 4096; }
    void mark(Lane &L, int i) {
        if (!L.profiled) return;
        hipEvent_t &e = L.ev[i];
        if (!e) ZK_HIP(hipEventCreate(&e));
        ZK_HIP(hipEventRecord(e, L.stream));
    }

    int window_bits(size_t n) const override { return pick_window_bits(n); }
    int max_in_flight() const override { return nlanes; }

    template <int C> void launch_prepare(Lane &L, const uint32_t *sc, const uint32_t *pt, uint32_t n, uint32_t n_pad) {
        hipLaunchKernelGGL((msm_prepare_kernel<F, C>), dim3(n_pad / (PREP_NT * PREP_PPT)), dim3(PREP_NT), 0, L.stream, sc, pt,
                           L.pts_m.template as<PackedAffine<F>>(), L.digits.template as<int16_t>(), L.cells.template as<uint32_t>(), err_dev(L), n, n_pad);
    }
    SortBufs sort_bufs(Lane &L) {
        SortBufs B;
        B.counts = L.counts.template as<uint32_t>();
        B.bucket_off = L.bucket_off.template as<uint32_t>();
        B.cell_total = L.cells.template as<uint32_t>();
        B.cell_base = B.cell_total + MAX_CELLS;
        B.cell_cnt = B.cell_total + 2 * MAX_CELLS;
        B.cell_cursor = B.cell_total + 3 * MAX_CELLS;
        B.e_idx = L.e_idx.template as<uint32_t>();
        B.e_loc = L.e_loc.template as<uint8_t>();
        B.sorted = L.sorted.template as<uint32_t>();
        B.zcount = L.zcount.template as<uint32_t>();
        B.size_hist = L.size_bins.template as<uint32_t>();
        B.size_base = B.size_hist + SIZE_BINS;
        B.size_cursor = B.size_hist + 2 * SIZE_BINS;
        B.perm = L.perm.template as<uint32_t>();
        B.heavy_th = 32;
        B.heavy_cap = heavy_cap;
        B.heavy_ctr = B.size_hist + 3 * SIZE_BINS;
        B.big_ctr = B.heavy_ctr + 2;
        B.big_cells = B.heavy_ctr + 4;
        B.heavy_tasks = L.heavy_tasks.template as<uint2>();
        B.heavy_buckets = L.heavy_buckets.template as<uint4>();
        B.err = err_dev(L);
        return B;
    }
    void launch_sort_accumulate(Lane &L, uint32_t n_pad, uint32_t nb, uint32_t W) {
        hipStream_t st = L.stream;
        const uint32_t G = (nb + SEG_BUCKETS - 1) / SEG_BUCKETS;
        if (G * W > MAX_CELLS) throw std::runtime_error("zk_msm: too many sort cells");
        SortBufs B = sort_bufs(L);
        B.heavy_th = std::max<uint32_t>(32, 8 * (n_pad / nb));
        const uint32_t nbuckets_all = W * nb;
        hipLaunchKernelGGL((msm_partition_kernel<0>), dim3((n_pad + PART_PTS - 1) / PART_PTS, W), dim3(PREP_NT), 0, st, L.digits.template as<int16_t>(), B,
                           n_pad, W, G);   // scans the cell totals itself
        hipLaunchKernelGGL((msm_cellsort_kernel<0>), dim3(G, W), dim3(CS_NT), 0, st, B, nb);
        hipLaunchKernelGGL((msm_segcount_kernel<0>), dim3(SEG_Z, SEG_LIST), dim3(SEG_NT), 0, st, B);   // cells beyond CS_MAX entries only
        hipLaunchKernelGGL((msm_segscatter_kernel<0>), dim3(SEG_Z, SEG_LIST), dim3(SEG_NT), 0, st, B, nb, G);
        hipLaunchKernelGGL((msm_scan_kernel<true>), dim3(1), dim3(1024), 0, st, B, G * W);
        hipLaunchKernelGGL((msm_rank_kernel<0>), dim3((nbuckets_all + 2047) / 2048), dim3(1024), 0, st, B, nbuckets_all);
        mark(L, 2);
        const uint32_t nbuckets = W * nb;
        mark(L, 5);
        // Heavy buckets (lists beyond heavy_th: skewed / witness-like scalars; normally none) are summed by the FIRST heavy_blocks
        // workgroups of the same grid -- wavefront tasks msm_rank_kernel has listed -- so their long dependent chains start first and
        // run beside the ordinary lists instead of after them; with no task listed those workgroups return at once.
        const uint32_t heavy_blocks = std::min<uint32_t>(heavy_cap, HEAVY_BLOCKS);
        hipLaunchKernelGGL((msm_accumulate_kernel<F>), dim3(heavy_blocks + (nbuckets + 63) / 64), dim3(64), 0, st, L.pts_m.template as<PackedAffine<F>>(),
                           L.sorted.template as<uint32_t>(), L.counts.template as<uint32_t>(), L.bucket_off.template as<uint32_t>(),
                           L.perm.template as<uint32_t>(), L.arena.template as<Xyzz<F>>(), nbuckets, B.heavy_th, B,
                           L.heavy_partial.template as<Xyzz<F>>(), heavy_blocks);
        mark(L, 6);
    }

    // Enqueues the whole GPU pipeline plus the 36 KiB read-back; returns a ticket.  The work runs on the lane's own
    // stream: it starts once everything queued on `st` so far has finished, and `st` resumes as soon as the prepare
    // kernel has consumed the caller's scalars and points (the only kernel that reads them).
    int submit(const void *d_scalars, const void *d_points, size_t n, hipStream_t st) override {
        if (!d_points && n) return submit_bound(d_scalars, 0, n, st);
        if (n > max_n) throw std::runtime_error("zk_msm: n exceeds the plan's max_n");
        if (big.active) throw std::runtime_error("zk_msm: a chunked (> 2^22 points) submission is outstanding; collect it first");
        if (n > cap_n) return submit_chunked(d_scalars, d_points, 0, n, st);
        return submit_lane(d_scalars, d_points, n, st);
    }
    // MSM over the bound bases [first, first + n)
    int submit_bound(const void *d_scalars, size_t first, size_t n, hipStream_t st) override {
        if (!fix_n) throw std::runtime_error("zk_msm: no bases bound (zk_msm_plan_bind_points)");
        if (first + n > fix_n) throw std::runtime_error("zk_msm: range exceeds the bound bases");
        if (big.active) throw std::runtime_error("zk_msm: a chunked (> 2^22 points) submission is outstanding; collect it first");
        if (n == 0) return submit_lane(d_scalars, nullptr, 0, st);
        if (n > cap_n) return submit_chunked(d_scalars, nullptr, first, n, st);
        return submit_lane_fixed(d_scalars, first, n, st);
    }

    // ---- bound-bases mode (msm_fixed.h)
    int bind_points(const void *d_points, size_t n, hipStream_t st) override {
        if (n == 0) {
            fix_n = 0;
            fix_table.release();
            return ZK_OK;
        }
        constexpr int FIX_W = FixOf<F>::W;
        if (pick_window_bits(cap_n) != 16) throw std::runtime_error("zk_msm_plan_bind_points: the plan must be created for more than 2^17 points");
        if ((uint64_t)FIX_W * n >= ((uint64_t)1 << 31)) throw std::runtime_error("zk_msm_plan_bind_points: too many bases (rows * n must stay below 2^31)");
        for (int i = 0; i < nlanes; i++)
            if (lanes[i].busy) throw std::runtime_error("zk_msm_plan_bind_points: submissions outstanding");
        fix_n = 0;
        fix_table.alloc((size_t)FIX_W * n * sizeof(PackedAffine<F>));
        hipLaunchKernelGGL((msm_fixed_table_kernel<F>), dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, static_cast<const uint32_t *>(d_points),
                           fix_table.template as<PackedAffine<F>>(), (uint32_t)n, n);
        ZK_HIP(hipGetLastError());
        ZK_HIP(hipStreamSynchronize(st));
        fix_n = n;
        return ZK_OK;
    }
    // The first free lane, scanning from next_lane (tickets are collected in any order, so the lane after the one used last may still
    // be busy while others are free); throws only when every lane holds an uncollected submission.
    int pick_lane() {
        for (int k = 0; k < nlanes; k++)
            if (!lanes[(next_lane + k) % nlanes].busy) return (next_lane + k) % nlanes;
        throw std::runtime_error("zk_msm: too many submissions in flight (zk_msm_plan_max_in_flight); collect one first");
    }
    int submit_lane_fixed(const void *d_scalars, size_t first, size_t n, hipStream_t st) {
        using Cfg = FixOf<F>;
        constexpr int FIX_C = Cfg::C, FIX_W = Cfg::W;
        constexpr uint32_t FIX_NB = Cfg::NB, FIX_G = Cfg::G;
        const int ticket = pick_lane();
        Lane &L = lanes[ticket];
        prepare_lane(L);
        L.stream = lane_stream_next(device);
        const uint32_t n_pad = (uint32_t)pad_n(n);
        if (L.digits32.bytes < (size_t)FIX_W * pad_n(cap_n) * sizeof(int32_t)) L.digits32.alloc((size_t)FIX_W * pad_n(cap_n) * sizeof(int32_t));
        L.busy = true;
        L.empty = false;
        L.profiled = profile;
        L.single_window = true;
        L.c = FIX_C;
        next_lane = (ticket + 1) % nlanes;
        LaneGuard guard{&L};   // a launch that throws must not leave the lane marked busy
        hipStream_t ls = L.stream;
        ZK_HIP(hipEventRecord(L.ev_in, st));
        ZK_HIP(hipStreamWaitEvent(ls, L.ev_in, 0));
        mark(L, 0);
        hipLaunchKernelGGL((msm_fixed_prepare_kernel<FIX_C>), dim3(n_pad / (PREP_NT * PREP_PPT)), dim3(PREP_NT), 0, ls, static_cast<const uint32_t *>(d_scalars),
                           L.digits32.template as<int32_t>(), L.cells.template as<uint32_t>(), (uint32_t)n, n_pad, F::CANON_WORDS == 8);
        ZK_HIP(hipEventRecord(L.ev_consumed, ls));
        ZK_HIP(hipStreamWaitEvent(st, L.ev_consumed, 0));
        mark(L, 1);
        SortBufs B = sort_bufs(L);
        B.heavy_th = std::max<uint32_t>(32, 8 * (uint32_t)(((size_t)FIX_W * n_pad) / FIX_NB));
        const PackedAffine<F> *table = fix_table.template as<PackedAffine<F>>();
#ifndef ZK_FIX_PPT
#define ZK_FIX_PPT 8
#endif
        constexpr int FIX_PPT = ZK_FIX_PPT;
        const uint32_t fix_total = (uint32_t)FIX_W * n_pad;
        hipLaunchKernelGGL((msm_fixed_partition_kernel<FIX_PPT, FIX_C>), dim3((fix_total + PREP_NT * FIX_PPT - 1) / (PREP_NT * FIX_PPT)), dim3(PREP_NT), 0, ls,
                           L.digits32.template as<int32_t>(), B, n_pad, fix_n, (uint32_t)first, fix_total, fixed_alias_mask());
        hipLaunchKernelGGL((msm_cellsort_kernel<0>), dim3(FIX_G, 1), dim3(CS_NT), 0, ls, B, FIX_NB);
        hipLaunchKernelGGL((msm_segcount_kernel<0>), dim3(SEG_Z, SEG_LIST), dim3(SEG_NT), 0, ls, B);
        hipLaunchKernelGGL((msm_segscatter_kernel<0>), dim3(SEG_Z, SEG_LIST), dim3(SEG_NT), 0, ls, B, FIX_NB, (uint32_t)FIX_G);
        hipLaunchKernelGGL((msm_scan_kernel<true>), dim3(1), dim3(1024), 0, ls, B, FIX_G);
        hipLaunchKernelGGL((msm_rank_kernel<0>), dim3((FIX_NB + 2047) / 2048), dim3(1024), 0, ls, B, FIX_NB);
        mark(L, 2);
        mark(L, 5);
        const uint32_t heavy_blocks = std::min<uint32_t>(heavy_cap, HEAVY_BLOCKS);
        hipLaunchKernelGGL((msm_accumulate_kernel<F>), dim3(heavy_blocks + (FIX_NB + 63) / 64), dim3(64), 0, ls, table, L.sorted.template as<uint32_t>(),
                           L.counts.template as<uint32_t>(), L.bucket_off.template as<uint32_t>(), L.perm.template as<uint32_t>(),
                           L.arena.template as<Xyzz<F>>(), FIX_NB, B.heavy_th, B, L.heavy_partial.template as<Xyzz<F>>(), heavy_blocks);
        mark(L, 6);
        mark(L, 3);
        // The block kernel takes the lowest RED_BL index bits (256 workgroups); the eight levels above them run in TWO launches of the
        // window kernel: 16 groups of 16 blocks reduce as if they were windows of their own (four steps, at most 96 team additions
        // each), then one workgroup takes the 16 group results as "blocks" of 2^(RED_BL+4) buckets and writes the level sums out.
        constexpr uint32_t levels = FIX_C - 1, BL = RED_BL, GL = (levels - BL) / 2;
        constexpr int WNT = RED_WINDOW_NT;
        hipLaunchKernelGGL((msm_reduce_block_kernel<F, RED_BLOCK_NT>), dim3(FIX_NB >> BL), dim3(RED_BLOCK_NT), 0, ls, L.arena.template as<Xyzz<F>>(), BL, static_cast<uint64_t *>(nullptr));
        hipLaunchKernelGGL((msm_reduce_window_kernel<F, WNT>), dim3(FIX_NB >> (BL + GL)), dim3(WNT), 0, ls, L.arena.template as<Xyzz<F>>(),
                           static_cast<Xyzz<F> *>(nullptr), 1u << (BL + GL), BL, BL + GL);
        hipLaunchKernelGGL((msm_reduce_window_kernel<F, WNT>), dim3(1), dim3(WNT), 0, ls, L.arena.template as<Xyzz<F>>(), out_dev(L), FIX_NB, BL + GL, levels);
        mark(L, 4);
        ZK_HIP(hipMemcpyAsync(L.h_out.p, L.out.p, OUT_HDR + (size_t)(levels + 1) * sizeof(Xyzz<F>), hipMemcpyDeviceToHost, ls));
        ZK_HIP(hipEventRecord(L.done, ls));
        ZK_HIP(hipGetLastError());
        guard.lane = nullptr;
        return ticket;
    }
    // n > CHUNK: every chunk is an ordinary submission; when the lanes run out the oldest chunk is collected (this call
    // then blocks for it).  The ticket stands for the whole MSM; no other submission may be outstanding meanwhile.
    int submit_chunked(const void *d_scalars, const void *d_points, size_t first, size_t n, hipStream_t st) {
        // The chunks rotate through the lanes that are free now: submissions made earlier (a prover's other queries) stay in
        // flight in theirs and are collected by their own tickets, in any order.  With every lane taken there is nowhere to run.
        int nfree = 0;
        for (int i = 0; i < nlanes; i++) nfree += lanes[i].busy ? 0 : 1;
        if (!nfree) throw std::runtime_error("zk_msm: an MSM of more than 2^22 points needs a free lane; collect an outstanding submission first");
        big.active = true;
        big.acc = Xyzz<HF>::inf();
        big.npend = 0;
        const uint32_t *sc = static_cast<const uint32_t *>(d_scalars), *pt = static_cast<const uint32_t *>(d_points);
        try {
            for (size_t off = 0; off < n; off += cap_n) {
                const size_t m = std::min(cap_n, n - off);
                int lane = -1;
                for (int k = 0; k < nlanes && lane < 0; k++)
                    if (!lanes[(next_lane + k) % nlanes].busy) lane = (next_lane + k) % nlanes;
                if (lane < 0) {  // all free lanes hold chunks of this MSM: collect the oldest, its lane is free again
                    lane = big.pend[0];
                    for (int i = 1; i < big.npend; i++) big.pend[i - 1] = big.pend[i];
                    big.npend--;
                    xyzz_add(big.acc, collect_lane(lane));
                }
                next_lane = lane;
                big.pend[big.npend++] = d_points ? submit_lane(sc + off * 8, pt + off * 2 * F::CANON_WORDS, m, st) : submit_lane_fixed(sc + off * 8, first + off, m, st);
            }
        } catch (...) {
            abandon_big();
            throw;
        }
        return BIG_TICKET;
    }
    // A chunk failed: wait for the chunks still in flight, free their lanes and forget the submission, so the plan stays usable.
    void abandon_big() {
        for (int i = 0; i < big.npend; i++) {
            try {
                (void)collect_lane(big.pend[i]);
            } catch (...) {
            }
        }
        big.npend = 0;
        big.active = false;
    }
    int submit_lane(const void *d_scalars, const void *d_points, size_t n, hipStream_t st) {
        const int ticket = pick_lane();
        Lane &L = lanes[ticket];
        prepare_lane(L);
        L.stream = lane_stream_next(device);
        L.busy = true;
        L.empty = (n == 0);
        L.profiled = profile;
        L.single_window = false;
        next_lane = (ticket + 1) % nlanes;
        if (n == 0) return ticket;
        LaneGuard guard{&L};
        const int c = pick_window_bits(n);
        L.c = c;
        const uint32_t W = (255 + c - 1) / c, nb = 1u << (c - 1), levels = c - 1;
        const uint32_t n_pad = (uint32_t)pad_n(n);
        const uint32_t *sc = static_cast<const uint32_t *>(d_scalars), *pt = static_cast<const uint32_t *>(d_points);
        ZK_HIP(hipEventRecord(L.ev_in, st));
        ZK_HIP(hipStreamWaitEvent(L.stream, L.ev_in, 0));
        mark(L, 0);
        switch (c) {
            case 8: launch_prepare<8>(L, sc, pt, (uint32_t)n, n_pad); break;
            case 15: launch_prepare<15>(L, sc, pt, (uint32_t)n, n_pad); break;
            default: launch_prepare<16>(L, sc, pt, (uint32_t)n, n_pad); break;
        }
        ZK_HIP(hipEventRecord(L.ev_consumed, L.stream));
        ZK_HIP(hipStreamWaitEvent(st, L.ev_consumed, 0));
        mark(L, 1);
        launch_sort_accumulate(L, n_pad, nb, W);
        mark(L, 3);
        {
            Xyzz<F> *ar = L.arena.template as<Xyzz<F>>();
            const uint32_t BL = std::min<uint32_t>(RED_BL, levels);
            constexpr int WNT = RED_WINDOW_NT;
            hipLaunchKernelGGL((msm_reduce_block_kernel<F, RED_BLOCK_NT>), dim3((W * nb) >> BL), dim3(RED_BLOCK_NT), 0, L.stream, ar, BL, static_cast<uint64_t *>(nullptr));
            hipLaunchKernelGGL((msm_reduce_window_kernel<F, WNT>), dim3(W), dim3(WNT), 0, L.stream, ar, out_dev(L), nb, BL, levels);
        }
        mark(L, 4);
        const size_t out_bytes = OUT_HDR + (size_t)W * (levels + 1) * sizeof(Xyzz<F>);
        ZK_HIP(hipMemcpyAsync(L.h_out.p, L.out.p, out_bytes, hipMemcpyDeviceToHost, L.stream));
        ZK_HIP(hipEventRecord(L.done, L.stream));
        ZK_HIP(hipGetLastError());
        guard.lane = nullptr;
        return ticket;
    }

    // Waits for a submission and folds its window/level sums on the host.
    Xyzz<HF> collect(int ticket) {
        if (ticket == BIG_TICKET && big.active) {
            try {
                while (big.npend) {
                    const int oldest = big.pend[0];
                    for (int i = 1; i < big.npend; i++) big.pend[i - 1] = big.pend[i];
                    big.npend--;
                    xyzz_add(big.acc, collect_lane(oldest));
                }
            } catch (...) {
                abandon_big();
                throw;
            }
            big.active = false;
            return big.acc;
        }
        return collect_lane(ticket);
    }
    Xyzz<HF> collect_lane(int ticket) {
        if (ticket < 0 || ticket >= nlanes || !lanes[ticket].busy) throw std::runtime_error("zk_msm: bad ticket");
        Lane &L = lanes[ticket];
        L.busy = false;
        if (L.empty) return Xyzz<HF>::inf();
        ZK_HIP(hipEventSynchronize(L.done));
        if (L.profiled) {
            ZK_HIP(hipEventElapsedTime(&stage_ms[0], L.ev[0], L.ev[1]));
            ZK_HIP(hipEventElapsedTime(&stage_ms[1], L.ev[1], L.ev[2]));
            ZK_HIP(hipEventElapsedTime(&stage_ms[2], L.ev[5], L.ev[6]));  // the accumulate kernel alone
            ZK_HIP(hipEventElapsedTime(&stage_ms[3], L.ev[3], L.ev[4]));
        }
        const uint32_t err_now = *L.h_out.template as<uint32_t>();
        if (err_now != L.err_seen) {
            L.err_seen = err_now;
            throw std::runtime_error("zk_msm: invalid input detected on the device (a non-canonical scalar near or above 2^255 whose signed digits "
                                     "do not fit the windows, or a heavy bucket beyond the plan's task capacity); no result");
        }
        const int c = L.c;
        const uint32_t W = L.single_window ? 1u : (uint32_t)((255 + c - 1) / c), levels = c - 1;
        // Host fold: result = sum_w 2^(c w) * (T_w + sum_l 2^l O_{w,l}); one Horner pass over bit positions.
        const Xyzz<F> *h = out_host(L);
        auto conv = [](const Xyzz<F> &p) { return Xyzz<HF>{HF::from_dev(p.x), HF::from_dev(p.y), HF::from_dev(p.zz), HF::from_dev(p.zzz)}; };
        Xyzz<HF> acc = Xyzz<HF>::inf();
        for (int pos = (int)(c * (W - 1) + levels - 1); pos >= 0; pos--) {
            acc = xyzz_dbl(acc);
            const uint32_t w = pos / c, l = pos % c;
            if (l < levels) xyzz_add(acc, conv(h[(size_t)w * (levels + 1) + l]));
            if (l == 0) xyzz_add(acc, conv(h[(size_t)w * (levels + 1) + levels]));
        }
        return acc;
    }
    int collect_affine(int ticket, uint64_t *out_xy, int *out_is_inf) override {
        write_affine<F>(collect(ticket), out_xy, out_is_inf);
        return ZK_OK;
    }
    int collect_partial(int ticket, uint64_t *out_xyzz) override {
        Xyzz<HF> r = collect(ticket);
        memcpy(out_xyzz, &r, sizeof(r));
        return ZK_OK;
    }
    Xyzz<HF> run(const void *d_scalars, const void *d_points, size_t n, hipStream_t st) { return collect(submit(d_scalars, d_points, n, st)); }

    int run_affine(const void *d_scalars, const void *d_points, size_t n, uint64_t *out_xy, int *out_is_inf, hipStream_t st) override {
        Xyzz<HF> r = run(d_scalars, d_points, n, st);
        write_affine<F>(r, out_xy, out_is_inf);
        return ZK_OK;
    }
    int run_partial(const void *d_scalars, const void *d_points, size_t n, uint64_t *out_xyzz, hipStream_t st) override {
        Xyzz<HF> r = run(d_scalars, d_points, n, st);
        memcpy(out_xyzz, &r, sizeof(r));
        return ZK_OK;
    }
};

}  // namespace zk
