// msm_reduce.h -- bucket reduction of the Pippenger MSM:  R_w = sum_j (j+1) * B_j  per window, in place, no serial running sum.
//
// Replaces (as arithmetic) the tail of what the reference computes term by term in zkp/plonk/kzg.py:59-65 and
// zkp/groth16/proving.py:23-75; there is no counterpart in the reference (its "MSM" has no buckets).
//
//   sum_j (j+1) B_j = T + sum_l 2^l O_l,      T = sum_j B_j,     O_l = sum of the buckets whose index has bit l set.
// The kernels leave T in x[0] and O_l in x[2^l]; the 2^l weights are applied by the host's Horner fold together with the window
// shifts.  All level sums come out of ONE folding scheme, applied to the index bits from the TOP down:
//   step s (half = nb >> (s+1)):   for every live region [base, base + 2 half):   x[base + i] += x[base + i + half],  i < half.
// Folding the region that starts at 0 is the sum over bit (top - s) of the index; the upper half it consumed, [half, 2 half), still
// holds the entries whose bit was SET -- a new live region, whose plain sum (by the same folding, from the next step on) is O of
// that bit.  Live regions at step s: base = 0 and base = nb >> g for g = 1..s; every one has `half` independent additions with
// consecutive lanes on consecutive buckets (coalesced loads, unlike the bottom-up order of rounds 1-4, in which a lane owned 2^m
// CONSECUTIVE buckets), (s+1) * half additions per step, 2 nb in total, depth log2 nb.
//
// Round 5.  A general addition is ~3400 instructions in one lane (G1; 10 100 in G2) and a lone wavefront issues them at 83 % of the
// SIMD's rate (tools/reduce_probe.hip ilp): a level of the tree with fewer additions than lanes costs one whole addition, 6-7 us,
// whatever the occupancy -- 19 + 10 of them in rounds 1-4.  The only way to shorten such a level is more lanes per addition:
//   * a step's whole rounds of NT additions take one lane each (the cheapest form while every SIMD is loaded: the lower levels,
//     at two wavefronts per SIMD, the first two steps without barriers: half >= NT);
//   * what is left of a step -- and every step above -- goes to TEAMS (curve.h team2_add / team4_add: the products of one addition
//     side by side in two lanes, ~2100 instructions deep, or in four, ~1300), two lanes while more than NT/4 additions remain,
//     four below that.
// msm_reduce_block_kernel:   one workgroup per block of 2^BL buckets (BL = 11): steps 0 .. BL-1 inside the block.
// msm_reduce_window_kernel:  one workgroup per window (or per aligned group of blocks): the steps over the block index -- the fold
//                            of region 0 carries every block's T and its BL partial O_l along (BL + 1 + s) * half additions at step s.
#pragma once
#include "curve.h"

namespace zk {

constexpr uint32_t RED_BL = 11;        // log2 buckets per block of msm_reduce_block_kernel
constexpr int RED_BLOCK_NT = 512;      // its workgroup: two wavefronts per SIMD, so that the F_p^2 addition (252 registers) fits
constexpr int RED_WINDOW_NT = 512;     // the window kernel: at most 16 (BL + 5) = 256 additions per step, all by teams of four

template <class F> __device__ __forceinline__ void reduce_add(Xyzz<F> *dst, const Xyzz<F> *src) {
    Xyzz<F> a = *dst;
    const Xyzz<F> b = *src;
    xyzz_add(a, b);
    *dst = a;
}
// One step's additions over the threads of a workgroup: `tasks` independent additions, task q -> (dst, src) by `where`.
// Whole rounds of NT additions (and a remainder beyond NT/2) take one lane each -- with every SIMD loaded that is the cheapest form,
// 3400 lane-instructions per addition in G1 against ~4200 for two lanes of a team --; a remainder of up to NT/2 goes to teams of
// two and one of up to NT/4 to teams of four: the fewer additions a round has, the more its latency and the less its
// instruction count is what the step costs (measured per step: tools/reduce_probe.hip).
template <class F, int NT, bool TEAMS_ONLY, class Where>
__device__ __forceinline__ void reduce_step(uint32_t tasks, Where where) {
    const uint32_t t = threadIdx.x;
    uint32_t done = 0;
    if (!TEAMS_ONLY) {
#pragma unroll 1
        while (tasks - done > NT / 2) {
            if (done + t < tasks) {
                Xyzz<F> *dst, *src;
                where(done + t, dst, src);
                reduce_add(dst, src);
            }
            done += min(tasks - done, (uint32_t)NT);
        }
        if (tasks - done > NT / 4) {
            if (done + (t >> 1) < tasks) {
                Xyzz<F> *dst, *src;
                where(done + (t >> 1), dst, src);
                team2_add(t & 1u, dst, src, QuadDpp{});
            }
            return;
        }
    }
#pragma unroll 1
    for (uint32_t q = done + (t >> 2); q < tasks; q += NT / 4) {
        Xyzz<F> *dst, *src;
        where(q, dst, src);
        team4_add(t & 3u, dst, src, QuadDpp{});
    }
}

template <class F, int NT>
__global__ __launch_bounds__(NT) void msm_reduce_block_kernel(Xyzz<F> *x, uint32_t BL, uint64_t *stamps /* nullptr; tools/reduce_probe.hip: [block][BL + 1] wall-clock ticks */) {
    Xyzz<F> *blk = x + ((size_t)blockIdx.x << BL);
    const uint32_t t = threadIdx.x;
#pragma unroll 1
    for (uint32_t s = 0; s < BL; s++) {
        const uint32_t sh = BL - 1 - s, half = 1u << sh;
        if (stamps != nullptr) {   // when thread 0 starts step s (the barrier-free steps: its own share of them)
            if (half < NT) __syncthreads();
            if (t == 0) stamps[(size_t)blockIdx.x * (BL + 1) + s] = wall_clock64();
        }
        if (half >= NT) {
            // thread t owns the residue class i = t (mod NT) of every region: what it reads now it wrote itself (NT divides half)
#pragma unroll 1
            for (uint32_t g = 0; g <= s; g++) {
                Xyzz<F> *reg = blk + (g ? (1u << (BL - g)) : 0u);
#pragma unroll 1
                for (uint32_t i = t; i < half; i += NT) reduce_add(reg + i, reg + i + half);
            }
            continue;
        }
        __syncthreads();
        reduce_step<F, NT, false>((s + 1) << sh, [&](uint32_t q, Xyzz<F> *&dst, Xyzz<F> *&src) {
            const uint32_t g = q >> sh, i = q & (half - 1u);
            dst = blk + (g ? (1u << (BL - g)) : 0u) + i;
            src = dst + half;
        });
    }
    if (stamps != nullptr) {
        __syncthreads();
        if (t == 0) stamps[(size_t)blockIdx.x * (BL + 1) + BL] = wall_clock64();
    }
}

// One workgroup per window of nb buckets whose blocks of 2^BL buckets are already reduced (T at [0], O_l at [2^l] of every block):
// the steps over the block index h < 2^UL, UL = levels - BL.  Region 0 of step s folds block i + half onto block i in all BL + 1
// values; the regions above fold block totals only.  Then out[w][l] = O_l (l < levels), out[w][levels] = T.  A "window" may also
// be an aligned GROUP of blocks of a larger window (out == nullptr: the results stay in place), and its "blocks" may be such groups
// already reduced (BL = log2 of the group size): the bound-bases mode reduces its 2^19-bucket window in two such launches.
template <class F, int NT>
__global__ __launch_bounds__(NT) void msm_reduce_window_kernel(Xyzz<F> *x, Xyzz<F> *__restrict__ out, uint32_t nb, uint32_t BL, uint32_t levels) {
    Xyzz<F> *win = x + (size_t)blockIdx.x * nb;
    const uint32_t t = threadIdx.x, UL = levels - BL;
#pragma unroll 1
    for (uint32_t s = 0; s < UL; s++) {
        const uint32_t sh = UL - 1 - s, half = 1u << sh;
        if (s) __syncthreads();
        reduce_step<F, NT, true>((BL + 1 + s) << sh, [&](uint32_t q, Xyzz<F> *&dst, Xyzz<F> *&src) {
            const uint32_t grp = q >> sh, i = q & (half - 1u);
            if (grp <= BL)   // region 0: value grp of block i (grp == BL: the block total)
                dst = win + ((size_t)i << BL) + (grp == BL ? 0u : (1u << grp));
            else             // region of the upper bit folded grp - BL steps ago: block totals
                dst = win + ((size_t)((1u << (UL - (grp - BL))) + i) << BL);
            src = dst + ((size_t)half << BL);
        });
    }
    if (out != nullptr) {
        __syncthreads();
        if (t <= levels) out[(size_t)blockIdx.x * (levels + 1) + t] = (t < levels) ? win[1u << t] : win[0];
    }
}

}  // namespace zk
