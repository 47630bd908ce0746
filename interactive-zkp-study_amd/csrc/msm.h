// msm.h -- internal interface between the C-ABI (api.hip) and the MSM pipeline (msm.hip).
#pragma once
#include <stdexcept>
#include "common.h"
#include "curve.h"
#include "host_field.h"

namespace zk {

struct MsmPlanBase {
    int group = 0;
    int device = 0;                        // the device the workspace lives on (set at creation; calls from another current device are refused)
    bool profile = false;                  // record HIP events around the pipeline stages
    float stage_ms[4] = {0.f, 0.f, 0.f, 0.f};  // last run: prepare, bucket sort, accumulate, reduce (device ms)
    virtual ~MsmPlanBase() {}
    virtual int window_bits(size_t n) const = 0;
    virtual int max_in_flight() const = 0;  // submissions that may be outstanding before one must be collected
    virtual int run_affine(const void *d_scalars, const void *d_points, size_t n, uint64_t *out_xy, int *out_is_inf, hipStream_t st) = 0;
    virtual int run_partial(const void *d_scalars, const void *d_points, size_t n, uint64_t *out_xyzz, hipStream_t st) = 0;
    // pipelined form: up to max_in_flight() submissions outstanding per plan
    virtual int submit(const void *d_scalars, const void *d_points, size_t n, hipStream_t st) = 0;
    virtual int collect_affine(int ticket, uint64_t *out_xy, int *out_is_inf) = 0;
    virtual int collect_partial(int ticket, uint64_t *out_xyzz) = 0;
    // bound-bases mode: expand the points into the plan's table (n == 0 unbinds); afterwards d_points == nullptr selects it
    virtual int bind_points(const void *d_points, size_t n, hipStream_t st) = 0;
    virtual int submit_bound(const void *d_scalars, size_t first, size_t n, hipStream_t st) = 0;
};
// The streams MSMs run on: three per device for ALL plans, created together the first time one is needed and never destroyed;
// every submission takes the next one in turn (whatever plan and lane it belongs to: a lane's workspace is free again only once
// its previous submission has been collected, so it may change stream from one submission to the next).
// Why: the HIP runtime spreads streams over its four hardware queues in creation order, and MSMs whose streams share a queue run
// strictly one after the other -- the same Groth16 proof took 9.07 / 9.15 / 9.29 / 9.44 ms with 1 / 0 / 2 / 3 unrelated streams created
// before the prover's plans (the "in-process penalty" of bench.py, whose secondaries create and destroy plans before the prover),
// and with one stream per lane the kernel timelines show the second G1 MSM of a proof, or its G2 one, waiting behind another MSM
// of the same proof on a shared queue (profiles/r05_experiments.md section 2).  Three streams plus the caller's own fill the four
// queues one each, and the three MSMs a Groth16 proof submits back to back (or the three commitments of a PLONK round) always
// get three different ones.
hipStream_t lane_stream_next(int device);

// all_lanes: allocate every lane's workspace now (a plan that will see a stream of MSMs) instead of on first use;
// chunk_log: log2 of the chunk size for MSMs beyond it (0 = the default 2^22; zk_msm_plan_create_ex)
MsmPlanBase *msm_plan_new_g1(size_t max_n, bool all_lanes, int chunk_log);
MsmPlanBase *msm_plan_new_g2(size_t max_n, bool all_lanes, int chunk_log);
inline MsmPlanBase *msm_plan_new(int group, size_t max_n, bool all_lanes = false, int chunk_log = 0) {
    return group == ZK_GROUP_G1 ? msm_plan_new_g1(max_n, all_lanes, chunk_log) : group == ZK_GROUP_G2 ? msm_plan_new_g2(max_n, all_lanes, chunk_log) : nullptr;
}

// Host XYZZ (Montgomery) -> canonical affine limbs; infinity -> zeros + flag.
inline void write_fe_canonical(const HFp &a, uint64_t *out) {
    HFp c = fe_from_mont(a);
    memcpy(out, c.l, 32);
}
inline void write_fe_canonical(const HFp2 &a, uint64_t *out) {
    write_fe_canonical(a.c0, out);
    write_fe_canonical(a.c1, out + 4);
}
template <class F> inline void write_affine(const Xyzz<typename HostOf<F>::type> &p, uint64_t *out_xy, int *out_is_inf) {
    constexpr int L = F::CANON_WORDS / 2;  // 64-bit limbs per coordinate
    if (p.is_inf()) {
        memset(out_xy, 0, 2 * L * 8);
        if (out_is_inf) *out_is_inf = 1;
        return;
    }
    auto a = xyzz_to_affine(p);
    write_fe_canonical(a.x, out_xy);
    write_fe_canonical(a.y, out_xy + L);
    if (out_is_inf) *out_is_inf = 0;
}

inline HFp read_fe_canonical_fp(const uint64_t *in) {
    HFp a;
    memcpy(a.l, in, 32);
    return fe_to_mont(a);
}
template <class HF> inline HF read_fe_canonical(const uint64_t *in);
template <> inline HFp read_fe_canonical<HFp>(const uint64_t *in) { return read_fe_canonical_fp(in); }
template <> inline HFp2 read_fe_canonical<HFp2>(const uint64_t *in) { return HFp2{read_fe_canonical_fp(in), read_fe_canonical_fp(in + 4)}; }

}  // namespace zk
