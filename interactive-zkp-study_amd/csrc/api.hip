// api.hip -- the extern "C" surface of libzkhip.so (declared in include/zkhip.h).
#include <string.h>
#include <array>
#include <map>
#include <memory>
#include <mutex>
#include <vector>
#include "common.h"
#include "curve.h"
#include "host_field.h"
#include "msm.h"
#include "ntt.h"
#include "frvec.h"

namespace zk {

static thread_local std::string g_last_error;
void set_last_error(const std::string &msg) { g_last_error = msg; }

int require_device() {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_last_error("no HIP device available (libzkhip has no CPU fallback)");
        return ZK_ERR_NO_DEVICE;
    }
    return ZK_OK;
}

// msm.h: the process-wide pool of lane streams, three per device, created together on first use (under a lock: plans may be
// created and used from several threads) and left to the runtime at process exit; handed out in turn.
hipStream_t lane_stream_next(int device) {
    struct Pool {
        std::array<hipStream_t, 3> st{};
        unsigned next = 0;
    };
    static std::mutex mu;
    static std::map<int, Pool> pools;
    std::lock_guard<std::mutex> lock(mu);
    auto it = pools.find(device);
    if (it == pools.end()) {
        int cur = -1;
        ZK_HIP(hipGetDevice(&cur));
        if (cur != device) ZK_HIP(hipSetDevice(device));
        Pool p;
        for (auto &st : p.st) ZK_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        if (cur != device) ZK_HIP(hipSetDevice(cur));
        it = pools.emplace(device, p).first;
    }
    Pool &p = it->second;
    const hipStream_t st = p.st[p.next];
    p.next = (p.next + 1) % 3;
    return st;
}

// Plans own device memory, streams and (NTT) a per-device kernel attribute: they work on the device they were created on only.
static int check_plan_device(int plan_device, const char *who) {
    int cur = -1;
    ZK_HIP(hipGetDevice(&cur));
    if (cur == plan_device) return ZK_OK;
    char buf[160];
    snprintf(buf, sizeof(buf), "%s: the plan belongs to device %d but the calling thread's current device is %d", who, plan_device, cur);
    return invalid(buf);
}

static bool fr_canonical_nonzero(const uint64_t v[4]) {
    const HFr m = HFr::modulus();
    if ((v[0] | v[1] | v[2] | v[3]) == 0) return false;
    for (int k = 3; k >= 0; k--) {
        if (v[k] < m.l[k]) return true;
        if (v[k] > m.l[k]) return false;
    }
    return false;
}

static bool scalars_canonical(const uint64_t *s, size_t n) {
    const HFr m = HFr::modulus();
    for (size_t i = 0; i < n; i++) {
        const uint64_t *v = s + 4 * i;
        bool lt = false;
        for (int k = 3; k >= 0; k--) {
            if (v[k] < m.l[k]) { lt = true; break; }
            if (v[k] > m.l[k]) break;
        }
        if (!lt) return false;
    }
    return true;
}

// ------------------------------------------------------------------ batch group-op kernels
template <class F> __device__ __forceinline__ Affine<F> load_affine_canonical(const uint32_t *p) {
    constexpr int NW = F::CANON_WORDS;
    uint32_t w[2 * NW];
    for (int i = 0; i < 2 * NW; i++) w[i] = p[i];
    const F x = fe_load_canonical(w, (F *)nullptr), y = fe_load_canonical(w + NW, (F *)nullptr);
    if (x.is_zero() && y.is_zero()) return Affine<F>::inf();
    return Affine<F>{fe_to_mont(x), fe_to_mont(y)};
}
template <class F> __device__ __forceinline__ void store_affine_canonical(uint32_t *o, const Affine<F> &a) {
    constexpr int NW = F::CANON_WORDS;
    uint32_t w[2 * NW];
    fe_store_canonical(w, fe_from_mont(a.x));
    fe_store_canonical(w + NW, fe_from_mont(a.y));
    for (int i = 0; i < 2 * NW; i++) o[i] = w[i];
}

// op 0: out[i] = p[i] + q[i];  op 1: out[i] = k[i] * p[i];  op 2: out[i] = k[i] * p[0] (fixed base)
template <class F>
__global__ __launch_bounds__(64) void group_op_kernel(int op, const uint32_t *p, const uint32_t *q, uint32_t *out, uint32_t n) {
    constexpr int PW = 2 * F::CANON_WORDS;
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    Xyzz<F> r;
    if (op == 0) {
        r = Xyzz<F>::from_affine(load_affine_canonical<F>(p + (size_t)i * PW));
        xyzz_add_affine(r, load_affine_canonical<F>(q + (size_t)i * PW));
    } else {
        const Affine<F> base = load_affine_canonical<F>(p + (op == 2 ? 0 : (size_t)i * PW));
        uint32_t k[8];
        for (int j = 0; j < 8; j++) k[j] = q[(size_t)i * 8 + j];
        r = xyzz_scalar_mul(base, k);
    }
    store_affine_canonical(out + (size_t)i * PW, xyzz_to_affine(r));
}

// Fixed-base batch k_i * P (setup side: powers of tau / x in the exponent):
// a table T[j][d-1] = d * 2^(8j) * P (32 byte-windows x 255 digits, affine) is built once per call, then
// every scalar costs at most 32 mixed additions and no doubling (the double-and-add it replaces: 256 + ~128).
constexpr int FB_WINDOWS = 32, FB_DIGITS = 255;
template <class F>
__global__ __launch_bounds__(64) void fixed_base_table_kernel(const uint32_t *base, Affine<F> *table) {
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= FB_WINDOWS * FB_DIGITS) return;
    const uint32_t j = i / FB_DIGITS, d = i % FB_DIGITS + 1;
    uint32_t k[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    k[j >> 2] = d << (8 * (j & 3));
    table[i] = xyzz_to_affine(xyzz_scalar_mul(load_affine_canonical<F>(base), k));
}
// Four scalars per thread: their XYZZ results go back to affine with ONE inversion (Montgomery's trick over the four ZZZ) -- the
// inversion (~380 field products) used to cost more than the 32 mixed additions (~290) of a scalar.
constexpr int FB_PER_THREAD = 4;
template <class F>
__global__ __launch_bounds__(64) void fixed_base_eval_kernel(const Affine<F> *__restrict__ table, const uint32_t *__restrict__ scalars,
                                                             uint32_t *__restrict__ out, uint32_t n) {
    constexpr int PW = 2 * F::CANON_WORDS;
    const uint32_t i0 = (blockIdx.x * 64 + threadIdx.x) * FB_PER_THREAD;
    if (i0 >= n) return;
    Xyzz<F> res[FB_PER_THREAD];
    F before[FB_PER_THREAD];
    F run = F::one();
#pragma unroll 1
    for (int r = 0; r < FB_PER_THREAD; r++) {
        Xyzz<F> acc = Xyzz<F>::inf();
        if (i0 + r < n) {
            uint32_t k[8];
            for (int j = 0; j < 8; j++) k[j] = scalars[(size_t)(i0 + r) * 8 + j];
#pragma unroll 1
            for (int j = 0; j < FB_WINDOWS; j++) {
                const uint32_t d = (k[j >> 2] >> (8 * (j & 3))) & 0xffu;
                if (d) xyzz_add_affine(acc, table[j * FB_DIGITS + d - 1]);
            }
        }
        res[r] = acc;
        before[r] = run;
        if (!acc.is_inf()) run = fe_mul(run, acc.zzz);
    }
    F inv = fe_inv(run);
#pragma unroll 1
    for (int r = FB_PER_THREAD - 1; r >= 0; r--) {
        if (i0 + r >= n) continue;
        const Xyzz<F> p = res[r];
        Affine<F> a = Affine<F>::inf();
        if (!p.is_inf()) {
            const F izzz = fe_mul(inv, before[r]);
            inv = fe_mul(inv, p.zzz);
            const F izz = fe_sqr(fe_mul(p.zz, izzz));       // 1/ZZ = (ZZ/ZZZ)^2
            a = Affine<F>{fe_mul(p.x, izz), fe_mul(p.y, izzz)};
        }
        store_affine_canonical(out + (size_t)(i0 + r) * PW, a);
    }
}

template <class F> static int group_op_host(int op, const uint64_t *p, const uint64_t *q, size_t n, uint64_t *out);
template <class F> static int fixed_base_host(const uint64_t *base, const uint64_t *scalars, size_t n, uint64_t *out) {
    constexpr size_t PB = 8 * F::CANON_WORDS;
    if (n == 0) return ZK_OK;
    if (n < 4096) return group_op_host<F>(2, base, scalars, n, out);  // the table alone is 8160 scalar multiplications
    DevBuf dbase(PB), table((size_t)FB_WINDOWS * FB_DIGITS * sizeof(Affine<F>)), dk(n * 32), dout(n * PB);
    ZK_HIP(hipMemcpy(dbase.p, base, PB, hipMemcpyHostToDevice));
    ZK_HIP(hipMemcpy(dk.p, scalars, n * 32, hipMemcpyHostToDevice));
    hipLaunchKernelGGL((fixed_base_table_kernel<F>), dim3((FB_WINDOWS * FB_DIGITS + 63) / 64), dim3(64), 0, 0, dbase.as<uint32_t>(), table.as<Affine<F>>());
    hipLaunchKernelGGL((fixed_base_eval_kernel<F>), dim3((unsigned)((n + 64 * FB_PER_THREAD - 1) / (64 * FB_PER_THREAD))), dim3(64), 0, 0, table.as<Affine<F>>(), dk.as<uint32_t>(),
                       dout.as<uint32_t>(), (uint32_t)n);
    ZK_HIP(hipGetLastError());
    ZK_HIP(hipMemcpy(out, dout.p, n * PB, hipMemcpyDeviceToHost));
    return ZK_OK;
}

// The same batch on DEVICE buffers (at-scale setup: the scalars come out of F_r vector kernels and the points go straight into an MSM
// plan's table, so neither visits the host).  Enqueued on `st`; the table is the call's own, hence one synchronisation of `st`
// before it is released -- this is key generation, not the proving path.
template <class F> static int fixed_base_dev(const uint64_t *base, const void *d_scalars, size_t n, void *d_out, hipStream_t st) {
    constexpr size_t PB = 8 * F::CANON_WORDS;
    if (n == 0) return ZK_OK;
    if (n > 0xffffffffull) return invalid("zk_fixed_base_dev: n must fit 32 bits");
    DevBuf dbase(PB);
    ZK_HIP(hipMemcpyAsync(dbase.p, base, PB, hipMemcpyHostToDevice, st));
    if (n < 4096) {
        hipLaunchKernelGGL((group_op_kernel<F>), dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, 2, dbase.as<uint32_t>(), static_cast<const uint32_t *>(d_scalars),
                           static_cast<uint32_t *>(d_out), (uint32_t)n);
        ZK_HIP(hipGetLastError());
        ZK_HIP(hipStreamSynchronize(st));
        return ZK_OK;
    }
    DevBuf table((size_t)FB_WINDOWS * FB_DIGITS * sizeof(Affine<F>));
    hipLaunchKernelGGL((fixed_base_table_kernel<F>), dim3((FB_WINDOWS * FB_DIGITS + 63) / 64), dim3(64), 0, st, dbase.as<uint32_t>(), table.as<Affine<F>>());
    hipLaunchKernelGGL((fixed_base_eval_kernel<F>), dim3((unsigned)((n + 64 * FB_PER_THREAD - 1) / (64 * FB_PER_THREAD))), dim3(64), 0, st, table.as<Affine<F>>(),
                       static_cast<const uint32_t *>(d_scalars), static_cast<uint32_t *>(d_out), (uint32_t)n);
    ZK_HIP(hipGetLastError());
    ZK_HIP(hipStreamSynchronize(st));
    return ZK_OK;
}

// Arithmetic-rate probes (zk_measure_rate): the integer-ALU ceilings the MSM kernels are priced against.
// Every thread runs a dependent chain, the grid oversubscribes the chip, so the rate is the chip-wide
// issue limit of the field multiplication / the mixed addition as compiled into this library.
__global__ __launch_bounds__(256) void rate_modmul_kernel(Fp *out, int iters, uint32_t sink_thread) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    Fp x = Fp::one(), y = Fp::one();
    x.l[0] = (x.l[0] + i) & 0x1fffffffu;
    y.l[1] = (y.l[1] + (i >> 3)) & 0x1fffffffu;
#pragma unroll 1
    for (int k = 0; k < iters; k++) {
        x = fe_mul(x, y);
        y = fe_mul(y, x);
    }
    if (i == sink_thread) out[0] = fe_add(x, y);  // sink_thread is outside the grid: keeps the chain alive, never stores
}
__global__ __launch_bounds__(64, 3) void rate_madd_kernel(const uint32_t *gen_xy, Xyzz<Fp> *out, int iters, uint32_t sink_thread) {
    Affine<Fp> g = load_affine_canonical<Fp>(gen_xy);
    if (threadIdx.x & 1) g.y = fe_neg<2>(g.y);  // lane-dependent data: keeps the chain on the vector ALU
    Xyzz<Fp> acc = Xyzz<Fp>::inf();
#pragma unroll 1
    for (int k = 0; k < iters; k++) xyzz_add_affine(acc, g);
    if (blockIdx.x * 64 + threadIdx.x == sink_thread) out[0] = acc;
}

// The chip's bare v_mad_u64_u32 issue rate: four independent accumulator chains per thread, nothing else in the loop
// (the probe of tools/ubench.hip).  This ceiling does not depend on how field.h arranges a modular product.
__global__ __launch_bounds__(256) void rate_mad_kernel(uint32_t *out, int iters, uint32_t sink_thread) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    const uint32_t a = i * 2654435761u + 12345u, b = a ^ 0x9e3779b9u;
    uint64_t q0 = a, q1 = b, q2 = a + 3, q3 = b + 5;
#pragma unroll 1
    for (int k = 0; k < iters; k++) {
#pragma unroll
        for (int u = 0; u < 8; u++)
            asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n"
                         "v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3"
                         : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3)
                         : "v"(a), "v"(b)
                         : "vcc");
    }
    const uint64_t x = q0 ^ q1 ^ q2 ^ q3;
    if (i == sink_thread) out[0] = (uint32_t)x ^ (uint32_t)(x >> 32);
}

static int measure_rate(int what, double *out_per_sec) {
    if (!out_per_sec) return invalid("zk_measure_rate: null output");
    const int iters = what == 0 ? 512 : what == 1 ? 96 : 2048;
    DevBuf sink(sizeof(Xyzz<Fp>)), gen(64);
    const uint64_t g[8] = {1, 0, 0, 0, 2, 0, 0, 0};
    ZK_HIP(hipMemcpy(gen.p, g, 64, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    ZK_HIP(hipEventCreate(&e0));
    ZK_HIP(hipEventCreate(&e1));
    double units = 0;
    auto launch = [&]() {
        if (what == 0) {
            const unsigned blocks = 256 * 16;
            hipLaunchKernelGGL(rate_modmul_kernel, dim3(blocks), dim3(256), 0, 0, sink.as<Fp>(), iters, 0xffffffffu);
            units = (double)blocks * 256 * iters * 2;
        } else if (what == 2) {
            const unsigned blocks = 256 * 8;
            hipLaunchKernelGGL(rate_mad_kernel, dim3(blocks), dim3(256), 0, 0, sink.as<uint32_t>(), iters, 0xffffffffu);
            units = (double)blocks * 256 * iters * 32;
        } else {
            const unsigned blocks = 256 * 4 * 12;  // the accumulate kernel's shape: one wave per workgroup
            hipLaunchKernelGGL(rate_madd_kernel, dim3(blocks), dim3(64), 0, 0, gen.as<uint32_t>(), sink.as<Xyzz<Fp>>(), iters, 0xffffffffu);
            units = (double)blocks * 64 * iters;
        }
    };
    launch();
    ZK_HIP(hipDeviceSynchronize());
    const int reps = 3;
    ZK_HIP(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; r++) launch();
    ZK_HIP(hipEventRecord(e1, 0));
    ZK_HIP(hipEventSynchronize(e1));
    float ms = 0;
    ZK_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    ZK_HIP(hipGetLastError());
    *out_per_sec = units * reps / (ms * 1e-3);
    return ZK_OK;
}

template <class F> static int group_op_host(int op, const uint64_t *p, const uint64_t *q, size_t n, uint64_t *out) {
    constexpr size_t PB = 8 * F::CANON_WORDS;  // bytes of one canonical affine point
    if (n == 0) return ZK_OK;
    const size_t p_bytes = (op == 2 ? 1 : n) * PB, q_bytes = n * (op == 0 ? PB : 32);
    DevBuf dp(p_bytes), dq(q_bytes), dout(n * PB);
    ZK_HIP(hipMemcpy(dp.p, p, p_bytes, hipMemcpyHostToDevice));
    ZK_HIP(hipMemcpy(dq.p, q, q_bytes, hipMemcpyHostToDevice));
    hipLaunchKernelGGL((group_op_kernel<F>), dim3((unsigned)((n + 63) / 64)), dim3(64), 0, 0, op, dp.as<uint32_t>(), dq.as<uint32_t>(),
                       dout.as<uint32_t>(), (uint32_t)n);
    ZK_HIP(hipGetLastError());
    ZK_HIP(hipMemcpy(out, dout.p, n * PB, hipMemcpyDeviceToHost));
    return ZK_OK;
}

// ------------------------------------------------------------------ plan caches behind the host-buffer entry points
// zk_msm_g1 / zk_msm_g2 / zk_ntt_fr are what the reference-signature facade calls (commit, proof_a/b/c, fft / ifft): one call per
// primitive, host buffers in and out.  Building a plan per call cost milliseconds (NTT: host power tables, uploads, a device
// synchronisation; MSM: workspace allocation) -- more than the transform of a toy-sized proof.  Plans are therefore kept per
// thread and device: NTT plans by log_n, MSM plans by group and size class (next power of two), a few of each (least recently
// used goes first), together with their staging buffers.  zk_cache_clear() drops the calling thread's; zk_cache_stats() counts.
struct CacheStats {
    uint64_t ntt_builds = 0, ntt_hits = 0, msm_builds = 0, msm_hits = 0;
};
static thread_local CacheStats g_cache_stats;
struct HostMsmCtx {
    std::unique_ptr<MsmPlanBase> plan;
    DevBuf in;
    size_t cls = 0;
    uint64_t used = 0;
};
struct HostNttCtx {
    std::unique_ptr<NttPlan> plan;
    DevBuf buf;
    uint64_t used = 0;
};
static thread_local std::map<std::pair<int, int>, std::vector<std::unique_ptr<HostMsmCtx>>> g_msm_cache;   // (device, group)
static thread_local std::map<std::pair<int, unsigned>, std::unique_ptr<HostNttCtx>> g_ntt_cache;            // (device, log_n)
static thread_local uint64_t g_cache_clock = 0;
constexpr size_t MSM_CACHE_PER_GROUP = 3, NTT_CACHE_PLANS = 6;
// Calls beyond these sizes build their plan for the call and release it afterwards, so that what a thread can leave pinned in HBM
// stays modest: per group at most three size classes of at most 2^20 points (staging 96 / 160 B per point plus one lane's workspace,
// about 0.5 GB for G1 and 1.2 GB for G2 at the largest class) and six NTT plans of at most 2^22 elements (128 MB of staging plus
// as much scratch each) -- a few GB in the worst case, against several GB PER CALL SIZE before round 4.  The facade works at the
// reference's toy sizes; provers at scale hold their own plans (zk_msm_plan_create / zk_ntt_plan_create).
constexpr size_t MSM_CACHE_MAX_POINTS = (size_t)1 << 20;
constexpr unsigned NTT_CACHE_MAX_LOG = 22;

static HostMsmCtx &host_msm_ctx(int group, size_t n, size_t point_bytes) {
    int dev = 0;
    ZK_HIP(hipGetDevice(&dev));
    size_t cls = 4096;                       // the facade's toy-size proofs and verifier combinations share the smallest class
    while (cls < n) cls <<= 1;
    auto &slots = g_msm_cache[std::make_pair(dev, group)];
    for (auto &c : slots)
        if (c->cls == cls) {
            g_cache_stats.msm_hits++;
            c->used = ++g_cache_clock;
            return *c;
        }
    if (slots.size() >= MSM_CACHE_PER_GROUP) {
        size_t lru = 0;
        for (size_t i = 1; i < slots.size(); i++)
            if (slots[i]->used < slots[lru]->used) lru = i;
        slots.erase(slots.begin() + lru);
    }
    std::unique_ptr<HostMsmCtx> c(new HostMsmCtx);
    c->cls = cls;
    c->plan.reset(msm_plan_new(group, cls));
    c->in.alloc(cls * (32 + point_bytes));
    c->used = ++g_cache_clock;
    g_cache_stats.msm_builds++;
    slots.push_back(std::move(c));
    return *slots.back();
}
static HostNttCtx &host_ntt_ctx(unsigned log_n) {
    int dev = 0;
    ZK_HIP(hipGetDevice(&dev));
    auto it = g_ntt_cache.find(std::make_pair(dev, log_n));
    if (it != g_ntt_cache.end()) {
        g_cache_stats.ntt_hits++;
        it->second->used = ++g_cache_clock;
        return *it->second;
    }
    if (g_ntt_cache.size() >= NTT_CACHE_PLANS) {
        auto lru = g_ntt_cache.begin();
        for (auto j = g_ntt_cache.begin(); j != g_ntt_cache.end(); ++j)
            if (j->second->used < lru->second->used) lru = j;
        g_ntt_cache.erase(lru);
    }
    std::unique_ptr<HostNttCtx> c(new HostNttCtx);
    c->plan.reset(new NttPlan(log_n));
    c->buf.alloc(((size_t)1 << log_n) * 32);
    c->used = ++g_cache_clock;
    g_cache_stats.ntt_builds++;
    return *(g_ntt_cache[std::make_pair(dev, log_n)] = std::move(c));
}

// Host-buffer MSM (cached plan of the size class, see above).
template <class F> static int msm_host(int group, const uint64_t *scalars, const uint64_t *points, size_t n, uint64_t *out_xy, int *out_is_inf) {
    constexpr size_t PB = 8 * F::CANON_WORDS;
    if (n == 0) {
        memset(out_xy, 0, PB);
        if (out_is_inf) *out_is_inf = 1;
        return ZK_OK;
    }
    if (!scalars_canonical(scalars, n)) return invalid("zk_msm: scalar not canonical (>= r)");
    std::unique_ptr<HostMsmCtx> once;
    if (n > MSM_CACHE_MAX_POINTS) {
        once.reset(new HostMsmCtx);
        once->plan.reset(msm_plan_new(group, n));
        once->in.alloc(n * (32 + PB));
        g_cache_stats.msm_builds++;
    }
    HostMsmCtx &ctx = once ? *once : host_msm_ctx(group, n, PB);
    char *ds = ctx.in.as<char>(), *dp = ds + n * 32;
    ZK_HIP(hipMemcpy(ds, scalars, n * 32, hipMemcpyHostToDevice));
    ZK_HIP(hipMemcpy(dp, points, n * PB, hipMemcpyHostToDevice));
    return ctx.plan->run_affine(ds, dp, n, out_xy, out_is_inf, 0);
}

template <class F> static int fold_partials(const uint64_t *partials, size_t count, uint64_t *out_xy, int *out_is_inf) {
    typedef typename HostOf<F>::type HF;
    Xyzz<HF> acc = Xyzz<HF>::inf();
    for (size_t i = 0; i < count; i++) {
        Xyzz<HF> p;
        memcpy(&p, partials + i * (sizeof(Xyzz<HF>) / 8), sizeof(p));
        xyzz_add(acc, p);
    }
    write_affine<F>(acc, out_xy, out_is_inf);
    return ZK_OK;
}

int pairing_product(const uint64_t *g1_points, const uint64_t *g2_points, size_t n, uint64_t *out, int *is_one);  // pairing.hip

}  // namespace zk

using namespace zk;

struct zk_msm_plan {
    std::unique_ptr<MsmPlanBase> impl;
};
struct zk_ntt_plan {
    std::unique_ptr<NttPlan> impl;
};
struct zk_frvec {
    FrVecScratch impl;
    int device = 0;  // the scratch buffers are allocated on the device that was current at creation
};

extern "C" {

const char *zk_last_error(void) { return g_last_error.c_str(); }
int zk_version(void) { return 1; }

int zk_device_count(int *count) {
    if (!count) return invalid("zk_device_count: null pointer");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
    *count = n;
    return ZK_OK;
}
int zk_set_device(int device) {
    return guarded([&] {
        int rc = require_device();
        if (rc) return rc;
        ZK_HIP(hipSetDevice(device));
        return ZK_OK;
    });
}

int zk_msm_g1(const uint64_t *scalars, const uint64_t *points, size_t n, uint64_t out_xy[8], int *out_is_inf) {
    return guarded([&] {
        if (!out_xy || (n && (!scalars || !points))) return invalid("zk_msm_g1: null pointer");
        int rc = require_device();
        if (rc) return rc;
        return msm_host<Fp>(ZK_GROUP_G1, scalars, points, n, out_xy, out_is_inf);
    });
}
int zk_msm_g2(const uint64_t *scalars, const uint64_t *points, size_t n, uint64_t out_xy[16], int *out_is_inf) {
    return guarded([&] {
        if (!out_xy || (n && (!scalars || !points))) return invalid("zk_msm_g2: null pointer");
        int rc = require_device();
        if (rc) return rc;
        return msm_host<Fp2>(ZK_GROUP_G2, scalars, points, n, out_xy, out_is_inf);
    });
}

int zk_msm_plan_create(int group, size_t max_n, zk_msm_plan **plan) {
    return guarded([&] {
        if (!plan || (group != ZK_GROUP_G1 && group != ZK_GROUP_G2) || max_n == 0 || max_n > ((size_t)1 << 30))
            return invalid("zk_msm_plan_create: bad argument");
        int rc = require_device();
        if (rc) return rc;
        zk_msm_plan *p = new zk_msm_plan;
        p->impl.reset(msm_plan_new(group, max_n, true));
        *plan = p;
        return ZK_OK;
    });
}
int zk_msm_plan_create_ex(int group, size_t max_n, int chunk_log2, zk_msm_plan **plan) {
    return guarded([&] {
        if (!plan || (group != ZK_GROUP_G1 && group != ZK_GROUP_G2) || max_n == 0 || max_n > ((size_t)1 << 30))
            return invalid("zk_msm_plan_create_ex: bad argument");
        if (chunk_log2 != 0 && (chunk_log2 < 12 || chunk_log2 > 24)) return invalid("zk_msm_plan_create_ex: chunk_log2 must be 0 (default) or 12..24");
        int rc = require_device();
        if (rc) return rc;
        zk_msm_plan *p = new zk_msm_plan;
        p->impl.reset(msm_plan_new(group, max_n, true, chunk_log2));
        *plan = p;
        return ZK_OK;
    });
}
int zk_msm_plan_destroy(zk_msm_plan *plan) {
    delete plan;
    return ZK_OK;
}
int zk_msm_plan_profile(zk_msm_plan *plan, int enable) {
    if (!plan) return ZK_ERR_INVALID;
    plan->impl->profile = enable != 0;
    return ZK_OK;
}
int zk_msm_plan_stage_ms(const zk_msm_plan *plan, float out_ms[4]) {
    if (!plan || !out_ms) return ZK_ERR_INVALID;
    for (int i = 0; i < 4; i++) out_ms[i] = plan->impl->stage_ms[i];
    return ZK_OK;
}
int zk_msm_plan_window_bits(const zk_msm_plan *plan, size_t n) { return plan ? plan->impl->window_bits(n) : ZK_ERR_INVALID; }
int zk_msm_plan_bind_points(zk_msm_plan *plan, const void *d_points, size_t n, void *stream) {
    return guarded([&] {
        if (!plan || (n && !d_points)) return invalid("zk_msm_plan_bind_points: null pointer");
        if (int rc = check_plan_device(plan->impl->device, "zk_msm_plan_bind_points")) return rc;
        return plan->impl->bind_points(d_points, n, (hipStream_t)stream);
    });
}
int zk_msm_plan_max_in_flight(const zk_msm_plan *plan) { return plan ? plan->impl->max_in_flight() : ZK_ERR_INVALID; }
int zk_msm_dev(zk_msm_plan *plan, const void *d_scalars, const void *d_points, size_t n, uint64_t *out_xy, int *out_is_inf, void *stream) {
    return guarded([&] {
        if (!plan || !out_xy || (n && !d_scalars)) return invalid("zk_msm_dev: null pointer");
        if (int rc = check_plan_device(plan->impl->device, "zk_msm_dev")) return rc;
        return plan->impl->run_affine(d_scalars, d_points, n, out_xy, out_is_inf, (hipStream_t)stream);
    });
}
int zk_msm_dev_partial(zk_msm_plan *plan, const void *d_scalars, const void *d_points, size_t n, uint64_t *out_xyzz, void *stream) {
    return guarded([&] {
        if (!plan || !out_xyzz || (n && !d_scalars)) return invalid("zk_msm_dev_partial: null pointer");
        if (int rc = check_plan_device(plan->impl->device, "zk_msm_dev_partial")) return rc;
        return plan->impl->run_partial(d_scalars, d_points, n, out_xyzz, (hipStream_t)stream);
    });
}
int zk_msm_submit(zk_msm_plan *plan, const void *d_scalars, const void *d_points, size_t n, void *stream, int *out_ticket) {
    return guarded([&] {
        if (!plan || !out_ticket || (n && !d_scalars)) return invalid("zk_msm_submit: null pointer");
        if (int rc = check_plan_device(plan->impl->device, "zk_msm_submit")) return rc;
        *out_ticket = plan->impl->submit(d_scalars, d_points, n, (hipStream_t)stream);
        return ZK_OK;
    });
}
int zk_msm_submit_bound(zk_msm_plan *plan, const void *d_scalars, size_t first, size_t n, void *stream, int *out_ticket) {
    return guarded([&] {
        if (!plan || !out_ticket || (n && !d_scalars)) return invalid("zk_msm_submit_bound: null pointer");
        if (int rc = check_plan_device(plan->impl->device, "zk_msm_submit_bound")) return rc;
        *out_ticket = plan->impl->submit_bound(d_scalars, first, n, (hipStream_t)stream);
        return ZK_OK;
    });
}
int zk_msm_collect(zk_msm_plan *plan, int ticket, uint64_t *out_xy, int *out_is_inf) {
    return guarded([&] {
        if (!plan || !out_xy) return invalid("zk_msm_collect: null pointer");
        return plan->impl->collect_affine(ticket, out_xy, out_is_inf);
    });
}
int zk_msm_collect_partial(zk_msm_plan *plan, int ticket, uint64_t *out_xyzz) {
    return guarded([&] {
        if (!plan || !out_xyzz) return invalid("zk_msm_collect_partial: null pointer");
        return plan->impl->collect_partial(ticket, out_xyzz);
    });
}
int zk_msm_partial_limbs(int group) { return group == ZK_GROUP_G1 ? 16 : group == ZK_GROUP_G2 ? 32 : ZK_ERR_INVALID; }
int zk_msm_fold_partials(int group, const uint64_t *partials, size_t count, uint64_t *out_xy, int *out_is_inf) {
    return guarded([&] {
        if (!out_xy || (count && !partials)) return invalid("zk_msm_fold_partials: null pointer");
        if (group == ZK_GROUP_G1) return fold_partials<Fp>(partials, count, out_xy, out_is_inf);
        if (group == ZK_GROUP_G2) return fold_partials<Fp2>(partials, count, out_xy, out_is_inf);
        return invalid("zk_msm_fold_partials: bad group");
    });
}

int zk_fixed_base_g1(const uint64_t base_xy[8], const uint64_t *scalars, size_t n, uint64_t *out_points) {
    return guarded([&] {
        if (!base_xy || (n && (!scalars || !out_points))) return invalid("zk_fixed_base_g1: null pointer");
        int rc = require_device();
        if (rc) return rc;
        return fixed_base_host<Fp>(base_xy, scalars, n, out_points);
    });
}
int zk_fixed_base_g2(const uint64_t base_xy[16], const uint64_t *scalars, size_t n, uint64_t *out_points) {
    return guarded([&] {
        if (!base_xy || (n && (!scalars || !out_points))) return invalid("zk_fixed_base_g2: null pointer");
        int rc = require_device();
        if (rc) return rc;
        return fixed_base_host<Fp2>(base_xy, scalars, n, out_points);
    });
}
int zk_fixed_base_g1_dev(const uint64_t base_xy[8], const void *d_scalars, size_t n, void *d_out_points, void *stream) {
    return guarded([&] {
        if (!base_xy || (n && (!d_scalars || !d_out_points))) return invalid("zk_fixed_base_g1_dev: null pointer");
        int rc = require_device();
        if (rc) return rc;
        return fixed_base_dev<Fp>(base_xy, d_scalars, n, d_out_points, static_cast<hipStream_t>(stream));
    });
}
int zk_fixed_base_g2_dev(const uint64_t base_xy[16], const void *d_scalars, size_t n, void *d_out_points, void *stream) {
    return guarded([&] {
        if (!base_xy || (n && (!d_scalars || !d_out_points))) return invalid("zk_fixed_base_g2_dev: null pointer");
        int rc = require_device();
        if (rc) return rc;
        return fixed_base_dev<Fp2>(base_xy, d_scalars, n, d_out_points, static_cast<hipStream_t>(stream));
    });
}
int zk_measure_rate(int what, double *out_per_sec) {
    return guarded([&]() -> int {
        int rc = require_device();
        if (rc) return rc;
        if (what < 0 || what > 2) return invalid("zk_measure_rate: what must be 0 (F_p multiplications), 1 (G1 mixed additions) or 2 (v_mad_u64_u32)");
        return measure_rate(what, out_per_sec);
    });
}

int zk_group_op(int group, int op, const uint64_t *p, const uint64_t *q_or_scalar, size_t n, uint64_t *out) {
    return guarded([&] {
        if ((n && (!p || !q_or_scalar || !out)) || (op != 0 && op != 1)) return invalid("zk_group_op: bad argument");
        int rc = require_device();
        if (rc) return rc;
        if (group == ZK_GROUP_G1) return group_op_host<Fp>(op, p, q_or_scalar, n, out);
        if (group == ZK_GROUP_G2) return group_op_host<Fp2>(op, p, q_or_scalar, n, out);
        return invalid("zk_group_op: bad group");
    });
}

int zk_pairing(const uint64_t g1_xy[8], const uint64_t g2_xy[16], uint64_t out_fq12[48]) {
    return guarded([&] {
        if (!g1_xy || !g2_xy || !out_fq12) return invalid("zk_pairing: null pointer");
        return pairing_product(g1_xy, g2_xy, 1, out_fq12, nullptr);
    });
}
int zk_pairing_check(const uint64_t *g1_points, const uint64_t *g2_points, size_t n, int *out_is_one) {
    return guarded([&] {
        if (!out_is_one || (n && (!g1_points || !g2_points))) return invalid("zk_pairing_check: null pointer");
        return pairing_product(g1_points, g2_points, n, nullptr, out_is_one);
    });
}

int zk_ntt_plan_create(unsigned log_n, zk_ntt_plan **plan) {
    return guarded([&] {
        if (!plan || log_n > 28) return invalid("zk_ntt_plan_create: log_n must be <= 28");
        int rc = require_device();
        if (rc) return rc;
        zk_ntt_plan *p = new zk_ntt_plan;
        p->impl.reset(new NttPlan(log_n));
        *plan = p;
        return ZK_OK;
    });
}
int zk_ntt_plan_destroy(zk_ntt_plan *plan) {
    delete plan;
    return ZK_OK;
}
int zk_ntt_dev(zk_ntt_plan *plan, void *d_data, int inverse, const uint64_t coset_shift[4], void *stream) {
    return guarded([&] {
        if (!plan || !d_data) return invalid("zk_ntt_dev: null pointer");
        if (int rc = check_plan_device(plan->impl->device(), "zk_ntt_dev")) return rc;
        if (coset_shift && !fr_canonical_nonzero(coset_shift)) return invalid("zk_ntt_dev: coset_shift must be a canonical non-zero element of F_r");
        plan->impl->run(d_data, inverse != 0, coset_shift, (hipStream_t)stream);
        return ZK_OK;
    });
}
int zk_ntt_dev_padded(zk_ntt_plan *plan, const void *d_in, void *d_out, size_t in_len, int inverse, const uint64_t coset_shift[4], void *stream) {
    return guarded([&] {
        if (!plan || !d_out || (in_len && !d_in)) return invalid("zk_ntt_dev_padded: null pointer");
        if (int rc = check_plan_device(plan->impl->device(), "zk_ntt_dev_padded")) return rc;
        if (coset_shift && !fr_canonical_nonzero(coset_shift)) return invalid("zk_ntt_dev_padded: coset_shift must be a canonical non-zero element of F_r");
        plan->impl->run_padded(d_in, d_out, in_len, inverse != 0, coset_shift, (hipStream_t)stream);
        return ZK_OK;
    });
}
int zk_ntt_dev_multi(zk_ntt_plan *plan, unsigned jobs, const void *const *d_in, void *const *d_out, size_t in_len, int inverse,
                     const uint64_t coset_shift[4], void *stream) {
    return guarded([&] {
        if (!plan || (jobs && (!d_in || !d_out))) return invalid("zk_ntt_dev_multi: null pointer");
        if (jobs > NTT_MULTI_MAX) return invalid("zk_ntt_dev_multi: at most 4 transforms per call");
        if (int rc = check_plan_device(plan->impl->device(), "zk_ntt_dev_multi")) return rc;
        // byte ranges: an output is n elements, an input min(in_len, n); a job may transform in place (same address), nothing else may overlap
        const size_t n = (size_t)1 << plan->impl->log_n(), out_bytes = n * 32, in_bytes = std::min(in_len, n) * 32;
        auto overlap = [](const void *p, size_t pl, const void *q, size_t ql) {
            const uintptr_t a = reinterpret_cast<uintptr_t>(p), b = reinterpret_cast<uintptr_t>(q);
            return pl && ql && a < b + ql && b < a + pl;
        };
        for (unsigned b = 0; b < jobs; b++) {
            if (!d_out[b] || (in_len && !d_in[b])) return invalid("zk_ntt_dev_multi: null buffer");
            if (d_in[b] != d_out[b] && overlap(d_in[b], in_bytes, d_out[b], out_bytes)) return invalid("zk_ntt_dev_multi: a transform's input and output overlap without being the same buffer");
            for (unsigned c = 0; c < b; c++)
                if (overlap(d_out[b], out_bytes, d_out[c], out_bytes) || overlap(d_out[b], out_bytes, d_in[c], in_bytes) || overlap(d_in[b], in_bytes, d_out[c], out_bytes))
                    return invalid("zk_ntt_dev_multi: a buffer is written by one transform and used by another");
        }
        if (coset_shift && !fr_canonical_nonzero(coset_shift)) return invalid("zk_ntt_dev_multi: coset_shift must be a canonical non-zero element of F_r");
        plan->impl->run_multi(d_in, d_out, jobs, in_len, inverse != 0, coset_shift, (hipStream_t)stream);
        return ZK_OK;
    });
}
int zk_ntt_dev_batch(zk_ntt_plan *plan, void *d_data, unsigned batch, int inverse, void *stream) {
    return guarded([&] {
        if (!plan || (batch && !d_data)) return invalid("zk_ntt_dev_batch: null pointer");
        if (int rc = check_plan_device(plan->impl->device(), "zk_ntt_dev_batch")) return rc;
        plan->impl->run(d_data, inverse != 0, nullptr, (hipStream_t)stream, batch);
        return ZK_OK;
    });
}
int zk_ntt_dev_io(zk_ntt_plan *plan, const void *d_in, void *d_out, unsigned batch, int inverse, int in_layout, int out_layout, unsigned log_block,
                  uint64_t row0, const zk_ntt_plan *big, int tw_inverse, void *stream) {
    return guarded([&] {
        if (!plan || (batch && (!d_in || !d_out))) return invalid("zk_ntt_dev_io: null pointer");
        if (int rc = check_plan_device(plan->impl->device(), "zk_ntt_dev_io")) return rc;
        if (big && big->impl->device() != plan->impl->device()) return invalid("zk_ntt_dev_io: the two plans live on different devices");
        plan->impl->run_io(d_in, d_out, inverse != 0, batch, in_layout, out_layout, log_block, row0, big ? big->impl.get() : nullptr, tw_inverse != 0,
                           (hipStream_t)stream);
        return ZK_OK;
    });
}
int zk_ntt_twiddle_dev(zk_ntt_plan *plan, void *d_data, unsigned log_cols, uint64_t rows, uint64_t row0, int inverse, void *stream) {
    return guarded([&] {
        if (!plan || (rows && !d_data)) return invalid("zk_ntt_twiddle_dev: null pointer");
        plan->impl->twiddle_2d(d_data, log_cols, rows, row0, inverse != 0, (hipStream_t)stream);
        return ZK_OK;
    });
}
int zk_ntt_fr(uint64_t *data, unsigned log_n, int inverse, const uint64_t coset_shift[4]) {
    return guarded([&] {
        if (!data || log_n > 28) return invalid("zk_ntt_fr: bad argument");
        int rc = require_device();
        if (rc) return rc;
        const size_t n = (size_t)1 << log_n;
        if (!scalars_canonical(data, n)) return invalid("zk_ntt_fr: element not canonical (>= r)");
        if (coset_shift && !fr_canonical_nonzero(coset_shift)) return invalid("zk_ntt_fr: coset_shift must be a canonical non-zero element of F_r");
        std::unique_ptr<HostNttCtx> once;
        if (log_n > NTT_CACHE_MAX_LOG) {
            once.reset(new HostNttCtx);
            once->plan.reset(new NttPlan(log_n));
            once->buf.alloc(n * 32);
            g_cache_stats.ntt_builds++;
        }
        HostNttCtx &ctx = once ? *once : host_ntt_ctx(log_n);   // plan (tables) + staging buffer of this size, built once per thread and device
        ZK_HIP(hipMemcpy(ctx.buf.p, data, n * 32, hipMemcpyHostToDevice));
        ctx.plan->run(ctx.buf.p, inverse != 0, coset_shift, 0);
        ZK_HIP(hipStreamSynchronize(0));
        ZK_HIP(hipMemcpy(data, ctx.buf.p, n * 32, hipMemcpyDeviceToHost));
        return ZK_OK;
    });
}
int zk_cache_clear(void) {
    return guarded([&] {
        g_msm_cache.clear();
        g_ntt_cache.clear();
        return ZK_OK;
    });
}
int zk_cache_stats(uint64_t out[4]) {
    if (!out) return invalid("zk_cache_stats: null pointer");
    out[0] = g_cache_stats.ntt_builds;
    out[1] = g_cache_stats.ntt_hits;
    out[2] = g_cache_stats.msm_builds;
    out[3] = g_cache_stats.msm_hits;
    return ZK_OK;
}
int zk_fr_spmv_dev(const void *d_row_ptr, const void *d_col, const void *d_vals, const void *d_x, void *d_y, size_t rows, void *stream) {
    return guarded([&] {
        if (rows && (!d_row_ptr || !d_col || !d_vals || !d_x || !d_y)) return invalid("zk_fr_spmv_dev: null pointer");
        fr_spmv(d_row_ptr, d_col, d_vals, d_x, d_y, rows, (hipStream_t)stream);
        return ZK_OK;
    });
}
int zk_fr_lincomb_dev(void *d_out, const void *const *d_in, const uint64_t *coeffs, unsigned k, const uint64_t constant[4], size_t n, void *stream) {
    return guarded([&] {
        if (n && (!d_out || (k && (!d_in || !coeffs)))) return invalid("zk_fr_lincomb_dev: null pointer");
        if (k > FR_LINCOMB_MAX) return invalid("zk_fr_lincomb_dev: at most 8 input vectors");
        for (unsigned j = 0; j < k; j++)
            if (!scalars_canonical(coeffs + 4 * j, 1)) return invalid("zk_fr_lincomb_dev: coefficient not canonical (>= r)");
        if (constant && !scalars_canonical(constant, 1)) return invalid("zk_fr_lincomb_dev: constant not canonical (>= r)");
        fr_lincomb(d_out, d_in, coeffs, k, constant, n, (hipStream_t)stream);
        return ZK_OK;
    });
}
int zk_fr_mul_dev(void *d_out, const void *d_a, const void *d_b, size_t n, void *stream) {
    return guarded([&] {
        if (n && (!d_out || !d_a || !d_b)) return invalid("zk_fr_mul_dev: null pointer");
        fr_mul(d_out, d_a, d_b, n, (hipStream_t)stream);
        return ZK_OK;
    });
}
int zk_plonk_quotient_dev(void *d_out, const void *const *d_in, const uint64_t *zh_inv, unsigned period, const uint64_t alpha[4], const uint64_t beta[4],
                          const uint64_t gamma[4], size_t n, void *stream) {
    return guarded([&] {
        if (!d_in || !zh_inv || !alpha || !beta || !gamma || (n && !d_out)) return invalid("zk_plonk_quotient_dev: null pointer");
        if (period == 0 || period > 8 || (period & (period - 1))) return invalid("zk_plonk_quotient_dev: period must be 1, 2, 4 or 8");
        if (!scalars_canonical(zh_inv, period) || !scalars_canonical(alpha, 1) || !scalars_canonical(beta, 1) || !scalars_canonical(gamma, 1))
            return invalid("zk_plonk_quotient_dev: scalar not canonical (>= r)");
        for (int k = 0; k < 15; k++)
            if (n && !d_in[k]) return invalid("zk_plonk_quotient_dev: null input vector");
        plonk_quotient(d_out, d_in, zh_inv, period, alpha, beta, gamma, n, (hipStream_t)stream);
        return ZK_OK;
    });
}
int zk_plonk_perm_factors_dev(void *d_num, void *d_den, const void *const *d_in, const uint64_t beta[4], const uint64_t gamma[4], size_t n, void *stream) {
    return guarded([&] {
        if (!d_in || !beta || !gamma || (n && (!d_num || !d_den))) return invalid("zk_plonk_perm_factors_dev: null pointer");
        if (!scalars_canonical(beta, 1) || !scalars_canonical(gamma, 1)) return invalid("zk_plonk_perm_factors_dev: scalar not canonical (>= r)");
        for (int k = 0; k < 7; k++)
            if (n && !d_in[k]) return invalid("zk_plonk_perm_factors_dev: null input vector");
        plonk_perm_factors(d_num, d_den, d_in, beta, gamma, n, (hipStream_t)stream);
        return ZK_OK;
    });
}
int zk_frvec_create(zk_frvec **ws) {
    return guarded([&] {
        if (!ws) return invalid("zk_frvec_create: null pointer");
        int rc = require_device();
        if (rc) return rc;
        *ws = new zk_frvec;
        ZK_HIP(hipGetDevice(&(*ws)->device));
        return ZK_OK;
    });
}
int zk_frvec_destroy(zk_frvec *ws) {
    delete ws;
    return ZK_OK;
}
int zk_fr_scale_powers_dev(zk_frvec *ws, void *d_data, size_t n, const uint64_t base[4], void *stream) {
    return guarded([&] {
        if (!ws || !base || (n && !d_data)) return invalid("zk_fr_scale_powers_dev: null pointer");
        if (!scalars_canonical(base, 1)) return invalid("zk_fr_scale_powers_dev: base not canonical (>= r)");
        if (int rc = check_plan_device(ws->device, "zk_fr_scale_powers_dev")) return rc;
        ws->impl.scale_powers(d_data, n, base, (hipStream_t)stream);
        return ZK_OK;
    });
}
int zk_fr_scan_dev(zk_frvec *ws, void *d_data, size_t n, int op, int reverse, void *stream) {
    return guarded([&] {
        if (!ws || (n && !d_data)) return invalid("zk_fr_scan_dev: null pointer");
        if (op != 0 && op != 1) return invalid("zk_fr_scan_dev: op must be 0 (sum) or 1 (product)");
        if (int rc = check_plan_device(ws->device, "zk_fr_scan_dev")) return rc;
        ws->impl.scan(d_data, n, op == 1, reverse != 0, (hipStream_t)stream);
        return ZK_OK;
    });
}
int zk_fr_eval_dev(zk_frvec *ws, const void *const *d_coefs, const size_t *counts, unsigned k, const uint64_t point[4], void *d_out, void *stream) {
    return guarded([&] {
        if (!ws || !point || (k && (!d_coefs || !counts || !d_out))) return invalid("zk_fr_eval_dev: null pointer");
        for (unsigned j = 0; j < k; j++)
            if (counts[j] && !d_coefs[j]) return invalid("zk_fr_eval_dev: null pointer");
        if (!scalars_canonical(point, 1)) return invalid("zk_fr_eval_dev: point not canonical (>= r)");
        if (int rc = check_plan_device(ws->device, "zk_fr_eval_dev")) return rc;
        ws->impl.eval(d_coefs, counts, k, point, d_out, (hipStream_t)stream);
        return ZK_OK;
    });
}
int zk_fr_quotient_dev(void *d_out, const void *d_a, const void *d_b, const void *d_c, const uint64_t zinv[4], size_t n, void *stream) {
    return guarded([&] {
        if (!d_out || !d_a || !d_b || !d_c || !zinv) return invalid("zk_fr_quotient_dev: null pointer");
        if (!scalars_canonical(zinv, 1)) return invalid("zk_fr_quotient_dev: zinv not canonical (>= r)");
        fr_quotient(d_out, d_a, d_b, d_c, zinv, n, (hipStream_t)stream);
        return ZK_OK;
    });
}

}  // extern "C"
