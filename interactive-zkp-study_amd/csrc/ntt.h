// ntt.h -- internal interface of the F_r NTT pipeline (ntt.hip).
#pragma once
#include <stdexcept>
#include <vector>
#include "common.h"
#include "field.h"
#include "host_field.h"

namespace zk {

struct NttPassParams {
    uint32_t L;          // log2 n
    uint32_t lp;         // log2 of this pass's digit
    uint32_t sp;         // log2 stride of the digit (sum of later digits' logs)
    uint32_t g;          // log2 of adjacent elements/rows carried per tile (coalescing)
    uint32_t l1;         // final pass: log2 of the first digit (0 when D == 1)
    uint32_t nmid;       // final pass: number of middle digits (0..2)
    uint32_t lmid[2];    // final pass: logs of the middle digits, q = 2 first
    uint32_t tw_shift;   // in-tile twiddle table stride shift (lmax - lp)
    uint32_t lh;         // low-part bits of the two-level inter-pass twiddle tables
    uint32_t apply_scale;  // final pass: multiply outputs by `scale` (D == 1 inverse)
    uint32_t direct_tw;    // inter-pass twiddles come ready-made from a per-(k, rem) table instead of the two-level one
    uint32_t gb;           // low bits of the carried dimension that run over the BATCH index (transposed input / output); g - gb run over elements
};

// How element i of transform b of a batch sits in the caller's buffers: the layouts of the four-step (multi-GPU) transform, read by
// the first pass's loads and written by the last pass's stores, so that no separate pack / transpose / twiddle pass is needed.
enum NttLayout : int {
    NTT_PLAIN = 0,       // b * n + i                                                        (transforms back to back)
    NTT_BLOCKED_TW = 1,  // (((i >> kbits) * batch + b) << kbits) | (i & (2^kbits - 1)), and the value is multiplied by
                         // w_N^(+-(row0 + b) * i) of the LARGE transform (N = tw plan's size): per-destination blocks + four-step twiddle
    NTT_TRANSPOSED = 2,  // i * batch + b                                                   (element-major: the matrix transpose)
};
constexpr unsigned NTT_MULTI_MAX = 4;   // transforms per run_multi call
struct NttIoArgs {
    const Fr *twA = nullptr, *twB = nullptr;  // two-level tables of w_N (Montgomery form), direction chosen by the host
    uint32_t lh = 0;                          // low bits of that table
    uint32_t kbits = 0;                       // log2 block length of NTT_BLOCKED_TW
    uint32_t batch = 1;                       // number of transforms in the batch (layout multiplier)
    uint64_t row0 = 0;                        // first row index of the batch inside the large transform
    // coset transforms (plain layout, one transform): x[i] *= k^i at the first pass's load (forward) or k^-i at the last pass's store
    // (inverse), from the plan's two-level power table -- the scaling passes coset_fft / coset_ifft (zkp/plonk/utils.py:145-205) run
    // before / after the transform cost a launch and a read + write of the vector each
    const Fr *cosA = nullptr, *cosB = nullptr;
    uint32_t cos_lh = 0;
    uint32_t cos_in = 0, cos_out = 0;
    uint32_t in_len = 0xffffffffu;            // plain first-pass loads: elements from in_len on are zero and are not read (zero-padded input)
    // Independent transforms of separate buffers in ONE launch per pass (run_multi; plain layouts): transform b reads in_multi[b] and
    // writes out_multi[b]; the scratch between the passes is the plan's, back to back as for a batch.
    uint32_t multi = 0;
    const void *in_multi[NTT_MULTI_MAX] = {nullptr, nullptr, nullptr, nullptr};
    void *out_multi[NTT_MULTI_MAX] = {nullptr, nullptr, nullptr, nullptr};
};

// Natural-order in/out radix-2 NTT over F_r of size 2^log_n; omega = 5^((r-1)/n).
class NttPlan {
  public:
    explicit NttPlan(unsigned log_n);
    // In-place transform of the device buffer (n * 32 bytes, canonical elements).  Enqueues only.
    // `batch` > 1: that many independent transforms stored back to back (no coset shift).
    void run(void *d_data, bool inverse, const uint64_t coset_shift[4], hipStream_t st, unsigned batch = 1);
    // One transform from d_in to d_out (d_out == d_in allowed): only the first in_len elements of d_in are read, the rest of the input
    // counts as zero -- a polynomial of in_len coefficients evaluated on a larger domain needs neither the zero fill nor the copy.
    void run_padded(const void *d_in, void *d_out, size_t in_len, bool inverse, const uint64_t coset_shift[4], hipStream_t st);
    // `jobs` (<= NTT_MULTI_MAX) independent transforms, job b from d_in[b] to d_out[b] (d_out[b] == d_in[b] allowed; no other overlap
    // between the buffers), all with the same in_len / direction / coset shift, in ONE launch per pass: the workgroups of the jobs
    // share the chip, so one job's load and store phases run under the others' butterflies (three 2^20-point transforms: 0.31 ms
    // against 0.36 one after the other, tools/ntt_batch_probe.py).
    void run_multi(const void *const *d_in, void *const *d_out, unsigned jobs, size_t in_len, bool inverse, const uint64_t coset_shift[4], hipStream_t st);
    // Batched transform between two buffers with the layouts above (in_layout read by the first pass, out_layout written by the
    // last).  `big` supplies w_N for NTT_BLOCKED_TW (the plan of the large transform; tw_inverse picks w_N^-1).  d_out may be
    // d_in only when both layouts are NTT_PLAIN.
    void run_io(const void *d_in, void *d_out, bool inverse, unsigned batch, int in_layout, int out_layout, unsigned kbits, uint64_t row0,
                const NttPlan *big, bool tw_inverse, hipStream_t st);
    // data[b * 2^log_cols + k] *= omega_n^(+-(row0 + b) * k), b < rows: the twiddle between the two dimensions of a
    // four-step transform of n = 2^log_n points whose second dimension has 2^log_cols points.
    void twiddle_2d(void *d_data, unsigned log_cols, uint64_t rows, uint64_t row0, bool inverse, hipStream_t st);
    unsigned log_n() const { return L_; }
    int device() const { return device_; }

  private:
    void launch_passes(const void *d_in, void *d_out, bool inverse, unsigned batch, int in_layout, int out_layout, const NttIoArgs &io, hipStream_t st);
    void build_tables();
    void coset_tables(const uint64_t k[4], bool inverse);
    unsigned L_;
    int device_ = 0;  // the device the tables live on; run() refuses any other current device
    std::vector<uint32_t> digits_;  // log2 of each pass's digit, pass 1 first
    uint32_t lmax_ = 0, lh_ = 0;
    DevBuf tmp_;
    // [0] forward, [1] inverse
    DevBuf tile_tw_[2], twA_[2], twB_[2], twB_scaled_inv_;
    DevBuf tw_direct_[2][3];  // ready-made inter-pass twiddles of the small pass boundaries
    Fr scale_inv_;  // n^-1 (Montgomery)
    // coset power tables (two-level) of the last shift k, one pair per direction ([0]: k^i, [1]: k^-i): a prover alternates
    // coset NTT and inverse with one fixed shift, and rebuilding means host work plus a blocking upload
    DevBuf cosA_[2], cosB_[2];
    uint64_t cos_k_[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    bool cos_valid_[2] = {false, false};
};

// y = M x for a CSR matrix over F_r (u32 row_ptr[rows+1], u32 col[nnz], canonical vals[nnz], x, y) on device buffers.
void fr_spmv(const void *d_row_ptr, const void *d_col, const void *d_vals, const void *d_x, void *d_y, size_t rows, hipStream_t st);

// out[i] = (a[i]*b[i] - c[i]) * zinv  on device buffers of canonical F_r elements.
void fr_quotient(void *d_out, const void *d_a, const void *d_b, const void *d_c, const uint64_t zinv[4], size_t n, hipStream_t st);

}  // namespace zk
