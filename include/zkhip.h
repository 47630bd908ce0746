/*
 * zkhip.h -- C ABI of libzkhip.so: MI355X (gfx950) MSM / NTT prover backend for BN254.
 *
 * The reference (tokamak-network/interactive-zkp-study) is pure Python and has NO FFI or
 * plugin interface for this path.  Its de-facto seam is a set
 * of Python call sites; each entry point below names the reference interface it replaces.
 * INTEGRATION.md shows the ctypes binding a reference maintainer would add.
 *
 * Conventions
 *   - Field elements: 4 x uint64 little-endian limbs, canonical residues (< modulus).
 *     This is int(FR(x)) / int(FQ(x)) of the reference, split into limbs.
 *   - G1 affine point: 8 limbs  x || y                      (py_ecc tuple (FQ, FQ)).
 *   - G2 affine point: 16 limbs x.c0 || x.c1 || y.c0 || y.c1 (py_ecc (FQ2, FQ2), coeffs[0], coeffs[1]).
 *   - Point at infinity (Python None): all coordinate limbs zero ((0,0) is on neither curve).
 *     Outputs additionally report it through *out_is_inf.
 *   - Scalars must be canonical (< r).  The reference reduces mod r before multiplying
 *     (zkp/plonk/field.py:86-88); the Python layer does the same before calling in.  Host-buffer
 *     entry points check it (ZK_ERR_INVALID).  "_dev" entry points cannot read device memory on the host:
 *     an MSM over device scalars is exact for every canonical scalar (and any value below 2^254 - 2^240) and
 *     is never silently wrong above: a scalar whose signed digits do not fit the windows (from 2^255; from
 *     about 2^254 for plans of at most 2^17 points) is detected on the device, and the call that collects that submission fails with
 *     ZK_ERR_INVALID and no result; F_r vector / NTT kernels require canonical
 *     device elements as a precondition.  Host-side scalar arguments (coset_shift, zinv, coefficients)
 *     are always checked.
 *   - A plan (zk_msm_plan, zk_ntt_plan) lives on the device that was current when it was created
 *     (zk_set_device); calls made while another device is current fail with ZK_ERR_INVALID.
 *   - Ownership: the caller allocates and frees every buffer it passes.  The library owns only
 *     the handles it returns (zk_*_create / zk_*_destroy).
 *   - Errors: 0 = ZK_OK, negative = failure; zk_last_error() returns a thread-local message.
 *   - There is no CPU fallback: without a usable HIP device every compute entry point fails
 *     with ZK_ERR_NO_DEVICE.
 *   - "_dev" entry points take DEVICE pointers (hipMalloc'd, or torch tensor .data_ptr()) and a
 *     hipStream_t passed as void* (NULL = default stream); they enqueue work and, where noted,
 *     synchronise that stream only to read back the O(1)-size result.
 */
#ifndef ZKHIP_H
#define ZKHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ZK_OK 0
#define ZK_ERR_INVALID (-1)   /* bad argument (null pointer, size, non-canonical input)        */
#define ZK_ERR_HIP (-2)       /* a HIP runtime call failed; see zk_last_error()                  */
#define ZK_ERR_NO_DEVICE (-3) /* no HIP device visible                                           */
#define ZK_ERR_NOMEM (-4)     /* host or device allocation failed                                */

#define ZK_GROUP_G1 1
#define ZK_GROUP_G2 2

const char *zk_last_error(void);
int zk_version(void);                 /* ABI version, currently 1 */
int zk_device_count(int *count);      /* number of HIP devices (0 is not an error) */
int zk_set_device(int device);        /* hipSetDevice for the calling thread */

/* ------------------------------------------------------------------------------------------
 * MSM  sum_i scalars[i] * points[i]
 * Replaces the reference's scalar-mul-and-add loops:
 *   zkp/plonk/kzg.py:59-65 (commit), zkp/groth16/proving.py:27-31,39-43,56-60,66-73 (proof_a/b/c).
 * Terms with a zero scalar or an infinity point contribute nothing (kzg.py:62-63).
 * n == 0 gives infinity (commit of the zero polynomial is None: tests/plonk/test_crypto.py:132-136).
 */
int zk_msm_g1(const uint64_t *scalars /* n*4 */, const uint64_t *points /* n*8 */, size_t n,
              uint64_t out_xy[8], int *out_is_inf);
int zk_msm_g2(const uint64_t *scalars /* n*4 */, const uint64_t *points /* n*16 */, size_t n,
              uint64_t out_xy[16], int *out_is_inf);

/* Device-resident MSM.  A plan owns the workspace for MSMs of up to max_n points. */
typedef struct zk_msm_plan zk_msm_plan;
int zk_msm_plan_create(int group /* ZK_GROUP_G1|G2 */, size_t max_n, zk_msm_plan **plan);
/* The same with the chunk size for MSMs of more than 2^chunk_log2 points chosen per plan (12..24; 0 = the default 22 that
 * zk_msm_plan_create uses).  No reference counterpart: smaller chunks exercise the chunked path at sizes an oracle can check. */
int zk_msm_plan_create_ex(int group, size_t max_n, int chunk_log2, zk_msm_plan **plan);
int zk_msm_plan_destroy(zk_msm_plan *plan);
/* Window width the plan will use for n points (Pippenger c); informational. */
int zk_msm_plan_window_bits(const zk_msm_plan *plan, size_t n);
/* Measurement hooks: with profiling enabled each run records HIP events on the pipeline's own
 * stream; zk_msm_plan_stage_ms returns, for the submission collected last, the spans in ms of
 * {prepare, bucket sort, the bucket-accumulation kernel alone, bucket reduction}.  With several
 * submissions outstanding the spans other than the accumulation kernel's include time spent waiting
 * for the GPU. */
int zk_msm_plan_profile(zk_msm_plan *plan, int enable);
int zk_msm_plan_stage_ms(const zk_msm_plan *plan, float out_ms[4]);
/* Runs the whole MSM; returns after the (tiny) window sums have been read back and folded, i.e. the
 * result is final.  out_xy: 8 (G1) or 16 (G2) limbs on the HOST.  Stream semantics (all device calls
 * of a plan): the kernels run on a stream owned by the plan; they start after everything queued on
 * `stream` so far, and `stream` continues once the first kernel has consumed d_scalars / d_points,
 * so the caller may overwrite its inputs in stream order as usual. */
int zk_msm_dev(zk_msm_plan *plan, const void *d_scalars, const void *d_points, size_t n,
               uint64_t *out_xy, int *out_is_inf, void *stream);
/* Same, but returns the result as a projective partial sum (HOST, 16 limbs per coordinate set:
 * G1 = 4*4 limbs X,Y,ZZ,ZZZ Montgomery form; G2 = 4*8 limbs) for multi-GPU folding. */
int zk_msm_dev_partial(zk_msm_plan *plan, const void *d_scalars, const void *d_points, size_t n,
                       uint64_t *out_xyzz, void *stream);
/* Pipelined form of the two calls above: zk_msm_submit enqueues the whole GPU pipeline (and the
 * 36 KiB read-back) and returns at once with a ticket; zk_msm_collect / _collect_partial wait for
 * that submission and do the host fold.  Up to zk_msm_plan_max_in_flight(plan) (3) submissions may be
 * outstanding; each runs in its own workspace and stream, so consecutive MSMs overlap on the GPU and
 * the host fold of MSM k hides behind MSM k+1 (a prover issues 5-17 MSMs back to back).  Tickets are
 * collected in any order.  An MSM of more than 2^22 points runs as consecutive 2^22-point chunks in
 * the lanes that are free at that moment (partial sums added on the host; at least one lane must be free): zk_msm_submit
 * may block for its early chunks, submissions made before it stay collectable by their tickets, and no further
 * submission is accepted until it has been collected. */
/* Bound-bases mode.  The bases of a prover's queries never change (CRS / SRS): zk_msm_plan_bind_points expands n device
 * points once into the plan's table T[w][i] = 2^(20 w) * P_i (13 * n packed points: 832 B per G1 point, 1664 B per G2
 * point; the plan must have been created for more than 2^17 points, and 13 n < 2^31).  Afterwards every MSM call of the plan
 * that passes d_points == NULL uses the first n bound bases: one window of 2^19 buckets over the 13 n table entries the
 * signed 20-bit digits select -- 13 n bucket additions instead of 16 n and a single window to reduce.  n == 0 unbinds.
 * Results are identical to the unbound calls.  (G1, here and in the unbound calls: a canonical scalar k >= 2^240 is run as k + m r with a small m taken
 * from its index, which evens out the top row's bucket load; every point of the curve y^2 = x^3 + 3 has order r, so the sum
 * is the same -- for bases that are not on the curve the two modes may differ, and neither matches anything.) */
int zk_msm_plan_bind_points(zk_msm_plan *plan, const void *d_points, size_t n, void *stream);
/* Pipelined MSM over the bound bases [first, first + n): several queries (sigma1_2 | sigma1_4 | sigma1_5 ...) can be bound
 * as one concatenated array and addressed by range; collect with zk_msm_collect / zk_msm_collect_partial.  The bound array
 * may be longer than the plan's max_n (ranges of more than 2^22 bases run as chunks). */
int zk_msm_submit_bound(zk_msm_plan *plan, const void *d_scalars, size_t first, size_t n, void *stream, int *out_ticket);
int zk_msm_plan_max_in_flight(const zk_msm_plan *plan);
int zk_msm_submit(zk_msm_plan *plan, const void *d_scalars, const void *d_points, size_t n, void *stream, int *out_ticket);
int zk_msm_collect(zk_msm_plan *plan, int ticket, uint64_t *out_xy, int *out_is_inf);
int zk_msm_collect_partial(zk_msm_plan *plan, int ticket, uint64_t *out_xyzz);
/* Folds `count` partial sums (as written by zk_msm_dev_partial, e.g. all-gathered over RCCL
 * from the ranks that each hold a chunk of the points) in rank order into one affine point.
 * Host-side, O(count) group additions; this is the "all-reduce of partial sums" epilogue
 * (RCCL has no elliptic-curve reduction operator). */
int zk_msm_fold_partials(int group, const uint64_t *partials, size_t count, uint64_t *out_xy, int *out_is_inf);
/* Limbs per partial for a group (16 for G1, 32 for G2). */
int zk_msm_partial_limbs(int group);

/* ------------------------------------------------------------------------------------------
 * NTT over F_r, natural order in and out, omega_n = 5^((r-1)/n)   (zkp/plonk/field.py:178-180)
 * Replaces fft / ifft (zkp/plonk/polynomial.py:292-341, 344-378) and, with a coset shift k,
 * coset_fft / coset_ifft (zkp/plonk/utils.py:145-205; the reference's default k is 5):
 *   inverse == 0: data[i] <- sum_j (k^j data[j]) * omega^(i j)
 *   inverse != 0: data[j] <- k^-j * n^-1 * sum_i data[i] * omega^(-i j)
 * coset_shift == NULL means k = 1 (plain fft / ifft).  1 <= 2^log_n <= 2^28 (field.py:169-172).
 */
int zk_ntt_fr(uint64_t *data /* n*4, in place, HOST */, unsigned log_n, int inverse,
              const uint64_t coset_shift[4] /* nullable */);

/* The host-buffer entry points (zk_msm_g1 / zk_msm_g2 / zk_ntt_fr: what the reference-signature facade calls once per commit /
 * proof element / fft) keep their plans -- NTT tables per log_n, MSM workspaces per group and power-of-two size class, plus the
 * staging buffers -- per calling thread and device, a handful of each (least recently used first out), so that only the first
 * call of a size pays for table building and allocation (MSMs of more than 2^20 points and transforms of more than 2^22
 * elements are not kept: their plan lives for the call; what a thread retains is bounded by three MSM size classes per group of at
 * most 2^20 points and six NTT plans of at most 2^22 elements -- a few GB of HBM in the worst case, returned by zk_cache_clear or at
 * thread exit).  zk_cache_clear drops the calling thread's cache (device memory is
 * returned); zk_cache_stats reports {NTT plans built, NTT cache hits, MSM plans built, MSM cache hits} of the calling thread --
 * a call beyond the cached sizes counts as one more plan BUILT every time, so a caller that sees the builds grow with its calls is
 * on the uncached path and should hold a plan of its own (zk_msm_plan_create / zk_ntt_plan_create). */
int zk_cache_clear(void);
int zk_cache_stats(uint64_t out[4]);

typedef struct zk_ntt_plan zk_ntt_plan;
int zk_ntt_plan_create(unsigned log_n, zk_ntt_plan **plan);
int zk_ntt_plan_destroy(zk_ntt_plan *plan);
/* In-place transform of a DEVICE buffer of n*4 limbs.  Enqueues on `stream` and returns without
 * synchronising.  coset_shift is a HOST pointer (nullable). */
int zk_ntt_dev(zk_ntt_plan *plan, void *d_data, int inverse, const uint64_t coset_shift[4], void *stream);
/* The same transform from d_in to d_out (DEVICE buffers; d_out == d_in allowed) of an input that is zero from element in_len on: only
 * the first min(in_len, n) elements of d_in are read.  What Polynomial.from_evaluations / evaluate-on-a-larger-domain do with a
 * coefficient list shorter than the domain (zkp/plonk/polynomial.py:263-285, 292-341; the coset form zkp/plonk/utils.py:145-177)
 * without the zero fill and the copy into a domain-sized buffer. */
int zk_ntt_dev_padded(zk_ntt_plan *plan, const void *d_in, void *d_out, size_t in_len, int inverse, const uint64_t coset_shift[4], void *stream);
/* `jobs` (<= 4) independent transforms of the plan's size in ONE launch per pass: job b reads d_in[b] (its first min(in_len, n)
 * elements; the rest counts as zero) and writes d_out[b]; d_out[b] == d_in[b] is allowed, any other overlap between the buffers is
 * ZK_ERR_INVALID.  d_in / d_out are HOST arrays of DEVICE pointers.  Direction, in_len and coset shift are shared.  What a prover does
 * with the three polynomials of a round -- zkp/groth16/poly_utils.py:116-125 (A, B, C to coefficients, then onto the coset),
 * zkp/plonk/prover/round1.py (a, b, c) -- one fft()/ifft() call each (zkp/plonk/polynomial.py:316-378): transforms that share
 * their launches share the chip, one job's load and store phases run under the others' butterflies (three 2^20-point transforms:
 * 0.31 ms against 0.36 one after the other). */
int zk_ntt_dev_multi(zk_ntt_plan *plan, unsigned jobs, const void *const *d_in, void *const *d_out, size_t in_len, int inverse,
                     const uint64_t coset_shift[4], void *stream);
/* `batch` independent transforms of the plan's size stored back to back in d_data (no coset shift). */
int zk_ntt_dev_batch(zk_ntt_plan *plan, void *d_data, unsigned batch, int inverse, void *stream);
/* Batched transform between two DEVICE buffers whose layouts are those of the four-step (multi-GPU) transform, so that the
 * per-destination pack, the transpose after the exchange and the twiddle between the two dimensions happen inside the first
 * pass's loads and the last pass's stores instead of in passes of their own.  Element i of
 * transform b (of `batch`) sits at
 *   ZK_NTT_PLAIN       b * n + i                                          transforms back to back;
 *   ZK_NTT_BLOCKED_TW  (((i >> log_block) * batch + b) << log_block) | (i mod 2^log_block), and the value read / written is
 *                      multiplied by w_N^(+-(row0 + b) * i), N = size of `big` (tw_inverse != 0: the inverse root): the buffer an
 *                      all-to-all sends / has received, block s of it belonging to rank s;
 *   ZK_NTT_TRANSPOSED  i * batch + b                                      the matrix transpose (element-major).
 * At most one side may be ZK_NTT_BLOCKED_TW; d_out may equal d_in only when both are ZK_NTT_PLAIN.  Shape it stays exact
 * with: zkp/plonk/polynomial.py:316-378. */
#define ZK_NTT_PLAIN 0
#define ZK_NTT_BLOCKED_TW 1
#define ZK_NTT_TRANSPOSED 2
int zk_ntt_dev_io(zk_ntt_plan *plan, const void *d_in, void *d_out, unsigned batch, int inverse, int in_layout, int out_layout,
                  unsigned log_block, uint64_t row0, const zk_ntt_plan *big /* nullable unless a side is BLOCKED_TW */, int tw_inverse,
                  void *stream);
/* Twiddle between the two dimensions of a four-step transform of the PLAN's size n = 2^log_n whose second
 * dimension has 2^log_cols points:  d_data[b * 2^log_cols + k] *= omega_n^(+-(row0 + b) * k)  for b < rows.
 * Together with zk_ntt_dev_batch this is the local work of the multi-GPU single large NTT
 * (interactive-zkp-study_amd/zkhip/distributed.py: DistNtt). */
int zk_ntt_twiddle_dev(zk_ntt_plan *plan, void *d_data, unsigned log_cols, uint64_t rows, uint64_t row0, int inverse, void *stream);

/* ------------------------------------------------------------------------------------------
 * F_r vector helpers on DEVICE buffers (the pointwise part of the at-scale quotient,
 * replacing the O(n^2) zkp/groth16/poly_utils.py:17-45,116-125 path):
 *   out[i] = (a[i]*b[i] - c[i]) * zinv        (zinv: HOST pointer to one element)
 */
int zk_fr_quotient_dev(void *d_out, const void *d_a, const void *d_b, const void *d_c,
                       const uint64_t zinv[4], size_t n, void *stream);
/*
 *   y = M x over F_r for a CSR matrix (u32 row_ptr[rows+1], u32 col[nnz], vals[nnz*4 limbs]): the scalar
 *   collapse A.w, B.w, C.w of an R1CS with its witness, which zkp/groth16/proving.py:27-31 carries out in
 *   the group (W*G scalar multiplications).
 */
int zk_fr_spmv_dev(const void *d_row_ptr, const void *d_col, const void *d_vals, const void *d_x, void *d_y,
                   size_t rows, void *stream);

/* ------------------------------------------------------------------------------------------
 * F_r vector primitives on DEVICE buffers of canonical elements -- what the coefficient / evaluation algebra of
 * a prover is made of once its vectors live in HBM.  They replace, element-wise and without host round trips,
 * Polynomial.__add__ / __sub__ / scale / evaluate (zkp/plonk/polynomial.py:85-162,189-198), poly_div by a linear
 * factor (polynomial.py:385-436) and the grand product compute_accumulator (zkp/plonk/permutation.py:89-140):
 *   zk_fr_lincomb_dev       out[i] = constant + sum_{j<k} coeffs[j] * in[j][i]     (k <= 8; coeffs, constant: HOST)
 *   zk_fr_mul_dev           out[i] = a[i] * b[i]
 *   zk_fr_scale_powers_dev  data[i] *= base^i            (evaluate(z) = sum of the scaled vector; coset shifts)
 *   zk_fr_scan_dev          in-place inclusive scan, op 0: running sums, op 1: running products; reverse != 0 scans
 *                           from the last element down (suffix sums / products).  Synthetic division by (x - z):
 *                           q[i] = z^-(i+1) * sum_{j>i} c[j] z^j;  grand product: prefix products of the numerators
 *                           times suffix products of the denominators over their total.
 *   zk_fr_eval_dev          out[j] = sum_i coefs[j][i] * point^i for k <= 8 polynomials at ONE point (Polynomial.evaluate,
 *                           polynomial.py:85-106, called once per opening in zkp/plonk/prover/round4.py:40-79): one pass over the
 *                           coefficients; d_out: k canonical elements on the DEVICE.
 * A zk_frvec holds the scratch of the last three (power tables, per-level chunk totals, block sums); one per thread and stream of use:
 * every call rewrites it with kernels on `stream` (nothing is built on the host, no call synchronises), so only stream
 * order keeps consecutive calls apart.
 */
/* The PLONK round-3 quotient in one pass (zkp/plonk/prover/round3.py:114-147 builds the numerator by polynomial products and
 * divides by Z_H with poly_div): for every point of the evaluation coset
 *   out = (q_L a + q_R b + q_O c + q_M a b + q_C
 *          + alpha [ (a + beta x + gamma)(b + 2 beta x + gamma)(c + 3 beta x + gamma) z
 *                  - (a + beta s1 + gamma)(b + beta s2 + gamma)(c + beta s3 + gamma) zw ]
 *          + alpha^2 (z - 1) L1) * zh_inv[i mod period]
 * d_in: 15 device vectors in the order a b c z zw | q_L q_R q_O q_M q_C | s1 s2 s3 | x L1 (zw = z at omega x);
 * zh_inv: `period` (1, 2, 4 or 8) HOST elements, 1 / Z_H repeats with that period on the coset. */
int zk_plonk_quotient_dev(void *d_out, const void *const *d_in, const uint64_t *zh_inv, unsigned period, const uint64_t alpha[4],
                          const uint64_t beta[4], const uint64_t gamma[4], size_t n, void *stream);
/* The per-row factors of PLONK's grand product (zkp/plonk/permutation.py:89-137 forms them in its accumulator loop), fused:
 *   num[i] = (a + beta x + gamma)(b + 2 beta x + gamma)(c + 3 beta x + gamma),  den[i] = (a + beta s1 + gamma)(b + beta s2 + gamma)(c + beta s3 + gamma)
 * d_in: HOST-side array of the 7 device vectors a b c | s1 s2 s3 | x (x[i] = omega^i, the identity labels); canonical elements in and out. */
int zk_plonk_perm_factors_dev(void *d_num, void *d_den, const void *const *d_in, const uint64_t beta[4], const uint64_t gamma[4], size_t n, void *stream);
typedef struct zk_frvec zk_frvec;
int zk_frvec_create(zk_frvec **ws);
int zk_frvec_destroy(zk_frvec *ws);
int zk_fr_lincomb_dev(void *d_out, const void *const *d_in /* k HOST-side array of device pointers */, const uint64_t *coeffs /* k*4 */,
                      unsigned k, const uint64_t constant[4] /* nullable */, size_t n, void *stream);
int zk_fr_mul_dev(void *d_out, const void *d_a, const void *d_b, size_t n, void *stream);
int zk_fr_scale_powers_dev(zk_frvec *ws, void *d_data, size_t n, const uint64_t base[4], void *stream);
int zk_fr_scan_dev(zk_frvec *ws, void *d_data, size_t n, int op, int reverse, void *stream);
int zk_fr_eval_dev(zk_frvec *ws, const void *const *d_coefs /* k HOST-side array of device pointers */, const size_t *counts /* k */, unsigned k,
                   const uint64_t point[4], void *d_out, void *stream);

/* ------------------------------------------------------------------------------------------
 * Fixed-base batch scalar multiplication out[i] = scalars[i] * base  (HOST buffers).
 * Replaces the setup-side loops zkp/groth16/setup.py:18-23,56-60,65-69 and
 * zkp/plonk/srs.py:77-85 (one bn128.multiply(G, k_i) per element).
 */
int zk_fixed_base_g1(const uint64_t base_xy[8], const uint64_t *scalars, size_t n, uint64_t *out_points /* n*8 */);
int zk_fixed_base_g2(const uint64_t base_xy[16], const uint64_t *scalars, size_t n, uint64_t *out_points /* n*16 */);
/* The same batch with the scalars and the points in DEVICE buffers (base_xy stays a HOST pointer): key generation at scale, where
 * the exponents x^j, (beta A_i + alpha B_i + C_i)(x) / delta, x^k Z(x) / delta (setup.py:18-60) and tau^i (srs.py:68-85) are produced
 * by the F_r vector kernels above and the points go straight into zk_msm_plan_bind_points.  Enqueued on `stream`, which is
 * synchronised once before the call returns (the call owns its window table). */
int zk_fixed_base_g1_dev(const uint64_t base_xy[8], const void *d_scalars, size_t n, void *d_out_points /* n*8 limbs */, void *stream);
int zk_fixed_base_g2_dev(const uint64_t base_xy[16], const void *d_scalars, size_t n, void *d_out_points /* n*16 limbs */, void *stream);

/* ------------------------------------------------------------------------------------------
 * Single group operations on the device (HOST buffers; batch of n independent operations).
 * Replace the seam zkp/plonk/field.py:72-115 (ec_mul / ec_add / ec_neg) where the Python layer
 * needs individual points (proof_c's prf_A*s, verifier-side glue).  op: 0 = P+Q, 1 = k*P.
 */
int zk_group_op(int group, int op, const uint64_t *p, const uint64_t *q_or_scalar, size_t n, uint64_t *out);

/* ------------------------------------------------------------------------------------------
 * Pairings for the verifiers -- HOST code (a proof needs 2-4 pairings; never a GPU target).
 * Replace py_ecc.bn128.pairing(Q, P) as used by zkp/groth16/verifying.py:17-40,
 * zkp/plonk/field.py:118-138 and zkp/plonk/kzg.py:117-160.  Infinity inputs give the identity.
 * Like py_ecc's pairing (which asserts is_on_curve for both arguments) every input must be a canonical
 * point of its curve -- G1: y^2 = x^3 + 3, G2: the twist y^2 = x^3 + 3/(9+i) -- or the all-zero infinity
 * encoding; anything else fails with ZK_ERR_INVALID before any arithmetic (the reference raises
 * AssertionError).  As in the reference, subgroup membership of a twist point is not tested.
 *   zk_pairing        e(P, Q) as 12 canonical F_p coefficients (4 limbs each) of
 *                     F_p[w]/(w^12 - 18 w^6 + 82), the reference's FQ12 coefficient order.
 *   zk_pairing_check  *out_is_one = [ prod_i e(P_i, Q_i) == 1 ]  (one shared final exponentiation).
 */
int zk_pairing(const uint64_t g1_xy[8], const uint64_t g2_xy[16], uint64_t out_fq12[48]);
int zk_pairing_check(const uint64_t *g1_points /* n*8 */, const uint64_t *g2_points /* n*16 */, size_t n, int *out_is_one);

/* ------------------------------------------------------------------------------------------
 * Measurement aid (no reference counterpart): chip-wide rate of the library's own arithmetic,
 * the integer-ALU ceiling bench.py prices the MSM against.
 *   what = 0: Montgomery multiplications in F_p per second;  1: G1 mixed (XYZZ += affine) additions per second;
 *   what = 2: bare v_mad_u64_u32 lane-operations per second (independent chains, nothing else in the loop) -- the
 *             hardware multiply-add issue rate, a ceiling that does not depend on this library's field arithmetic.
 */
int zk_measure_rate(int what, double *out_per_sec);

#ifdef __cplusplus
}
#endif
#endif /* ZKHIP_H */
