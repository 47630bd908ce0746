// frvec.h -- internal interface of the F_r vector primitives (frvec.hip).
#pragma once
#include <stdexcept>
#include "common.h"
#include "field.h"

namespace zk {

constexpr unsigned FR_LINCOMB_MAX = 8;

// out[i] = constant + sum_{j<k} coeffs[j] * in[j][i]   (coeffs: k*4 HOST limbs; constant: HOST pointer or null)
void fr_lincomb(void *d_out, const void *const *d_in, const uint64_t *coeffs, unsigned k, const uint64_t constant[4], size_t n, hipStream_t st);
// out[i] = a[i] * b[i]
void fr_mul(void *d_out, const void *d_a, const void *d_b, size_t n, hipStream_t st);

// Fused PLONK quotient on the evaluation coset; d_in: 15 device vectors a b c z zw | ql qr qo qm qc | s1 s2 s3 | x l1;
// zh_inv: `period` HOST elements (1 / Z_H on the coset repeats with that period).
void plonk_quotient(void *d_out, const void *const *d_in, const uint64_t *zh_inv, unsigned period, const uint64_t alpha[4], const uint64_t beta[4],
                    const uint64_t gamma[4], size_t n, hipStream_t st);

// The per-row factors of PLONK's grand product, fused; d_in: 7 device vectors a b c | s1 s2 s3 | x (x[i] = omega^i).
void plonk_perm_factors(void *d_num, void *d_den, const void *const *d_in, const uint64_t beta[4], const uint64_t gamma[4], size_t n, hipStream_t st);

// Scratch owned by whoever issues the calls (one per thread of use): power tables and the scan's per-level chunk totals.
// One scratch object serves one stream at a time: its tables and scan levels are rewritten by every call and only stream
// order keeps an earlier call's kernels ahead of the next call's writes.
struct FrVecScratch {
    DevBuf tables;
    DevBuf levels[4];  // n <= 2^28: 2^17, 2^6 and 1 chunk totals
    // x[i] *= base^i
    void scale_powers(void *d_data, size_t n, const uint64_t base[4], hipStream_t st);
    // in-place inclusive scan under + (mul == false) or * (mul == true); reverse: from the last element down
    void scan(void *d_data, size_t n, bool mul, bool reverse, hipStream_t st);
    // out[j] = sum_i coefs[j][i] * point^i for k <= 8 polynomials (counts[j] coefficients each); d_out: k canonical elements
    void eval(const void *const *d_coefs, const size_t *counts, unsigned k, const uint64_t point[4], void *d_out, hipStream_t st);
};

}  // namespace zk
