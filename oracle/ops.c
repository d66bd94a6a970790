/*
 * oracle/ops.c — CpuHal op semantics on host arrays + NTT + poly helpers.
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see zkoracle.h).
 *
 * Follows risc0-zkp 3.0.2 (un-vendored; /root/reference/Cargo.lock:5393):
 *   src/core/ntt.rs  — interpolate_ntt (DIF, natural in -> bit-reversed out, * n^-1),
 *                      evaluate_ntt (DIT, bit-reversed in -> natural out, first `expand_bits` layers skipped),
 *                      expand, bit_reverse
 *   src/hal/cpu.rs   — one function per Hal method below (same names)
 *   src/core/poly.rs — poly_interpolate, poly_eval, poly_divide
 * as summarised in SURVEY.md Appendix A.3/A.4.
 */
#include <stdlib.h>
#include <string.h>
#include "field.h"
#include "zkoracle.h"

uint32_t zko_fp_mul(uint32_t a, uint32_t b) { return fp_mul(a, b); }
uint32_t zko_fp_encode(uint32_t x) { return fp_from_u32(x); }
uint32_t zko_fp_decode(uint32_t a) { return fp_to_u32(a); }
uint32_t zko_fp_inv(uint32_t a) { return fp_inv(a); }
uint32_t zko_rou_fwd(unsigned k) { return fp_rou_fwd(k); }
uint32_t zko_rou_rev(unsigned k) { return fp_rou_rev(k); }
void zko_fp4_mul(const uint32_t a[4], const uint32_t b[4], uint32_t out[4]) {
    fp4 x, y; memcpy(&x, a, 16); memcpy(&y, b, 16);
    fp4 r = fp4_mul(x, y); memcpy(out, &r, 16);
}
void zko_fp4_inv(const uint32_t a[4], uint32_t out[4]) {
    fp4 x; memcpy(&x, a, 16);
    fp4 r = fp4_inv(x); memcpy(out, &r, 16);
}
void zko_free(void* p) { free(p); }

/* ---- NTT (ntt.rs) ---- */
static void interpolate_ntt(fp* io, size_t n) {
    unsigned bits = log2_ceil(n);
    for (unsigned N = bits; N >= 1; N--) {
        size_t len = (size_t)1 << N, half = len >> 1;
        fp step = fp_rou_rev(N);
        for (size_t s = 0; s < n; s += len) {
            fp cur = fp_from_u32(1);
            for (size_t i = 0; i < half; i++) {
                fp a = io[s + i], b = io[s + i + half];
                io[s + i] = fp_add(a, b);
                io[s + i + half] = fp_mul(fp_sub(a, b), cur);
                cur = fp_mul(cur, step);
            }
        }
    }
    fp norm = fp_inv(fp_from_u32((uint32_t)n));
    for (size_t i = 0; i < n; i++) io[i] = fp_mul(io[i], norm);
}
static void evaluate_ntt(fp* io, size_t n, unsigned expand_bits) {
    unsigned bits = log2_ceil(n);
    for (unsigned N = expand_bits + 1; N <= bits; N++) {
        size_t len = (size_t)1 << N, half = len >> 1;
        fp step = fp_rou_fwd(N);
        for (size_t s = 0; s < n; s += len) {
            fp cur = fp_from_u32(1);
            for (size_t i = 0; i < half; i++) {
                fp a = io[s + i], b = fp_mul(io[s + i + half], cur);
                io[s + i] = fp_add(a, b);
                io[s + i + half] = fp_sub(a, b);
                cur = fp_mul(cur, step);
            }
        }
    }
}

void zko_batch_interpolate_ntt(uint32_t* io, size_t size, size_t count) {
    size_t n = size / count;
#pragma omp parallel for schedule(dynamic)
    for (size_t c = 0; c < count; c++) interpolate_ntt(io + c * n, n);
}
void zko_batch_expand_into_evaluate_ntt(uint32_t* out, size_t out_size, const uint32_t* in, size_t in_size,
                                        size_t count, size_t expand_bits) {
    size_t n_out = out_size / count, n_in = in_size / count;
    (void)n_in;
#pragma omp parallel for schedule(dynamic)
    for (size_t c = 0; c < count; c++) {
        fp* o = out + c * n_out;
        const fp* p = in + c * (in_size / count);
        for (size_t i = 0; i < n_out; i++) o[i] = p[i >> expand_bits];   /* expand */
        evaluate_ntt(o, n_out, (unsigned)expand_bits);
    }
}
void zko_batch_bit_reverse(uint32_t* io, size_t size, size_t count) {
    size_t n = size / count;
    unsigned bits = log2_ceil(n);
    if (bits == 0) return;
#pragma omp parallel for schedule(dynamic)
    for (size_t c = 0; c < count; c++) {
        fp* p = io + c * n;
        for (size_t i = 0; i < n; i++) {
            size_t r = bit_rev32((uint32_t)i) >> (32 - bits);
            if (i < r) { fp t = p[i]; p[i] = p[r]; p[r] = t; }
        }
    }
}
/* cpu.rs zk_shift: io[idx] *= 3^(bitrev(idx mod n)) — coefficients are in bit-reversed order */
void zko_zk_shift(uint32_t* io, size_t size, size_t count) {
    size_t n = size / count;
    unsigned bits = log2_ceil(n);
    fp three = fp_from_u32(3);
    /* table of 3^(2^k) */
    fp pw[32];
    pw[0] = three;
    for (int k = 1; k < 32; k++) pw[k] = fp_mul(pw[k - 1], pw[k - 1]);
#pragma omp parallel for schedule(static)
    for (size_t idx = 0; idx < size; idx++) {
        uint32_t pos = (uint32_t)(idx & (n - 1));
        uint32_t e = bits ? bit_rev32(pos) >> (32 - bits) : 0;
        fp m = fp_from_u32(1);
        for (unsigned k = 0; k < bits; k++) if ((e >> k) & 1) m = fp_mul(m, pw[k]);
        io[idx] = fp_mul(io[idx], m);
    }
}

/* ---- hashing ops ---- */
void zko_hash_rows(uint32_t* out, size_t rows, const uint32_t* matrix, size_t matrix_size) {
    size_t cols = matrix_size / rows;
#pragma omp parallel for schedule(static)
    for (size_t r = 0; r < rows; r++) zko_hash_elem_slice(matrix + r, cols, rows, out + 8 * r);
}
void zko_hash_fold(uint32_t* io, size_t input_size, size_t output_size) {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < output_size; i++)
        zko_hash_pair(io + 8 * (input_size + 2 * i), io + 8 * (input_size + 2 * i + 1), io + 8 * (output_size + i));
}

/* ---- polynomial ops ---- */
static inline fp4 ld4(const uint32_t* p) { fp4 r; memcpy(&r, p, 16); return r; }
static inline void st4(uint32_t* p, fp4 v) { memcpy(p, &v, 16); }

void zko_batch_evaluate_any(const uint32_t* coeffs, size_t coeffs_size, size_t poly_count, const uint32_t* which,
                            const uint32_t* xs, size_t eval_count, uint32_t* out) {
    size_t po = coeffs_size / poly_count;
#pragma omp parallel for schedule(dynamic)
    for (size_t k = 0; k < eval_count; k++) {
        const fp* c = coeffs + (size_t)which[k] * po;
        fp4 x = ld4(xs + 4 * k), tot = fp4_zero();
        for (size_t j = po; j-- > 0;) tot = fp4_add(fp4_mul(tot, x), fp4_from_fp(c[j]));   /* Horner */
        st4(out + 4 * k, tot);
    }
}
void zko_mix_poly_coeffs(uint32_t* out, const uint32_t mix_start[4], const uint32_t mix[4], const uint32_t* in,
                         const uint32_t* combos, size_t input_size, size_t count) {
    fp4 m = ld4(mix);
#pragma omp parallel for schedule(static)
    for (size_t idx = 0; idx < count; idx++) {
        fp4 cur = ld4(mix_start);
        for (size_t i = 0; i < input_size; i++) {
            uint32_t* o = out + 4 * ((size_t)combos[i] * count + idx);
            st4(o, fp4_add(ld4(o), fp4_mul_fp(cur, in[i * count + idx])));
            cur = fp4_mul(cur, m);
        }
    }
}
void zko_eltwise_add_elem(uint32_t* out, const uint32_t* a, const uint32_t* b, size_t n) {
    for (size_t i = 0; i < n; i++) out[i] = fp_add(a[i], b[i]);
}
void zko_eltwise_sum_extelem(uint32_t* out, size_t out_size, const uint32_t* in, size_t in_elems) {
    size_t count = out_size / 4, k = in_elems / count;
#pragma omp parallel for schedule(static)
    for (size_t idx = 0; idx < count; idx++) {
        fp4 s = fp4_zero();
        for (size_t j = 0; j < k; j++) s = fp4_add(s, ld4(in + 4 * (j * count + idx)));
        for (int i = 0; i < 4; i++) out[i * count + idx] = s.c[i];
    }
}
/* cpu.rs fri_fold: ext elems stored as 4 planes; 16 -> 1 with bit-reversed slice order */
void zko_fri_fold(uint32_t* out, size_t out_size, const uint32_t* in, const uint32_t mix[4]) {
    size_t count = out_size / 4;
    fp4 m = ld4(mix);
#pragma omp parallel for schedule(static)
    for (size_t idx = 0; idx < count; idx++) {
        fp4 tot = fp4_zero(), cur = fp4_one();
        for (unsigned i = 0; i < ZKO_FRI_FOLD; i++) {
            unsigned r = bit_rev32(i) >> (32 - ZKO_FRI_FOLD_PO2);
            fp4 f;
            for (int p = 0; p < 4; p++) f.c[p] = in[(size_t)p * count * ZKO_FRI_FOLD + r * count + idx];
            tot = fp4_add(tot, fp4_mul(cur, f));
            cur = fp4_mul(cur, m);
        }
        for (int p = 0; p < 4; p++) out[p * count + idx] = tot.c[p];
    }
}
void zko_gather_sample(uint32_t* dst, const uint32_t* src, size_t idx, size_t size, size_t stride) {
    for (size_t g = 0; g < size; g++) dst[g] = src[g * stride + idx];
}
/* cpu.rs scatter: for each cycle i: into[index[i]..] pattern — values[offsets[j]] style preload.
 * into[index[j]] = values[j] for offsets[i] <= j < offsets[i+1] is the upstream shape; here the flat form. */
void zko_scatter(uint32_t* into, const uint32_t* index, const uint32_t* offsets, const uint32_t* values, size_t n_idx) {
    for (size_t i = 0; i < n_idx; i++)
        for (uint32_t j = offsets[i]; j < offsets[i + 1]; j++) into[index[j]] = values[j];
}
void zko_prefix_products(uint32_t* io, size_t n_ext) {
    for (size_t i = 1; i < n_ext; i++) st4(io + 4 * i, fp4_mul(ld4(io + 4 * i), ld4(io + 4 * (i - 1))));
}

/* poly.rs */
void zko_poly_eval(const uint32_t* coeffs, size_t n, const uint32_t x[4], uint32_t out[4]) {
    fp4 xx = ld4(x), tot = fp4_zero();
    for (size_t j = n; j-- > 0;) tot = fp4_add(fp4_mul(tot, xx), ld4(coeffs + 4 * j));
    st4(out, tot);
}
void zko_poly_divide(uint32_t* poly, size_t n, const uint32_t z[4], uint32_t rem[4]) {
    fp4 zz = ld4(z), cur = fp4_zero();
    for (size_t i = n; i-- > 0;) {
        fp4 c = ld4(poly + 4 * i);
        st4(poly + 4 * i, cur);
        cur = fp4_add(fp4_mul(cur, zz), c);
    }
    st4(rem, cur);
}
/* Lagrange interpolation to coefficient form (degree < size); the result is unique. */
void zko_poly_interpolate(uint32_t* out, const uint32_t* xs, const uint32_t* fx, size_t size) {
    if (size == 1) { memcpy(out, fx, 16); return; }
    /* ft(x) = prod (x - x_i), size+1 coeffs */
    fp4* ft = (fp4*)calloc(size + 1, sizeof(fp4));
    fp4* fr = (fp4*)calloc(size + 1, sizeof(fp4));
    ft[0] = fp4_one();
    for (size_t i = 0; i < size; i++) {
        fp4 xi = ld4(xs + 4 * i);
        for (size_t j = i + 1; j >= 1; j--) ft[j] = fp4_sub(ft[j - 1], fp4_mul(ft[j], xi));
        ft[0] = fp4_sub(fp4_zero(), fp4_mul(ft[0], xi));
    }
    for (size_t i = 0; i < size; i++) st4(out + 4 * i, fp4_zero());
    for (size_t i = 0; i < size; i++) {
        fp4 xi = ld4(xs + 4 * i);
        /* fr = ft / (x - x_i) */
        memcpy(fr, ft, (size + 1) * sizeof(fp4));
        fp4 rem;
        zko_poly_divide((uint32_t*)fr, size + 1, (const uint32_t*)&xi, (uint32_t*)&rem);
        fp4 d; zko_poly_eval((const uint32_t*)fr, size, (const uint32_t*)&xi, (uint32_t*)&d);
        fp4 mul = fp4_mul(ld4(fx + 4 * i), fp4_inv(d));
        for (size_t j = 0; j < size; j++) st4(out + 4 * j, fp4_add(ld4(out + 4 * j), fp4_mul(mul, fr[j])));
    }
    free(ft); free(fr);
}

#ifdef _OPENMP
#include <omp.h>
int zko_num_threads(void) { return omp_get_max_threads(); }
void zko_set_num_threads(int n) { if (n > 0) omp_set_num_threads(n); }
#else
int zko_num_threads(void) { return 1; }
void zko_set_num_threads(int n) { (void)n; }
#endif
