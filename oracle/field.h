/*
 * oracle/field.h — BabyBear field and its degree-4 extension, CPU restatement.
 *
 * TEST INFRASTRUCTURE ONLY. Nothing under oracle/ is part of the product path; only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it (as the checker).
 *
 * PARITY UNPINNED: the algorithm lives in the un-vendored crate risc0-core 3.0.0
 * (src/field/baby_bear.rs; pinned at /root/reference/Cargo.lock:5338) which is absent from
 * /root/reference. This file restates its published algorithm:
 *   P = 15*2^27 + 1, Elem = u32 in Montgomery form (a * 2^32 mod P), M = P^-1 mod 2^32,
 *   R2 = 2^64 mod P, ExtElem = Fp[x]/(x^4 + 11), ROU_FWD[k] = 137^(2^(27-k)).
 * The reference call sites that reach it: /root/reference/crates/host/src/lib.rs:137
 * (default_prover().prove) and /root/reference/crates/host/src/bin/cli.rs:103 (receipt.verify).
 */
#ifndef ZKORACLE_FIELD_H
#define ZKORACLE_FIELD_H
#include <stdint.h>
#include <stddef.h>

#define FP_P 2013265921u
#define FP_M 0x88000001u     /* P^-1 mod 2^32 */
#define FP_R2 1172168163u    /* 2^64 mod P */
#define FP_INVALID 0xffffffffu
#define FP_BETA 11u
#define EXT_SIZE 4

typedef uint32_t fp;                     /* Montgomery form, always < P */
typedef struct { fp c[4]; } fp4;

static inline fp fp_add(fp a, fp b) { uint32_t r = a + b; return r >= FP_P ? r - FP_P : r; }
static inline fp fp_sub(fp a, fp b) { uint32_t r = a - b; return a < b ? r + FP_P : r; }
static inline fp fp_neg(fp a) { return a ? FP_P - a : 0; }
/* Montgomery product: a*b*2^-32 mod P (baby_bear.rs `mul`). */
static inline fp fp_mul(fp a, fp b) {
    uint64_t o = (uint64_t)a * b;
    uint32_t low = 0u - (uint32_t)o;
    uint32_t red = FP_M * low;
    o += (uint64_t)red * FP_P;
    uint32_t r = (uint32_t)(o >> 32);
    return r >= FP_P ? r - FP_P : r;
}
static inline fp fp_from_u32(uint32_t x) { return fp_mul(FP_R2, x % FP_P); }   /* encode */
static inline uint32_t fp_to_u32(fp a) { return fp_mul(1u, a); }                /* decode */
static inline fp fp_pow(fp a, uint64_t e) {
    fp r = fp_from_u32(1);
    while (e) { if (e & 1) r = fp_mul(r, a); a = fp_mul(a, a); e >>= 1; }
    return r;
}
static inline fp fp_inv(fp a) { return fp_pow(a, FP_P - 2); }

static inline fp4 fp4_zero(void) { fp4 r = {{0, 0, 0, 0}}; return r; }
static inline fp4 fp4_from_fp(fp a) { fp4 r = {{a, 0, 0, 0}}; return r; }
static inline fp4 fp4_one(void) { return fp4_from_fp(fp_from_u32(1)); }
static inline int fp4_eq(fp4 a, fp4 b) {
    return a.c[0] == b.c[0] && a.c[1] == b.c[1] && a.c[2] == b.c[2] && a.c[3] == b.c[3];
}
static inline fp4 fp4_add(fp4 a, fp4 b) {
    fp4 r; for (int i = 0; i < 4; i++) r.c[i] = fp_add(a.c[i], b.c[i]); return r;
}
static inline fp4 fp4_sub(fp4 a, fp4 b) {
    fp4 r; for (int i = 0; i < 4; i++) r.c[i] = fp_sub(a.c[i], b.c[i]); return r;
}
static inline fp4 fp4_mul_fp(fp4 a, fp b) {
    fp4 r; for (int i = 0; i < 4; i++) r.c[i] = fp_mul(a.c[i], b); return r;
}
/* (a0 + a1 x + a2 x^2 + a3 x^3)(b0 + ...) mod x^4 + 11  (x^4 = -11 =: NBETA). */
static inline fp4 fp4_mul(fp4 a, fp4 b) {
    const fp nbeta = fp_from_u32(FP_P - FP_BETA);
    fp4 r;
    r.c[0] = fp_add(fp_mul(a.c[0], b.c[0]),
                    fp_mul(nbeta, fp_add(fp_add(fp_mul(a.c[1], b.c[3]), fp_mul(a.c[2], b.c[2])),
                                         fp_mul(a.c[3], b.c[1]))));
    r.c[1] = fp_add(fp_add(fp_mul(a.c[0], b.c[1]), fp_mul(a.c[1], b.c[0])),
                    fp_mul(nbeta, fp_add(fp_mul(a.c[2], b.c[3]), fp_mul(a.c[3], b.c[2]))));
    r.c[2] = fp_add(fp_add(fp_add(fp_mul(a.c[0], b.c[2]), fp_mul(a.c[1], b.c[1])),
                           fp_mul(a.c[2], b.c[0])),
                    fp_mul(nbeta, fp_mul(a.c[3], b.c[3])));
    r.c[3] = fp_add(fp_add(fp_mul(a.c[0], b.c[3]), fp_mul(a.c[1], b.c[2])),
                    fp_add(fp_mul(a.c[2], b.c[1]), fp_mul(a.c[3], b.c[0])));
    return r;
}
static inline fp4 fp4_pow(fp4 a, uint64_t e) {
    fp4 r = fp4_one();
    while (e) { if (e & 1) r = fp4_mul(r, a); a = fp4_mul(a, a); e >>= 1; }
    return r;
}
/* Inverse via the norm tower: a(x)*a(-x) = b0 + b2 x^2 ; (b0 + b2 x^2)(b0 - b2 x^2) = b0^2 + 11 b2^2 in Fp. */
static inline fp4 fp4_inv(fp4 a) {
    const fp beta = fp_from_u32(FP_BETA);
    fp a0 = a.c[0], a1 = a.c[1], a2 = a.c[2], a3 = a.c[3];
    /* a(x) a(-x): even part E = a0 + a2 x^2, odd part O = a1 + a3 x^2 (times x). E^2 - x^2 O^2 */
    /* E^2 = a0^2 + 2 a0 a2 x^2 + a2^2 x^4 ; x^2 O^2 = a1^2 x^2 + 2 a1 a3 x^4 + a3^2 x^6, x^4 = -11 */
    fp b0 = fp_add(fp_mul(a0, a0), fp_mul(beta, fp_sub(fp_mul(fp_add(a1, a1), a3), fp_mul(a2, a2))));
    fp b2 = fp_add(fp_sub(fp_mul(fp_add(a0, a0), a2), fp_mul(a1, a1)), fp_mul(beta, fp_mul(a3, a3)));
    fp c = fp_add(fp_mul(b0, b0), fp_mul(beta, fp_mul(b2, b2)));
    fp ic = fp_inv(c);
    b0 = fp_mul(b0, ic); b2 = fp_mul(b2, ic);
    /* inv = a(-x) * (b0 - b2 x^2) */
    fp4 am = {{a0, fp_neg(a1), a2, fp_neg(a3)}};
    fp4 t = {{b0, 0, fp_neg(b2), 0}};
    return fp4_mul(am, t);
}

/* Roots of unity: ROU_FWD[k] has order 2^k; ROU_REV[k] is its inverse (k <= 27). */
static inline fp fp_rou_fwd(unsigned k) { return fp_pow(fp_from_u32(137), 1ull << (27 - k)); }
static inline fp fp_rou_rev(unsigned k) { return fp_inv(fp_rou_fwd(k)); }

static inline uint32_t bit_rev32(uint32_t x) {
    x = ((x & 0xaaaaaaaau) >> 1) | ((x & 0x55555555u) << 1);
    x = ((x & 0xccccccccu) >> 2) | ((x & 0x33333333u) << 2);
    x = ((x & 0xf0f0f0f0u) >> 4) | ((x & 0x0f0f0f0fu) << 4);
    x = ((x & 0xff00ff00u) >> 8) | ((x & 0x00ff00ffu) << 8);
    return (x >> 16) | (x << 16);
}
static inline unsigned log2_ceil(size_t x) { unsigned r = 0; while (((size_t)1 << r) < x) r++; return r; }

#endif
