/*
 * oracle/hash.c — Poseidon2 over BabyBear (t = 24, rate 16, out 8), sponge, 2->1 compression, RNG.
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see zkoracle.h).
 *
 * Follows risc0-zkp 3.0.2 (un-vendored; /root/reference/Cargo.lock:5393):
 *   src/core/hash/poseidon2/mod.rs  — poseidon2_mix, multiply_by_m_ext / m_int, unpadded_hash, hash_pair
 *   src/core/hash/poseidon2/rng.rs  — Poseidon2Rng::{mix, random_bits, random_elem, random_ext_elem}
 * as summarised in SURVEY.md Appendix A.6.  Constant tables are data (include/zkh_poseidon2_consts.h).
 */
#include <string.h>
#include "field.h"
#include "zkoracle.h"
#include "../include/zkh_poseidon2_consts.h"

#define CELLS 24
#define RATE 16
#define OUT 8
#define HALF_FULL 4
#define PARTIAL 21

static fp g_rc[CELLS * (2 * HALF_FULL + PARTIAL)];
static fp g_diag[CELLS];
static int g_init = 0;

void zko_poseidon2_set_constants(const uint32_t* rc, const uint32_t* diag) {
    for (int i = 0; i < CELLS * (2 * HALF_FULL + PARTIAL); i++) g_rc[i] = fp_from_u32(rc[i]);
    for (int i = 0; i < CELLS; i++) g_diag[i] = fp_from_u32(diag[i]);
    g_init = 1;
}
static void ensure_init(void) {
    if (!g_init) zko_poseidon2_set_constants(ZKH_P2_ROUND_CONSTANTS, ZKH_P2_M_INT_DIAG);
}

static inline fp sbox(fp x) { /* x^7 */
    fp x2 = fp_mul(x, x), x4 = fp_mul(x2, x2), x6 = fp_mul(x4, x2);
    return fp_mul(x6, x);
}

/* M4 = [[5,7,1,3],[4,6,1,1],[1,3,5,7],[1,1,4,6]] applied to each 4-chunk, then every cell gets the sum of
 * the same-position cells of all chunks added (Poseidon2 paper appendix B; mod.rs multiply_by_m_ext). */
static void m_ext(fp* c) {
    fp sums[4] = {0, 0, 0, 0};
    for (int i = 0; i < CELLS; i += 4) {
        fp x0 = c[i], x1 = c[i + 1], x2 = c[i + 2], x3 = c[i + 3];
        fp t0 = fp_add(x0, x1), t1 = fp_add(x2, x3);
        fp t2 = fp_add(fp_add(x1, x1), t1), t3 = fp_add(fp_add(x3, x3), t0);
        fp t1_4 = fp_add(fp_add(t1, t1), fp_add(t1, t1)), t0_4 = fp_add(fp_add(t0, t0), fp_add(t0, t0));
        fp t4 = fp_add(t1_4, t3), t5 = fp_add(t0_4, t2);
        c[i] = fp_add(t3, t5); c[i + 1] = t5; c[i + 2] = fp_add(t2, t4); c[i + 3] = t4;
        for (int j = 0; j < 4; j++) sums[j] = fp_add(sums[j], c[i + j]);
    }
    for (int i = 0; i < CELLS; i++) c[i] = fp_add(c[i], sums[i & 3]);
}
/* M_int = J + diag: cells[i] = sum + diag[i]*cells[i]  (mod.rs multiply_by_m_int) */
static void m_int(fp* c) {
    fp sum = 0;
    for (int i = 0; i < CELLS; i++) sum = fp_add(sum, c[i]);
    for (int i = 0; i < CELLS; i++) c[i] = fp_add(sum, fp_mul(g_diag[i], c[i]));
}

void zko_poseidon2_mix(uint32_t c[24]) {
    ensure_init();
    int round = 0;
    m_ext(c);
    for (int r = 0; r < HALF_FULL; r++, round++) {
        for (int i = 0; i < CELLS; i++) c[i] = sbox(fp_add(c[i], g_rc[round * CELLS + i]));
        m_ext(c);
    }
    for (int r = 0; r < PARTIAL; r++, round++) {
        c[0] = sbox(fp_add(c[0], g_rc[round * CELLS]));
        m_int(c);
    }
    for (int r = 0; r < HALF_FULL; r++, round++) {
        for (int i = 0; i < CELLS; i++) c[i] = sbox(fp_add(c[i], g_rc[round * CELLS + i]));
        m_ext(c);
    }
}

/* mod.rs unpadded_hash: overwrite-absorb rate cells, permute every 16; zero-pad the tail (or empty input). */
void zko_hash_elem_slice(const uint32_t* in, size_t n, size_t stride, uint32_t out[8]) {
    fp st[CELLS];
    memset(st, 0, sizeof st);
    size_t unmixed = 0;
    for (size_t i = 0; i < n; i++) {
        st[unmixed++] = in[i * stride];
        if (unmixed == RATE) { zko_poseidon2_mix(st); unmixed = 0; }
    }
    if (unmixed != 0 || n == 0) {
        for (size_t i = unmixed; i < RATE; i++) st[i] = 0;
        zko_poseidon2_mix(st);
    }
    memcpy(out, st, OUT * sizeof(uint32_t));
}

void zko_hash_pair(const uint32_t a[8], const uint32_t b[8], uint32_t out[8]) {
    uint32_t both[16];
    memcpy(both, a, 32); memcpy(both + 8, b, 32);
    zko_hash_elem_slice(both, 16, 1, out);
}

/* rng.rs */
void zko_rng_init(zko_rng* r) { memset(r, 0, sizeof *r); }
void zko_rng_mix(zko_rng* r, const uint32_t d[8]) {
    if (r->pool_used != 0) { zko_poseidon2_mix(r->cells); r->pool_used = 0; }
    for (int i = 0; i < OUT; i++) r->cells[i] = fp_add(r->cells[i], d[i]);
    zko_poseidon2_mix(r->cells);
}
uint32_t zko_rng_random_elem(zko_rng* r) {
    if (r->pool_used == RATE) { zko_poseidon2_mix(r->cells); r->pool_used = 0; }
    return r->cells[r->pool_used++];
}
uint32_t zko_rng_random_bits(zko_rng* r, unsigned bits) {
    uint32_t val = fp_to_u32(zko_rng_random_elem(r));
    for (int i = 0; i < 3; i++) {
        uint32_t nv = fp_to_u32(zko_rng_random_elem(r));
        if (val == 0) val = nv;
    }
    return (bits >= 32 ? 0xffffffffu : ((1u << bits) - 1)) & val;
}
void zko_rng_random_ext_elem(zko_rng* r, uint32_t out[4]) {
    for (int i = 0; i < 4; i++) out[i] = zko_rng_random_elem(r);
}
