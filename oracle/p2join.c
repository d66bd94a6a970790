/* oracle/p2join.c — CPU witness generator of the P2-JOIN circuit (TEST INFRASTRUCTURE ONLY; the checker, never the
 * product).  Every 31 active rows are one Poseidon2 permutation (risc0-zkp 3.0.2 src/core/hash/poseidon2/mod.rs,
 * un-vendored: /root/reference/Cargo.lock:5393) laid out round by round as zeth_amd/circuits/p2_join.py states; block 0 is
 * hash_pair(left, right) = the parent claim of a join (stands in for the in-circuit hashing of risc0-circuit-recursion 4.0.2,
 * /root/reference/Cargo.lock:5305).  Mirrored by k_p2join_* in zeth_amd/csrc/circuit.hip. */
#include <stdlib.h>
#include <string.h>

#include "../include/zkh_poseidon2_consts.h"
#include "circuit.h"
#include "field.h"

#define PJ_T 24
#define PJ_HALF 4
#define PJ_RP 21
#define PJ_ROUNDS 29
#define PJ_BLOCK 31

static void pj_m_ext(fp* c) {
    static const unsigned M4[4][4] = {{5, 7, 1, 3}, {4, 6, 1, 1}, {1, 3, 5, 7}, {1, 1, 4, 6}};
    fp y[PJ_T], sums[4] = {0, 0, 0, 0};
    for (int b = 0; b < PJ_T; b += 4)
        for (int i = 0; i < 4; i++) {
            fp e = 0;
            for (int j = 0; j < 4; j++) e = fp_add(e, fp_mul(fp_from_u32(M4[i][j]), c[b + j]));
            y[b + i] = e;
            sums[i] = fp_add(sums[i], e);
        }
    for (int k = 0; k < PJ_T; k++) c[k] = fp_add(y[k], sums[k & 3]);
}
static int pj_is_full(unsigned rnd) { return rnd < PJ_HALF || rnd >= PJ_HALF + PJ_RP; }

/* rows[k][0..24) = S, rows[k][24..48) = Q for k < 31; in: 24 Montgomery words */
void zko_p2_rows(const uint32_t in[PJ_T], uint32_t rows[PJ_BLOCK][2 * PJ_T]) {
    memset(rows, 0, sizeof(fp) * PJ_BLOCK * 2 * PJ_T);
    fp s[PJ_T];
    memcpy(rows[0], in, sizeof(fp) * PJ_T);
    memcpy(s, in, sizeof s);
    pj_m_ext(s);
    for (unsigned rnd = 0; rnd < PJ_ROUNDS; rnd++) {
        fp* row = rows[1 + rnd];
        memcpy(row, s, sizeof s);
        if (pj_is_full(rnd)) {
            for (int j = 0; j < PJ_T; j++) {
                fp u = fp_add(s[j], fp_from_u32(ZKH_P2_ROUND_CONSTANTS[rnd * PJ_T + j]));
                fp q = fp_mul(fp_mul(u, u), u);
                row[PJ_T + j] = q;
                s[j] = fp_mul(fp_mul(q, q), u);
            }
            pj_m_ext(s);
        } else {
            fp u = fp_add(s[0], fp_from_u32(ZKH_P2_ROUND_CONSTANTS[rnd * PJ_T]));
            fp q = fp_mul(fp_mul(u, u), u);
            row[PJ_T] = q;
            fp x7 = fp_mul(fp_mul(q, q), u), tot = x7;
            for (int j = 1; j < PJ_T; j++) tot = fp_add(tot, s[j]);
            s[0] = fp_add(tot, fp_mul(fp_from_u32(ZKH_P2_M_INT_DIAG[0]), x7));
            for (int j = 1; j < PJ_T; j++) s[j] = fp_add(tot, fp_mul(fp_from_u32(ZKH_P2_M_INT_DIAG[j]), s[j]));
        }
    }
    memcpy(rows[PJ_BLOCK - 1], s, sizeof s);
}

void zko_p2join_code(const zko_circuit* c, unsigned po2, unsigned zk, uint32_t* code) {
    size_t n = (size_t)1 << po2, A = n - zk, K = A / PJ_BLOCK;
    size_t wc = c->group_size[ZKC_GROUP_CODE];
    fp one = fp_from_u32(1);
    memset(code, 0, 4 * wc * n);
    for (size_t r = 0; r < n; r++) {
        code[0 * n + r] = r < A ? one : 0;
        code[1 * n + r] = r == 0 ? one : 0;
        code[2 * n + r] = (r > 0 && r < A) ? one : 0;
        if (r >= PJ_BLOCK * K) continue;
        const unsigned k = (unsigned)(r % PJ_BLOCK);
        const int full = k >= 1 && k <= PJ_ROUNDS && pj_is_full(k - 1), part = k >= 1 && k <= PJ_ROUNDS && !pj_is_full(k - 1);
        code[3 * n + r] = r == 0 ? one : 0;
        code[4 * n + r] = (k == 0 && r > 0) ? one : 0;
        code[5 * n + r] = k == 1 ? one : 0;
        code[6 * n + r] = full ? one : 0;
        code[7 * n + r] = part ? one : 0;
        code[8 * n + r] = (k >= 2 && pj_is_full(k - 2)) ? one : 0;             /* the previous row did a full round */
        code[9 * n + r] = (k >= 2 && !pj_is_full(k - 2)) ? one : 0;
        code[10 * n + r] = r == PJ_BLOCK - 1 ? one : 0;
        if (full) for (int j = 0; j < PJ_T; j++) code[(11 + j) * n + r] = fp_from_u32(ZKH_P2_ROUND_CONSTANTS[(k - 1) * PJ_T + j]);
        if (part) code[11 * n + r] = fp_from_u32(ZKH_P2_ROUND_CONSTANTS[(k - 1) * PJ_T]);
        if (k == 0 && r > 0) for (int j = 0; j < 8; j++) code[(35 + j) * n + r] = zko_syn_cell(ZKO_SYN_CODE_SEED, ZKC_GROUP_CODE, 35 + j, (uint32_t)r);
    }
}

/* children: 16 words = left claim ‖ right claim (Montgomery words).  out_global: parent (8) ‖ left (8) ‖ right (8). */
void zko_p2join_witgen(const zko_circuit* c, unsigned po2, unsigned zk, const uint32_t* noise_key, const uint32_t* children,
                       uint32_t* code, uint32_t* data, uint32_t* out_global) {
    size_t n = (size_t)1 << po2, A = n - zk, K = A / PJ_BLOCK;
    size_t wd = c->group_size[ZKC_GROUP_DATA];
    zko_p2join_code(c, po2, zk, code);
    memset(data, 0, 4 * wd * n);
    fp (*rows)[2 * PJ_T] = malloc(sizeof(fp) * PJ_BLOCK * 2 * PJ_T);
    fp in[PJ_T], parent[8] = {0};
    for (size_t p = 0; p < K; p++) {
        memset(in, 0, sizeof in);
        if (p == 0) memcpy(in, children, 64);
        else {
            memcpy(in, parent, 32);
            for (int j = 0; j < 8; j++) in[8 + j] = code[(35 + j) * n + PJ_BLOCK * p];
        }
        zko_p2_rows(in, rows);
        if (p == 0) memcpy(parent, rows[PJ_BLOCK - 1], 32);
        for (size_t k = 0; k < PJ_BLOCK; k++)
            for (size_t col = 0; col < 2 * PJ_T; col++) data[col * n + PJ_BLOCK * p + k] = rows[k][col];
    }
    free(rows);
    memcpy(out_global, parent, 32);
    memcpy(out_global + 8, children, 64);
    for (size_t col = 0; col < wd; col++)
        for (size_t r = A; r < n; r++)
            data[col * n + r] = zko_noise_cell(noise_key, ZKC_GROUP_DATA, (uint32_t)col, (uint32_t)r);
}
