/* oracle/recursion.c — CPU witness generator of the RECURSION circuit (TEST INFRASTRUCTURE ONLY; the checker, never the
 * product).  zeth_amd/circuits/recursion.py states the circuit (six Fp4 wires + one gate per row, Poseidon2 blocks of 31
 * rows, a PLONK-style copy argument in the accum group) and the program blob; this file executes a program's witness schedule
 * op by op, fills the trace, and computes the copy argument's running products.  Stands in for the preflight + witness
 * generator of risc0-circuit-recursion 4.0.2 (un-vendored: /root/reference/Cargo.lock:5305).  Mirrored by
 * zeth_amd/csrc/recursion.hip. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/zkh_poseidon2_consts.h"
#include "circuit.h"
#include "field.h"

static inline fp4 ld4(const uint32_t* p) { fp4 r; memcpy(&r, p, 16); return r; }

#define RC_T 24
#define RC_BLOCK 12
#define RC_NW 6
#define RC_WD 72
#define RC_WA 12
#define RC_WC 58
#define RC_ROW_WORDS 13
#define RC_MAGIC 0x5a4b5231u
#define RC_VERSION 2u
enum { RO_INPUT = 1, RO_GEN, RO_MUX, RO_PACK, RO_UNPACK, RO_INV, RO_BITS, RO_P2, RO_EQ, RO_ISZ };
enum { RG_MUX = 1, RG_BOOL = 2, RG_EMB = 4, RG_PACK0 = 8, RG_PUB = 128, RG_SWAP = 256 };

typedef struct {
    uint32_t po2, zk, A, n_vars, n_consts, n_ops, n_inputs, n_p2, n_gates;
    const uint32_t *table, *pos, *consts, *ops;
} rec_prog;

static const char* rec_parse(const uint32_t* b, size_t words, rec_prog* p) {
    if (words < 16 || b[0] != RC_MAGIC || b[1] != RC_VERSION) return "recursion program: bad header";
    p->po2 = b[2]; p->zk = b[3]; p->A = b[4]; p->n_vars = b[5]; p->n_consts = b[6]; p->n_ops = b[7]; p->n_inputs = b[8];
    p->n_p2 = b[9]; p->n_gates = b[10];
    if (p->po2 < 1 || p->po2 > 24 || (size_t)p->A + p->zk != (size_t)1 << p->po2) return "recursion program: bad shape";
    size_t need = 16 + (size_t)p->A * (RC_ROW_WORDS + RC_NW) + p->n_consts + (size_t)8 * p->n_ops;
    if (words != need) return "recursion program: length does not match the header";
    p->table = b + 16;
    p->pos = p->table + (size_t)p->A * RC_ROW_WORDS;
    p->consts = p->pos + (size_t)p->A * RC_NW;
    p->ops = p->consts + p->n_consts;
    for (size_t i = 0; i < (size_t)p->A * RC_NW; i++) if (p->pos[i] > p->n_vars) return "recursion program: position names an unknown variable";
    return NULL;
}

static void rc_m_ext(fp* c) {
    static const unsigned M4[4][4] = {{5, 7, 1, 3}, {4, 6, 1, 1}, {1, 3, 5, 7}, {1, 1, 4, 6}};
    fp y[RC_T], sums[4] = {0, 0, 0, 0};
    for (int b = 0; b < RC_T; b += 4)
        for (int i = 0; i < 4; i++) {
            fp e = 0;
            for (int j = 0; j < 4; j++) e = fp_add(e, fp_mul(fp_from_u32(M4[i][j]), c[b + j]));
            y[b + i] = e;
            sums[i] = fp_add(sums[i], e);
        }
    for (int k = 0; k < RC_T; k++) c[k] = fp_add(y[k], sums[k & 3]);
}
/* one permutation as the 12 rows of a block (zeth_amd/circuits/recursion.py block_rows): rows[k] = S[24] ‖ Q[24] */
void zko_rec_p2_rows(const uint32_t in[RC_T], uint32_t rows[RC_BLOCK][2 * RC_T]) {
    memset(rows, 0, sizeof(fp) * RC_BLOCK * 2 * RC_T);
    fp s[RC_T];
    memcpy(rows[0], in, sizeof s);
    memcpy(s, in, sizeof s);
    rc_m_ext(s);
    unsigned rnd = 0, k = 1;
    for (int half = 0; half < 2; half++) {
        for (int f = 0; f < 4; f++, rnd++, k++) {
            fp* row = rows[k];
            memcpy(row, s, sizeof s);
            for (int j = 0; j < RC_T; j++) {
                fp u = fp_add(s[j], fp_from_u32(ZKH_P2_ROUND_CONSTANTS[rnd * RC_T + j]));
                fp q = fp_mul(fp_mul(u, u), u);
                row[RC_T + j] = q;
                s[j] = fp_mul(fp_mul(q, q), u);
            }
            rc_m_ext(s);
        }
        if (half) break;
        for (unsigned m = 12; m >= 9; m -= 3, k++) {
            fp* row = rows[k];
            memcpy(row, s, sizeof s);
            for (unsigned i = 0; i < m; i++, rnd++) {
                fp u = fp_add(s[0], fp_from_u32(ZKH_P2_ROUND_CONSTANTS[rnd * RC_T]));
                fp q = fp_mul(fp_mul(u, u), u), x7 = fp_mul(fp_mul(q, q), u), tot = x7;
                row[RC_T + 2 * i] = q; row[RC_T + 2 * i + 1] = x7;
                for (int j = 1; j < RC_T; j++) tot = fp_add(tot, s[j]);
                s[0] = fp_add(tot, fp_mul(fp_from_u32(ZKH_P2_M_INT_DIAG[0]), x7));
                for (int j = 1; j < RC_T; j++) s[j] = fp_add(tot, fp_mul(fp_from_u32(ZKH_P2_M_INT_DIAG[j]), s[j]));
            }
        }
    }
    memcpy(rows[RC_BLOCK - 1], s, sizeof s);
}

/* the input state of a conditional-swap block (recursion.py swap_state): cell 16 is the bit t (Montgomery 0 / 1); t set: the
 * first two digests change places; cell 16 itself enters the permutation as zero.  -> 0 if t is not a bit */
static int rc_swap_state(uint32_t cells[RC_T]) {
    const fp t = cells[16];
    if (t != 0 && t != fp_from_u32(1)) return 0;
    if (t) for (int j = 0; j < 8; j++) { uint32_t x = cells[j]; cells[j] = cells[j + 8]; cells[j + 8] = x; }
    cells[16] = 0;
    return 1;
}

/* the code group: a function of the program alone (its Merkle root is the program's control root) */
const char* zko_rec_code(const uint32_t* blob, size_t words, uint32_t* code) {
    rec_prog p;
    const char* e = rec_parse(blob, words, &p);
    if (e) return e;
    size_t n = (size_t)1 << p.po2, A = p.A, K = A / RC_BLOCK;
    fp one = fp_from_u32(1);
    memset(code, 0, 4 * (size_t)RC_WC * n);
#pragma omp parallel for schedule(static)
    for (size_t r = 0; r < A; r++) {
        const uint32_t* row = p.table + r * RC_ROW_WORDS;
        code[0 * n + r] = one;
        code[1 * n + r] = r == 0 ? one : 0;
        code[2 * n + r] = r > 0 ? one : 0;
        code[3 * n + r] = r == A - 1 ? one : 0;
        code[4 * n + r] = fp_from_u32((uint32_t)r);
        for (int w = 0; w < RC_NW; w++) code[(5 + w) * n + r] = fp_from_u32(row[7 + w]);
        for (int i = 0; i < 6; i++) code[(11 + i) * n + r] = fp_from_u32(row[i]);
        const uint32_t fl = row[6];
        code[17 * n + r] = (fl & RG_MUX) ? one : 0;
        code[18 * n + r] = (fl & RG_BOOL) ? one : 0;
        code[19 * n + r] = (fl & RG_EMB) ? one : 0;
        for (int j = 0; j < 4; j++) code[(20 + j) * n + r] = (fl & (RG_PACK0 << j)) ? one : 0;
        code[25 * n + r] = (fl & RG_PUB) ? one : 0;
        if (r >= RC_BLOCK * K) continue;
        const unsigned k = (unsigned)(r % RC_BLOCK);
        const int full = (k >= 1 && k <= 4) || (k >= 7 && k <= 10);
        const unsigned rnd = k <= 4 ? k - 1 : k == 5 ? 4 : k == 6 ? 16 : k + 18;      /* first round this row performs (k = 7: 25) */
        const int swap = k == 0 && (fl & RG_SWAP);
        code[24 * n + r] = ((k == 0 && !swap) || k == RC_BLOCK - 1) ? one : 0;
        code[57 * n + r] = swap ? one : 0;
        code[26 * n + r] = k == 1 ? one : 0;
        code[27 * n + r] = full ? one : 0;
        code[28 * n + r] = k == 5 ? one : 0;
        code[29 * n + r] = k == 6 ? one : 0;
        code[30 * n + r] = ((k >= 2 && k <= 5) || (k >= 8 && k <= 11)) ? one : 0;      /* the previous row did a full round */
        code[31 * n + r] = k == 6 ? one : 0;                                            /* ... the twelve partial rounds */
        code[32 * n + r] = k == 7 ? one : 0;                                            /* ... the nine partial rounds */
        if (full) for (int j = 0; j < RC_T; j++) code[(33 + j) * n + r] = fp_from_u32(ZKH_P2_ROUND_CONSTANTS[rnd * RC_T + j]);
        if (k == 5 || k == 6) for (unsigned i = 0; i < (k == 5 ? 12u : 9u); i++) code[(33 + i) * n + r] = fp_from_u32(ZKH_P2_ROUND_CONSTANTS[(rnd + i) * RC_T]);
    }
    return NULL;
}

static char rec_err[256];

/* Executes the witness schedule on `inputs` (raw Montgomery words, e.g. child seals) and fills code (58 x n), data (72 x n)
 * and out_global (16 words = the wires of the PUB row).  NULL on success; a message when the witness does not exist (an
 * assertion of the program fails: the child seal is not valid). */
const char* zko_rec_witgen(const uint32_t* blob, size_t words, const uint32_t* inputs, size_t n_inputs, const uint32_t* noise_key,
                           uint32_t* code, uint32_t* data, uint32_t* out_global) {
    rec_prog p;
    const char* e = rec_parse(blob, words, &p);
    if (e) return e;
    if (n_inputs < p.n_inputs) return "recursion witgen: the program reads more input words than were given";
    if (n_inputs > p.n_inputs) return "recursion witgen: more input words than the program reads (a seal with trailing words is not a valid seal)";
    if ((e = zko_rec_code(blob, words, code))) return e;
    size_t n = (size_t)1 << p.po2, A = p.A, K = A / RC_BLOCK;
    fp4* val = (fp4*)calloc(p.n_vars ? p.n_vars : 1, sizeof(fp4));
    fp* kk = (fp*)malloc(sizeof(fp) * (p.n_consts ? p.n_consts : 1));
    for (size_t i = 0; i < p.n_consts; i++) kk[i] = fp_from_u32(p.consts[i]);
    const char* fail = NULL;
    for (size_t i = 0; i < p.n_ops && !fail; i++) {
        const uint32_t* o = p.ops + 8 * i;
        const uint32_t op = o[0] & 0xff, aux = o[0] >> 8, out = o[1];
        const uint32_t* in = o + 2;
#define BAD(msg) do { snprintf(rec_err, sizeof rec_err, "recursion witgen: op %zu: %s", i, msg); fail = rec_err; } while (0)
        switch (op) {
        case RO_INPUT: {
            fp4 v = fp4_zero();
            for (uint32_t t = 0; t < aux && t < 4; t++) {
                v.c[t] = inputs[in[0] + t];
                if (v.c[t] >= FP_P) BAD("input word is not a reduced field element");
            }
            val[out] = v;
            break;
        }
        case RO_GEN: {
            const fp* q = kk + in[3];
            const fp4 a = val[in[0]], b = val[in[1]], c = val[in[2]];
            fp4 r = q[0] ? fp4_mul_fp(fp4_mul(a, b), q[0]) : fp4_zero();
            r = fp4_add(r, fp4_add(fp4_mul_fp(a, q[1]), fp4_add(fp4_mul_fp(b, q[2]), fp4_mul_fp(c, q[3]))));
            r.c[0] = fp_add(r.c[0], q[4]);
            val[out] = r;
            break;
        }
        case RO_MUX: {
            const fp4 a = val[in[0]], b = val[in[1]], c = val[in[2]];
            val[out] = fp4_add(b, fp4_mul_fp(fp4_sub(c, b), a.c[0]));
            break;
        }
        case RO_PACK: {
            fp4 r = {{val[in[0]].c[aux & 3], val[in[1]].c[aux & 3], val[in[2]].c[aux & 3], val[in[3]].c[aux & 3]}};
            val[out] = r;
            break;
        }
        case RO_UNPACK:
            for (int t = 0; t < 4; t++) val[out + t] = fp4_from_fp(val[in[0]].c[t]);
            break;
        case RO_INV: {
            const fp4 a = val[in[0]];
            if (!(a.c[0] | a.c[1] | a.c[2] | a.c[3])) BAD("inverse of zero");
            else val[out] = fp4_inv(a);
            break;
        }
        case RO_ISZ: val[out] = fp4_from_fp(val[in[0]].c[0] ? fp_inv(val[in[0]].c[0]) : 0); break;
        case RO_BITS: {
            const uint32_t x = fp_to_u32(val[in[0]].c[0]);
            for (int t = 0; t < 31; t++) val[out + t] = fp4_from_fp(((x >> t) & 1) ? fp_from_u32(1) : 0);
            break;
        }
        case RO_P2: {
            uint32_t cells[RC_T];
            for (int w = 0; w < RC_NW; w++) for (int t = 0; t < 4; t++) cells[4 * w + t] = val[in[w]].c[t];
            if ((aux & 1) && !rc_swap_state(cells)) { BAD("the selector of a conditional swap is not a bit"); break; }
            zko_poseidon2_mix(cells);
            for (int w = 0; w < RC_NW; w++) for (int t = 0; t < 4; t++) val[out + w].c[t] = cells[4 * w + t];
            break;
        }
        case RO_EQ:
            if (!fp4_eq(val[in[0]], val[in[1]])) BAD("two wires that the program ties together differ (the input is not a valid seal)");
            break;
        default: BAD("unknown opcode");
        }
#undef BAD
    }
    if (!fail) {
        memset(data, 0, 4 * (size_t)RC_WD * n);
#pragma omp parallel for schedule(static)
        for (size_t r = 0; r < A; r++)
            for (int w = 0; w < RC_NW; w++) {
                const uint32_t v = p.pos[r * RC_NW + w];
                if (v) for (int t = 0; t < 4; t++) data[(size_t)(4 * w + t) * n + r] = val[v - 1].c[t];
            }
#pragma omp parallel
        {
            fp (*rows)[2 * RC_T] = malloc(sizeof(fp) * RC_BLOCK * 2 * RC_T);
#pragma omp for schedule(static)
            for (size_t b = 0; b < K; b++) {
                fp in[RC_T];
                const size_t r0 = RC_BLOCK * b;
                for (int j = 0; j < RC_T; j++) in[j] = data[(size_t)j * n + r0];
                if (p.table[r0 * RC_ROW_WORDS + 6] & RG_SWAP) (void)rc_swap_state(in);
                zko_rec_p2_rows(in, rows);
                for (size_t k = 0; k < RC_BLOCK; k++)
                    for (size_t col = 0; col < 2 * RC_T; col++) data[(RC_T + col) * n + r0 + k] = rows[k][col];
                for (int j = 0; j < RC_T; j++) data[(size_t)j * n + r0 + RC_BLOCK - 1] = rows[RC_BLOCK - 1][j];
            }
            free(rows);
        }
        for (size_t col = 0; col < RC_WD; col++)
            for (size_t r = A; r < n; r++) data[col * n + r] = zko_noise_cell(noise_key, ZKC_GROUP_DATA, (uint32_t)col, (uint32_t)r);
        memset(out_global, 0, 64);
        for (size_t r = 0; r < A; r++)
            if (p.table[r * RC_ROW_WORDS + 6] & RG_PUB) { for (int i = 0; i < 16; i++) out_global[i] = data[(size_t)i * n + r]; break; }
    }
    free(val); free(kk);
    return fail;
}

/* the copy argument: Z_k(r) = Z_k(r - 1) * prod_{w in {2k, 2k+1}} F(id) / F(sigma); mix = beta_1..4, gamma (20 words) */
void zko_rec_accum(const zko_circuit* c, unsigned po2, unsigned zk, const uint32_t* noise_key, const uint32_t* code, const uint32_t* data,
                   const uint32_t* mix, uint32_t* accum) {
    (void)c;
    size_t n = (size_t)1 << po2, A = n - zk;
    fp4 beta[4], gamma = ld4(mix + 16);
    for (int i = 0; i < 4; i++) beta[i] = ld4(mix + 4 * i);
    fp4* ratio = (fp4*)malloc(sizeof(fp4) * A * 3);
#pragma omp parallel for schedule(static)
    for (size_t r = 0; r < A; r++)
        for (int k = 0; k < 3; k++) {
            fp4 num = fp4_one(), den = fp4_one();
            for (int w = 2 * k; w < 2 * k + 2; w++) {
                fp4 f = gamma;
                for (int i = 0; i < 4; i++) f = fp4_add(f, fp4_mul_fp(beta[i], data[(size_t)(4 * w + i) * n + r]));
                fp4 fi = f, fs = f;
                fi.c[0] = fp_add(fi.c[0], fp_from_u32((uint32_t)(RC_NW * r + w)));
                fs.c[0] = fp_add(fs.c[0], code[(size_t)(5 + w) * n + r]);
                num = fp4_mul(num, fi); den = fp4_mul(den, fs);
            }
            ratio[3 * r + k] = fp4_mul(num, fp4_inv(den));
        }
    for (int k = 0; k < 3; k++) {
        fp4 z = fp4_one();
        for (size_t r = 0; r < A; r++) {
            z = fp4_mul(z, ratio[3 * r + k]);
            for (int t = 0; t < 4; t++) accum[(size_t)(4 * k + t) * n + r] = z.c[t];
        }
    }
    free(ratio);
    for (size_t col = 0; col < RC_WA; col++)
        for (size_t r = A; r < n; r++) accum[col * n + r] = zko_noise_cell(noise_key, ZKC_GROUP_ACCUM, (uint32_t)col, (uint32_t)r);
}

/* Row-by-row check of a trace against the circuit's own step list: every constraint must vanish on every row of the trace
 * domain.  -> index of the first failing row, or -1.  (Any circuit; taps at back b read row r - b mod n.) */
long zko_check_rows(const zko_circuit* c, unsigned po2, const uint32_t* const* groups, const uint32_t* const* globals, size_t row_lo,
                    size_t row_hi) {
    const size_t n = (size_t)1 << po2;
    const uint32_t poly_mix[4] = {fp_from_u32(0x1234567), fp_from_u32(0x2345678), fp_from_u32(0x3456789), fp_from_u32(0x456789a)};
    long bad = -1;
#pragma omp parallel
    {
        uint32_t* u = (uint32_t*)malloc(16 * c->n_taps);
#pragma omp for schedule(dynamic, 256)
        for (size_t r = row_lo; r < row_hi; r++) {
            for (size_t t = 0; t < c->n_taps; t++) {
                const zkc_tap* tp = &c->taps[t];
                fp4 v = fp4_from_fp(groups[tp->group][(size_t)tp->offset * n + ((r + n - tp->back) & (n - 1))]);
                memcpy(u + 4 * t, v.c, 16);
            }
            uint32_t tot[4];
            zko_poly_ext(c, poly_mix, u, globals, tot);
            if (tot[0] | tot[1] | tot[2] | tot[3]) {
#pragma omp critical
                if (bad < 0 || (long)r < bad) bad = (long)r;
            }
        }
        free(u);
    }
    return bad;
}
