/*
 * oracle/circuit.c — circuit description (TapSet + PolyExtStep list), constraint interpreter,
 * CPU eval_check, and the SYN-AIR witness generator.
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see zkoracle.h).
 *
 * Follows risc0-zkp 3.0.2 (un-vendored; /root/reference/Cargo.lock:5393):
 *   src/taps.rs     — TapSet: taps sorted by (group, offset, back); registers; combos
 *   src/adapter.rs  — PolyExtStep / PolyExtStepDef::step (MixState {tot, mul})
 * and the per-circuit CircuitHal::eval_check CPU semantics of risc0-circuit-rv32im 4.0.2 (:5320),
 * SURVEY.md Appendix A.7.  The rv32im constraint list itself is a Zirgen-generated artefact that is not
 * obtainable offline; the circuit is DATA here (desc blob) and SYN-AIR is the declared-synthetic stand-in.
 */
#include <stdlib.h>
#include <string.h>
#include "field.h"
#include "circuit.h"
#include "noise.h"

static inline fp4 ld4(const uint32_t* p) { fp4 r; memcpy(&r, p, 16); return r; }
static inline void st4(uint32_t* p, fp4 v) { memcpy(p, &v, 16); }

const char* zko_circuit_load(const uint32_t* d, size_t words, zko_circuit** out) {
    if (words < ZKC_HEADER_WORDS || d[0] != ZKC_MAGIC || d[1] != 1) return "bad circuit desc header";
    zko_circuit* c = (zko_circuit*)calloc(1, sizeof *c);
    for (int g = 0; g < 3; g++) c->group_size[g] = d[3 + g];
    c->global_size[0] = d[7]; c->global_size[1] = d[8];
    c->n_taps = d[9]; c->n_combos = d[10]; c->n_steps = d[11]; c->ret = d[12]; c->kind = d[13];
    size_t pos = ZKC_HEADER_WORDS;
    c->taps = (zkc_tap*)calloc(c->n_taps, sizeof(zkc_tap));
    for (size_t i = 0; i < c->n_taps; i++, pos += 3) {
        c->taps[i].group = d[pos]; c->taps[i].offset = d[pos + 1]; c->taps[i].back = d[pos + 2];
    }
    c->combo_begin = (uint32_t*)calloc(c->n_combos + 1, 4);
    size_t p2 = pos, tot = 0;
    for (size_t i = 0; i < c->n_combos; i++) { tot += d[p2]; p2 += 1 + d[p2]; }
    c->combo_backs = (uint32_t*)calloc(tot ? tot : 1, 4);
    for (size_t i = 0, k = 0; i < c->n_combos; i++) {
        uint32_t cnt = d[pos++];
        c->combo_begin[i] = (uint32_t)k;
        for (uint32_t j = 0; j < cnt; j++) c->combo_backs[k++] = d[pos++];
        c->combo_begin[i + 1] = (uint32_t)k;
    }
    c->tot_combo_backs = tot;
    c->steps = (zkc_step*)calloc(c->n_steps, sizeof(zkc_step));
    if (pos + 5 * c->n_steps > words) { zko_circuit_free(c); return "circuit desc truncated"; }
    for (size_t i = 0; i < c->n_steps; i++, pos += 5) {
        c->steps[i].op = d[pos];
        for (int j = 0; j < 4; j++) c->steps[i].a[j] = d[pos + 1 + j];
    }
    /* registers: maximal runs of taps with the same (group, offset) */
    c->regs = (zkc_reg*)calloc(c->n_taps, sizeof(zkc_reg));
    for (size_t i = 0; i < c->n_taps;) {
        size_t j = i;
        while (j < c->n_taps && c->taps[j].group == c->taps[i].group && c->taps[j].offset == c->taps[i].offset) j++;
        zkc_reg* r = &c->regs[c->n_regs++];
        r->group = c->taps[i].group; r->offset = c->taps[i].offset; r->tap_begin = (uint32_t)i; r->size = (uint32_t)(j - i);
        r->combo_id = 0xffffffffu;
        for (size_t k = 0; k < c->n_combos; k++) {
            uint32_t b = c->combo_begin[k], e = c->combo_begin[k + 1];
            if (e - b != r->size) continue;
            int same = 1;
            for (uint32_t m = 0; m < r->size; m++) if (c->combo_backs[b + m] != c->taps[i + m].back) same = 0;
            if (same) { r->combo_id = (uint32_t)k; break; }
        }
        if (r->combo_id == 0xffffffffu) { zko_circuit_free(c); return "register without combo"; }
        i = j;
    }
    *out = c;
    return NULL;
}
void zko_circuit_free(zko_circuit* c) {
    if (!c) return;
    free(c->taps); free(c->combo_begin); free(c->combo_backs); free(c->steps); free(c->regs); free(c);
}
size_t zko_circuit_group_size(const zko_circuit* c, unsigned g) { return c->group_size[g]; }
size_t zko_circuit_tap_count(const zko_circuit* c) { return c->n_taps; }

/* adapter.rs PolyExtStepDef::step — evaluated over ExtElem (the verifier's view). */
typedef struct { fp4 tot, mul; } mix_state;
static void poly_ext_buf(const zko_circuit* c, const uint32_t poly_mix[4], const uint32_t* u, const uint32_t* const* globals,
                         uint32_t out[4], fp4* fpv, mix_state* mv) {
    size_t nf = 0, nm = 0;
    fp4 pm = ld4(poly_mix);
    for (size_t i = 0; i < c->n_steps; i++) {
        const zkc_step* s = &c->steps[i];
        switch (s->op) {
        case ZKC_CONST: fpv[nf++] = fp4_from_fp(fp_from_u32(s->a[0])); break;
        case ZKC_CONST_EXT: {
            fp4 v; for (int j = 0; j < 4; j++) v.c[j] = fp_from_u32(s->a[j]);
            fpv[nf++] = v; break; }
        case ZKC_GET: fpv[nf++] = ld4(u + 4 * s->a[0]); break;
        case ZKC_GET_GLOBAL: fpv[nf++] = fp4_from_fp(globals[s->a[0]][s->a[1]]); break;
        case ZKC_ADD: fpv[nf] = fp4_add(fpv[s->a[0]], fpv[s->a[1]]); nf++; break;
        case ZKC_SUB: fpv[nf] = fp4_sub(fpv[s->a[0]], fpv[s->a[1]]); nf++; break;
        case ZKC_MUL: fpv[nf] = fp4_mul(fpv[s->a[0]], fpv[s->a[1]]); nf++; break;
        case ZKC_TRUE: mv[nm].tot = fp4_zero(); mv[nm].mul = fp4_one(); nm++; break;
        case ZKC_AND_EQZ: {
            mix_state x = mv[s->a[0]]; fp4 v = fpv[s->a[1]];
            mv[nm].tot = fp4_add(x.tot, fp4_mul(x.mul, v));
            mv[nm].mul = fp4_mul(x.mul, pm); nm++; break; }
        case ZKC_AND_COND: {
            mix_state x = mv[s->a[0]]; fp4 cond = fpv[s->a[1]]; mix_state y = mv[s->a[2]];
            mv[nm].tot = fp4_add(x.tot, fp4_mul(fp4_mul(cond, y.tot), x.mul));
            mv[nm].mul = fp4_mul(x.mul, y.mul); nm++; break; }
        }
    }
    st4(out, mv[c->ret].tot);
}
void zko_poly_ext(const zko_circuit* c, const uint32_t poly_mix[4], const uint32_t* u, const uint32_t* const* globals,
                  uint32_t out[4]) {
    fp4* fpv = (fp4*)malloc(sizeof(fp4) * (c->n_steps + 1));
    mix_state* mv = (mix_state*)malloc(sizeof(mix_state) * (c->n_steps + 1));
    poly_ext_buf(c, poly_mix, u, globals, out, fpv, mv);
    free(fpv); free(mv);
}

/* CircuitHal::eval_check CPU semantics: every domain point idx of the 4n coset 3*H_{4n}.  Get(tap) reads
 * group[col*4n + ((idx - 4*back) mod 4n)]; result * 1/((3 w^idx)^n - 1); stored as 4 Fp planes. */
void zko_eval_check(const zko_circuit* c, uint32_t* check, const uint32_t* const* groups,
                    const uint32_t* const* globals, const uint32_t poly_mix[4], unsigned po2) {
    size_t n = (size_t)1 << po2, dom = n * ZKO_INV_RATE;
    fp w = fp_rou_fwd(po2 + 2), three = fp_from_u32(3), one = fp_from_u32(1);
#pragma omp parallel
    {
        uint32_t* u = (uint32_t*)malloc(16 * c->n_taps);
        fp4* fpv = (fp4*)malloc(sizeof(fp4) * (c->n_steps + 1));          /* per-thread scratch of the step interpreter */
        mix_state* mv = (mix_state*)malloc(sizeof(mix_state) * (c->n_steps + 1));
#pragma omp for schedule(static)
        for (size_t idx = 0; idx < dom; idx++) {
            for (size_t t = 0; t < c->n_taps; t++) {
                const zkc_tap* tp = &c->taps[t];
                size_t pos = (idx + dom - (size_t)ZKO_INV_RATE * tp->back) & (dom - 1);
                fp4 v = fp4_from_fp(groups[tp->group][(size_t)tp->offset * dom + pos]);
                st4(u + 4 * t, v);
            }
            uint32_t tot[4];
            poly_ext_buf(c, poly_mix, u, globals, tot, fpv, mv);
            fp x = fp_pow(w, idx);
            fp y = fp_pow(fp_mul(three, x), n);
            fp4 r = fp4_mul_fp(ld4(tot), fp_inv(fp_sub(y, one)));
            for (int p = 0; p < 4; p++) check[(size_t)p * dom + idx] = r.c[p];
        }
        free(u); free(fpv); free(mv);
    }
}

/* ---- blinding rows (noise.h) ---- */
uint32_t zko_noise_cell(const uint32_t* noise_key, uint32_t group, uint32_t col, uint32_t row) { return zko_noise_cell_inline(noise_key, group, col, row); }
void zko_chacha_block(const uint32_t* key, const uint32_t* tail, int double_rounds, uint32_t* out) { zko_chacha_block_inline(key, tail, double_rounds, out); }

/* ---- SYN-AIR witness ---- */
uint32_t zko_syn_cell(uint64_t seed, uint32_t group, uint32_t col, uint32_t row) {
    uint64_t z = seed ^ ((uint64_t)(group + 1) * 0x9E3779B97F4A7C15ull);
    z += (uint64_t)col * 0xBF58476D1CE4E5B9ull;
    z += (uint64_t)row * 0x94D049BB133111EBull;
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return fp_from_u32((uint32_t)(z >> 32) % FP_P);
}

/* The code group depends only on (circuit, po2, zk_cycles) — like upstream, where it is the program's control
 * columns and its Merkle root is the control ID the verifier checks (verify/mod.rs check_code). */
void zko_syn_code(const zko_circuit* c, unsigned po2, unsigned zk, uint32_t* code) {
    if (c->kind == 2) { zko_keccak_code(c, po2, zk, code); return; }       /* built-in witness generators: 1 SYN-AIR, 2 KECCAK-F, 3 P2-JOIN */
    if (c->kind == 3) { zko_p2join_code(c, po2, zk, code); return; }
    size_t n = (size_t)1 << po2, A = n - zk;
    size_t wc = c->group_size[ZKC_GROUP_CODE];
    fp one = fp_from_u32(1);
    for (size_t col = 0; col < wc; col++)
        for (size_t r = 0; r < n; r++) {
            fp v;
            switch (col) {
            case 0: v = r < A ? one : 0; break;
            case 1: v = r == 0 ? one : 0; break;
            case 2: v = (r > 0 && r < A) ? one : 0; break;
            case 3: v = fp_from_u32((uint32_t)r); break;
            case 4: v = r == A - 1 ? one : 0; break;
            default: v = zko_syn_cell(ZKO_SYN_CODE_SEED, ZKC_GROUP_CODE, (uint32_t)col, (uint32_t)r);
            }
            code[col * n + r] = v;
        }
}

void zko_syn_witgen(const zko_circuit* c, unsigned po2, unsigned zk, uint64_t seed, const uint32_t* noise_key,
                    const uint32_t* pub, uint32_t* code, uint32_t* data, uint32_t* out_global) {
    if (c->kind == 2) { zko_keccak_witgen(c, po2, zk, seed, noise_key, pub, code, data, out_global); return; }
    if (c->kind == 3) { zko_p2join_witgen(c, po2, zk, noise_key, pub, code, data, out_global); return; }
    size_t n = (size_t)1 << po2, A = n - zk;
    size_t wd = c->group_size[ZKC_GROUP_DATA];
    size_t n_pub = c->global_size[ZKC_GLOBAL_OUT] - 4;
    zko_syn_code(c, po2, zk, code);
    size_t T = (wd - 2) / 3;
    for (size_t col = 0; col < wd; col++)
        for (size_t r = 0; r < n; r++)
            data[col * n + r] = (r < A ? zko_syn_cell(seed, ZKC_GROUP_DATA, (uint32_t)col, (uint32_t)r) : zko_noise_cell(noise_key, ZKC_GROUP_DATA, (uint32_t)col, (uint32_t)r));
    /* public inputs: word k sits in row 0 of data column 3k (the x cell of triple k) and is bound to out[4 + k] */
    for (size_t k = 0; k < n_pub; k++) data[(3 * k) * n] = pub[k];
    fp s = 0;
    for (size_t r = 0; r < A; r++) {
        for (size_t j = 0; j < T; j++)
            data[(3 * j + 2) * n + r] = fp_mul(data[(3 * j) * n + r], data[(3 * j + 1) * n + r]);
        fp d0 = data[r], d1 = data[n + r], d3 = data[3 * n + r], d4 = data[4 * n + r];
        data[(wd - 2) * n + r] = fp_mul(fp_mul(d0, d1), fp_mul(d3, d4));
        s = r == 0 ? d0 : fp_add(s, fp_add(d0, fp_mul(code[3 * n + r], d1)));
        data[(wd - 1) * n + r] = s;
    }
    out_global[0] = s; out_global[1] = out_global[2] = out_global[3] = 0;
    for (size_t k = 0; k < n_pub; k++) out_global[4 + k] = pub[k];
}

void zko_syn_accum(const zko_circuit* c, unsigned po2, unsigned zk, const uint32_t* noise_key, const uint32_t* data,
                   const uint32_t* mix_global, uint32_t* accum) {
    size_t n = (size_t)1 << po2, A = n - zk;
    size_t wa = c->group_size[ZKC_GROUP_ACCUM], wd = c->group_size[ZKC_GROUP_DATA];
    for (size_t e = 0; e < wa / 4; e++) {
        fp4 m = ld4(mix_global + 4 * e), acc = fp4_one();
        const fp* d = data + (e % wd) * n;
        for (size_t r = 0; r < A; r++) {
            fp4 term = m; term.c[0] = fp_add(term.c[0], d[r]);
            acc = fp4_mul(acc, term);
            for (int p = 0; p < 4; p++) accum[(4 * e + p) * n + r] = acc.c[p];
        }
    }
    for (size_t col = 0; col < wa; col++)
        for (size_t r = A; r < n; r++)
            accum[col * n + r] = zko_noise_cell(noise_key, ZKC_GROUP_ACCUM, (uint32_t)col, (uint32_t)r);
}
