/*
 * oracle/prover.c — CPU restatement of one segment seal.
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see zkoracle.h).
 *
 * Follows risc0-zkp 3.0.2 (un-vendored; /root/reference/Cargo.lock:5393):
 *   src/prove/prover.rs     — Prover::{new, commit_group, finalize}
 *   src/prove/poly_group.rs — PolyGroup::new (expand-NTT, bit-reverse coeffs, Merkle)
 *   src/prove/merkle.rs     — MerkleTreeProver::{new, commit, prove}; src/merkle.rs MerkleTreeParams
 *   src/prove/fri.rs        — fri_prove, ProveRoundInfo
 *   src/prove/write_iop.rs  — WriteIOP
 * driven the way risc0-circuit-rv32im 4.0.2 (:5320) src/prove SegmentProver does: header, code, data,
 * mix -> accum, finalize.  Reached from /root/reference/crates/host/src/lib.rs:137.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "field.h"
#include "circuit.h"

static inline fp4 ld4(const uint32_t* p) { fp4 r; memcpy(&r, p, 16); return r; }
static inline void st4(uint32_t* p, fp4 v) { memcpy(p, &v, 16); }

/* Stage hook (tests/golden/make_golden_large.py): lets the checker hash every intermediate buffer of a seal, so that the
 * HIP path can be compared op by op at the BASELINE shape without keeping gigabytes of fixtures. */
static zko_stage_hook g_hook = NULL;
void zko_set_stage_hook(zko_stage_hook h) { g_hook = h; }
static void stage(const char* name, const uint32_t* p, size_t words) { if (g_hook) g_hook(name, p, words); }
static const char* GROUP_NAME[3] = {"accum", "code", "data"};

/* ---- WriteIOP ---- */
typedef struct { uint32_t* w; size_t n, cap; zko_rng rng; } iop_t;
static void iop_write(iop_t* io, const uint32_t* p, size_t n) {
    if (io->n + n > io->cap) { io->cap = (io->n + n) * 2 + 1024; io->w = (uint32_t*)realloc(io->w, io->cap * 4); }
    memcpy(io->w + io->n, p, n * 4); io->n += n;
}
static void iop_commit(iop_t* io, const uint32_t d[8]) { zko_rng_mix(&io->rng, d); }

/* ---- Merkle ---- */
typedef struct { size_t rows, cols, layers, top_layer, top_size; uint32_t* nodes; const uint32_t* matrix; } merkle_t;
static void merkle_params(merkle_t* m, size_t rows, size_t cols) {
    m->rows = rows; m->cols = cols; m->layers = log2_ceil(rows);
    m->top_layer = 0;
    for (size_t i = 1; i < m->layers; i++) { if (((size_t)1 << i) > ZKO_QUERIES) break; m->top_layer = i; }
    m->top_size = (size_t)1 << m->top_layer;
}
static void merkle_build(merkle_t* m, const uint32_t* matrix, size_t rows, size_t cols) {
    merkle_params(m, rows, cols);
    m->matrix = matrix;
    m->nodes = (uint32_t*)calloc(rows * 2, 32);
    zko_hash_rows(m->nodes + 8 * rows, rows, matrix, rows * cols);
    for (size_t i = m->layers; i-- > 0;) { size_t ls = (size_t)1 << i; zko_hash_fold(m->nodes, ls * 2, ls); }
}
static void merkle_commit(const merkle_t* m, iop_t* io) {
    iop_write(io, m->nodes + 8 * m->top_size, 8 * m->top_size);
    iop_commit(io, m->nodes + 8);
}
static void merkle_prove(const merkle_t* m, iop_t* io, size_t idx) {
    uint32_t* col = (uint32_t*)malloc(4 * m->cols);
    zko_gather_sample(col, m->matrix, idx, m->cols, m->rows);
    iop_write(io, col, m->cols);
    free(col);
    idx += m->rows;
    while (idx >= 2 * m->top_size) { iop_write(io, m->nodes + 8 * (idx ^ 1), 8); idx /= 2; }
}

/* ---- PolyGroup ---- */
typedef struct { uint32_t *coeffs, *evaluated; size_t count, n; merkle_t merkle; } polygroup_t;
static void polygroup_new(polygroup_t* pg, uint32_t* coeffs, size_t count, size_t n) {
    size_t dom = n * ZKO_INV_RATE;
    pg->coeffs = coeffs; pg->count = count; pg->n = n;
    pg->evaluated = (uint32_t*)malloc(4 * count * dom);
    zko_batch_expand_into_evaluate_ntt(pg->evaluated, count * dom, coeffs, count * n, count, log2_ceil(ZKO_INV_RATE));
    zko_batch_bit_reverse(coeffs, count * n, count);
    merkle_build(&pg->merkle, pg->evaluated, dom, count);
}
static void polygroup_stage(const polygroup_t* pg, const char* group) {
    char nm[64];
    size_t dom = pg->n * ZKO_INV_RATE;
    snprintf(nm, sizeof nm, "coeffs.%s", group); stage(nm, pg->coeffs, pg->count * pg->n);      /* natural order */
    snprintf(nm, sizeof nm, "evaluated.%s", group); stage(nm, pg->evaluated, pg->count * dom);
    snprintf(nm, sizeof nm, "nodes.%s", group); stage(nm, pg->merkle.nodes + 8, (2 * dom - 1) * 8);  /* root .. leaves */
}
static void polygroup_free(polygroup_t* pg) { free(pg->coeffs); free(pg->evaluated); free(pg->merkle.nodes); }
/* Prover::commit_group: takes ownership of a copy of the trace columns */
static void commit_group(polygroup_t* pg, iop_t* io, const uint32_t* trace, size_t count, size_t n) {
    uint32_t* coeffs = (uint32_t*)malloc(4 * count * n);
    memcpy(coeffs, trace, 4 * count * n);
    zko_batch_interpolate_ntt(coeffs, count * n, count);
    zko_zk_shift(coeffs, count * n, count);
    polygroup_new(pg, coeffs, count, n);
    merkle_commit(&pg->merkle, io);
}

/* ---- FRI ---- */
typedef struct { size_t domain; uint32_t* coeffs; size_t coeffs_size; uint32_t* evaluated; merkle_t merkle; } fri_round_t;
static void fri_round_new(fri_round_t* r, iop_t* io, const uint32_t* coeffs, size_t coeffs_size) {
    size_t size = coeffs_size / EXT_SIZE, domain = size * ZKO_INV_RATE;
    r->domain = domain;
    r->evaluated = (uint32_t*)malloc(4 * domain * EXT_SIZE);
    zko_batch_expand_into_evaluate_ntt(r->evaluated, domain * EXT_SIZE, coeffs, coeffs_size, EXT_SIZE,
                                       log2_ceil(ZKO_INV_RATE));
    merkle_build(&r->merkle, r->evaluated, domain / ZKO_FRI_FOLD, ZKO_FRI_FOLD * EXT_SIZE);
    merkle_commit(&r->merkle, io);
    uint32_t fold_mix[4];
    zko_rng_random_ext_elem(&io->rng, fold_mix);
    r->coeffs_size = size / ZKO_FRI_FOLD * EXT_SIZE;
    r->coeffs = (uint32_t*)malloc(4 * r->coeffs_size);
    zko_fri_fold(r->coeffs, r->coeffs_size, coeffs, fold_mix);
}

void zko_control_root(const zko_circuit* c, unsigned po2, unsigned zk, uint32_t root[8]) {
    size_t n = (size_t)1 << po2, wc = c->group_size[ZKC_GROUP_CODE];
    uint32_t* code = (uint32_t*)malloc(4 * wc * n);
    zko_syn_code(c, po2, zk, code);
    iop_t io; memset(&io, 0, sizeof io); zko_rng_init(&io.rng);
    polygroup_t pg;
    commit_group(&pg, &io, code, wc, n);
    memcpy(root, pg.merkle.nodes + 8, 32);
    polygroup_free(&pg); free(code); free(io.w);
}

/* Merkle root of a given code trace (wc x n): the control root of a circuit whose code group is a program (kind 4) */
void zko_root_of_code(const zko_circuit* c, unsigned po2, const uint32_t* code, uint32_t root[8]) {
    size_t n = (size_t)1 << po2, wc = c->group_size[ZKC_GROUP_CODE];
    iop_t io; memset(&io, 0, sizeof io); zko_rng_init(&io.rng);
    polygroup_t pg;
    commit_group(&pg, &io, code, wc, n);
    memcpy(root, pg.merkle.nodes + 8, 32);
    polygroup_free(&pg); free(io.w);
}

uint32_t* zko_prove_segment(const zko_circuit* c, unsigned po2, unsigned zk, uint64_t seed, const uint32_t* noise_key,
                            const uint32_t* pub, size_t* seal_words, const char** err) {
    *err = NULL;
    size_t n = (size_t)1 << po2;
    size_t wc = c->group_size[1], wd = c->group_size[2];
    if (n <= zk + 1) { *err = "po2 too small for zk_cycles"; return NULL; }
    if (c->kind == 4) { *err = "the recursion circuit has no seed-driven witness generator: zko_rec_witgen + zko_prove_traces"; return NULL; }
    /* witgen (SegmentProver step 2) */
    uint32_t* code = (uint32_t*)malloc(4 * wc * n);
    uint32_t* data = (uint32_t*)malloc(4 * wd * n);
    size_t out_size = c->global_size[ZKC_GLOBAL_OUT];
    uint32_t* out_global = (uint32_t*)malloc(4 * (out_size + 1));
    zko_syn_witgen(c, po2, zk, seed, noise_key, pub, code, data, out_global);
    uint32_t* seal = zko_prove_traces(c, po2, zk, noise_key, code, data, out_global, seal_words, err);
    free(code); free(data); free(out_global);
    return seal;
}

/* the seal of given code / data traces and out globals (the accum group is generated here, after the mix challenge) */
uint32_t* zko_prove_traces(const zko_circuit* c, unsigned po2, unsigned zk, const uint32_t* noise_key, const uint32_t* code,
                           const uint32_t* data, const uint32_t* out_words, size_t* seal_words, const char** err) {
    *err = NULL;
    size_t n = (size_t)1 << po2, dom = n * ZKO_INV_RATE;
    size_t wa = c->group_size[0], wc = c->group_size[1], wd = c->group_size[2];
    if (n <= zk + 1) { *err = "po2 too small for zk_cycles"; return NULL; }
    iop_t io; memset(&io, 0, sizeof io); zko_rng_init(&io.rng);
    size_t out_size = c->global_size[ZKC_GLOBAL_OUT];
    uint32_t* out_global = (uint32_t*)malloc(4 * (out_size + 1));
    memcpy(out_global, out_words, 4 * out_size);
    stage("trace.code", code, wc * n); stage("trace.data", data, wd * n);

    /* step 3: header — out globals + po2 as field elements (write_field_elem_slice), committed */
    {
        out_global[out_size] = fp_from_u32(po2);
        iop_write(&io, out_global, out_size + 1);
        uint32_t dg[8]; zko_hash_elem_slice(out_global, out_size + 1, 1, dg);
        iop_commit(&io, dg);
    }
    /* step 4: commit code, data */
    polygroup_t groups[3];
    commit_group(&groups[ZKC_GROUP_CODE], &io, code, wc, n);
    commit_group(&groups[ZKC_GROUP_DATA], &io, data, wd, n);
    polygroup_stage(&groups[ZKC_GROUP_CODE], GROUP_NAME[ZKC_GROUP_CODE]);
    polygroup_stage(&groups[ZKC_GROUP_DATA], GROUP_NAME[ZKC_GROUP_DATA]);
    /* step 5: accum mix + accum */
    uint32_t* mix_global = (uint32_t*)malloc(4 * (wa + c->global_size[ZKC_GLOBAL_MIX] + 1));
    for (size_t i = 0; i < c->global_size[ZKC_GLOBAL_MIX]; i++) mix_global[i] = zko_rng_random_elem(&io.rng);
    uint32_t* accum = (uint32_t*)malloc(4 * wa * n);
    if (c->kind == 4) zko_rec_accum(c, po2, zk, noise_key, code, data, mix_global, accum);
    else zko_syn_accum(c, po2, zk, noise_key, data, mix_global, accum);
    stage("global.mix", mix_global, c->global_size[ZKC_GLOBAL_MIX]);
    stage("trace.accum", accum, wa * n);
    commit_group(&groups[ZKC_GROUP_ACCUM], &io, accum, wa, n);
    polygroup_stage(&groups[ZKC_GROUP_ACCUM], GROUP_NAME[ZKC_GROUP_ACCUM]);
    free(accum);

    /* step 6: finalize */
    uint32_t poly_mix[4]; zko_rng_random_ext_elem(&io.rng, poly_mix);
    uint32_t* check = (uint32_t*)calloc(EXT_SIZE * dom, 4);
    const uint32_t* gev[3] = {groups[0].evaluated, groups[1].evaluated, groups[2].evaluated};
    const uint32_t* globals[2] = {out_global, mix_global};
    stage("poly_mix", poly_mix, 4);
    zko_eval_check(c, check, gev, globals, poly_mix, po2);
    stage("check.evaluated", check, EXT_SIZE * dom);
    zko_batch_interpolate_ntt(check, EXT_SIZE * dom, EXT_SIZE);
    polygroup_t check_group;
    polygroup_new(&check_group, check, ZKO_CHECK_SIZE, n);   /* 4 polys of 4n reinterpreted as 16 of n */
    merkle_commit(&check_group.merkle, &io);
    polygroup_stage(&check_group, "check");

    uint32_t zw[4]; zko_rng_random_ext_elem(&io.rng, zw);
    fp4 z = ld4(zw);
    fp4 back_one = fp4_from_fp(fp_rou_rev(po2));
    size_t n_taps = c->n_taps;
    fp4* all_xs = (fp4*)malloc(sizeof(fp4) * n_taps);
    fp4* eval_u = (fp4*)malloc(sizeof(fp4) * n_taps);
    {
        size_t pos = 0;
        for (unsigned g = 0; g < 3; g++) {
            size_t cnt = 0;
            for (size_t t = 0; t < n_taps; t++) if (c->taps[t].group == g) cnt++;
            uint32_t* which = (uint32_t*)malloc(4 * (cnt ? cnt : 1));
            size_t k = 0;
            for (size_t t = 0; t < n_taps; t++) if (c->taps[t].group == g) {
                which[k] = c->taps[t].offset;
                all_xs[pos + k] = fp4_mul(fp4_pow(back_one, c->taps[t].back), z);
                k++;
            }
            zko_batch_evaluate_any(groups[g].coeffs, groups[g].count * n, groups[g].count, which,
                                   (const uint32_t*)(all_xs + pos), cnt, (uint32_t*)(eval_u + pos));
            pos += cnt; free(which);
        }
    }
    size_t n_u = n_taps + ZKO_CHECK_SIZE;
    fp4* coeff_u = (fp4*)calloc(n_u, sizeof(fp4));
    {
        size_t pos = 0;
        for (size_t r = 0; r < c->n_regs; r++) {
            zko_poly_interpolate((uint32_t*)(coeff_u + pos), (const uint32_t*)(all_xs + pos),
                                 (const uint32_t*)(eval_u + pos), c->regs[r].size);
            pos += c->regs[r].size;
        }
        fp4 z_pow = fp4_pow(z, EXT_SIZE);
        uint32_t which[ZKO_CHECK_SIZE]; fp4 xs[ZKO_CHECK_SIZE];
        for (int i = 0; i < ZKO_CHECK_SIZE; i++) { which[i] = i; xs[i] = z_pow; }
        zko_batch_evaluate_any(check_group.coeffs, ZKO_CHECK_SIZE * n, ZKO_CHECK_SIZE, which, (const uint32_t*)xs,
                               ZKO_CHECK_SIZE, (uint32_t*)(coeff_u + pos));
    }
    stage("z", zw, 4);
    stage("coeff_u", (const uint32_t*)coeff_u, 4 * n_u);
    iop_write(&io, (const uint32_t*)coeff_u, 4 * n_u);
    {
        uint32_t dg[8]; zko_hash_elem_slice((const uint32_t*)coeff_u, 4 * n_u, 1, dg);   /* hash_ext_elem_slice */
        iop_commit(&io, dg);
    }
    uint32_t mixw[4]; zko_rng_random_ext_elem(&io.rng, mixw);
    fp4 mix = ld4(mixw);
    size_t combo_count = c->n_combos;
    uint32_t* combos = (uint32_t*)calloc(n * (combo_count + 1) * 4, 4);
    fp4 cur_mix = fp4_one();
    for (unsigned g = 0; g < 3; g++) {
        size_t gs = c->group_size[g];
        uint32_t* which = (uint32_t*)malloc(4 * (gs ? gs : 1));
        size_t k = 0;
        for (size_t r = 0; r < c->n_regs; r++) if (c->regs[r].group == g) which[k++] = c->regs[r].combo_id;
        if (k != gs) { *err = "every column of a group must have a register"; return NULL; }
        zko_mix_poly_coeffs(combos, (const uint32_t*)&cur_mix, mixw, groups[g].coeffs, which, gs, n);
        cur_mix = fp4_mul(cur_mix, fp4_pow(mix, gs));
        free(which);
    }
    {
        uint32_t which[ZKO_CHECK_SIZE];
        for (int i = 0; i < ZKO_CHECK_SIZE; i++) which[i] = (uint32_t)combo_count;
        zko_mix_poly_coeffs(combos, (const uint32_t*)&cur_mix, mixw, check_group.coeffs, which, ZKO_CHECK_SIZE, n);
    }
    stage("mix", mixw, 4);
    stage("combos.mixed", combos, n * (combo_count + 1) * 4);
    /* combos_prepare: subtract the U polys; combos_divide: divide by prod (x - z*back_one^back) */
    {
        size_t cur_pos = 0; fp4 cur = fp4_one();
        for (size_t r = 0; r < c->n_regs; r++) {
            for (size_t i = 0; i < c->regs[r].size; i++) {
                uint32_t* p = combos + 4 * (n * c->regs[r].combo_id + i);
                st4(p, fp4_sub(ld4(p), fp4_mul(cur, coeff_u[cur_pos + i])));
            }
            cur = fp4_mul(cur, mix); cur_pos += c->regs[r].size;
        }
        for (int i = 0; i < ZKO_CHECK_SIZE; i++) {
            uint32_t* p = combos + 4 * (n * combo_count);
            st4(p, fp4_sub(ld4(p), fp4_mul(cur, coeff_u[cur_pos])));
            cur_pos++; cur = fp4_mul(cur, mix);
        }
        fp4 z_pow = fp4_pow(z, EXT_SIZE);
        for (size_t i = 0; i <= combo_count; i++) {
            uint32_t* poly = combos + 4 * n * i;
            fp4 rem;
            if (i == combo_count) {
                zko_poly_divide(poly, n, (const uint32_t*)&z_pow, (uint32_t*)&rem);
                if (!fp4_eq(rem, fp4_zero())) { *err = "check combo remainder != 0"; return NULL; }
            } else {
                for (uint32_t b = c->combo_begin[i]; b < c->combo_begin[i + 1]; b++) {
                    fp4 pt = fp4_mul(z, fp4_pow(back_one, c->combo_backs[b]));
                    zko_poly_divide(poly, n, (const uint32_t*)&pt, (uint32_t*)&rem);
                    if (!fp4_eq(rem, fp4_zero())) { *err = "tap combo remainder != 0"; return NULL; }
                }
            }
        }
    }
    uint32_t* final_coeffs = (uint32_t*)malloc(4 * n * EXT_SIZE);
    zko_eltwise_sum_extelem(final_coeffs, n * EXT_SIZE, combos, n * (combo_count + 1));
    stage("combos.divided", combos, n * (combo_count + 1) * 4);
    zko_batch_bit_reverse(final_coeffs, n * EXT_SIZE, EXT_SIZE);
    stage("final_coeffs", final_coeffs, n * EXT_SIZE);
    free(combos);

    /* fri_prove */
    {
        size_t orig_domain = n * ZKO_INV_RATE;
        fri_round_t rounds[8]; size_t n_rounds = 0;
        const uint32_t* cur = final_coeffs; size_t cur_size = n * EXT_SIZE;
        while (cur_size / EXT_SIZE > ZKO_FRI_MIN_DEGREE) {
            fri_round_new(&rounds[n_rounds], &io, cur, cur_size);
            cur = rounds[n_rounds].coeffs; cur_size = rounds[n_rounds].coeffs_size; n_rounds++;
        }
        uint32_t* fin = (uint32_t*)malloc(4 * cur_size);
        memcpy(fin, cur, 4 * cur_size);
        zko_batch_bit_reverse(fin, cur_size, EXT_SIZE);
        iop_write(&io, fin, cur_size);
        uint32_t dg[8]; zko_hash_elem_slice(fin, cur_size, 1, dg);
        iop_commit(&io, dg);
        free(fin);
        for (int q = 0; q < ZKO_QUERIES; q++) {
            uint32_t rng = zko_rng_random_bits(&io.rng, log2_ceil(orig_domain));
            size_t pos = rng % orig_domain;
            for (unsigned g = 0; g < 3; g++) merkle_prove(&groups[g].merkle, &io, pos);
            merkle_prove(&check_group.merkle, &io, pos);
            for (size_t r = 0; r < n_rounds; r++) {
                size_t group = pos % (rounds[r].domain / ZKO_FRI_FOLD);
                merkle_prove(&rounds[r].merkle, &io, group);
                pos = group;
            }
        }
        for (size_t r = 0; r < n_rounds; r++) { free(rounds[r].coeffs); free(rounds[r].evaluated); free(rounds[r].merkle.nodes); }
    }
    free(final_coeffs); free(all_xs); free(eval_u); free(coeff_u); free(mix_global); free(out_global);
    for (int g = 0; g < 3; g++) polygroup_free(&groups[g]);
    polygroup_free(&check_group);
    *seal_words = io.n;
    return io.w;
}
