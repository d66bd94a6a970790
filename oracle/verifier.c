/*
 * oracle/verifier.c — independent restatement of the STARK verifier: the acceptance test for every seal.
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see zkoracle.h).
 *
 * Follows risc0-zkp 3.0.2 (un-vendored; /root/reference/Cargo.lock:5393):
 *   src/verify/mod.rs      — verify (DEEP-ALI check at z, combo_u, fri_eval_taps)
 *   src/verify/merkle.rs   — MerkleTreeVerifier::{new, verify}
 *   src/verify/fri.rs      — fri_verify, VerifyRoundInfo::verify_query, fold_eval
 *   src/verify/read_iop.rs — ReadIOP
 * This is what /root/reference/crates/host/src/bin/cli.rs:103 (`receipt.verify(image_id)`) runs per segment.
 */
#include <stdlib.h>
#include <string.h>
#include "field.h"
#include "circuit.h"

static inline fp4 ld4(const uint32_t* p) { fp4 r; memcpy(&r, p, 16); return r; }

typedef struct { const uint32_t* w; size_t n, pos; zko_rng rng; int bad; } riop_t;
static const uint32_t* riop_read(riop_t* io, size_t n) {
    static const uint32_t zeros[256] = {0};
    if (io->pos + n > io->n) { io->bad = 1; return n <= 256 ? zeros : NULL; }
    const uint32_t* p = io->w + io->pos; io->pos += n; return p;
}
static int elems_reduced(const uint32_t* p, size_t n) {
    for (size_t i = 0; i < n; i++) if (p[i] >= FP_P) return 0;
    return 1;
}

typedef struct { size_t rows, cols, layers, top_layer, top_size; uint32_t* top; } mverif_t;
static const char* mverif_new(mverif_t* m, riop_t* io, size_t rows, size_t cols) {
    m->rows = rows; m->cols = cols; m->layers = log2_ceil(rows); m->top_layer = 0;
    for (size_t i = 1; i < m->layers; i++) { if (((size_t)1 << i) > ZKO_QUERIES) break; m->top_layer = i; }
    m->top_size = (size_t)1 << m->top_layer;
    m->top = (uint32_t*)calloc(m->top_size * 2, 32);
    const uint32_t* p = riop_read(io, 8 * m->top_size);
    if (io->bad || !p) return "seal truncated (merkle top)";
    memcpy(m->top + 8 * m->top_size, p, 32 * m->top_size);
    for (size_t i = m->top_size; i-- > 1;) zko_hash_pair(m->top + 8 * (2 * i), m->top + 8 * (2 * i + 1), m->top + 8 * i);
    zko_rng_mix(&io->rng, m->top + 8);
    return NULL;
}
static const char* mverif_verify(const mverif_t* m, riop_t* io, size_t idx, const uint32_t** out) {
    if (idx >= m->rows) return "merkle index out of range";
    const uint32_t* col = riop_read(io, m->cols);
    if (io->bad || !col) return "seal truncated (merkle column)";
    if (!elems_reduced(col, m->cols)) return "unreduced field element in seal";
    uint32_t cur[8], other[8];
    zko_hash_elem_slice(col, m->cols, 1, cur);
    idx += m->rows;
    while (idx >= 2 * m->top_size) {
        size_t low = idx & 1;
        const uint32_t* o = riop_read(io, 8);
        if (io->bad) return "seal truncated (merkle path)";
        memcpy(other, o, 32);
        idx /= 2;
        uint32_t nx[8];
        if (low) zko_hash_pair(other, cur, nx); else zko_hash_pair(cur, other, nx);
        memcpy(cur, nx, 32);
    }
    if (memcmp(cur, m->top + 8 * idx, 32) != 0) return "merkle path mismatch";
    *out = col;
    return NULL;
}

/* verify/fri.rs fold_eval: interpolate the 16 values to coefficients, evaluate at the mix with shift */
static void ext_interpolate_ntt16(fp4* io) {
    /* interpolate_ntt over ExtElem values with Fp roots (ntt.rs is generic over the value type) */
    for (unsigned N = ZKO_FRI_FOLD_PO2; N >= 1; N--) {
        size_t len = (size_t)1 << N, half = len >> 1;
        fp step = fp_rou_rev(N);
        for (size_t s = 0; s < ZKO_FRI_FOLD; s += len) {
            fp cur = fp_from_u32(1);
            for (size_t i = 0; i < half; i++) {
                fp4 a = io[s + i], b = io[s + i + half];
                io[s + i] = fp4_add(a, b);
                io[s + i + half] = fp4_mul_fp(fp4_sub(a, b), cur);
                cur = fp_mul(cur, step);
            }
        }
    }
    fp norm = fp_inv(fp_from_u32(ZKO_FRI_FOLD));
    for (int i = 0; i < ZKO_FRI_FOLD; i++) io[i] = fp4_mul_fp(io[i], norm);
}
static fp4 fold_eval(fp4* io, fp4 x, fp inv_wk) {
    ext_interpolate_ntt16(io);
    for (unsigned i = 0; i < ZKO_FRI_FOLD; i++) {
        unsigned r = bit_rev32(i) >> (32 - ZKO_FRI_FOLD_PO2);
        if (i < r) { fp4 t = io[i]; io[i] = io[r]; io[r] = t; }
    }
    fp4 tot = fp4_zero(), mul_x = fp4_one();
    fp mul_wk = fp_from_u32(1);
    for (int i = 0; i < ZKO_FRI_FOLD; i++) {
        tot = fp4_add(tot, fp4_mul(fp4_mul_fp(io[i], mul_wk), mul_x));
        mul_x = fp4_mul(mul_x, x);
        mul_wk = fp_mul(mul_wk, inv_wk);
    }
    return tot;
}

typedef struct { size_t domain; mverif_t merkle; fp4 mix; } vround_t;

#define FAIL(msg) do { ret = (msg); goto done; } while (0)

const char* zko_verify_segment(const zko_circuit* c, const uint32_t* seal, size_t seal_words,
                               const uint32_t control_root[8]) {
    const char* ret = NULL;
    riop_t io; memset(&io, 0, sizeof io); io.w = seal; io.n = seal_words; zko_rng_init(&io.rng);
    mverif_t mg[3], mcheck; memset(mg, 0, sizeof mg); memset(&mcheck, 0, sizeof mcheck);
    vround_t rounds[8]; size_t n_rounds = 0; memset(rounds, 0, sizeof rounds);
    fp4 *eval_u = NULL, *combo_u = NULL, *tap_mix_pows = NULL, *poly_buf = NULL, *tot = NULL;
    uint32_t* mix_global = NULL;

    /* header */
    if (!control_root) return "no control root given (check_code needs the expected code commitment)";
    size_t out_size = c->global_size[ZKC_GLOBAL_OUT];
    if (out_size + 1 > 256) return "output global too large";
    const uint32_t* out_global = riop_read(&io, out_size + 1);     /* out words, then po2 as an Elem */
    if (io.bad) return "seal truncated (header)";
    if (!elems_reduced(out_global, out_size + 1)) return "unreduced output";
    unsigned po2 = fp_to_u32(out_global[out_size]);
    if (po2 > 24 || po2 < 1) return "bad po2";
    {
        uint32_t dg[8]; zko_hash_elem_slice(out_global, out_size + 1, 1, dg); zko_rng_mix(&io.rng, dg);
    }
    size_t size = (size_t)1 << po2, domain = ZKO_INV_RATE * size;
    const char* e;
    if ((e = mverif_new(&mg[ZKC_GROUP_CODE], &io, domain, c->group_size[ZKC_GROUP_CODE]))) FAIL(e);
    /* check_code: the code commitment must be the one registered for (circuit, po2) */
    if (memcmp(mg[ZKC_GROUP_CODE].top + 8, control_root, 32) != 0) FAIL("code root does not match the control root");
    if ((e = mverif_new(&mg[ZKC_GROUP_DATA], &io, domain, c->group_size[ZKC_GROUP_DATA]))) FAIL(e);
    mix_global = (uint32_t*)malloc(4 * (c->global_size[ZKC_GLOBAL_MIX] + 1));
    for (size_t i = 0; i < c->global_size[ZKC_GLOBAL_MIX]; i++) mix_global[i] = zko_rng_random_elem(&io.rng);
    if ((e = mverif_new(&mg[ZKC_GROUP_ACCUM], &io, domain, c->group_size[ZKC_GROUP_ACCUM]))) FAIL(e);
    uint32_t poly_mix[4]; zko_rng_random_ext_elem(&io.rng, poly_mix);
    if ((e = mverif_new(&mcheck, &io, domain, ZKO_CHECK_SIZE))) FAIL(e);
    uint32_t zw[4]; zko_rng_random_ext_elem(&io.rng, zw);
    fp4 z = ld4(zw);
    fp back_one = fp_rou_rev(po2);
    size_t num_taps = c->n_taps;
    const uint32_t* cu_words = riop_read(&io, 4 * (num_taps + ZKO_CHECK_SIZE));
    if (io.bad || !cu_words) FAIL("seal truncated (coeff_u)");
    if (!elems_reduced(cu_words, 4 * (num_taps + ZKO_CHECK_SIZE))) FAIL("unreduced coeff_u");
    const fp4* coeff_u = (const fp4*)cu_words;
    {
        uint32_t dg[8]; zko_hash_elem_slice(cu_words, 4 * (num_taps + ZKO_CHECK_SIZE), 1, dg);
        zko_rng_mix(&io.rng, dg);
    }
    /* U polys: coefficient form -> evaluations at z * back_one^back */
    eval_u = (fp4*)malloc(sizeof(fp4) * num_taps);
    {
        size_t cur_pos = 0;
        for (size_t r = 0; r < c->n_regs; r++) {
            const zkc_reg* reg = &c->regs[r];
            for (size_t i = 0; i < reg->size; i++) {
                fp4 x = fp4_mul_fp(z, fp_pow(back_one, c->taps[reg->tap_begin + i].back));
                zko_poly_eval((const uint32_t*)(coeff_u + cur_pos), reg->size, (const uint32_t*)&x,
                              (uint32_t*)&eval_u[cur_pos + i]);
            }
            cur_pos += reg->size;
        }
    }
    /* constraint polynomial at z vs check polynomial at z */
    {
        const uint32_t* globals[2] = {out_global, mix_global};
        uint32_t resw[4];
        zko_poly_ext(c, poly_mix, (const uint32_t*)eval_u, globals, resw);
        fp4 result = ld4(resw), check = fp4_zero();
        static const int remap[4] = {0, 2, 1, 3};
        fp one = fp_from_u32(1);
        for (int i = 0; i < 4; i++) {
            int rmi = remap[i];
            fp4 zi = fp4_pow(z, i);
            for (int k = 0; k < 4; k++) {
                fp4 basis = fp4_zero(); basis.c[k] = one;
                check = fp4_add(check, fp4_mul(fp4_mul(coeff_u[num_taps + rmi + 4 * k], zi), basis));
            }
        }
        fp4 three_z = fp4_mul_fp(z, fp_from_u32(3));
        check = fp4_mul(check, fp4_sub(fp4_pow(three_z, size), fp4_one()));
        if (!fp4_eq(check, result)) FAIL("constraint check failed: check(z) != poly_ext(z)");
    }
    uint32_t mixw[4]; zko_rng_random_ext_elem(&io.rng, mixw);
    fp4 mix = ld4(mixw);
    size_t combo_count = c->n_combos;
    combo_u = (fp4*)calloc(c->tot_combo_backs + 1, sizeof(fp4));
    tap_mix_pows = (fp4*)malloc(sizeof(fp4) * (c->n_regs + ZKO_CHECK_SIZE));
    {
        fp4 cur_mix = fp4_one(); size_t cur_pos = 0;
        for (size_t r = 0; r < c->n_regs; r++) {
            const zkc_reg* reg = &c->regs[r];
            for (size_t i = 0; i < reg->size; i++) {
                fp4* p = &combo_u[c->combo_begin[reg->combo_id] + i];
                *p = fp4_add(*p, fp4_mul(cur_mix, coeff_u[cur_pos + i]));
            }
            tap_mix_pows[r] = cur_mix;
            cur_mix = fp4_mul(cur_mix, mix); cur_pos += reg->size;
        }
        for (int i = 0; i < ZKO_CHECK_SIZE; i++) {
            combo_u[c->tot_combo_backs] = fp4_add(combo_u[c->tot_combo_backs], fp4_mul(cur_mix, coeff_u[cur_pos]));
            cur_pos++;
            tap_mix_pows[c->n_regs + i] = cur_mix;
            cur_mix = fp4_mul(cur_mix, mix);
        }
    }
    /* fri_verify */
    {
        size_t degree = size, dom = domain, orig_domain = domain;
        while (degree > ZKO_FRI_MIN_DEGREE) {
            vround_t* r = &rounds[n_rounds];
            r->domain = dom;
            if ((e = mverif_new(&r->merkle, &io, dom / ZKO_FRI_FOLD, ZKO_FRI_FOLD * EXT_SIZE))) { n_rounds++; FAIL(e); }
            uint32_t m[4]; zko_rng_random_ext_elem(&io.rng, m); r->mix = ld4(m);
            n_rounds++;
            dom /= ZKO_FRI_FOLD; degree /= ZKO_FRI_FOLD;
        }
        const uint32_t* final_coeffs = riop_read(&io, EXT_SIZE * degree);
        if (io.bad || !final_coeffs) FAIL("seal truncated (final coeffs)");
        if (!elems_reduced(final_coeffs, EXT_SIZE * degree)) FAIL("unreduced final coeffs");
        {
            uint32_t dg[8]; zko_hash_elem_slice(final_coeffs, EXT_SIZE * degree, 1, dg); zko_rng_mix(&io.rng, dg);
        }
        fp gen = fp_rou_fwd(log2_ceil(dom));
        fp gen_orig = fp_rou_fwd(log2_ceil(orig_domain));
        poly_buf = (fp4*)malloc(sizeof(fp4) * degree);
        for (size_t i = 0; i < degree; i++) for (int j = 0; j < 4; j++) poly_buf[i].c[j] = final_coeffs[j * degree + i];
        tot = (fp4*)malloc(sizeof(fp4) * (combo_count + 1));
        for (int q = 0; q < ZKO_QUERIES; q++) {
            uint32_t rng = zko_rng_random_bits(&io.rng, log2_ceil(orig_domain));
            size_t pos = rng % orig_domain;
            /* inner: DEEP quotient value at x = gen^pos from the opened rows (fri_eval_taps) */
            fp4 goal;
            {
                fp4 x = fp4_from_fp(fp_pow(gen_orig, pos));
                const uint32_t* rows[3]; const uint32_t* check_row;
                for (unsigned g = 0; g < 3; g++) if ((e = mverif_verify(&mg[g], &io, pos, &rows[g]))) FAIL(e);
                if ((e = mverif_verify(&mcheck, &io, pos, &check_row))) FAIL(e);
                for (size_t i = 0; i <= combo_count; i++) tot[i] = fp4_zero();
                for (size_t r = 0; r < c->n_regs; r++) {
                    const zkc_reg* reg = &c->regs[r];
                    tot[reg->combo_id] = fp4_add(tot[reg->combo_id], fp4_mul_fp(tap_mix_pows[r], rows[reg->group][reg->offset]));
                }
                for (int i = 0; i < ZKO_CHECK_SIZE; i++)
                    tot[combo_count] = fp4_add(tot[combo_count], fp4_mul_fp(tap_mix_pows[c->n_regs + i], check_row[i]));
                fp4 acc = fp4_zero();
                for (size_t i = 0; i < combo_count; i++) {
                    uint32_t b = c->combo_begin[i], en = c->combo_begin[i + 1];
                    fp4 px; zko_poly_eval((const uint32_t*)(combo_u + b), en - b, (const uint32_t*)&x, (uint32_t*)&px);
                    fp4 divisor = fp4_one();
                    for (uint32_t k = b; k < en; k++)
                        divisor = fp4_mul(divisor, fp4_sub(x, fp4_mul_fp(z, fp_pow(back_one, c->combo_backs[k]))));
                    acc = fp4_add(acc, fp4_mul(fp4_sub(tot[i], px), fp4_inv(divisor)));
                }
                fp4 check_num = fp4_sub(tot[combo_count], combo_u[c->tot_combo_backs]);
                fp4 check_div = fp4_sub(x, fp4_pow(z, ZKO_INV_RATE));
                acc = fp4_add(acc, fp4_mul(check_num, fp4_inv(check_div)));
                goal = acc;
            }
            for (size_t r = 0; r < n_rounds; r++) {
                vround_t* vr = &rounds[r];
                size_t per = vr->domain / ZKO_FRI_FOLD;
                size_t quot = pos / per, group = pos % per;
                const uint32_t* data;
                if ((e = mverif_verify(&vr->merkle, &io, group, &data))) FAIL(e);
                fp4 data_ext[ZKO_FRI_FOLD];
                for (int i = 0; i < ZKO_FRI_FOLD; i++) for (int j = 0; j < 4; j++) data_ext[i].c[j] = data[j * ZKO_FRI_FOLD + i];
                if (!fp4_eq(data_ext[quot], goal)) FAIL("FRI round goal mismatch");
                fp inv_wk = fp_pow(fp_rou_rev(log2_ceil(vr->domain)), group);
                goal = fold_eval(data_ext, vr->mix, inv_wk);
                pos = group;
            }
            fp4 x = fp4_from_fp(fp_pow(gen, pos)), fx;
            zko_poly_eval((const uint32_t*)poly_buf, degree, (const uint32_t*)&x, (uint32_t*)&fx);
            if (!fp4_eq(fx, goal)) FAIL("FRI final polynomial mismatch");
        }
    }
    if (io.pos != io.n) FAIL("seal has trailing words");
done:
    for (int g = 0; g < 3; g++) free(mg[g].top);
    free(mcheck.top);
    for (size_t r = 0; r < n_rounds; r++) free(rounds[r].merkle.top);
    free(eval_u); free(combo_u); free(tap_mix_pows); free(poly_buf); free(tot); free(mix_global);
    return ret;
}
