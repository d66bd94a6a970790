/*
 * oracle/zkoracle.h — CPU restatement of the STARK seal path behind risc0_zkp::hal::Hal.
 *
 * TEST INFRASTRUCTURE ONLY (the checker, never the thing shipped or measured as the product).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load libzkoracle.so.
 *
 * PARITY UNPINNED.  /root/reference (risc0/zeth) contains none of this arithmetic: it calls
 * risc0_zkvm::default_prover().prove(env, elf) (/root/reference/crates/host/src/lib.rs:137) and
 * receipt.verify(image_id) (/root/reference/crates/host/src/bin/cli.rs:103).  The algorithm lives in
 * un-vendored crates pinned in /root/reference/Cargo.lock: risc0-zkp 3.0.2 (:5393), risc0-core 3.0.0
 * (:5338), risc0-circuit-rv32im 4.0.2 (:5320), risc0-zkvm 3.0.3 (:5418).  No golden vector, KAT or
 * fixture for this path exists in the reference (its 3 tests are chain-spec equality,
 * /root/reference/crates/chainspec/src/lib.rs:208-221), and no reference binary can be built here
 * (no cargo/rustc/r0vm).  Every function below restates the published upstream algorithm as
 * summarised in SURVEY.md Appendix A and cites the upstream file it follows; the pins we do have are
 * first-principles identities (tests/test_oracle_*.py), an independent verifier restatement
 * (verifier.c) that must accept every seal, and Poseidon2 tables (include/zkh_poseidon2_consts.h) that are
 * the output of the published parameter-generation procedure (tools/gen_poseidon2_consts.py), which
 * reproduces every value of the published instance on record — derived, not yet diffed against consts.rs.
 *
 * All buffers are host arrays of raw Montgomery-form u32 words, column-major exactly like upstream
 * Buffer<T>: element (row r, column c) at c*rows + r.  ExtElem buffers are AoS (4 words per element)
 * unless a comment says "planes".
 */
#ifndef ZKORACLE_H
#define ZKORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* protocol constants — risc0-zkp 3.0.2 src/lib.rs */
#define ZKO_INV_RATE 4
#define ZKO_QUERIES 50
#define ZKO_FRI_FOLD 16
#define ZKO_FRI_FOLD_PO2 4
#define ZKO_FRI_MIN_DEGREE 256
#define ZKO_ZK_CYCLES 1994
#define ZKO_CHECK_SIZE 16
#define ZKO_DIGEST_WORDS 8

/* ---- field helpers exported for the python tests (field.h) ---- */
uint32_t zko_fp_mul(uint32_t a, uint32_t b);
uint32_t zko_fp_encode(uint32_t x);
uint32_t zko_fp_decode(uint32_t a);
uint32_t zko_fp_inv(uint32_t a);
uint32_t zko_rou_fwd(unsigned k);
uint32_t zko_rou_rev(unsigned k);
void zko_fp4_mul(const uint32_t a[4], const uint32_t b[4], uint32_t out[4]);
void zko_fp4_inv(const uint32_t a[4], uint32_t out[4]);

/* ---- Poseidon2 (risc0-zkp src/core/hash/poseidon2/{mod.rs,consts.rs,rng.rs}) ---- */
void zko_poseidon2_set_constants(const uint32_t* rc_canonical /*24*29*/, const uint32_t* diag_canonical /*24*/);
void zko_poseidon2_mix(uint32_t cells[24]);
/* hash_elem_slice over n elems read at in[i*stride] */
void zko_hash_elem_slice(const uint32_t* in, size_t n, size_t stride, uint32_t out[8]);
void zko_hash_pair(const uint32_t a[8], const uint32_t b[8], uint32_t out[8]);

typedef struct { uint32_t cells[24]; uint32_t pool_used; } zko_rng;
void zko_rng_init(zko_rng*);
void zko_rng_mix(zko_rng*, const uint32_t digest[8]);
uint32_t zko_rng_random_elem(zko_rng*);
uint32_t zko_rng_random_bits(zko_rng*, unsigned bits);
void zko_rng_random_ext_elem(zko_rng*, uint32_t out[4]);

/* ---- Hal ops, CpuHal semantics (risc0-zkp src/hal/cpu.rs, src/core/ntt.rs) ---- */
void zko_batch_interpolate_ntt(uint32_t* io, size_t size, size_t count);
void zko_batch_expand_into_evaluate_ntt(uint32_t* out, size_t out_size, const uint32_t* in, size_t in_size,
                                        size_t count, size_t expand_bits);
void zko_batch_bit_reverse(uint32_t* io, size_t size, size_t count);
void zko_zk_shift(uint32_t* io, size_t size, size_t count);
void zko_hash_rows(uint32_t* out_digests, size_t rows, const uint32_t* matrix, size_t matrix_size);
void zko_hash_fold(uint32_t* io_digests, size_t input_size, size_t output_size);
void zko_batch_evaluate_any(const uint32_t* coeffs, size_t coeffs_size, size_t poly_count, const uint32_t* which,
                            const uint32_t* xs /*ext*/, size_t eval_count, uint32_t* out /*ext*/);
void zko_mix_poly_coeffs(uint32_t* out /*ext*/, const uint32_t mix_start[4], const uint32_t mix[4],
                         const uint32_t* in, const uint32_t* combos, size_t input_size, size_t count);
void zko_eltwise_add_elem(uint32_t* out, const uint32_t* a, const uint32_t* b, size_t n);
void zko_eltwise_sum_extelem(uint32_t* out, size_t out_size, const uint32_t* in /*ext*/, size_t in_elems);
void zko_fri_fold(uint32_t* out, size_t out_size, const uint32_t* in, const uint32_t mix[4]);
void zko_gather_sample(uint32_t* dst, const uint32_t* src, size_t idx, size_t size, size_t stride);
void zko_scatter(uint32_t* into, const uint32_t* index, const uint32_t* offsets, const uint32_t* values,
                 size_t n_idx);
void zko_prefix_products(uint32_t* io /*ext*/, size_t n_ext);
/* poly helpers (risc0-zkp src/core/poly.rs) */
void zko_poly_interpolate(uint32_t* out /*ext*/, const uint32_t* xs, const uint32_t* fx, size_t size);
void zko_poly_eval(const uint32_t* coeffs /*ext*/, size_t n, const uint32_t x[4], uint32_t out[4]);
/* in-place synthetic division of an ext poly by (x - z); returns remainder in rem */
void zko_poly_divide(uint32_t* poly /*ext*/, size_t n, const uint32_t z[4], uint32_t rem[4]);

/* ---- circuit description (flat u32 blob, see zeth_amd/circuits/desc.py) ---- */
typedef struct zko_circuit zko_circuit;
const char* zko_circuit_load(const uint32_t* desc, size_t words, zko_circuit** out); /* NULL on success */
void zko_circuit_free(zko_circuit*);
size_t zko_circuit_group_size(const zko_circuit*, unsigned group);
size_t zko_circuit_tap_count(const zko_circuit*);
/* PolyExtStep interpreter at an ext point (risc0-zkp src/adapter.rs PolyExtStepDef::step):
 * u = tap evaluations in tap order; globals[g] = global group g words.  Returns tot. */
void zko_poly_ext(const zko_circuit*, const uint32_t poly_mix[4], const uint32_t* u /*ext per tap*/,
                  const uint32_t* const* globals, uint32_t out[4]);
/* CircuitHal::eval_check, CPU semantics: check (4 planes x 4n) from evaluated groups (each W x 4n). */
void zko_eval_check(const zko_circuit*, uint32_t* check, const uint32_t* const* groups,
                    const uint32_t* const* globals, const uint32_t poly_mix[4], unsigned po2);

/* ---- SYN-AIR witness (definition in DESIGN.md §SYN-AIR; mirrored by the HIP witgen kernels) ---- */
uint32_t zko_syn_cell(uint64_t seed, uint32_t group, uint32_t col, uint32_t row);
/* blinding rows (noise.h): cell = fold mod P of six words of ChaCha12(key; counter = (row, col), nonce = (group, "ZKN1")); every
 * noise_key below is 8 words and REQUIRED (the oracle is deterministic: it never draws from the OS) */
uint32_t zko_noise_cell(const uint32_t* noise_key, uint32_t group, uint32_t col, uint32_t row);
/* RFC 8439 section 2.3 block function with `double_rounds` double rounds (10 = ChaCha20); tail = state words 12..15 */
void zko_chacha_block(const uint32_t* key, const uint32_t* tail, int double_rounds, uint32_t* out);
#define ZKO_SYN_CODE_SEED 0xC0DEC0DE5EEDull
/* code group (wc x n): a function of (circuit, po2, zk_cycles) only — its Merkle root is the control root */
void zko_syn_code(const zko_circuit*, unsigned po2, unsigned zk_cycles, uint32_t* code);
/* fills code (wc x n) and data (wd x n), out global (OUTPUT_SIZE = 4 + n_pub words: s,0,0,0, pub...);
 * pub = n_pub public input words (Montgomery), may be NULL when n_pub == 0 */
void zko_syn_witgen(const zko_circuit*, unsigned po2, unsigned zk_cycles, uint64_t seed, const uint32_t* noise_key,
                    const uint32_t* pub, uint32_t* code, uint32_t* data, uint32_t* out_global);
/* fills accum (wa x n) given data and the mix global (wa words) */
void zko_syn_accum(const zko_circuit*, unsigned po2, unsigned zk_cycles, const uint32_t* noise_key,
                   const uint32_t* data, const uint32_t* mix_global, uint32_t* accum);

/* ---- KECCAK-F witness (zeth_amd/circuits/keccak_f.py; circuit kind 2): every 25 active rows = one keccak-f[1600] ----
 * zko_syn_code / zko_syn_witgen dispatch here for kind 2; `pub` is then the optional input state of the LAST permutation
 * (50 words = 25 lanes, low word first; NULL = seeded like the others) and out_global its output state (100 16-bit limbs). */
uint64_t zko_keccak_lane(uint64_t seed, uint64_t perm, uint32_t lane);
void zko_keccak_code(const zko_circuit*, unsigned po2, unsigned zk_cycles, uint32_t* code);
void zko_keccak_witgen(const zko_circuit*, unsigned po2, unsigned zk_cycles, uint64_t seed, const uint32_t* noise_key,
                       const uint32_t* last_input, uint32_t* code, uint32_t* data, uint32_t* out_global);

/* ---- P2-JOIN witness (zeth_amd/circuits/p2_join.py; circuit kind 3): every 31 active rows = one Poseidon2 permutation,
 * block 0 = hash_pair(left, right).  zko_syn_code / zko_syn_witgen dispatch here for kind 3; `pub` = the two child claims
 * (16 Montgomery words, required), out_global = parent (8) ‖ left (8) ‖ right (8); the seed is unused. */
void zko_p2join_code(const zko_circuit*, unsigned po2, unsigned zk_cycles, uint32_t* code);
void zko_p2join_witgen(const zko_circuit*, unsigned po2, unsigned zk_cycles, const uint32_t* noise_key, const uint32_t* children,
                       uint32_t* code, uint32_t* data, uint32_t* out_global);

/* ---- whole seal: restates SegmentProver::prove + risc0_zkp::prove::Prover ---- */
/* returns malloc'd seal words (caller frees with zko_free); NULL + *err on failure */
uint32_t* zko_prove_segment(const zko_circuit*, unsigned po2, unsigned zk_cycles, uint64_t seed,
                            const uint32_t* noise_key, const uint32_t* pub, size_t* seal_words, const char** err);
/* the same from given code (wc x n) and data (wd x n) traces and the out globals (OUTPUT_SIZE words) */
/* ---- trace-driven witness (preflight.c): the sequential per-cycle machine and the row fill ----
 * records: 4 words per ACTIVE row (2^po2 - zk_cycles rows); ram_image: zko_syn_preflight_ram_words() words (may be NULL) */
size_t zko_syn_preflight_ram_words(void);
void zko_syn_preflight(uint64_t seed, unsigned po2, unsigned zk_cycles, uint32_t* records, uint32_t* ram_image);
void zko_syn_witgen_trace(const zko_circuit*, unsigned po2, unsigned zk_cycles, const uint32_t* noise_key, const uint32_t* records,
                          const uint32_t* ram_image, uint32_t* code, uint32_t* data, uint32_t* out_global);
uint32_t* zko_prove_traces(const zko_circuit*, unsigned po2, unsigned zk_cycles, const uint32_t* noise_key, const uint32_t* code,
                           const uint32_t* data, const uint32_t* out_words, size_t* seal_words, const char** err);
void zko_root_of_code(const zko_circuit*, unsigned po2, const uint32_t* code, uint32_t root[8]);

/* ---- RECURSION circuit (zeth_amd/circuits/recursion.py; circuit kind 4): the code group is a PROGRAM ---- */
/* code (55 x n) of a program blob; NULL on success */
const char* zko_rec_code(const uint32_t* prog, size_t prog_words, uint32_t* code);
/* runs the program's witness schedule on `inputs` (raw Montgomery words: child seals ...), fills code (55 x n), data (72 x n)
 * and out_global (16 words); a message if an assertion of the program fails (the inputs are not what the program verifies) */
const char* zko_rec_witgen(const uint32_t* prog, size_t prog_words, const uint32_t* inputs, size_t n_inputs, const uint32_t* noise_key,
                           uint32_t* code, uint32_t* data, uint32_t* out_global);
/* the copy argument's running products (12 x n) from code, data and the 20 mix words */
void zko_rec_accum(const zko_circuit*, unsigned po2, unsigned zk_cycles, const uint32_t* noise_key, const uint32_t* code,
                   const uint32_t* data, const uint32_t* mix_global, uint32_t* accum);
/* every constraint of the circuit's step list on rows [row_lo, row_hi) of a trace (groups[g]: W_g x n): first failing row or -1 */
long zko_check_rows(const zko_circuit*, unsigned po2, const uint32_t* const* groups, const uint32_t* const* globals, size_t row_lo,
                    size_t row_hi);
/* one Poseidon2 permutation as the 31 trace rows of P2-JOIN / RECURSION: rows[k] = S[24] ‖ Q[24] */
void zko_p2_rows(const uint32_t in[24], uint32_t rows[31][48]);
/* ... and as the 12 rows of a RECURSION block (rows 5 and 6 hold twelve and nine partial rounds) */
void zko_rec_p2_rows(const uint32_t in[24], uint32_t rows[12][48]);

/* Merkle root of the committed code group for (circuit, po2, zk_cycles): the control-ID analogue */
void zko_control_root(const zko_circuit*, unsigned po2, unsigned zk_cycles, uint32_t root[8]);
/* restates risc0_zkp::verify::verify (incl. check_code: the code root must equal control_root).
 * NULL on success, static error string otherwise */
const char* zko_verify_segment(const zko_circuit*, const uint32_t* seal, size_t seal_words,
                               const uint32_t control_root[8]);
/* called by zko_prove_segment with every intermediate buffer of the seal (name, words) when set; NULL disables */
typedef void (*zko_stage_hook)(const char* name, const uint32_t* data, size_t words);
void zko_set_stage_hook(zko_stage_hook hook);
void zko_free(void*);
int zko_num_threads(void);   /* OpenMP threads the oracle will use */
void zko_set_num_threads(int n);   /* the loops stop scaling well before a 2-socket host is full: let the caller pick */

#ifdef __cplusplus
}
#endif
#endif
