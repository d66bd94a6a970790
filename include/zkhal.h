/*
 * zkhal.h — C ABI of libzkhal_mi355x.so: an MI355X (gfx950) implementation of the operator set behind
 * risc0_zkp::hal::{Hal, CircuitHal, Buffer}, plus the per-segment prover that drives it.
 *
 * What this replaces.  risc0/zeth reaches the prover through exactly one call,
 *     default_prover().prove(env, elf)            /root/reference/crates/host/src/lib.rs:137
 * (imports at lib.rs:26; segment size chosen at lib.rs:132-135; result checked by receipt.verify at
 * /root/reference/crates/host/src/bin/cli.rs:103-107).  Below that call sits the un-vendored
 * risc0-zkp 3.0.2 (/root/reference/Cargo.lock:5393): `prove::Prover<H: Hal>` issues every op declared
 * here through `hal::Hal` (src/hal/mod.rs) — upstream backends: CpuHal (src/hal/cpu.rs), CudaHal
 * (src/hal/cuda.rs + risc0-sys kernels).  Each entry point below names the trait method it stands in for.
 * A Rust `impl Hal for HipHal` binds these 1:1 (INTEGRATION.md shows the extern block).
 *
 * Conventions (same as upstream's risc0-sys C exports):
 *   - every function returns `const char*`: NULL on success, otherwise a heap error string that the
 *     caller releases with zkh_free_error();
 *   - element words are raw Montgomery-form BabyBear u32 exactly as upstream stores `Elem` in memory and in
 *     seals; Elem = 1 word, ExtElem = 4 words (AoS), Digest = 8 words.  Every op expects REDUCED words (< P) and
 *     produces reduced words; upstream's Elem::INVALID marker (0xffffffff) is not an input (upstream asserts on it);
 *   - matrices are column-major like upstream Buffer<T>: element (row r, column c) at c*rows + r;
 *   - a zkh_ctx is one GPU + one HIP stream, driven by one host thread at a time (upstream HALs are driven
 *     by a single prover thread).  Ops are enqueued in order on the ctx stream and are asynchronous;
 *     zkh_read / zkh_sync are the only host-visible sync points;
 *   - zkh_buf handles are reference counted views {allocation, word offset, word length}; slices share the
 *     allocation.  Host pointers are borrowed for the duration of the call only.
 *   - There is NO CPU fallback in this library: without a HIP device every call fails.
 */
#ifndef ZKHAL_H
#define ZKHAL_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct zkh_ctx zkh_ctx;
typedef struct zkh_buf zkh_buf;
typedef struct zkh_circuit zkh_circuit;
typedef struct zkh_prover zkh_prover;

/* protocol constants — risc0-zkp 3.0.2 src/lib.rs */
#define ZKH_INV_RATE 4
#define ZKH_QUERIES 50
#define ZKH_FRI_FOLD 16
#define ZKH_FRI_MIN_DEGREE 256
#define ZKH_ZK_CYCLES 1994
#define ZKH_CHECK_SIZE 16
#define ZKH_EXT_SIZE 4
#define ZKH_DIGEST_WORDS 8

void zkh_free_error(const char* err);
/* library/ABI version, the gfx arch the kernels were compiled for ("gfx950"), and where the shipped Poseidon2 tables come
 * from: "poseidon2_consts=derived" (output of the published parameter-generation procedure, tools/gen_poseidon2_consts.py,
 * which reproduces every value of the published instance on record; not yet compared with upstream's consts.rs itself),
 * "=upstream" once that comparison has been made, "=placeholder" for filler tables (digests cannot match upstream's) */
const char* zkh_version(void);

/* ---- context: CudaHal::new / HalPair (hal/cuda.rs) ---- */
/* hash_suite must be "poseidon2" (the suite zeth's default ProverOpts selects). */
const char* zkh_ctx_create(int device_ordinal, const char* hash_suite, zkh_ctx** out);
void zkh_ctx_destroy(zkh_ctx*);
const char* zkh_sync(zkh_ctx*);
/* Device memory of this context: bytes held by live buffers, bytes cached by the stream-ordered free list (blocks are
 * recycled by exact size, so a host that proves many segment sizes accumulates cache), and the live high-water mark.
 * zkh_ctx_trim drains the stream and returns the cached blocks to the driver; allocation does the same on its own
 * before reporting out-of-memory. */
void zkh_ctx_memory(const zkh_ctx*, size_t* live_bytes, size_t* cached_bytes, size_t* peak_live_bytes);
const char* zkh_ctx_trim(zkh_ctx*);
/* raw hipStream_t of the context (for callers that interleave their own work) */
void* zkh_ctx_stream(zkh_ctx*);
/* Replace the Poseidon2 tables (canonical residues): rc[24*29], diag[24] — consts.rs as data. */
const char* zkh_poseidon2_set_constants(zkh_ctx*, const uint32_t* rc, const uint32_t* diag);

/* ---- Buffer<T>: alloc_* / copy_from_* / slice / view / get_at (hal/mod.rs trait Buffer) ---- */
const char* zkh_alloc(zkh_ctx*, const char* name, size_t n_words, int zero, zkh_buf** out);
const char* zkh_copy_from(zkh_ctx*, const char* name, const uint32_t* host, size_t n_words, zkh_buf** out);
/* wrap device memory owned by the caller (e.g. a torch tensor's data_ptr); never freed by the library */
const char* zkh_wrap(zkh_ctx*, void* device_ptr, size_t n_words, zkh_buf** out);
const char* zkh_slice(zkh_buf*, size_t off_words, size_t n_words, zkh_buf** out);
void zkh_retain(zkh_buf*);
void zkh_release(zkh_buf*);
size_t zkh_size(const zkh_buf*);
void* zkh_device_ptr(const zkh_buf*);
/* view(): synchronises the stream, then D2H */
const char* zkh_read(zkh_ctx*, const zkh_buf*, uint32_t* host, size_t off_words, size_t n_words);
/* view_mut()/copy: H2D ordered on the stream */
const char* zkh_write(zkh_ctx*, zkh_buf*, const uint32_t* host, size_t off_words, size_t n_words);
/* Witness ingress for hosts that run preflight + witgen on the CPU (upstream: SegmentProver steps 1-2 before
 * Prover::commit_group; 0.94 GB of code + data per po2-20 SYN-A segment): pinned host memory the caller fills in place,
 * and an upload that is only ENQUEUED on the context's stream (no host sync; the block must not be rewritten before the
 * next zkh_sync / zkh_read of this context).  The DMA overlaps kernels of other contexts on the same GPU. */
const char* zkh_host_alloc(zkh_ctx*, size_t n_words, uint32_t** host);
void zkh_host_free(zkh_ctx*, uint32_t* host);
const char* zkh_write_async(zkh_ctx*, zkh_buf*, const uint32_t* pinned_host, size_t off_words, size_t n_words);

/* ---- trait Hal ops (hal/mod.rs); semantics = CpuHal (hal/cpu.rs) ---- */
/* Hal::batch_interpolate_ntt(io, count): per column inverse NTT, natural in -> bit-reversed coeffs, * n^-1 */
const char* zkh_batch_interpolate_ntt(zkh_ctx*, zkh_buf* io, size_t count);
/* Hal::batch_expand_into_evaluate_ntt(out, in, count, expand_bits) */
const char* zkh_batch_expand_into_evaluate_ntt(zkh_ctx*, zkh_buf* out, const zkh_buf* in, size_t count,
                                               size_t expand_bits);
/* Hal::batch_bit_reverse(io, count) */
const char* zkh_batch_bit_reverse(zkh_ctx*, zkh_buf* io, size_t count);
/* Hal::zk_shift(io, count): io[c][i] *= 3^bitrev(i) */
const char* zkh_zk_shift(zkh_ctx*, zkh_buf* io, size_t count);
/* Fused Prover::commit_group prefix: interpolate_ntt + zk_shift in one pass over HBM (same result as the
 * two calls above in sequence). */
const char* zkh_batch_interpolate_ntt_zk_shift(zkh_ctx*, zkh_buf* io, size_t count);
/* Out-of-place variant that also absorbs commit_group's eltwise_copy_elem: out = iNTT(in) [* 3^bitrev(i)];
 * `in` (the witness) is left untouched. */
const char* zkh_batch_interpolate_ntt_from(zkh_ctx*, zkh_buf* out, const zkh_buf* in, size_t count, int zk_shift);
/* Hal::hash_rows(output, matrix): rows = output digests; leaf r = Poseidon2 sponge over matrix[c*rows + r] */
const char* zkh_hash_rows(zkh_ctx*, zkh_buf* out_digests, const zkh_buf* matrix);
/* Hal::hash_fold(io, input_size, output_size): io[out+i] = H(io[in+2i], io[in+2i+1]) */
const char* zkh_hash_fold(zkh_ctx*, zkh_buf* io_digests, size_t input_size, size_t output_size);
/* Fused MerkleTreeProver::new tail: every hash_fold layer from `rows` leaves down to the root.
 * nodes has 2*rows digests, leaves already at [rows, 2*rows). */
const char* zkh_merkle_fold_all(zkh_ctx*, zkh_buf* nodes, size_t rows);
/* MerkleTreeProver::new as one call: nodes[rows .. 2 rows) = hash_rows(matrix), then every layer above down to the root at nodes[1];
 * digests are identical to zkh_hash_rows + zkh_merkle_fold_all. */
const char* zkh_merkle_build(zkh_ctx*, zkh_buf* nodes, const zkh_buf* matrix, size_t rows);
/* The bare permutation (risc0_zkp::core::hash::poseidon2::poseidon2_mix): `count` states of 24 Montgomery words each,
 * in place — on the device with the context's tables, or on the host (rc / diag canonical residues, NULL = the shipped
 * tables).  What the published known-answer vector of the instance is checked against (tests/golden/poseidon2_kat.json). */
const char* zkh_poseidon2_mix(zkh_ctx*, zkh_buf* states, size_t count);
const char* zkh_poseidon2_mix_host(const uint32_t* rc, const uint32_t* diag, uint32_t* states, size_t count);
/* Hal::batch_evaluate_any(coeffs, poly_count, which, xs, out): out[k] = sum_j coeffs[which[k]][j] xs[k]^j */
const char* zkh_batch_evaluate_any(zkh_ctx*, const zkh_buf* coeffs, size_t poly_count, const zkh_buf* which,
                                   const zkh_buf* xs, zkh_buf* out);
/* Same, for coefficient columns still in the bit-reversed order batch_interpolate_ntt produced (position p holds
 * coefficient bitrev(p)); column length a power of two >= 2^14.  Lets PolyGroup::new skip its W x n batch_bit_reverse. */
const char* zkh_batch_evaluate_any_bitrev(zkh_ctx*, const zkh_buf* coeffs, size_t poly_count, const zkh_buf* which,
                                          const zkh_buf* xs, zkh_buf* out);
/* batch_bit_reverse for ExtElem (AoS) columns: used on the few combo polynomials instead of on every coefficient column */
const char* zkh_batch_bit_reverse_extelem(zkh_ctx*, zkh_buf* io_ext, size_t count);
/* Hal::mix_poly_coeffs(output, mix_start, mix, input, combos, input_size, count) */
const char* zkh_mix_poly_coeffs(zkh_ctx*, zkh_buf* out, const uint32_t mix_start[4], const uint32_t mix[4],
                                const zkh_buf* in, const zkh_buf* combos, size_t input_size, size_t count);
/* Hal::combos_prepare: combos[combo_of_reg*cycles + i] -= mix^reg * coeff_u[...]  (flattened on the host into
 * (position, value) pairs: combos[pos[k]] -= val[k]) */
const char* zkh_combos_prepare(zkh_ctx*, zkh_buf* combos, const uint32_t* pos, const uint32_t* vals_ext,
                               size_t n_entries);
/* Hal::combos_prepare with upstream's own argument list (combos, coeff_u, combo_count, cycles, regs_count, reg_sizes,
 * reg_combo_ids, mix), all operands device buffers as in `impl Hal`:
 *   cur = 1; for r < regs_count: combos[cycles*reg_combo_ids[r] + i] -= cur * coeff_u[pos + i] (i < reg_sizes[r]); cur *= mix; pos += reg_sizes[r];
 *   then ZKH_CHECK_SIZE times: combos[cycles*combo_count] -= cur * coeff_u[pos]; pos += 1; cur *= mix.
 * combos: (combo_count + 1) x cycles ExtElems; registers of any size 1 .. cycles.  Nothing is read back: the register list is
 * VALIDATED ON THE DEVICE before it is used (sizes, combo ids, and that coeff_u holds sum(sizes) + ZKH_CHECK_SIZE coefficients); an
 * inconsistent list leaves combos untouched — never an out-of-bounds read — and is REPORTED BY THE NEXT zkh_sync / zkh_read of the
 * context (the error names the register).  (zkh_combos_prepare above is the host-flattened form the in-library prover uses.) */
const char* zkh_combos_prepare_regs(zkh_ctx*, zkh_buf* combos, const zkh_buf* coeff_u, size_t combo_count, size_t cycles,
                                    size_t regs_count, const zkh_buf* reg_sizes, const zkh_buf* reg_combo_ids,
                                    const uint32_t mix[4]);
/* Hal::combos_divide: synthetic division of combo polynomial `combo` (cycles ExtElems at combos[combo*cycles..])
 * by (x - pt) for every pt in pts; remainders (must be 0) are written to rem_out (n_pts ExtElems, device). */
const char* zkh_combos_divide(zkh_ctx*, zkh_buf* combos, size_t combo, size_t cycles, const uint32_t* pts_ext,
                              size_t n_pts, zkh_buf* rem_out);
/* The same for every combo of the buffer in one call: combo i (i < n_combos) is divided by (x - pt) for each of its points
 * pts_ext[4*pts_begin[i] .. 4*pts_begin[i+1]); remainders go to rem_out in point order.  The r-th divisions of all
 * combos share their kernel launches (they are independent; only the divisions of one combo are sequential). */
const char* zkh_combos_divide_all(zkh_ctx*, zkh_buf* combos, size_t cycles, size_t n_combos, const uint32_t* pts_ext,
                                  const uint32_t* pts_begin, zkh_buf* rem_out);
/* Hal::eltwise_add_elem / eltwise_copy_elem / eltwise_zeroize_elem / eltwise_sum_extelem */
const char* zkh_eltwise_add_elem(zkh_ctx*, zkh_buf* out, const zkh_buf* a, const zkh_buf* b);
const char* zkh_eltwise_copy_elem(zkh_ctx*, zkh_buf* out, const zkh_buf* in);
const char* zkh_eltwise_zeroize_elem(zkh_ctx*, zkh_buf* io);   /* INVALID (0xffffffff) -> 0 */
const char* zkh_eltwise_sum_extelem(zkh_ctx*, zkh_buf* out_elem, const zkh_buf* in_ext);
/* Hal::fri_fold(output, input, mix) */
const char* zkh_fri_fold(zkh_ctx*, zkh_buf* out, const zkh_buf* in, const uint32_t mix[4]);
/* Hal::gather_sample(dst, src, idx, size, stride): dst[g] = src[g*stride + idx] */
const char* zkh_gather_sample(zkh_ctx*, zkh_buf* dst, const zkh_buf* src, size_t idx, size_t size, size_t stride);
/* Hal::scatter(into, index, offsets, values) */
const char* zkh_scatter(zkh_ctx*, zkh_buf* into, const uint32_t* index, const uint32_t* offsets,
                        const uint32_t* values, size_t n_idx, size_t n_val);
/* Hal::prefix_products(io): io[i] *= io[i-1] over ExtElems */
const char* zkh_prefix_products(zkh_ctx*, zkh_buf* io_ext);
/* batched MerkleTreeProver::prove for many indices at once (query phase): for each idx writes
 * (cols column words, then the path digests down to the top layer) consecutively into `out` (device).
 * words per query = cols + 8*(log2(rows) - top_layer). */
const char* zkh_merkle_open(zkh_ctx*, const zkh_buf* matrix, const zkh_buf* nodes, size_t rows, size_t cols,
                            const uint32_t* idx, size_t n_idx, zkh_buf* out);

/* ---- CircuitHal (risc0-circuit-rv32im 4.0.2 src/prove/hal/; circuit is data: zeth_amd/circuits/desc.py) ---- */
const char* zkh_circuit_load(zkh_ctx*, const uint32_t* desc, size_t n_words, zkh_circuit** out);
void zkh_circuit_destroy(zkh_circuit*);
/* 2 if a code object was attached at run time (below), 1 if a build-time generated straight-line eval_check
 * kernel matches this desc, 0 if the on-device step interpreter will be used. */
int zkh_circuit_has_compiled_kernel(const zkh_circuit*);
/* Attach a gfx950 code object (ELF or clang offload bundle, borrowed for the call) that holds
 *   extern "C" __global__ void <kernel_name>(zkh::EvalCheckArgs)        -- csrc/circuit.h
 * generated for this circuit's PolyExtStep list; zkh_eval_check then launches it instead of the interpreter.
 * Replaces upstream's per-circuit machine-generated eval_check kernel (risc0-circuit-rv32im-sys 4.0.2
 * kernels/…/eval_check.cu, un-vendored: /root/reference/Cargo.lock:5320) for circuits that arrive as data.
 * Not thread-safe against a concurrent zkh_eval_check on the same circuit. */
const char* zkh_circuit_attach_code_object(zkh_circuit*, const void* image, size_t len, const char* kernel_name);
/* A constraint system of realistic size is generated as n_parts kernels over disjoint constraint ranges (upstream splits
 * its generated eval_check over many translation units for the same reason); each part arrives as its own code object.
 * zkh_eval_check uses them once all n_parts are attached: part 0 writes `check`, the others add their share.
 * A code object may export `<kernel_name>_exps` (device data: {count, e_0, e_1, ...}): the kernel then reads slot k of
 * EvalCheckArgs::mix_pows as poly_mix^e_k — its powers gathered into the order its code touches them — and zkh_eval_check
 * builds that table per call (all parts of a circuit export it, or none: zkh_eval_check refuses a mixed set). */
const char* zkh_circuit_attach_code_object_part(zkh_circuit*, const void* image, size_t len, const char* kernel_name,
                                                size_t part, size_t n_parts);
/* number of kernels zkh_eval_check will launch for this circuit (0: step interpreter) */
size_t zkh_circuit_compiled_parts(const zkh_circuit*);
/* CircuitHal::eval_check(check, groups, globals, poly_mix, po2, steps).  groups = evaluated accum, code, data
 * (each W x 4n; n_groups must be 3); globals = out, mix (n_globals must be 2); steps = 2^po2 like upstream.
 * use_interpreter != 0 forces the generic interpreter kernel. */
const char* zkh_eval_check(zkh_ctx*, const zkh_circuit*, zkh_buf* check, const zkh_buf* const* groups, size_t n_groups,
                           const zkh_buf* const* globals, size_t n_globals, const uint32_t poly_mix[4], size_t po2,
                           size_t steps, int use_interpreter);

/* ---- built-in witness generators on the device, by circuit kind (desc word 13) ----
 *   kind 1 SYN-AIR   stands in for risc0-circuit-rv32im's witgen (declared synthetic)
 *   kind 2 KECCAK-F  every 25 active rows are one real keccak-f[1600] permutation (zeth_amd/circuits/keccak_f.py; stands
 *                    in for risc0-circuit-keccak 4.0.2, /root/reference/Cargo.lock:5289).  For this kind `pub` is the optional
 *                    input state of the LAST permutation (25 lanes = 50 words, low word first; NULL = seeded like the others)
 *                    and out_global receives its output state as 100 16-bit limbs (lane l, limb j at 4 l + j).
 *   kind 3 P2-JOIN   every 31 active rows are one Poseidon2 permutation; block 0 = hash_pair(left, right) (zeth_amd/circuits/
 *                    p2_join.py; stands in for the in-circuit hashing of risc0-circuit-recursion 4.0.2, Cargo.lock:5305).
 *                    `pub` = the two child claims (16 words, required); out_global = parent (8) ‖ left (8) ‖ right (8). */
/* BLINDING ROWS.  The last zk_cycles rows of every data / accum column are zero-knowledge blinding noise.  Upstream draws each
 * cell from the OS RNG (`Elem::random(&mut rng)`); here the randomness enters once per call as a 256-bit key — `noise_key`, 8
 * words — and every cell is ChaCha12(key; counter = (row, column), nonce = (group, "ZKN1")) folded mod P the way upstream's
 * Elem::random folds six u32 draws (csrc/noise.h; CPU twin oracle/noise.h).  noise_key == NULL or all-zero: 256 fresh bits from
 * getrandom() for this call — the product default; the call FAILS if the OS cannot deliver.  A fixed key makes seals reproducible
 * (tests, bench.py, golden fixtures).
 * The two host functions below expose the generator itself (no GPU): one blinding cell, and the underlying RFC 8439 block function
 * with `double_rounds` double rounds (10 = ChaCha20: what the RFC's test vector pins; the stream uses 6). */
uint32_t zkh_noise_cell_host(const uint32_t noise_key[8], uint32_t group, uint32_t column, uint32_t row);
void zkh_chacha_block_host(const uint32_t key[8], const uint32_t counter_nonce[4], int double_rounds, uint32_t out[16]);
/* code group only: a function of (circuit, po2, zk_cycles) — what the control root commits to */
const char* zkh_syn_code(zkh_ctx*, const zkh_circuit*, size_t po2, size_t zk_cycles, zkh_buf* code);
/* pub: OUTPUT_SIZE - 4 public input words (Montgomery; NULL if the circuit has none); out_global: OUTPUT_SIZE words;
 * code may be NULL for kinds 1 and 2 (the caller already holds this size's code group, e.g. resident in its prover) */
const char* zkh_syn_witgen(zkh_ctx*, const zkh_circuit*, size_t po2, size_t zk_cycles, uint64_t seed,
                           const uint32_t noise_key[8], const uint32_t* pub, zkh_buf* code, zkh_buf* data, uint32_t* out_global);
const char* zkh_syn_accum(zkh_ctx*, const zkh_circuit*, size_t po2, size_t zk_cycles, const uint32_t noise_key[8],
                          const zkh_buf* data, const uint32_t* mix_global, zkh_buf* accum);
/* Chained sessions (SYN-C: a kind-1 circuit with ONE public input = the segment's pre-state; out = (post, 0, 0, 0, pre)): what each
 * of n segments (seed, po2) adds to the running state — its post-state when started from state 0 — in one launch.  The executor's
 * part: with these a host fixes every segment's pre-state before any segment is proven (upstream's executor fixes
 * ReceiptClaim.pre / .post the same way), which keeps the segments independent for the provers. */
const char* zkh_syn_chain_contributions(zkh_ctx*, const zkh_circuit*, const uint64_t* seeds, const uint32_t* po2s, size_t n,
                                        size_t zk_cycles, uint32_t* contributions);

/* ---- trace-driven witness (csrc/preflight.hip).  Upstream: the rv32im preflight replays a segment's cycles
 * on one host thread into per-cycle records, witgen kernels fill the trace rows from them (one lane per cycle) and Hal::scatter
 * places the preloaded memory image (risc0-circuit-rv32im 4.0.2 prove/witgen, un-vendored: /root/reference/Cargo.lock:5320; the
 * guest input enters at /root/reference/crates/host/src/lib.rs:132-136).  Here the same pipeline for SYN-AIR circuits (kind 1,
 * no public inputs) with a stand-in machine: zkh_syn_preflight is the SEQUENTIAL host producer — 4 words (16 bytes) per active
 * cycle (value, operand, pc | op | rd | address, running state digest), all valid Elem words, plus the 1024-word RAM image before
 * the first cycle — and zkh_syn_witgen_trace expands records
 * that are already on the device (zkh_write_async from pinned memory: 16.7 MB per po2-20 segment instead of the 0.94 GB full
 * trace) into the data group: one row-fill launch, the running-sum scan, the preload through zkh_scatter. ---- */
size_t zkh_syn_preflight_ram_words(void);
const char* zkh_syn_preflight(uint64_t seed, size_t po2, size_t zk_cycles, uint32_t* records, uint32_t* ram_image,
                              double* cpu_seconds);
const char* zkh_syn_witgen_trace(zkh_ctx*, const zkh_circuit*, size_t po2, size_t zk_cycles, const uint32_t noise_key[8],
                                 const zkh_buf* records, const uint32_t* ram_image, zkh_buf* code, zkh_buf* data,
                                 uint32_t* out_global);

/* ---- segment prover: SegmentProver::prove_segment + risc0_zkp::prove::Prover ---- */
const char* zkh_prover_create(zkh_ctx*, const zkh_circuit*, zkh_prover** out);
void zkh_prover_destroy(zkh_prover*);   /* before zkh_ctx_destroy: a prover may hold device buffers (zkh_prover_cache_code) */
/* Seal one segment of a SYN-AIR-family circuit (kind 1: the accum witness generator is zkh_syn_accum) whose code/data
 * traces are already resident in HBM (W x 2^po2 each); out_global has OUTPUT_SIZE words.  On success *seal is a
 * malloc'd word array (release with zkh_free_seal). */
const char* zkh_prove_segment(zkh_prover*, size_t po2, size_t zk_cycles, const uint32_t noise_key[8], const zkh_buf* code,
                              const zkh_buf* data, const uint32_t* out_global, uint32_t** seal,
                              size_t* seal_words);
void zkh_free_seal(uint32_t* seal);
/* The same seal in the two halves upstream's SegmentProver drives `Prover` in, for ANY circuit and for traces the caller
 * produced itself (uploaded with zkh_copy_from / zkh_write_async):
 *   zkh_prove_begin : header, Prover::commit_group(code), commit_group(data); returns the accum mix challenges
 *                     (global_size[mix] words into mix_global, may be NULL) drawn from the transcript;
 *   (caller)        : fills the accum trace (W_accum x 2^po2) from data + mix — CircuitHal::accumulate, circuit-specific;
 *   zkh_prove_finish: commit_group(accum) + Prover::finalize (eval_check, DEEP, FRI, queries).  Consumes the job whether
 *                     or not it succeeds; zkh_prove_abort drops a job that will not be finished. */
typedef struct zkh_seal_job zkh_seal_job;
/* The code (control) group is a function of (circuit, po2) alone — it is what the control ID commits to — yet upstream's
 * SegmentProver re-commits it for every segment.  A prover that seals many segments of one size can keep that group's
 * committed form (coefficients, 4n evaluations, Merkle nodes: 0.6 GB at po2 20, W_code 16) resident in HBM:
 * zkh_prover_cache_code commits `code` once for this po2 (replacing an earlier entry); afterwards zkh_prove_begin /
 * zkh_prove_segment accept code == NULL for that po2 and share the resident group read-only.  Seals are byte-identical
 * to the ones made from the same code trace.  The caller vouches that the trace is the circuit's code trace for that
 * size (the verifier still checks the root against the control root).  zkh_prover_drop_code_cache releases the entries. */
const char* zkh_prover_cache_code(zkh_prover*, size_t po2, const zkh_buf* code);
void zkh_prover_drop_code_cache(zkh_prover*);
/* Root of the resident code group of this size.  The entry is keyed by po2 alone although a code trace also depends on
 * zk_cycles: a host that changes zk_cycles must re-cache, and can compare this root with the control root it expects. */
const char* zkh_prover_cached_code_root(zkh_prover*, size_t po2, uint32_t root[8]);
const char* zkh_prove_begin(zkh_prover*, size_t po2, const zkh_buf* code, const zkh_buf* data, const uint32_t* out_global,
                            zkh_seal_job** job, uint32_t* mix_global);
const char* zkh_prove_finish(zkh_seal_job*, const zkh_buf* accum, uint32_t** seal, size_t* seal_words);
void zkh_prove_abort(zkh_seal_job*);
/* Control root = Merkle root of the committed code group (the control-ID analogue: risc0-zkp verify/mod.rs check_code).
 * zkh_code_root commits a caller-supplied code trace; zkh_syn_control_root generates SYN-AIR's for (po2, zk_cycles). */
const char* zkh_code_root(zkh_prover*, const zkh_buf* code, size_t po2, uint32_t root[8]);
const char* zkh_syn_control_root(zkh_prover*, size_t po2, size_t zk_cycles, uint32_t root[8]);

/* ---- verifier: risc0_zkp::verify::verify — what `receipt.verify(image_id)` (cli.rs:103) runs per segment ----
 * Pure host code, no GPU needed: the circuit may be loaded with ctx == NULL (zkh_circuit_load(NULL, ...)).
 * control_root: the expected code commitment for (circuit, po2) — REQUIRED (check_code: a seal whose code group is
 * anything else, e.g. all-zero selectors, is rejected).  rc / diag: canonical Poseidon2 tables (24*29 / 24 words) or
 * NULL for the shipped ones.  NULL = seal accepted. */
const char* zkh_verify_segment(const zkh_circuit*, const uint32_t* seal, size_t seal_words, const uint32_t control_root[8],
                               const uint32_t* rc, const uint32_t* diag);

/* Claim digest of a sealed segment = Poseidon2(out globals, po2, control root): what a join receipt commits to for each
 * of its two children (zeth_amd/host.py; upstream: ReceiptClaim digests inside risc0-zkvm's lift/join).  Host only. */
const char* zkh_receipt_claim(const zkh_circuit*, const uint32_t* seal, size_t seal_words, const uint32_t control_root[8],
                              const uint32_t* rc, const uint32_t* diag, uint32_t claim[8]);

/* Receipt container: a versioned little-endian word envelope around a seal carrying what upstream's SegmentReceipt does
 * (circuit hash, po2, hash function, segment index, control root, claim digest, seal, checksum; layout: csrc/verifier.hip).
 * Stands in for the bincode `SegmentReceipt` inside `ProveInfo` (risc0-zkvm 3.0.3, decoded by zeth at
 * /root/reference/crates/host/src/lib.rs:137).  Host only.  *blob is malloc'd: release with zkh_free_seal. */
const char* zkh_receipt_encode(const zkh_circuit*, const uint32_t* seal, size_t seal_words, uint32_t segment_index,
                               const uint32_t control_root[8], uint32_t** blob, size_t* blob_words);
/* Parse + integrity-check (checksum, version, hash-suite, and with a circuit: desc hash, output size, po2, claim digest).
 * The checksum is FNV-1a: it detects corruption, not forgery; with circuit == NULL only the envelope is checked and NOTHING
 * is authenticated.  Authenticity comes from zkh_verify_segment on the seal against the EXPECTED control root, never from here.
 * info receives header words [0, 26); the seal sits at blob + *seal_offset (info[9] words).  Does NOT verify the seal. */
const char* zkh_receipt_decode(const zkh_circuit*, const uint32_t* blob, size_t blob_words, uint32_t info[26],
                               size_t* seal_offset);

/* ---- RECURSION circuit (kind 4): lift / join as programs of an in-circuit STARK verifier ----
 * Replaces risc0-circuit-recursion 4.0.2 src/prove/{mod.rs Prover::run, program.rs} + its witness generator (un-vendored:
 * /root/reference/Cargo.lock:5305), reached from default_prover().prove (/root/reference/crates/host/src/lib.rs:137) once per
 * lift and per join (BASELINE.json config 5).  The circuit: zeth_amd/circuits/recursion.py (six Fp4 wires + one gate per row,
 * Poseidon2 blocks, a copy argument in the accum group); a PROGRAM = the code group, produced by
 * zeth_amd/circuits/rec_verify.py (build_lift / build_lift2 / build_join = this library's verifier restated gate by gate) as a u32 blob.
 * zkh_rec_program_load validates the blob, sorts its witness schedule into dependency levels, uploads it, generates the code
 * group and commits it (resident); `circuit` = the RECURSION description loaded on the same context.  zkh_rec_prove runs the
 * program on `inputs` (raw Montgomery words: the child seal(s) and the program's other witness words), which FAILS unless every
 * assertion of the in-circuit verifier holds, and seals the trace; out_global (16 words) = claim (8) ‖ allowed-programs root.
 * A program handle owns its witness buffers and the hipGraph of its schedule: one zkh_rec_witgen / zkh_rec_prove at a time per
 * handle (every lane loads its own). */
typedef struct zkh_rec_program zkh_rec_program;
/* The circuit DESCRIPTIONS compiled into the library (the shipped circuits of zeth_amd/circuits/: syn_a, syn_small, syn_tiny, syn_join,
 * syn_chain, syn_heavy, keccak_f, p2_join, recursion), for hosts without Python; the words are static data (do not free). */
size_t zkh_shipped_circuit_count(void);
const char* zkh_shipped_circuit_name(size_t i);                       /* NULL past the end */
const char* zkh_shipped_circuit_desc(const char* name, const uint32_t** words, size_t* n_words);
/* The program BUILDER on the host, in C++ (csrc/rec_builder.hip; no GPU, no Python): what zeth_amd/circuits/rec_verify.py +
 * recursion.py Program.finish produce, word for word (upstream ships its lift / join programs as precompiled .zkr files; a host
 * without Python builds them here).  kind 0 lift: child_desc = the SEGMENT circuit, po2s[0], control_roots = the 8 words of
 * its control root at that size as the library hands them out (zkh_syn_control_root: Montgomery form); kind 2 lift2: po2s[0..2),
 * control_roots = 16 words (left, right); kind 1 join /
 * kind 3 join3: child_desc = the RECURSION circuit, po2s[0..2) / [0..3), control_roots = NULL; kind 4 union (two receipts of any
 * claims -> the digest of the sorted pair; inputs: seal, membership path per child, then the swap bit) / kind 5 resolve (the
 * conditional receipt, opened, bound to its assumption receipt): child_desc = the RECURSION circuit, po2s[0..2).  The program is placed at the
 * smallest po2 that holds it (blob[2]); *blob is malloc'd: release with zkh_free_seal. */
const char* zkh_rec_build_program(uint32_t kind, const uint32_t* child_desc, size_t child_desc_words, const uint32_t* po2s,
                                  const uint32_t* control_roots, uint32_t zk_cycles, uint32_t** blob, size_t* words);
const char* zkh_rec_program_load(zkh_ctx*, const zkh_circuit* circuit, const uint32_t* blob, size_t words, zkh_rec_program** out);
void zkh_rec_program_destroy(zkh_rec_program*);
/* root: the program's control root (Merkle root of its code group); info: po2, zk_cycles, input words, permutations, gates,
 * witness ops, dependency levels, variables */
const char* zkh_rec_program_info(const zkh_rec_program*, uint32_t root[8], uint32_t info[8]);
/* > 0: the witness schedule is replayed as a hipGraph (opt-in: ZKH_REC_GRAPH=1 when the program is loaded); the value is the
 * number of steps of its launch plan (runs of narrow levels in one persistent workgroup + single wide levels).  0: the plan is
 * launched kernel by kernel (the default: measured equal, and profilers cope with it). */
int zkh_rec_program_has_graph(const zkh_rec_program*);
const char* zkh_rec_code(const zkh_rec_program*, zkh_buf* code /* 58 x 2^po2 */);
const char* zkh_rec_witgen(const zkh_rec_program*, const uint32_t* inputs, size_t n_inputs, const uint32_t noise_key[8],
                           zkh_buf* data /* 72 x 2^po2 */, uint32_t out_global[16]);
const char* zkh_rec_accum(const zkh_rec_program*, const uint32_t noise_key[8], const zkh_buf* data, const uint32_t* mix_global /* 20 */,
                          zkh_buf* accum /* 12 x 2^po2 */);
const char* zkh_rec_prove(const zkh_rec_program*, const uint32_t* inputs, size_t n_inputs, const uint32_t noise_key[8],
                          uint32_t out_global[16] /* may be NULL */, uint32_t** seal, size_t* seal_words);

/* ---- session executor: ProverServer::prove_session / ProverImpl::{prove_segment, join} (risc0-zkvm 3.0.3, un-vendored:
 * /root/reference/Cargo.lock:5418) — what default_prover().prove(env, elf) runs once the executor has cut the guest's run
 * into segments (/root/reference/crates/host/src/lib.rs:137), as ONE call: every segment sealed on G devices x K lanes through
 * one shared work index (segments are independent: no exchange between devices), receipts in index order, optionally folded
 * through the P2-JOIN tree to one root receipt; zkh_session_verify is receipt.verify (cli.rs:103) for the result.
 * A session owns its lanes (one zkh_ctx + circuit + prover each) and runs one zkh_session_prove at a time. ---- */
typedef struct zkh_session zkh_session;
typedef struct {
    uint32_t po2;                 /* segment size 2^po2 cycles */
    uint64_t seed;                /* built-in witness generators (circuit kind 1..3): witness seed */
    uint32_t noise_key[8];        /* blinding rows: 256-bit ChaCha12 key; all-zero = fresh OS randomness per segment (the product default, like upstream) */
    const uint32_t* pub;          /* public inputs of the built-in generators (see zkh_syn_witgen), may be NULL */
    size_t n_pub;
    /* caller-produced traces instead (CPU preflight + witgen, upstream's flow): W_code x 2^po2, W_data x 2^po2 words and
     * OUTPUT_SIZE out globals; accum comes from the session's accumulate callback (built-in for kinds 1..3) */
    const uint32_t* host_code;
    const uint32_t* host_data;
    const uint32_t* out_global;
} zkh_segment;
typedef struct {
    size_t n_segments;
    uint32_t** seals;             /* n_segments malloc'd seals in index order (the composite receipt) */
    size_t* seal_words;
    uint32_t* root_seal;          /* the root join receipt, or NULL (no join tree requested / one segment) */
    size_t root_seal_words;
    size_t n_joins;
    double wall_s, leaves_s, join_s;            /* whole call, leaf phase, join tree */
    double witgen_s_sum, seal_s_sum;            /* summed over segments (lane seconds) */
    size_t n_lifts;                             /* join_tree == 2: proofs of the bottom level (lifts, or lift2 per pair) + one lift per assumption receipt; the root is a RECURSION seal (n_joins then counts joins, join3s, unions and the resolve) */
    size_t root_program;                        /* ... and the index of the program the root was sealed under */
    double lift_s;                              /* ... the bottom level (join_s is then the joins alone) — with the streamed fold: what
                                                 * each still took AFTER the last segment was sealed (most of it overlapped the leaves) */
    size_t n_retries;                           /* segments handed to another lane after a failed attempt (ZKH_SEGMENT_RETRIES, default 1) */
    double fold_tail_s;                         /* join_tree == 2: last segment sealed -> root receipt */
    double fold_busy_s_sum;                     /* ... lane seconds spent inside lift / lift2 / join proofs */
    int streamed;                               /* ... 1 = fold nodes were proven as soon as their children existed (the default) */
    double preflight_cpu_s_sum;                 /* witness source 1: host CPU seconds spent in the sequential preflight, summed over segments */
    double trace_bytes;                         /* ... bytes that crossed PCIe as witness input (16 per cycle + the RAM image), summed */
    uint32_t root_core[8];                      /* join_tree == 2: the OPENING of the claim' the root seal publishes (claim' = hash_pair(core, */
    uint32_t root_pre, root_post;               /* (pre, post, 0..))): what a further join needs as witness for this receipt; never trusted */
} zkh_prove_info;
/* CircuitHal::accumulate for circuits without a built-in accum generator: fill `accum` (W_accum x 2^po2) from data + mix */
typedef const char* (*zkh_accumulate_fn)(void* user, zkh_ctx*, const zkh_circuit*, size_t po2, const zkh_buf* data,
                                         const uint32_t* mix_global, zkh_buf* accum);
/* devices[n_devices] x lanes_per_device lanes; join_desc: a P2-JOIN description (kind 3) or NULL */
const char* zkh_session_create(const int* devices, size_t n_devices, size_t lanes_per_device, const uint32_t* desc,
                               size_t desc_words, const uint32_t* join_desc, size_t join_desc_words, zkh_session** out);
void zkh_session_destroy(zkh_session*);
size_t zkh_session_lanes(const zkh_session*);
/* the circuit handle of a lane (join != 0: its join circuit), e.g. to attach code objects before proving */
zkh_circuit* zkh_session_circuit(zkh_session*, size_t lane, int join);
void zkh_session_set_accumulate(zkh_session*, zkh_accumulate_fn fn, void* user);
/* Built-in circuits (kinds 1, 2): every lane commits the code (control) group of a segment size ONCE and keeps the committed form
 * resident in HBM (zkh_prover_cache_code) — the default; seals are byte-identical.  on = 0: re-commit it per segment, as
 * upstream's SegmentProver does. */
void zkh_session_set_resident_code(zkh_session*, int on);
/* The lift / join programs of the RECURSION circuit (zkh_rec_program_*; blobs from `python -m zeth_amd.circuits.rec_verify dir`):
 * rec_desc = the RECURSION description; program i is blobs[i] (words[i] words) of kind kinds[3 i .. 3 i + 3) = {0, segment po2,
 * circuit family (0 = the session's)} for a lift, {1, left po2, right po2} for a join of two recursion seals, {2, left po2,
 * right po2} for a lift2 (two SEGMENT seals verified by one program: lift + lift + join fused; used for the bottom level when
 * every pair of the session has one), {3, po2 of the first two children, po2 of the third} for a join3 (three recursion seals,
 * out = what join(join(a, b), c) publishes: used above the bottom level wherever three neighbours have these sizes).
 * Every lane loads every program (code groups resident). */
const char* zkh_session_set_recursion(zkh_session*, const uint32_t* rec_desc, size_t rec_desc_words, const uint32_t* const* blobs,
                                      const size_t* words, const uint32_t* kinds, size_t n_programs);
/* The same, with the programs BUILT by the library (zkh_rec_build_program; the RECURSION description is compiled in): for a block
 * whose segments have the sizes po2s[0] > po2s[1] > .. (built-in circuits, kinds 1..3: the control roots come from their own code
 * generators) — a lift per size, a lift2 per pair, joins until the set of program sizes closes, and (with_join3) the join3 of the
 * largest size if it fits that size again: the set and the order of zeth_amd/recursion.py build_programs.  No Python, no files. */
const char* zkh_session_build_recursion(zkh_session*, const uint32_t* po2s, size_t n_po2s, int with_join3);
/* ASSUMPTION receipts of the session (upstream: the keccak batch receipts a guest's accelerator calls leave behind; ProverServer::
 * {prove_keccak, lift, union, resolve}, risc0-zkvm 3.0.3, /root/reference/Cargo.lock:5418): n seals of the circuit `desc` (no state
 * words; KECCAK-F), sizes po2s[i], proven beforehand (zkh_prove_segment), control_roots = n x 8 words (that circuit's control root at
 * each size).  Every receipt is VERIFIED here on the host; the session keeps copies.  With join_tree == 2 they are lifted (lift
 * programs of family 1), united pairwise into one receipt (kind 4: every node the digest of the SORTED pair; an odd one moves up) and
 * the session's join-tree root is resolved against the union root (kind 5): the root receipt of zkh_session_prove is the RESOLVED
 * one, zkh_session_verify recomputes its claim from the segments AND these receipts (zkh_succinct_verify_resolved: the same from
 * claims alone).  Call BEFORE zkh_session_build_recursion, which then also builds those programs (zkh_session_set_recursion: kinds
 * {0, po2, 1}, {4, a, b}, {5, session size, union size}).  n = 0 clears. */
const char* zkh_session_set_assumptions(zkh_session*, const uint32_t* desc, size_t desc_words, const uint32_t* const* seals,
                                        const size_t* seal_words, const uint32_t* po2s, const uint32_t* control_roots, size_t n);
/* join_tree == 2 runs as ONE pipeline by default: a lift2 / join is proven the moment both children exist, on the fold lanes while
 * the sealing lanes are still busy with segments, on every lane afterwards (upstream joins as receipts arrive too).  on = 0: two
 * phases (seal everything, then fold).  Same tree, same receipts either way. */
void zkh_session_set_streamed_fold(zkh_session*, int on);
/* Chained session (SYN-C circuits): zkh_session_prove runs the executor's pass first (zkh_syn_chain_contributions), gives segment i
 * the pre-state initial + sum of the contributions of segments 0 .. i-1 as its public input (any `pub` of the caller is replaced),
 * and zkh_session_verify additionally checks CONTINUITY on the seals: the first segment starts from initial_state (canonical
 * residue), every segment's pre-state (out[4]) is its predecessor's post-state (out[0]) — `CompositeReceipt::verify_integrity`.
 * SYN-S circuits additionally get their exit-code pair and journal-digest limbs here and have them checked (see below). */
const char* zkh_session_set_chained(zkh_session*, int on, uint32_t initial_state);
/* The journal of a chained SYN-S session: the bytes the guest commits — zeth's guest commits the 32-byte block hash
 * (/root/reference/guests/stateless-client/src/lib.rs:33 `env::commit_slice(block_hash)`), which the CLI then compares with the hash
 * it computed itself (/root/reference/crates/host/src/bin/cli.rs:103-107).  The last seal binds Output{SHA-256(journal), assumptions};
 * zkh_session_verify checks the seals against THESE bytes.  journal == NULL (default): the session's final state word, 4 bytes LE.
 * journal != NULL with journal_len == 0: an empty journal.  Call before zkh_session_prove. */
const char* zkh_session_set_journal(zkh_session*, const uint8_t* journal, size_t journal_len);
/* Where a segment's witness comes from (SYN-AIR circuits without public inputs).  0 (default): the closed-form generator on the
 * device (zkh_syn_witgen).  1: upstream's shape — a SEQUENTIAL host preflight per segment (zkh_syn_preflight) running ahead of the
 * seals on `producers_per_lane` host threads per sealing lane (0 = 2), its compact records (16 bytes per cycle) uploaded from pinned
 * memory and expanded on the GPU (zkh_syn_witgen_trace).  zkh_prove_info reports the host CPU seconds and the PCIe bytes. */
const char* zkh_session_set_witness_source(zkh_session*, int source, size_t producers_per_lane);
/* join_tree == 1: fold the receipts through the P2-JOIN tree (joins at 2^join_po2; join_noise_key NULL = a fresh OS key per proof);
 * join_tree == 2: lift every receipt and join level by level with the RECURSION programs - every node verifies its child
 * seal(s) in-circuit; the root receipt is a RECURSION seal with out = claim tree root ‖ allowed-programs root.  The tree: the
 * first level pairs the segments, every level above takes three nodes at a time (a remainder of two is a join, of one moves up);
 * a group of three is ONE proof where the session has a join3 program for its sizes, else join(join(a, b), c) - the same node. */
const char* zkh_session_prove(zkh_session*, const zkh_segment* segs, size_t n, int join_tree, size_t join_po2,
                              const uint32_t join_noise_key[8], zkh_prove_info* info);
void zkh_prove_info_free(zkh_prove_info*);
/* every leaf seal against the control root of its size; with a root receipt also the root seal and the claim tree
 * (hash_pair over the leaf claims, recomputed on the host) against the root's public output */
const char* zkh_session_verify(zkh_session*, const zkh_segment* segs, const zkh_prove_info* info, size_t join_po2);

/* Sessions that TERMINATE (SYN-S circuits: zeth_amd/circuits/syn_air.py syn_session; shipped as "syn_session").  Upstream's ReceiptClaim
 * carries an exit code and an output (journal) digest per segment, and `receipt.verify(image_id)` + the journal comparison
 * (/root/reference/crates/host/src/bin/cli.rs:103-107) rely on them: every segment but the last ends in SystemSplit, the last in
 * Halted(0) with the digest of the journal (risc0-zkvm 3.0.3 receipt/composite.rs verify_integrity, recalled).  A SYN-S segment
 * publishes out = (post, 0, 0, 0, pre, exit_sys, exit_user, j_0 .. j_15) — j = the session's OUTPUT digest
 * tagged_struct("risc0.Output", [SHA-256(journal), Assumptions digest]) as sixteen 16-bit limbs (round 6; until then SHA-256(journal)
 * alone) — every word a bound public input; zkh_session_set_chained fills them (the journal of a session = its final state word, 4 bytes
 * LE; the assumptions = the receipts handed to zkh_session_set_assumptions, in that order) and zkh_session_verify checks them.
 * zkh_session_check_output is that check for a verifier that holds VERIFIED seals: a receipt whose trailing segments were cut off ends in
 * a SystemSplit and is refused; journal == NULL: the final state word; assumptions_digest == NULL: the session assumed nothing —
 * otherwise zkh_assumptions_digest over (claim digest = zkh_receipt_claim, control root) of the assumption receipts THE VERIFIER holds:
 * a session cannot be resolved against other receipts than the ones its own seal names.  zkh_session_check_termination = the same with
 * no assumptions.  Host only. */
void zkh_sha256(const uint8_t* data, size_t len, uint8_t out[32]);
const char* zkh_session_check_termination(const zkh_circuit*, const uint32_t* const* seals, const size_t* seal_words, size_t n,
                                          const uint8_t* journal, size_t journal_len);
const char* zkh_session_check_output(const zkh_circuit*, const uint32_t* const* seals, const size_t* seal_words, size_t n,
                                     const uint8_t* journal, size_t journal_len, const uint32_t assumptions_digest[8]);
/* Assumptions([Assumption{claim, control_root}, ..]).digest() (recalled layout): claims / control_roots = n x 8 words; n == 0: zero digest */
void zkh_assumptions_digest(const uint32_t* claims, const uint32_t* control_roots, size_t n, uint32_t out[8]);

/* `receipt.verify` for a SUCCINCT receipt on the host alone — no GPU, no session (upstream: SuccinctReceipt::verify_integrity, reached
 * from /root/reference/crates/host/src/bin/cli.rs:103): ONE seal of the RECURSION circuit (description compiled in) under the control
 * root of program `root_program` of the allowed set (allowed_roots: n_allowed x 8 words, in the order the prover loaded its
 * programs), the allowed-programs root the receipt carries, and its claim = the root of the leaves' claim tree.  leaves: n_leaves x
 * 10 words = (receipt claim digest [8], pre, post) per segment, taken from VERIFIED segment receipts or from the statement being
 * checked; ranks = 1, or N when the block was folded as N contiguous equal ranges whose roots were folded again (§7). */
const char* zkh_succinct_verify(const uint32_t* root_seal, size_t root_words, const uint32_t* allowed_roots, size_t n_allowed,
                                size_t root_program, const uint32_t* leaves, size_t n_leaves, size_t ranks);
/* The same for a RESOLVED receipt (upstream: ProverServer::{union, resolve}, risc0-zkvm 3.0.3, /root/reference/Cargo.lock:5418 — the
 * session's assumption receipts, e.g. keccak batches, are lifted, united pairwise into one receipt whose claim is the digest of the
 * SORTED pair at every node, and the session's root is resolved against it; recursion programs of kinds 4 / 5, zkh_rec_build_program):
 * assumption_claims = n_assumptions x 8 words, the receipt claim digests of the assumption receipts (any order within a pair: the
 * union sorts).  The claim must be wrap(hash_pair(claim' of the leaves' join tree, claim' of the union tree), pre, post of the
 * session).  n_leaves == 0 (leaves may be NULL): the receipt is the union-tree root alone. */
const char* zkh_succinct_verify_resolved(const uint32_t* root_seal, size_t root_words, const uint32_t* allowed_roots, size_t n_allowed,
                                         size_t root_program, const uint32_t* leaves, size_t n_leaves, size_t ranks,
                                         const uint32_t* assumption_claims, size_t n_assumptions);

/* ---- host placement (topology.hip): one process per GPU / one lane thread per context should run on the cores of the NUMA node
 * the GPU's root port hangs off, and allocate its pinned witness blocks there (upstream leaves placement to the operator:
 * /root/reference/run-parallel.sh:15 starts one prover per GPU and pins nothing).  Host only; sysfs + sched_setaffinity. ---- */
/* "0-15,64-79" -> sorted CPU ids (cpus may be NULL to count); anything that is not a cpulist is an error */
const char* zkh_parse_cpulist(const char* text, int* cpus, size_t cap, size_t* n);
/* NUMA node of PCI function bdf ("0000:c1:00.0") and that node's CPUs, read under sysfs_root ("/sys"); *node = -1 and
 * *n_cpus = 0 when the kernel reports none */
const char* zkh_pci_numa_cpus(const char* sysfs_root, const char* bdf, int* node, int* cpus, size_t cap, size_t* n_cpus);
/* NUMA node of a HIP device (-1 = unknown) and, optionally, its PCI bus id */
const char* zkh_device_numa_node(int device, int* node, char pci_bus_id[32]);
/* What tells one GPU from another across processes (bench.py gathers it from every rank: an N-GPU line lists N distinct devices):
 * PCI bus id, the device UUID as 32 hex digits (independent of HIP_VISIBLE_DEVICES renumbering; empty if the runtime has none),
 * NUMA node (-1 = unknown), marketing name, and the number of devices this process sees.  Any out pointer may be NULL. */
const char* zkh_device_identity(int device, char pci_bus_id[32], char uuid_hex[40], int* numa_node, char name[64], int* visible_devices);
/* Bind the CALLING thread (and the threads it creates afterwards) to slice `slot` of `share` equal slices of the device's
 * NUMA-node CPUs (share <= 1: the whole node) and make that node its preferred memory node.  ZKH_AFFINITY=off, or a host that
 * reports no node: nothing is changed and *node = -1. */
const char* zkh_bind_thread_to_device(int device, size_t slot, size_t share, int* node, size_t* n_cpus);

/* ---- profiling: per-kernel HIP-event timing on the ctx stream ---- */
const char* zkh_prof_enable(zkh_ctx*, int on);
/* writes up to cap records; returns count via *n.  Each record: name (<=47 chars), calls, total_ms */
typedef struct { char name[48]; uint64_t calls; double total_ms; double alg_bytes; } zkh_prof_rec;
const char* zkh_prof_get(zkh_ctx*, zkh_prof_rec* recs, size_t cap, size_t* n);
const char* zkh_prof_reset(zkh_ctx*);

#ifdef __cplusplus
}
#endif
#endif
