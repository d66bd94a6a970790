// hash.hip — Poseidon2 (BabyBear, t = 24, rate 16, out 8) Merkle hashing for gfx950:
// Hal::hash_rows / Hal::hash_fold and the fused tree build (risc0-zkp 3.0.2 src/hal/mod.rs; hash semantics
// src/core/hash/poseidon2/mod.rs; un-vendored: /root/reference/Cargo.lock:5393).  Reached from
// /root/reference/crates/host/src/lib.rs:137 (MerkleTreeProver::new inside Prover::commit_group).
//
// This is 31-bit modular integer work: no MFMA.  One lane owns one sponge; the 24-word state lives in VGPRs for
// the whole permutation (all cell loops are fully unrolled, round constants arrive through scalar loads), rows
// are read column-major so a wave's 64 lanes read 64 consecutive words per column; four resident workgroups per CU
// hide the column loads behind other waves' permutations (no software prefetch: it only cost VGPRs).  VALU-bound
// by construction (~1356 Montgomery products per 64 absorbed bytes); the HBM side only has to keep up with
// 16*W*n bytes per tree.  The permutation is poseidon2.h's lane-per-permutation form: signed Montgomery s-boxes with
// no canonicalisation, M_ext on exact doubles (v_add_f64 / v_fma_f64 issue at the rate of a 32-bit multiply and have
// the headroom a 31-bit prime denies a 32-bit word), partial rounds three at a time with unreduced weighted sums:
// 6.4 k VALU instructions per 64 permutations.
#include "common.h"
#include "poseidon2.h"
#include "poseidon2_wide.h"

using namespace zkh;

// common.h sizes the host / device / verifier copies of the partial-round table by hand; poseidon2.h lays the table out.
static_assert(zkh::ZKH_P2_PTAB == zkh::P2_TAB_WORDS, "common.h ZKH_P2_PTAB must equal poseidon2.h P2_TAB_WORDS");

namespace {

// Hal::hash_rows — one lane per leaf.
// (register budgets for 2, 3 or 4 workgroups per CU compile to the same speed; capping residency at 2 or 3 workgroups
// with LDS padding is 5 % slower alone and 5 % slower with three seals in flight: no SMT-style gain from leaving room)
__global__ __launch_bounds__(256, 4) void k_hash_rows(uint32_t* __restrict__ out, const uint32_t* __restrict__ matrix,
                                                   size_t rows, uint32_t cols, const uint32_t* __restrict__ rc,
                                                   const uint32_t* __restrict__ diag) {
    const size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    uint32_t s[CELLS];
#pragma unroll
    for (int i = 0; i < CELLS; i++) s[i] = 0;
    const uint32_t* src = matrix + r;
    const uint32_t full = cols / RATE, tail = cols % RATE;
    // No register double-buffer for the next 16 columns: the grouped partial rounds need the VGPRs, and with >= 5
    // waves per SIMD the loads of one wave hide under the permutations of the others.
    for (uint32_t b = 0; b < full; b++) {
        const uint32_t* bsrc = src + (size_t)b * RATE * rows;
#pragma unroll
        for (int i = 0; i < RATE; i++) s[i] = bsrc[(size_t)i * rows];
        poseidon2_mix_raw(s, rc, diag);
#pragma unroll
        for (int i = RATE; i < CELLS; i++) s[i] = p2_finish(s[i], diag);     // the capacity is all the next block keeps
    }
    if (tail || cols == 0) {
        const uint32_t* tsrc = src + (size_t)full * RATE * rows;
#pragma unroll
        for (int i = 0; i < RATE; i++) s[i] = (uint32_t)i < tail ? tsrc[(size_t)i * rows] : 0u;
        poseidon2_mix_raw(s, rc, diag);
    }
#pragma unroll
    for (int i = 0; i < OUT; i++) s[i] = p2_finish(s[i], diag);
    uint4* o = (uint4*)(out + r * 8);
    o[0] = make_uint4(s[0], s[1], s[2], s[3]);
    o[1] = make_uint4(s[4], s[5], s[6], s[7]);
}

// Hal::hash_fold — one lane per parent: io[out+i] = H(io[in+2i] || io[in+2i+1])
__global__ __launch_bounds__(256, 4) void k_hash_fold(uint32_t* __restrict__ io, size_t input_size, size_t output_size,
                                                   const uint32_t* __restrict__ rc, const uint32_t* __restrict__ diag) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= output_size) return;
    const uint4* src = (const uint4*)(io + (input_size + 2 * i) * 8);
    uint32_t s[CELLS];
    const uint4 a = src[0], b = src[1], c = src[2], d = src[3];
    s[0] = a.x; s[1] = a.y; s[2] = a.z; s[3] = a.w; s[4] = b.x; s[5] = b.y; s[6] = b.z; s[7] = b.w;
    s[8] = c.x; s[9] = c.y; s[10] = c.z; s[11] = c.w; s[12] = d.x; s[13] = d.y; s[14] = d.z; s[15] = d.w;
#pragma unroll
    for (int k = RATE; k < CELLS; k++) s[k] = 0;
    poseidon2_mix_raw(s, rc, diag);
#pragma unroll
    for (int k = 0; k < OUT; k++) s[k] = p2_finish(s[k], diag);
    uint4* o = (uint4*)(io + (output_size + i) * 8);
    o[0] = make_uint4(s[0], s[1], s[2], s[3]);
    o[1] = make_uint4(s[4], s[5], s[6], s[7]);
}

// ---------------------------------------------------------------------------------------------------------
// Wavefront-cooperative permutation for the narrow layers of a tree.  With one lane per permutation a layer
// cannot finish faster than one serial permutation (~36k VALU cycles, ~20-30 us) however few parents it has, and
// a 2^22-leaf tree has 16 such layers.  Here EIGHT lanes share one permutation: lane j < 6 owns state cells
// 4j..4j+3 (one M4 block), the column sums of M_ext and the partial-round state sum are 3-step DPP butterflies
// (quad_perm xor 1, xor 2, row_half_mirror) that never touch LDS, round constants sit in LDS (one ds_read_b128 per
// full round).  ~3.5x lower latency per layer; used only where the layer is too narrow to fill the chip.
// (The permutation itself: poseidon2_wide.h.)
// ---------------------------------------------------------------------------------------------------------
// one 8-lane group folds one parent: io[out + parent] = H(io[in + 2 parent] || io[in + 2 parent + 1]); rcs = rc in LDS
__device__ __forceinline__ void wide_fold_one(uint32_t* __restrict__ io, size_t input_size, size_t output_size, uint32_t gid,
                                              const uint32_t* rcs, const uint32_t* __restrict__ pc) {
    const uint32_t j = gid & 7;
    size_t parent = gid >> 3;
    const bool live = parent < output_size;
    if (!live) parent = output_size - 1;                        // keep the whole wave in the butterflies
    uint32_t c[4] = {0, 0, 0, 0};
    if (j < 4) {
        const uint4 v = *(const uint4*)(io + (input_size + 2 * parent) * 8 + 4 * j);
        c[0] = v.x; c[1] = v.y; c[2] = v.z; c[3] = v.w;
    }
    wide_permute(c, j, rcs, pc);
    if (live && j < 2) *(uint4*)(io + (output_size + parent) * 8 + 4 * j) = make_uint4(c[0], c[1], c[2], c[3]);
}
constexpr int WIDE_LOG = 15;      // layers with <= 2^15 parents use the 8-lane permutation
constexpr int TAIL_LOG = 7;       // a 1024-lane workgroup = 2^7 eight-lane groups; the last 8 layers (<= 2^7 parents) are one launch
// Middle of the tree: workgroup b owns the 128 consecutive parents [128 b, 128 b + 128) of the layer with `first_parents`
// parents and the `levels` - 1 layers above them (64, 32, ... parents of its own subtree), so the layers with
// 2^15 .. 2^8 parents are ONE launch.  Each level still costs one 8-lane permutation of latency; what disappears is the
// launch round trip between them.
__global__ __launch_bounds__(8 << TAIL_LOG) void k_hash_fold_subtree(uint32_t* __restrict__ io, size_t first_parents, uint32_t levels,
                                                                     const uint32_t* __restrict__ rc, const uint32_t* __restrict__ pc) {
    __shared__ __attribute__((aligned(16))) uint32_t rcs[ROUNDS_TOTAL * CELLS + 8];
    for (uint32_t w = threadIdx.x; w < ROUNDS_TOTAL * CELLS; w += blockDim.x) rcs[w] = rc[w];
    __syncthreads();
    for (uint32_t l = 0; l < levels; l++) {
        const size_t parents = first_parents >> l;
        const uint32_t local = (1u << TAIL_LOG) >> l;           // parents of this level inside the workgroup
        if ((threadIdx.x >> 6) * 8 < local) {                   // wave-uniform
            // lanes beyond `local` groups alias the last parent of the level (no write): live == false there
            const uint32_t grp = threadIdx.x >> 3;
            const uint32_t gid = grp < local ? (uint32_t)((blockIdx.x * local + grp) * 8 + (threadIdx.x & 7)) : 0xffffffffu;
            wide_fold_one(io, 2 * parents, parents, gid, rcs, pc);
        }
        __threadfence_block();
        __syncthreads();
    }
}
// Tree top: parents = first_parents, first_parents / 2, ..., 1 inside one workgroup (each layer reads what the previous
// one wrote: __syncthreads orders the global writes of a workgroup for its own later reads).
__global__ __launch_bounds__(8 << TAIL_LOG) void k_hash_fold_tail(uint32_t* __restrict__ io, size_t first_parents,
                                                                  const uint32_t* __restrict__ rc, const uint32_t* __restrict__ pc) {
    __shared__ __attribute__((aligned(16))) uint32_t rcs[ROUNDS_TOTAL * CELLS + 8];
    for (uint32_t w = threadIdx.x; w < ROUNDS_TOTAL * CELLS; w += blockDim.x) rcs[w] = rc[w];
    __syncthreads();
    for (size_t parents = first_parents; parents >= 1; parents >>= 1) {
        if ((threadIdx.x >> 6) * 8 < parents)                    // wave-uniform: waves with no live group skip the layer
            wide_fold_one(io, 2 * parents, parents, threadIdx.x, rcs, pc);
        __threadfence_block();
        __syncthreads();
    }
}

// the bare permutation, one lane per state (24 words each)
__global__ __launch_bounds__(256) void k_poseidon2_mix(uint32_t* __restrict__ io, size_t count, const uint32_t* __restrict__ rc,
                                                       const uint32_t* __restrict__ diag) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    uint4* p = (uint4*)(io + i * CELLS);
    uint32_t s[CELLS];
#pragma unroll
    for (int q = 0; q < CELLS / 4; q++) { const uint4 v = p[q]; s[4 * q] = v.x; s[4 * q + 1] = v.y; s[4 * q + 2] = v.z; s[4 * q + 3] = v.w; }
    poseidon2_mix(s, rc, diag);
#pragma unroll
    for (int q = 0; q < CELLS / 4; q++) p[q] = make_uint4(s[4 * q], s[4 * q + 1], s[4 * q + 2], s[4 * q + 3]);
}

}  // namespace

extern "C" const char* zkh_poseidon2_mix(zkh_ctx* c, zkh_buf* states, size_t count) {
    ZKH_REQUIRE(states && states->len == count * CELLS, "poseidon2_mix: buffer is not count x 24 words");
    if (!count) return nullptr;
    bind_thread(c);
    k_poseidon2_mix<<<(unsigned)((count + 255) / 256), 256, 0, c->stream>>>(states->ptr(), count, c->tab.rc, c->tab.diag);
    return last_launch_error("poseidon2_mix");
}
extern "C" const char* zkh_hash_rows(zkh_ctx* c, zkh_buf* out, const zkh_buf* matrix) {
    ZKH_REQUIRE(out->len % 8 == 0 && out->len, "hash_rows: output is not a digest array");
    const size_t rows = out->len / 8;
    ZKH_REQUIRE(matrix->len % rows == 0, "hash_rows: matrix size %zu not a multiple of rows %zu", matrix->len, rows);
    const size_t cols = matrix->len / rows;
    ProfScope prof(c, "hash_rows", 4.0 * matrix->len + 32.0 * rows);
    k_hash_rows<<<(unsigned)((rows + 255) / 256), 256, 0, c->stream>>>(out->ptr(), matrix->ptr(), rows, (uint32_t)cols,
                                                                      c->tab.rc, c->tab.diag);
    return last_launch_error("hash_rows");
}
extern "C" const char* zkh_hash_fold(zkh_ctx* c, zkh_buf* io, size_t input_size, size_t output_size) {
    ZKH_REQUIRE(io->len % 8 == 0, "hash_fold: not a digest array");
    ZKH_REQUIRE((input_size + 2 * output_size) * 8 <= io->len && output_size * 2 <= input_size, "hash_fold: ranges out of bounds");
    if (!output_size) return nullptr;
    ProfScope prof(c, "hash_fold", 96.0 * output_size);
    k_hash_fold<<<(unsigned)((output_size + 255) / 256), 256, 0, c->stream>>>(io->ptr(), input_size, output_size, c->tab.rc,
                                                                             c->tab.diag);
    return last_launch_error("hash_fold");
}
static const char* merkle_fold_from(zkh_ctx* c, zkh_buf* nodes, size_t first_layer);
extern "C" const char* zkh_merkle_fold_all(zkh_ctx* c, zkh_buf* nodes, size_t rows) {
    ZKH_REQUIRE(nodes->len == rows * 16 && rows && (rows & (rows - 1)) == 0, "merkle_fold_all: nodes must hold 2*rows digests");
    return merkle_fold_from(c, nodes, rows);
}
// MerkleTreeProver::new: leaves = hash_rows(matrix), then every layer above.  (A fused leaves + first-layer pass was built in
// round 4, measured slower — 17.81 vs 17.68 ms of Poseidon2 per seal, profiles/r04_merkle_fused_ab.txt — and removed in round 6.)
extern "C" const char* zkh_merkle_build(zkh_ctx* c, zkh_buf* nodes, const zkh_buf* matrix, size_t rows) {
    ZKH_REQUIRE(nodes && matrix && nodes->len == rows * 16 && rows && (rows & (rows - 1)) == 0, "merkle_build: nodes must hold 2*rows digests");
    ZKH_REQUIRE(matrix->len % rows == 0, "merkle_build: matrix size %zu not a multiple of rows %zu", matrix->len, rows);
    zkh_buf* leaves = nullptr;
    ZKH_TRY(zkh_slice(nodes, rows * 8, rows * 8, &leaves));
    const char* err = zkh_hash_rows(c, leaves, matrix);
    zkh_release(leaves);
    ZKH_TRY(err);
    return merkle_fold_from(c, nodes, rows);
}
static const char* merkle_fold_from(zkh_ctx* c, zkh_buf* nodes, size_t first_layer) {
    for (size_t layer = first_layer; layer >= 2; layer /= 2) {       // layer = current input width, layer/2 parents
        const size_t parents = layer / 2;
        if (parents > ((size_t)1 << WIDE_LOG)) {
            ZKH_TRY(zkh_hash_fold(c, nodes, layer, parents));
        } else if (parents > ((size_t)1 << TAIL_LOG)) {
            uint32_t levels = 0;
            while ((parents >> levels) > ((size_t)1 << TAIL_LOG)) levels++;      // down to the layer with 2^(TAIL_LOG+1) parents
            ProfScope prof(c, "hash_fold_wide", 96.0 * (2 * parents - (parents >> (levels - 1))));
            k_hash_fold_subtree<<<(unsigned)(parents >> TAIL_LOG), 8 << TAIL_LOG, 0, c->stream>>>(nodes->ptr(), parents, levels, c->tab.rc,
                                                                                                   c->tab.diag);
            ZKH_TRY(last_launch_error("hash_fold_wide"));
            layer >>= (levels - 1);                              // the loop's own /2 steps over the last fused level
        } else {
            ProfScope prof(c, "hash_fold_tail", 96.0 * (2 * parents - 1));
            k_hash_fold_tail<<<1, 8 << TAIL_LOG, 0, c->stream>>>(nodes->ptr(), parents, c->tab.rc, c->tab.diag);
            ZKH_TRY(last_launch_error("hash_fold_tail"));
            break;
        }
    }
    return nullptr;
}
