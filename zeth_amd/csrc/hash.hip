// hash.hip — Poseidon2 (BabyBear, t = 24, rate 16, out 8) Merkle hashing for gfx950:
// Hal::hash_rows / Hal::hash_fold and the fused tree build (risc0-zkp 3.0.2 src/hal/mod.rs; hash semantics
// src/core/hash/poseidon2/mod.rs; un-vendored: /root/reference/Cargo.lock:5393).  Reached from
// /root/reference/crates/host/src/lib.rs:137 (MerkleTreeProver::new inside Prover::commit_group).
//
// This is 31-bit modular integer work: no MFMA.  One lane owns one sponge; the 24-word state lives in VGPRs for
// the whole permutation (all cell loops are fully unrolled, round constants arrive through scalar loads), rows
// are read column-major so a wave's 64 lanes read 64 consecutive words per column, and the next 16 column
// words are fetched while the current block is permuted.  VALU-bound by construction (~1356 Montgomery products
// per 64 absorbed bytes); the HBM side only has to keep up with 16*W*n bytes per tree.
#include "common.h"
#include "poseidon2.h"

using namespace zkh;

namespace {

// Hal::hash_rows — one lane per leaf.
__global__ __launch_bounds__(256, 5) void k_hash_rows(uint32_t* __restrict__ out, const uint32_t* __restrict__ matrix,
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
        poseidon2_mix(s, rc, diag);
    }
    if (tail || cols == 0) {
        const uint32_t* tsrc = src + (size_t)full * RATE * rows;
#pragma unroll
        for (int i = 0; i < RATE; i++) s[i] = (uint32_t)i < tail ? tsrc[(size_t)i * rows] : 0u;
        poseidon2_mix(s, rc, diag);
    }
    uint4* o = (uint4*)(out + r * 8);
    o[0] = make_uint4(s[0], s[1], s[2], s[3]);
    o[1] = make_uint4(s[4], s[5], s[6], s[7]);
}

// Hal::hash_fold — one lane per parent: io[out+i] = H(io[in+2i] || io[in+2i+1])
__global__ __launch_bounds__(256, 5) void k_hash_fold(uint32_t* __restrict__ io, size_t input_size, size_t output_size,
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
    poseidon2_mix(s, rc, diag);
    uint4* o = (uint4*)(io + (output_size + i) * 8);
    o[0] = make_uint4(s[0], s[1], s[2], s[3]);
    o[1] = make_uint4(s[4], s[5], s[6], s[7]);
}

// Top of the tree in ONE launch: a single workgroup folds layers width..1 through LDS (the last ~10 layers are
// latency-bound: one launch instead of ten).  nodes[width, 2*width) is the input layer.
constexpr int TOP_LOG = 9, TOP_W = 1 << TOP_LOG;
__global__ __launch_bounds__(256) void k_fold_top(uint32_t* __restrict__ nodes, uint32_t width, const uint32_t* __restrict__ rc,
                                                  const uint32_t* __restrict__ diag) {
    __shared__ uint32_t layer[TOP_W * 8];
    for (uint32_t w = threadIdx.x; w < width * 8; w += blockDim.x) layer[w] = nodes[(size_t)width * 8 + w];
    __syncthreads();
    for (uint32_t cur = width >> 1; cur >= 1; cur >>= 1) {
        uint32_t res[OUT];
        // cur <= 256 parents: one lane each
        const bool active = threadIdx.x < cur;
        if (active) {
            uint32_t s[CELLS];
#pragma unroll
            for (int k = 0; k < RATE; k++) s[k] = layer[threadIdx.x * 16 + k];
#pragma unroll
            for (int k = RATE; k < CELLS; k++) s[k] = 0;
            poseidon2_mix(s, rc, diag);
#pragma unroll
            for (int k = 0; k < OUT; k++) res[k] = s[k];
        }
        __syncthreads();
        if (active) {
#pragma unroll
            for (int k = 0; k < OUT; k++) {
                layer[threadIdx.x * 8 + k] = res[k];
                nodes[((size_t)cur + threadIdx.x) * 8 + k] = res[k];
            }
        }
        __syncthreads();
    }
}

}  // namespace

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
extern "C" const char* zkh_merkle_fold_all(zkh_ctx* c, zkh_buf* nodes, size_t rows) {
    ZKH_REQUIRE(nodes->len == rows * 16 && rows && (rows & (rows - 1)) == 0, "merkle_fold_all: nodes must hold 2*rows digests");
    size_t layer = rows;                       // current input layer width
    while (layer > (size_t)TOP_W) {
        ZKH_TRY(zkh_hash_fold(c, nodes, layer, layer / 2));
        layer /= 2;
    }
    if (layer >= 2) {
        ProfScope prof(c, "hash_fold_top", 96.0 * layer);
        k_fold_top<<<1, 256, 0, c->stream>>>(nodes->ptr(), (uint32_t)layer, c->tab.rc, c->tab.diag);
        ZKH_TRY(last_launch_error("hash_fold_top"));
    }
    return nullptr;
}
