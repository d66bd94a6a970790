// circuit.hip — CircuitHal side: circuit description loader, eval_check (generated kernels + generic on-device
// step interpreter) and the SYN-AIR witness generator.
// Stands in for risc0-circuit-rv32im 4.0.2 src/prove/hal/{mod.rs,cuda.rs} + the Zirgen-generated eval_check /
// witgen kernels of its -sys crate (un-vendored; /root/reference/Cargo.lock:5320), driven through
// risc0_zkp::hal::CircuitHal (risc0-zkp 3.0.2 src/hal/mod.rs).  Reached from
// /root/reference/crates/host/src/lib.rs:137.
#include "circuit.h"
#include "poseidon2.h"
#include "../../include/zkh_poseidon2_consts.h"

#include <algorithm>

using namespace zkh;

namespace zkh {
uint64_t desc_hash64(const uint32_t* w, size_t n) {   // FNV-1a over the words
    uint64_t h = 0xcbf29ce484222325ull;
    for (size_t i = 0; i < n; i++) {
        for (int b = 0; b < 4; b++) { h ^= (w[i] >> (8 * b)) & 0xff; h *= 0x100000001b3ull; }
    }
    return h;
}
const char* launch_ext_powers(zkh_ctx* c, uint32_t* out, const uint32_t start[4], const uint32_t base[4], uint32_t n);
const char* launch_ext_powers_at(zkh_ctx* c, uint32_t* out, const uint32_t base[4], const uint32_t* d_exps, uint32_t n);
const char* launch_ext_scale_at(zkh_ctx* c, uint32_t* out, const uint32_t* d_recs, uint32_t n);
const char* launch_ext_center_at(zkh_ctx* c, uint32_t* out, const uint32_t* d_exps, uint32_t n);
}
static const char* set_gather(zkh_circuit* c, int which, const std::vector<const uint32_t*>& lists, const std::vector<const uint32_t*>& consts);
namespace zkh {
const char* prefix_products_batched(zkh_ctx* c, uint32_t* io, size_t n0, size_t count, size_t col_stride);
}  // namespace zkh

namespace {

constexpr uint32_t INTERP_THREADS = 128;
constexpr uint32_t INSN_WORDS = 6;

// ---- generic interpreter: one lane per domain point, value slots in LDS ([slot][lane]) ----
// fp slots hold Fp values, wide slots (uint4) hold mix totals and the Fp4-valued values downstream of a ConstExt; taps,
// constants and globals are instruction operands (re-read where they are used), not slots.
__device__ __forceinline__ Fp4 u4_fp4(uint4 v) { return Fp4(Fp::raw(v.x), Fp::raw(v.y), Fp::raw(v.z), Fp::raw(v.w)); }
__device__ __forceinline__ uint4 fp4_u4(Fp4 v) { return make_uint4(v.c[0].v, v.c[1].v, v.c[2].v, v.c[3].v); }
__global__ __launch_bounds__(INTERP_THREADS) void k_eval_check_interp(EvalCheckArgs a, const uint32_t* __restrict__ prog,
                                                                      uint32_t n_insn, const uint32_t* __restrict__ taps,
                                                                      uint32_t n_fp_slots, uint32_t ret_slot) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    uint32_t* fps = lds;                                              // [n_fp_slots][THREADS]
    uint4* mxs = (uint4*)(lds + (size_t)n_fp_slots * INTERP_THREADS); // [n_mix_slots][THREADS]
    const uint32_t t = threadIdx.x;
    const uint32_t idx = blockIdx.x * INTERP_THREADS + t;
    if (idx >= a.dom) return;
    const uint32_t mask = a.dom - 1;
    auto fp_operand = [&](uint32_t kind, uint32_t x) -> uint32_t {
        switch (kind) {
        case OPK_FP: return fps[x * INTERP_THREADS + t];
        case OPK_TAP: {
            const uint32_t g = taps[3 * x], off = taps[3 * x + 1], back = taps[3 * x + 2];
            return a.groups[g][(size_t)off * a.dom + ((idx - 4 * back) & mask)]; }
        case OPK_CONST: return x;
        default: return a.globals[x >> 16][x & 0xffffu];
        }
    };
    auto ext_operand = [&](uint32_t kind, uint32_t x) -> Fp4 {
        if (kind == OPK_EXT) return u4_fp4(mxs[x * INTERP_THREADS + t]);
        return Fp4(Fp::raw(fp_operand(kind, x)));
    };
    for (uint32_t pc = 0; pc < n_insn; pc++) {
        const uint32_t* in = prog + pc * INSN_WORDS;
        const uint32_t opw = in[0], dst = in[1], x = in[2], y = in[3], z = in[4], w = in[5];
        const uint32_t op = opw & 0xffu, ka = (opw >> 8) & 7u, kb = (opw >> 11) & 7u;
        const bool dst_ext = (opw >> 14) & 1u;
        switch (op) {
        case OP_CONST_EXT: mxs[dst * INTERP_THREADS + t] = make_uint4(x, y, z, w); break;
        case OP_ADD: case OP_SUB: case OP_MUL:
            if (dst_ext) {
                const Fp4 l = ext_operand(ka, x), r = ext_operand(kb, y);
                mxs[dst * INTERP_THREADS + t] = fp4_u4(op == OP_ADD ? l + r : (op == OP_SUB ? l - r : l * r));
            } else {
                const uint32_t l = fp_operand(ka, x), r = fp_operand(kb, y);
                fps[dst * INTERP_THREADS + t] = op == OP_ADD ? add_mod(l, r) : (op == OP_SUB ? sub_mod(l, r) : mul_mod(l, r));
            }
            break;
        case OP_TRUE: mxs[dst * INTERP_THREADS + t] = make_uint4(0, 0, 0, 0); break;
        case OP_AND_EQZ: {   // tot = x.tot + mix^e(x) * v
            const uint4 xt = mxs[x * INTERP_THREADS + t];
            const uint4 pw = ((const uint4*)a.mix_pows)[w];
            if (kb == OPK_EXT) {
                mxs[dst * INTERP_THREADS + t] = fp4_u4(u4_fp4(xt) + u4_fp4(pw) * ext_operand(kb, y));
            } else {
                const uint32_t v = fp_operand(kb, y);
                mxs[dst * INTERP_THREADS + t] = make_uint4(add_mod(xt.x, mul_mod(pw.x, v)), add_mod(xt.y, mul_mod(pw.y, v)),
                                                           add_mod(xt.z, mul_mod(pw.z, v)), add_mod(xt.w, mul_mod(pw.w, v)));
            }
            break; }
        case OP_AND_COND: {  // tot = x.tot + cond * y.tot * mix^e(x)
            const uint4 xt = mxs[x * INTERP_THREADS + t], yt = mxs[z * INTERP_THREADS + t];
            const uint4 pw = ((const uint4*)a.mix_pows)[w];
            const Fp4 inner = kb == OPK_EXT ? u4_fp4(yt) * ext_operand(kb, y) : u4_fp4(yt) * Fp::raw(fp_operand(kb, y));
            mxs[dst * INTERP_THREADS + t] = fp4_u4(u4_fp4(xt) + inner * u4_fp4(pw));
            break; }
        }
    }
    const uint4 tot = mxs[ret_slot * INTERP_THREADS + t];
    const uint32_t zi = a.zinv[idx & 3];
    a.check[idx] = mul_mod(tot.x, zi);
    a.check[(size_t)a.dom + idx] = mul_mod(tot.y, zi);
    a.check[2 * (size_t)a.dom + idx] = mul_mod(tot.z, zi);
    a.check[3 * (size_t)a.dom + idx] = mul_mod(tot.w, zi);
}

// ---- SYN-AIR witness (CPU twin: oracle/circuit.c) ----
__device__ __forceinline__ uint32_t syn_cell(uint64_t seed, uint32_t group, uint32_t col, uint32_t row) {
    uint64_t z = seed ^ ((uint64_t)(group + 1) * 0x9E3779B97F4A7C15ull);
    z += (uint64_t)col * 0xBF58476D1CE4E5B9ull;
    z += (uint64_t)row * 0x94D049BB133111EBull;
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return mul_mod(R2, (uint32_t)(z >> 32) % P);
}
// The code group is a function of (circuit, po2, zk_cycles) only, like upstream's control columns: its Merkle root is the
// control root the verifier checks.
constexpr uint64_t SYN_CODE_SEED = 0xC0DEC0DE5EEDull;
__global__ void k_syn_code(uint32_t* code, uint32_t wc, uint32_t n, uint32_t A, uint64_t seed) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x, col = blockIdx.y;
    if (r >= n) return;
    uint32_t v;
    switch (col) {
    case 0: v = r < A ? R1 : 0; break;
    case 1: v = r == 0 ? R1 : 0; break;
    case 2: v = (r > 0 && r < A) ? R1 : 0; break;
    case 3: v = mul_mod(R2, r); break;
    case 4: v = r == A - 1 ? R1 : 0; break;
    default: v = syn_cell(seed, GROUP_CODE, col, r);
    }
    code[(size_t)col * n + r] = v;
}
// one lane per row: free cells, products, and the running-sum increment (scanned afterwards)
// pub: n_pub public input words; word k replaces the x cell of triple k in row 0 (bound to out[4 + k] by a constraint)
__global__ void k_syn_data(uint32_t* data, uint32_t wd, uint32_t n, uint32_t A, uint64_t seed, NoiseKey nk,
                           const uint32_t* __restrict__ pub, uint32_t n_pub) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const uint32_t T = (wd - 2) / 3;
    if (r >= A) {
        for (uint32_t c = 0; c < wd; c++) data[(size_t)c * n + r] = noise_cell(nk, GROUP_DATA, c, r);      // blinding rows: noise.h
        return;
    }
    uint32_t d0 = 0, d1 = 0, d3 = 0, d4 = 0;
    for (uint32_t j = 0; j < T; j++) {
        uint32_t x = syn_cell(seed, GROUP_DATA, 3 * j, r);
        if (r == 0 && j < n_pub) x = pub[j];
        const uint32_t y = syn_cell(seed, GROUP_DATA, 3 * j + 1, r);
        const uint32_t pr = mul_mod(x, y);
        data[(size_t)(3 * j) * n + r] = x; data[(size_t)(3 * j + 1) * n + r] = y; data[(size_t)(3 * j + 2) * n + r] = pr;
        if (j == 0) { d0 = x; d1 = y; }
        if (j == 1) { d3 = x; d4 = y; }
    }
    for (uint32_t c = 3 * T; c < wd - 2; c++) data[(size_t)c * n + r] = syn_cell(seed, GROUP_DATA, c, r);
    data[(size_t)(wd - 2) * n + r] = mul_mod(mul_mod(d0, d1), mul_mod(d3, d4));
    // increment of the running sum; row 0 starts the sum at d0
    const uint32_t inc = r == 0 ? d0 : add_mod(d0, mul_mod(mul_mod(R2, r), d1));
    data[(size_t)(wd - 1) * n + r] = inc;
}
// Chained sessions (SYN-C, circuits/syn_air.py syn_chain): what segment i adds to the running state = sum over its active rows
// r >= 1 of d0[r] + r d1[r] (row 0 holds the pre-state itself).  One workgroup per segment; the executor's pass of a session.
__global__ __launch_bounds__(1024) void k_syn_chain_contrib(uint32_t* out, const uint64_t* __restrict__ seeds, const uint32_t* __restrict__ po2s,
                                                            uint32_t zk_cycles) {
    __shared__ uint32_t part[1024];
    const uint32_t i = blockIdx.x, t = threadIdx.x;
    const uint64_t seed = seeds[i];
    const uint32_t A = (1u << po2s[i]) - zk_cycles;
    uint32_t acc = 0;
    for (uint32_t r = 1 + t; r < A; r += 1024) {
        const uint32_t d0 = syn_cell(seed, GROUP_DATA, 0, r), d1 = syn_cell(seed, GROUP_DATA, 1, r);
        acc = add_mod(acc, add_mod(d0, mul_mod(mul_mod(R2, r), d1)));
    }
    part[t] = acc;
    __syncthreads();
    for (uint32_t d = 512; d >= 1; d >>= 1) {
        if (t < d) part[t] = add_mod(part[t], part[t + d]);
        __syncthreads();
    }
    if (t == 0) out[i] = part[0];
}
// inclusive prefix sum (mod P) of the first A words of a column, three-level: per-1024-chunk scans + chunk totals,
// scan of the totals (one workgroup), carry add.  (witgen, reported separately from the seal)
__global__ __launch_bounds__(1024) void k_prefix_sum_chunks(uint32_t* col, uint32_t A, uint32_t* totals) {
    __shared__ uint32_t buf[2][1024];
    const uint32_t t = threadIdx.x, i = blockIdx.x * 1024 + t;
    buf[0][t] = i < A ? col[i] : 0;
    __syncthreads();
    int cur = 0;
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        uint32_t v = buf[cur][t];
        if (t >= d) v = add_mod(v, buf[cur][t - d]);
        buf[cur ^ 1][t] = v;
        cur ^= 1;
        __syncthreads();
    }
    if (i < A) col[i] = buf[cur][t];
    if (t == 1023 && totals) totals[blockIdx.x] = buf[cur][t];
}
// single workgroup over the (<= 2^14) chunk totals; also emits the grand total = s[A-1]
__global__ __launch_bounds__(1024) void k_prefix_sum_fp(uint32_t* col, uint32_t A, uint32_t* last_out) {
    __shared__ uint32_t buf[2][1024];
    __shared__ uint32_t carry_s;
    const uint32_t t = threadIdx.x;
    if (t == 0) carry_s = 0;
    __syncthreads();
    for (uint32_t base = 0; base < A; base += 1024) {
        const uint32_t i = base + t;
        buf[0][t] = i < A ? col[i] : 0;
        __syncthreads();
        int cur = 0;
        for (uint32_t d = 1; d < 1024; d <<= 1) {
            uint32_t v = buf[cur][t];
            if (t >= d) v = add_mod(v, buf[cur][t - d]);
            buf[cur ^ 1][t] = v;
            cur ^= 1;
            __syncthreads();
        }
        const uint32_t v = add_mod(buf[cur][t], carry_s);
        if (i < A) col[i] = v;
        __syncthreads();
        if (t == 1023) carry_s = v;
        __syncthreads();
    }
    if (t == 0 && last_out) *last_out = carry_s;   // prefix through the last chunk
}
__global__ __launch_bounds__(1024) void k_prefix_sum_carry(uint32_t* col, uint32_t A, const uint32_t* totals_scan) {
    const uint32_t i = blockIdx.x * 1024 + threadIdx.x;
    if (blockIdx.x == 0 || i >= A) return;
    col[i] = add_mod(col[i], totals_scan[blockIdx.x - 1]);
}
// accum: terms[e][r] = mix_e + d_{e mod wd}[r] (active rows) or 1 (noise rows), AoS ExtElems
__global__ void k_syn_accum_terms(uint32_t* terms, const uint32_t* data, const uint32_t* mix, uint32_t wd, uint32_t n, uint32_t A) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x, e = blockIdx.y;
    if (r >= n) return;
    uint4 v;
    if (r < A) {
        const uint4 m = ((const uint4*)mix)[e];
        v = make_uint4(add_mod(m.x, data[(size_t)(e % wd) * n + r]), m.y, m.z, m.w);
    } else {
        v = make_uint4(R1, 0, 0, 0);
    }
    ((uint4*)terms)[(size_t)e * n + r] = v;
}
__global__ void k_syn_accum_store(uint32_t* accum, const uint32_t* prods, uint32_t n, uint32_t A, NoiseKey nk) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x, e = blockIdx.y;
    if (r >= n) return;
    uint32_t v[4];
    if (r < A) {
        const uint4 p = ((const uint4*)prods)[(size_t)e * n + r];
        v[0] = p.x; v[1] = p.y; v[2] = p.z; v[3] = p.w;
    } else {
        for (int i = 0; i < 4; i++) v[i] = noise_cell(nk, GROUP_ACCUM, 4 * e + i, r);
    }
    for (int i = 0; i < 4; i++) accum[(size_t)(4 * e + i) * n + r] = v[i];
}

// ---- KECCAK-F witness (zeth_amd/circuits/keccak_f.py; CPU twin: oracle/keccak.c) ----
// Stands in for risc0-circuit-keccak 4.0.2's witness generator (un-vendored: /root/reference/Cargo.lock:5289): every 25
// active rows are one FIPS 202 keccak-f[1600] permutation.  Two phases so that the wide (3840-column) trace is written with
// coalesced stores: (1) one lane per PERMUTATION runs the 24 rounds on 25 x u64 registers and leaves 60 packed lanes per
// trace row (state, theta parities, rho/pi output) in a scratch buffer; (2) one lane per (row, column) expands a bit.
constexpr uint32_t KF_ROUNDS = 24, KF_BLOCK = 25, KF_LANES = 60;
struct KeccakTables { uint64_t rc[KF_ROUNDS]; uint8_t rho[25]; };      // rho[x + 5 y]
__device__ __forceinline__ uint64_t rotl64(uint64_t v, uint32_t k) { k &= 63; return k ? (v << k) | (v >> (64 - k)) : v; }
__device__ __forceinline__ uint64_t keccak_lane(uint64_t seed, uint64_t perm, uint32_t lane) {
    uint64_t z = seed ^ 0x4B454343414B5F46ull;
    z += perm * 0xBF58476D1CE4E5B9ull;
    z += (uint64_t)(lane + 1) * 0x94D049BB133111EBull;
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return z;
}
__global__ void k_keccak_code(uint32_t* code, uint32_t n, uint32_t A, uint32_t K, KeccakTables tb) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x, col = blockIdx.y;
    if (r >= n) return;
    const bool in_blocks = r < KF_BLOCK * K;
    const uint32_t k = r % KF_BLOCK;
    bool v;
    switch (col) {
    case 0: v = r < A; break;
    case 1: v = r == 0; break;
    case 2: v = r > 0 && r < A; break;
    case 3: v = in_blocks && k < KF_ROUNDS; break;
    case 4: v = in_blocks && k >= 1; break;
    case 5: v = in_blocks && k == 0; break;
    case 6: v = K > 0 && r == KF_BLOCK * K - 1; break;
    case 14: v = K > 0 && r == KF_BLOCK * (K - 1); break;          // bind: row 0 of the last block (its input state is public too)
    default: {
        const uint32_t pos = (1u << (col - 7)) - 1;                 // 0, 1, 3, 7, 15, 31, 63
        v = in_blocks && k >= 1 && ((tb.rc[k - 1] >> pos) & 1);
    }
    }
    code[(size_t)col * n + r] = v ? R1 : 0;
}
// rows: [K][25][60] packed lanes; last_input: 25 lanes for permutation K-1 (device, may be null)
__global__ void k_keccak_perm(uint64_t* rows, uint32_t K, uint64_t seed, const uint64_t* __restrict__ last_input, KeccakTables tb) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= K) return;
    uint64_t a[25];
    for (uint32_t l = 0; l < 25; l++) a[l] = (last_input && p + 1 == K) ? last_input[l] : keccak_lane(seed, p, l);
    uint64_t* row = rows + (size_t)p * KF_BLOCK * KF_LANES;
    for (uint32_t r = 0; r < KF_ROUNDS; r++, row += KF_LANES) {
        uint64_t c[5], d[5], b[25];
        for (uint32_t x = 0; x < 5; x++) {
            const uint64_t t = a[x] ^ a[x + 5] ^ a[x + 10];
            row[25 + x] = t;
            row[30 + x] = c[x] = t ^ a[x + 15] ^ a[x + 20];
        }
        for (uint32_t x = 0; x < 5; x++) d[x] = c[(x + 4) % 5] ^ rotl64(c[(x + 1) % 5], 1);
        for (uint32_t x = 0; x < 5; x++)
            for (uint32_t y = 0; y < 5; y++) {
                row[x + 5 * y] = a[x + 5 * y];
                b[y + 5 * ((2 * x + 3 * y) % 5)] = rotl64(a[x + 5 * y] ^ d[x], tb.rho[x + 5 * y]);
            }
        for (uint32_t i = 0; i < 25; i++) row[35 + i] = b[i];
        for (uint32_t y = 0; y < 5; y++)
            for (uint32_t x = 0; x < 5; x++) a[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
        a[0] ^= tb.rc[r];
    }
    for (uint32_t i = 0; i < KF_LANES; i++) row[i] = i < 25 ? a[i] : 0;         // row 24: the output state
}
__global__ void k_keccak_expand(uint32_t* data, const uint64_t* __restrict__ rows, uint32_t n, uint32_t A, uint32_t K, NoiseKey nk) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x, col = blockIdx.y;
    if (r >= n) return;
    uint32_t v = 0;
    if (r >= A) v = noise_cell(nk, GROUP_DATA, col, r);
    else if (r < KF_BLOCK * K) v = ((rows[(size_t)r * KF_LANES + (col >> 6)] >> (col & 63)) & 1) ? R1 : 0;
    data[(size_t)col * n + r] = v;
}
KeccakTables keccak_tables() {
    KeccakTables tb{};
    uint32_t x = 1, y = 0;
    for (uint32_t t = 0; t < 24; t++) {
        tb.rho[x + 5 * y] = (uint8_t)(((t + 1) * (t + 2) / 2) % 64);
        const uint32_t nx = y, ny = (2 * x + 3 * y) % 5;
        x = nx; y = ny;
    }
    uint32_t reg = 1;                                        // LFSR x^8 + x^6 + x^5 + x^4 + 1 (FIPS 202 algorithm 5)
    for (uint32_t i = 0; i < KF_ROUNDS; i++)
        for (uint32_t j = 0; j < 7; j++) {
            if (reg & 1) tb.rc[i] |= 1ull << ((1u << j) - 1);
            reg <<= 1;
            if (reg & 0x100) reg ^= 0x171;
        }
    return tb;
}


// ---- P2-JOIN witness (zeth_amd/circuits/p2_join.py; CPU twin: oracle/p2join.c) ----
// Every 31 active rows are one Poseidon2 permutation laid out round by round (S[24], Q[24] = (S + rc)^3); block 0 hashes the
// two child claims into the parent claim, the other blocks hash (parent ‖ public sibling words).  Stands in for the in-circuit
// hashing of risc0-circuit-recursion 4.0.2 (un-vendored: /root/reference/Cargo.lock:5305).  A literal round-by-round
// permutation (the trace needs every intermediate state); tab = Montgomery words of rc[24 * 29] then diag[24].
constexpr uint32_t PJ_T = 24, PJ_HALF = 4, PJ_RP = 21, PJ_ROUNDS = 29, PJ_BLOCK = 31;
__device__ __forceinline__ bool pj_is_full(uint32_t rnd) { return rnd < PJ_HALF || rnd >= PJ_HALF + PJ_RP; }
__device__ void pj_m_ext(uint32_t (&c)[PJ_T]) {
    uint32_t sums[4] = {0, 0, 0, 0};
    for (uint32_t b = 0; b < PJ_T; b += 4) {
        m4(c[b], c[b + 1], c[b + 2], c[b + 3]);
        for (uint32_t i = 0; i < 4; i++) sums[i] = add_mod(sums[i], c[b + i]);
    }
    for (uint32_t k = 0; k < PJ_T; k++) c[k] = add_mod(c[k], sums[k & 3]);
}
__global__ void k_p2join_code(uint32_t* code, uint32_t n, uint32_t A, uint32_t K, const uint32_t* __restrict__ tab) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x, col = blockIdx.y;
    if (r >= n) return;
    uint32_t v = 0;
    const bool in_blocks = r < PJ_BLOCK * K;
    const uint32_t k = r % PJ_BLOCK;
    const bool round_row = in_blocks && k >= 1 && k <= PJ_ROUNDS;
    const bool full = round_row && pj_is_full(k - 1), part = round_row && !pj_is_full(k - 1);
    switch (col) {
    case 0: v = r < A ? R1 : 0; break;
    case 1: v = r == 0 ? R1 : 0; break;
    case 2: v = (r > 0 && r < A) ? R1 : 0; break;
    case 3: v = (in_blocks && r == 0) ? R1 : 0; break;
    case 4: v = (in_blocks && k == 0 && r > 0) ? R1 : 0; break;
    case 5: v = (in_blocks && k == 1) ? R1 : 0; break;
    case 6: v = full ? R1 : 0; break;
    case 7: v = part ? R1 : 0; break;
    case 8: v = (in_blocks && k >= 2 && pj_is_full(k - 2)) ? R1 : 0; break;
    case 9: v = (in_blocks && k >= 2 && !pj_is_full(k - 2)) ? R1 : 0; break;
    case 10: v = (in_blocks && r == PJ_BLOCK - 1) ? R1 : 0; break;
    default:
        if (col < 35) { if (full || (part && col == 11)) v = tab[(k - 1) * PJ_T + (col - 11)]; }
        else if (in_blocks && k == 0 && r > 0) v = syn_cell(SYN_CODE_SEED, GROUP_CODE, col, r);
    }
    code[(size_t)col * n + r] = v;
}
// one lane per block: p0 .. p0 + count; block 0 takes `children` (16 words) and leaves the parent in parent_out, blocks
// p >= 1 read the parent (written by an earlier launch) and their sibling words from the code group
__global__ void k_p2join_blocks(uint32_t* data, const uint32_t* __restrict__ code, uint32_t n, uint32_t p0, uint32_t count,
                                const uint32_t* __restrict__ children, uint32_t* parent, const uint32_t* __restrict__ tab) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const uint32_t p = p0 + i;
    const size_t r0 = (size_t)PJ_BLOCK * p;
    uint32_t s[PJ_T];
    for (uint32_t j = 0; j < PJ_T; j++) s[j] = 0;
    if (p == 0) { for (uint32_t j = 0; j < 16; j++) s[j] = children[j]; }
    else {
        for (uint32_t j = 0; j < 8; j++) { s[j] = parent[j]; s[8 + j] = code[(size_t)(35 + j) * n + r0]; }
    }
    for (uint32_t j = 0; j < PJ_T; j++) { data[(size_t)j * n + r0] = s[j]; data[(size_t)(PJ_T + j) * n + r0] = 0; }
    pj_m_ext(s);
    const uint32_t* diag = tab + PJ_T * PJ_ROUNDS;
    for (uint32_t rnd = 0; rnd < PJ_ROUNDS; rnd++) {
        const size_t r = r0 + 1 + rnd;
        for (uint32_t j = 0; j < PJ_T; j++) data[(size_t)j * n + r] = s[j];
        if (pj_is_full(rnd)) {
            for (uint32_t j = 0; j < PJ_T; j++) {
                const uint32_t u = add_mod(s[j], tab[rnd * PJ_T + j]);
                const uint32_t q = mul_mod(mul_mod(u, u), u);
                data[(size_t)(PJ_T + j) * n + r] = q;
                s[j] = mul_mod(mul_mod(q, q), u);
            }
            pj_m_ext(s);
        } else {
            const uint32_t u = add_mod(s[0], tab[rnd * PJ_T]);
            const uint32_t q = mul_mod(mul_mod(u, u), u);
            data[(size_t)PJ_T * n + r] = q;
            for (uint32_t j = 1; j < PJ_T; j++) data[(size_t)(PJ_T + j) * n + r] = 0;
            const uint32_t x7 = mul_mod(mul_mod(q, q), u);
            uint32_t tot = x7;
            for (uint32_t j = 1; j < PJ_T; j++) tot = add_mod(tot, s[j]);
            s[0] = add_mod(tot, mul_mod(diag[0], x7));
            for (uint32_t j = 1; j < PJ_T; j++) s[j] = add_mod(tot, mul_mod(diag[j], s[j]));
        }
    }
    const size_t rl = r0 + PJ_BLOCK - 1;
    for (uint32_t j = 0; j < PJ_T; j++) { data[(size_t)j * n + rl] = s[j]; data[(size_t)(PJ_T + j) * n + rl] = 0; }
    if (p == 0) for (uint32_t j = 0; j < 8; j++) parent[j] = s[j];
}
// rows past the last block: zero while active, blinding noise after
__global__ void k_p2join_tail(uint32_t* data, uint32_t n, uint32_t A, uint32_t first_row, NoiseKey nk) {
    const uint32_t r = first_row + blockIdx.x * blockDim.x + threadIdx.x, col = blockIdx.y;
    if (r >= n) return;
    data[(size_t)col * n + r] = r < A ? 0u : noise_cell(nk, GROUP_DATA, col, r);
}

}  // namespace

// =====================================================================================================
extern "C" const char* zkh_circuit_load(zkh_ctx* ctx, const uint32_t* d, size_t words, zkh_circuit** out) {
    ZKH_REQUIRE(words >= DESC_HEADER && d[0] == DESC_MAGIC && d[1] == 1 && d[2] == 3 && d[6] == 2, "circuit desc: bad header");
    zkh_circuit* c = new zkh_circuit();
    c->ctx = ctx;
    c->desc.assign(d, d + words);
    c->hash = desc_hash64(d, words);
    for (int g = 0; g < 3; g++) c->group_size[g] = d[3 + g];
    c->global_size[0] = d[7]; c->global_size[1] = d[8];
    const uint32_t n_taps = d[9], n_combos = d[10], n_steps = d[11];
    c->ret = d[12]; c->kind = d[13];
    size_t pos = DESC_HEADER;
    auto fail = [&](const char* msg) { delete c; return make_err("circuit desc: %s", msg); };
    if (pos + 3ull * n_taps > words) return fail("truncated taps");
    for (uint32_t i = 0; i < n_taps; i++, pos += 3) {
        Tap t{d[pos], d[pos + 1], d[pos + 2]};
        if (t.group > 2 || t.offset >= c->group_size[t.group]) return fail("tap out of range");
        c->taps.push_back(t);
    }
    c->tot_combo_backs = 0;
    for (uint32_t i = 0; i < n_combos; i++) {
        if (pos >= words || pos + 1 + d[pos] > words) return fail("truncated combos");
        uint32_t cnt = d[pos++];
        c->combos.emplace_back(d + pos, d + pos + cnt);
        pos += cnt; c->tot_combo_backs += cnt;
    }
    if (pos + 5ull * n_steps > words) return fail("truncated steps");
    for (uint32_t i = 0; i < n_steps; i++, pos += 5) c->steps.push_back(Step{d[pos], d[pos + 1], d[pos + 2], d[pos + 3], d[pos + 4]});
    // registers
    for (size_t i = 0; i < c->taps.size();) {
        size_t j = i;
        while (j < c->taps.size() && c->taps[j].group == c->taps[i].group && c->taps[j].offset == c->taps[i].offset) j++;
        Reg r{c->taps[i].group, c->taps[i].offset, (uint32_t)i, (uint32_t)(j - i), 0xffffffffu};
        for (size_t k = 0; k < c->combos.size(); k++) {
            if (c->combos[k].size() != r.size) continue;
            bool same = true;
            for (uint32_t m = 0; m < r.size; m++) same &= c->combos[k][m] == c->taps[i + m].back;
            if (same) { r.combo_id = (uint32_t)k; break; }
        }
        if (r.combo_id == 0xffffffffu) return fail("register without combo");
        c->regs.push_back(r);
        i = j;
    }
    for (int g = 0; g < 3; g++) {
        size_t cnt = 0;
        for (auto& r : c->regs) cnt += r.group == (uint32_t)g;
        if (cnt != c->group_size[g]) return fail("every column of a group needs a register");
    }
    // ---- analyse the step list: value types, static mix exponents, liveness, slot allocation ----
    const size_t n_steps_all = c->steps.size();
    struct V { bool is_mix; uint32_t id; };
    std::vector<V> def(n_steps_all);
    uint32_t nf = 0, nm = 0;
    for (size_t i = 0; i < n_steps_all; i++) {
        if (c->steps[i].op >= OP_TRUE) def[i] = {true, nm++}; else def[i] = {false, nf++};
    }
    // per value var: Fp4-valued?  operand form (slot / tap / const / global)?
    std::vector<uint8_t> f_ext(nf, 0), f_kind(nf, OPK_FP);
    std::vector<uint32_t> f_imm(nf, 0), mix_exp(nm, 0);
    std::vector<int> fp_last(nf, -1), mix_last(nm, -1);            // last step that reads var
    uint32_t max_pow = 0;
    {
        uint32_t cf = 0, cm = 0;
        for (size_t i = 0; i < n_steps_all; i++) {
            const Step& s = c->steps[i];
            auto usef = [&](uint32_t v) -> bool { if (v >= cf) return false; fp_last[v] = (int)i; return true; };
            auto usem = [&](uint32_t v) -> bool { if (v >= cm) return false; mix_last[v] = (int)i; return true; };
            bool ok = true;
            switch (s.op) {
            case OP_CONST: f_kind[cf] = OPK_CONST; f_imm[cf] = fp_encode(s.a).v; break;
            case OP_GET_GLOBAL:
                ok = s.a <= 1 && s.b < c->global_size[s.a] && s.b < 65536;
                f_kind[cf] = OPK_GLOBAL; f_imm[cf] = (s.a << 16) | s.b; break;
            case OP_GET: ok = s.a < c->taps.size(); f_kind[cf] = OPK_TAP; f_imm[cf] = s.a; break;
            case OP_CONST_EXT: f_ext[cf] = 1; f_kind[cf] = OPK_EXT; break;
            case OP_ADD: case OP_SUB: case OP_MUL:
                ok = usef(s.a) && usef(s.b);
                if (ok) { f_ext[cf] = f_ext[s.a] | f_ext[s.b]; f_kind[cf] = f_ext[cf] ? OPK_EXT : OPK_FP; }
                break;
            case OP_TRUE: mix_exp[cm] = 0; break;
            case OP_AND_EQZ: ok = usem(s.a) && usef(s.b); if (ok) { mix_exp[cm] = mix_exp[s.a] + 1; max_pow = std::max(max_pow, mix_exp[s.a]); } break;
            case OP_AND_COND: ok = usem(s.a) && usef(s.b) && usem(s.c);
                if (ok) { mix_exp[cm] = mix_exp[s.a] + mix_exp[s.c]; max_pow = std::max(max_pow, mix_exp[s.a]); } break;
            default: ok = false;
            }
            if (!ok) return fail(s.op == OP_GET_GLOBAL ? "global out of range" : "step operand out of range");
            if (s.op >= OP_TRUE) cm++; else cf++;
        }
        if (c->ret >= nm) return fail("ret out of range");
        mix_last[c->ret] = (int)n_steps_all;
    }
    c->n_mix_pows = max_pow + 1;
    {
        // slots: fp slots for Fp results of arithmetic; wide slots for mix totals and Fp4-valued values.  Dead steps are
        // dropped (a value var is dead if nothing reads it; liveness is not transitive here, which only costs slots).
        std::vector<uint32_t> fp_slot(nf, 0), mix_slot(nm, 0), free_f, free_w;
        uint32_t nfs = 0, nws = 0, cf = 0, cm = 0;
        std::vector<std::vector<uint32_t>> rel_f(n_steps_all + 1), rel_m(n_steps_all + 1);
        for (uint32_t v = 0; v < nf; v++) if (fp_last[v] >= 0 && (f_kind[v] == OPK_FP || f_kind[v] == OPK_EXT)) rel_f[fp_last[v]].push_back(v);
        for (uint32_t v = 0; v < nm; v++) if (mix_last[v] >= 0 && mix_last[v] < (int)n_steps_all) rel_m[mix_last[v]].push_back(v);
        auto take = [](std::vector<uint32_t>& fl, uint32_t& n) { if (fl.empty()) return n++; uint32_t x = fl.back(); fl.pop_back(); return x; };
        auto operand = [&](uint32_t v, uint32_t& kind) -> uint32_t {
            kind = f_kind[v];
            return (kind == OPK_FP || kind == OPK_EXT) ? fp_slot[v] : f_imm[v];
        };
        for (size_t i = 0; i < n_steps_all; i++) {
            const Step& s = c->steps[i];
            const bool is_mix = s.op >= OP_TRUE;
            const bool inline_operand = !is_mix && (s.op == OP_CONST || s.op == OP_GET || s.op == OP_GET_GLOBAL);
            const bool dead = is_mix ? (mix_last[cm] < 0) : (fp_last[cf] < 0);
            if (!dead && !inline_operand) {
                InterpInsn in{s.op, 0, 0, 0, 0, 0};
                uint32_t ka = 0, kb = 0;
                if (is_mix) { in.dst = take(free_w, nws); mix_slot[cm] = in.dst; }
                else if (f_ext[cf]) { in.dst = take(free_w, nws); fp_slot[cf] = in.dst; in.op |= 1u << 14; }
                else { in.dst = take(free_f, nfs); fp_slot[cf] = in.dst; }
                switch (s.op) {
                case OP_CONST_EXT: in.a = fp_encode(s.a).v; in.b = fp_encode(s.b).v; in.c = fp_encode(s.c).v; in.w = fp_encode(s.d).v; break;
                case OP_ADD: case OP_SUB: case OP_MUL: in.a = operand(s.a, ka); in.b = operand(s.b, kb); break;
                case OP_TRUE: break;
                case OP_AND_EQZ: in.a = mix_slot[s.a]; in.b = operand(s.b, kb); in.w = mix_exp[s.a]; break;
                case OP_AND_COND: in.a = mix_slot[s.a]; in.b = operand(s.b, kb); in.c = mix_slot[s.c]; in.w = mix_exp[s.a]; break;
                }
                in.op |= (ka << 8) | (kb << 11);
                c->prog.push_back(in);
            }
            // NOTE: a destination slot may not alias a source released by the same step: release AFTER allocating
            for (uint32_t v : rel_f[i]) (f_ext[v] ? free_w : free_f).push_back(fp_slot[v]);
            for (uint32_t v : rel_m[i]) free_w.push_back(mix_slot[v]);
            if (is_mix) cm++; else cf++;
        }
        c->n_fp_slots = nfs ? nfs : 1; c->n_mix_slots = nws ? nws : 1;
        c->ret_slot = mix_slot[c->ret];
    }
    const size_t lds = ((size_t)c->n_fp_slots * 4 + (size_t)c->n_mix_slots * 16) * INTERP_THREADS;
    // a step list too large for the interpreter still loads: it then needs a compiled kernel (built in, or attached)
    c->interp_ok = lds <= 160 * 1024;
    c->compiled = find_compiled_eval_check(c->hash);
    c->d_prog = nullptr; c->d_taps = nullptr;
    if (ctx) {       // ctx == NULL: host-only circuit (enough for zkh_verify_segment, which needs no GPU)
        bind_thread(ctx);
        static_assert(sizeof(InterpInsn) == INSN_WORDS * 4, "insn layout");
        hipError_t e = hipMalloc((void**)&c->d_prog, c->prog.size() * sizeof(InterpInsn) + 4);
        if (e == hipSuccess) e = hipMemcpy(c->d_prog, c->prog.data(), c->prog.size() * sizeof(InterpInsn), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMalloc((void**)&c->d_taps, c->taps.size() * sizeof(Tap) + 4);
        if (e == hipSuccess) e = hipMemcpy(c->d_taps, c->taps.data(), c->taps.size() * sizeof(Tap), hipMemcpyHostToDevice);
        if (e != hipSuccess) { const char* m = hipGetErrorString(e); return fail(m); }
        if (c->compiled && c->compiled->gather_exps) {
            std::vector<const uint32_t*> lists(c->compiled->gather_exps, c->compiled->gather_exps + c->compiled->n_parts);
            std::vector<const uint32_t*> consts(c->compiled->n_parts, nullptr);
            if (c->compiled->gather_consts) consts.assign(c->compiled->gather_consts, c->compiled->gather_consts + c->compiled->n_parts);
            if (const char* err = set_gather(c, 0, lists, consts)) { std::string m(err); zkh_free_error(err); return fail(m.c_str()); }
        }
    }
    *out = c;
    return nullptr;
}
// The parts' exported exponent lists ({count, e0, e1, ...} each) -> one device list + the parts' slot offsets.  All parts or none.
// consts: per part NULL or {count, (slot, c0, c1, c2, c3) x count}: slots whose power is multiplied by an Fp4 constant (Montgomery words).
static const char* set_gather(zkh_circuit* c, int which, const std::vector<const uint32_t*>& lists, const std::vector<const uint32_t*>& consts) {
    if (c->d_gather[which]) { (void)hipFree(c->d_gather[which]); c->d_gather[which] = nullptr; }
    if (c->d_gconst[which]) { (void)hipFree(c->d_gconst[which]); c->d_gconst[which] = nullptr; }
    c->n_gconst[which] = 0;
    c->gather_centered[which] = false;
    c->gather_off[which].clear();
    size_t exported = 0;
    for (const uint32_t* l : lists) exported += l != nullptr;
    if (!exported) return nullptr;
    ZKH_REQUIRE(exported == lists.size(), "eval_check kernels: %zu of %zu parts export a gathered power table", exported, lists.size());
    std::vector<uint32_t> all, off(1, 0), recs;
    for (size_t part = 0; part < lists.size(); part++) {
        const uint32_t* l = lists[part];
        for (uint32_t i = 0; i < l[0]; i++) {            // (bit 31: the kernel reads this slot centred)
            ZKH_REQUIRE((l[1 + i] & 0x7fffffffu) < c->n_mix_pows, "eval_check kernels: gathered mix power %u, the step list has %u", l[1 + i] & 0x7fffffffu, c->n_mix_pows);
            all.push_back(l[1 + i]);
            if (l[1 + i] >> 31) c->gather_centered[which] = true;
        }
        const uint32_t* k = part < consts.size() ? consts[part] : nullptr;
        for (uint32_t i = 0; k && i < k[0]; i++) {
            const uint32_t* r = k + 1 + 5 * (size_t)i;
            ZKH_REQUIRE(r[0] < l[0], "eval_check kernels: a slot constant names slot %u of a part with %u slots", r[0], l[0]);
            recs.push_back(off.back() + r[0]);
            for (int j = 1; j <= 4; j++) { ZKH_REQUIRE(r[j] < P, "eval_check kernels: a slot constant is not a reduced word"); recs.push_back(r[j]); }
        }
        off.push_back((uint32_t)all.size());
    }
    hipError_t e = hipMalloc((void**)&c->d_gather[which], all.size() * 4 + 4);
    if (e == hipSuccess) e = hipMemcpy(c->d_gather[which], all.data(), all.size() * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess && !recs.empty()) {
        e = hipMalloc((void**)&c->d_gconst[which], recs.size() * 4);
        if (e == hipSuccess) e = hipMemcpy(c->d_gconst[which], recs.data(), recs.size() * 4, hipMemcpyHostToDevice);
        c->n_gconst[which] = (uint32_t)(recs.size() / 5);
    }
    if (e != hipSuccess) return make_err("eval_check kernels: gathered power table: %s", hipGetErrorString(e));
    c->gather_off[which] = std::move(off);
    return nullptr;
}
extern "C" void zkh_circuit_destroy(zkh_circuit* c) {
    if (!c) return;
    if (c->ctx) bind_thread(c->ctx);
    for (int w = 0; w < 2; w++) if (c->d_gather[w]) (void)hipFree(c->d_gather[w]);
    for (int w = 0; w < 2; w++) if (c->d_gconst[w]) (void)hipFree(c->d_gconst[w]);
    if (c->d_prog) (void)hipFree(c->d_prog);
    if (c->d_taps) (void)hipFree(c->d_taps);
    for (hipModule_t m : c->jit_modules) if (m) (void)hipModuleUnload(m);
    delete c;
}
static bool jit_complete(const zkh_circuit* c) {
    if (c->jit_kernels.empty()) return false;
    for (hipFunction_t f : c->jit_kernels) if (!f) return false;
    return true;
}
extern "C" int zkh_circuit_has_compiled_kernel(const zkh_circuit* c) { return jit_complete(c) ? 2 : (c->compiled != nullptr ? 1 : 0); }
extern "C" size_t zkh_circuit_compiled_parts(const zkh_circuit* c) {
    return jit_complete(c) ? c->jit_kernels.size() : (c->compiled ? c->compiled->n_parts : 0);
}

// Attach a gfx950 code object holding `extern "C" __global__ void <kernel_name>(EvalCheckArgs)` generated for this
// circuit's step list (zeth_amd/circuits/jit.py produces it with the same generator the build uses).  Upstream ships
// one machine-generated kernel per circuit; a circuit that arrives as data gets the same treatment at load time.
extern "C" const char* zkh_circuit_attach_code_object_part(zkh_circuit* c, const void* image, size_t len, const char* kernel_name,
                                                           size_t part, size_t n_parts) {
    ZKH_REQUIRE(c && c->ctx, "attach_code_object: circuit was loaded without a device context");
    ZKH_REQUIRE(image && len >= 64 && kernel_name, "attach_code_object: empty code object");
    ZKH_REQUIRE(n_parts >= 1 && n_parts <= 4096 && part < n_parts, "attach_code_object: part %zu of %zu", part, n_parts);
    ZKH_REQUIRE(memcmp(image, "\x7f" "ELF", 4) == 0 || memcmp(image, "__CLANG_OFFLOAD_BUNDLE__", 24) == 0,
                "attach_code_object: not an ELF code object or offload bundle");
    bind_thread(c->ctx);
    hipModule_t mod = nullptr;
    hipFunction_t fn = nullptr;
    hipError_t e = hipModuleLoadData(&mod, image);
    // a failed module call leaves its code in the thread's last-error slot; clear it so the next launch check is clean
    if (e != hipSuccess) { (void)hipGetLastError(); return make_err("attach_code_object: hipModuleLoadData: %s", hipGetErrorString(e)); }
    e = hipModuleGetFunction(&fn, mod, kernel_name);
    if (e != hipSuccess) { (void)hipGetLastError(); (void)hipModuleUnload(mod); return make_err("attach_code_object: no kernel '%s': %s", kernel_name, hipGetErrorString(e)); }
    // a kernel generated with a gathered power table exports `<kernel>_exps` = {count, exponent of slot 0, 1, ...} and, when some
    // slots hold a power times an Fp4 constant, `<kernel>_pwc` = {count, (slot, c0, c1, c2, c3) x count}
    std::vector<uint32_t> exps, pwc;
    {
        hipDeviceptr_t dptr = nullptr;
        size_t bytes = 0;
        const std::string sym = std::string(kernel_name) + "_pwc";
        if (hipModuleGetGlobal(&dptr, &bytes, mod, sym.c_str()) == hipSuccess && bytes >= 4 && bytes % 4 == 0) {
            pwc.resize(bytes / 4);
            e = hipMemcpy(pwc.data(), dptr, bytes, hipMemcpyDeviceToHost);
            if (e != hipSuccess || (size_t)pwc[0] * 5 != pwc.size() - 1) {
                (void)hipGetLastError(); (void)hipModuleUnload(mod);
                return make_err("attach_code_object: '%s' is malformed", sym.c_str());
            }
        } else {
            (void)hipGetLastError();
        }
    }
    {
        hipDeviceptr_t dptr = nullptr;
        size_t bytes = 0;
        const std::string sym = std::string(kernel_name) + "_exps";
        if (hipModuleGetGlobal(&dptr, &bytes, mod, sym.c_str()) == hipSuccess && bytes >= 4 && bytes % 4 == 0) {
            exps.resize(bytes / 4);
            e = hipMemcpy(exps.data(), dptr, bytes, hipMemcpyDeviceToHost);
            if (e != hipSuccess || exps[0] != exps.size() - 1) {
                (void)hipGetLastError(); (void)hipModuleUnload(mod);
                return make_err("attach_code_object: '%s' is malformed", sym.c_str());
            }
        } else {
            (void)hipGetLastError();
        }
    }
    if (c->jit_kernels.size() != n_parts) {          // a new set of parts replaces whatever was attached before
        for (hipModule_t m : c->jit_modules) if (m) (void)hipModuleUnload(m);
        c->jit_modules.assign(n_parts, nullptr);
        c->jit_kernels.assign(n_parts, nullptr);
        c->jit_exps.assign(n_parts, {});
        c->jit_pwc.assign(n_parts, {});
    }
    if (c->jit_modules[part]) (void)hipModuleUnload(c->jit_modules[part]);
    c->jit_modules[part] = mod; c->jit_kernels[part] = fn; c->jit_exps[part] = std::move(exps); c->jit_pwc[part] = std::move(pwc);
    c->jit_mixed = false;
    if (jit_complete(c)) {
        std::vector<const uint32_t*> lists, consts;
        size_t exported = 0;
        for (const auto& l : c->jit_exps) { lists.push_back(l.empty() ? nullptr : l.data()); exported += !l.empty(); }
        for (const auto& l : c->jit_pwc) consts.push_back(l.empty() ? nullptr : l.data());
        // a set being replaced part by part passes through states where some parts carry a table and some do not: such a set
        // is never launched (zkh_eval_check refuses it), but attaching the remaining parts repairs it
        if (exported && exported != lists.size()) {
            c->jit_mixed = true;
            std::vector<const uint32_t*> none(lists.size(), nullptr);
            (void)set_gather(c, 1, none, none);
        } else if (const char* err = set_gather(c, 1, lists, consts)) {
            c->jit_kernels[part] = nullptr;
            return err;
        }
    }
    return nullptr;
}
extern "C" const char* zkh_circuit_attach_code_object(zkh_circuit* c, const void* image, size_t len, const char* kernel_name) {
    return zkh_circuit_attach_code_object_part(c, image, len, kernel_name, 0, 1);
}

extern "C" const char* zkh_eval_check(zkh_ctx* ctx, const zkh_circuit* c, zkh_buf* check, const zkh_buf* const* groups, size_t n_groups,
                                      const zkh_buf* const* globals, size_t n_globals, const uint32_t poly_mix[4], size_t po2,
                                      size_t steps, int use_interpreter) {
    ZKH_REQUIRE(c->ctx == ctx && c->d_prog, "eval_check: circuit was not loaded on this context");
    ZKH_REQUIRE(n_groups == 3 && n_globals == 2, "eval_check: expected 3 register groups (accum, code, data) and 2 global groups (out, mix), got %zu / %zu",
                n_groups, n_globals);
    const size_t n = (size_t)1 << po2, dom = n * ZKH_INV_RATE;
    ZKH_REQUIRE(steps == n, "eval_check: steps %zu != 2^po2 = %zu", steps, n);
    ZKH_REQUIRE(po2 + 2 <= (size_t)MAX_LOG_N, "eval_check: po2 %zu too large", po2);
    ZKH_REQUIRE(check->len == ZKH_EXT_SIZE * dom, "eval_check: check buffer must hold 4 x 4n words");
    for (int g = 0; g < 3; g++)
        ZKH_REQUIRE(groups[g]->len == (size_t)c->group_size[g] * dom, "eval_check: group %d has %zu words, expected %zu", g,
                    groups[g]->len, (size_t)c->group_size[g] * dom);
    for (int g = 0; g < 2; g++) ZKH_REQUIRE(globals[g]->len >= c->global_size[g], "eval_check: global %d too small", g);
    EvalCheckArgs a{};
    a.check = check->ptr();
    for (int g = 0; g < 3; g++) a.groups[g] = groups[g]->ptr();
    for (int g = 0; g < 2; g++) a.globals[g] = globals[g]->ptr();
    a.dom = (uint32_t)dom;
    // (3 w^idx)^n = 3^n * i^(idx mod 4), i = ROU_FWD[2]
    const Fp three_n = fp_pow(fp_encode(3), n), i4 = Fp::raw(ctx->rou_fwd[2]);
    Fp cur = Fp::one();
    for (int k = 0; k < 4; k++) { a.zinv[k] = fp_inv(three_n * cur - Fp::one()).v; cur = cur * i4; }
    // the mix powers: the plain table mix^0, mix^1, ... — or, for kernels generated that way, every part's own table, gathered
    // into the order its code reads them (one launch either way)
    const int which = jit_complete(c) ? 1 : 0;
    ZKH_REQUIRE(use_interpreter || which == 0 || !c->jit_mixed, "eval_check: the attached kernels disagree — some export a gathered power table, some do not (attach every part of ONE generated set)");
    const std::vector<uint32_t>* goff = (!use_interpreter && (which == 1 || c->compiled) && !c->gather_off[which].empty()) ? &c->gather_off[which] : nullptr;
    Tmp pows;
    if (goff) {
        ZKH_TRY(new_buf(ctx, 4 * (size_t)goff->back() + 4, false, pows.out()));
        ZKH_TRY(launch_ext_powers_at(ctx, pows->ptr(), poly_mix, c->d_gather[which], goff->back()));
        ZKH_TRY(launch_ext_scale_at(ctx, pows->ptr(), c->d_gconst[which], c->n_gconst[which]));
        if (c->gather_centered[which]) ZKH_TRY(launch_ext_center_at(ctx, pows->ptr(), c->d_gather[which], goff->back()));
    } else {
        ZKH_TRY(new_buf(ctx, 4 * (size_t)c->n_mix_pows, false, pows.out()));
        const uint32_t one[4] = {R1, 0, 0, 0};
        ZKH_TRY(launch_ext_powers(ctx, pows->ptr(), one, poly_mix, c->n_mix_pows));
    }
    a.mix_pows = pows->ptr();
    size_t total_w = 0;
    for (int g = 0; g < 3; g++) total_w += c->group_size[g];
    if (jit_complete(c) && !use_interpreter) {
        ProfScope prof(ctx, "eval_check", 4.0 * total_w * dom + 16.0 * dom);
        for (size_t part = 0; part < c->jit_kernels.size(); part++) {
            a.accumulate = part != 0;
            if (goff) a.mix_pows = pows->ptr() + 4 * (size_t)(*goff)[part];
            size_t arg_size = sizeof(a);
            void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &arg_size, HIP_LAUNCH_PARAM_END};
            const hipError_t e = hipModuleLaunchKernel(c->jit_kernels[part], (unsigned)((dom + 255) / 256), 1, 1, 256, 1, 1, 0, ctx->stream,
                                                       nullptr, config);
            if (e != hipSuccess) return make_err("eval_check: launching attached kernel %zu: %s", part, hipGetErrorString(e));
        }
    } else if (c->compiled && !use_interpreter) {
        ProfScope prof(ctx, "eval_check", 4.0 * total_w * dom + 16.0 * dom);
        for (uint32_t part = 0; part < c->compiled->n_parts; part++) {
            a.accumulate = part != 0;
            if (goff) a.mix_pows = pows->ptr() + 4 * (size_t)(*goff)[part];
            c->compiled->parts[part](a, ctx->stream);
        }
    } else {
        if (!c->interp_ok)
            return make_err("eval_check: the step list has more live values than the interpreter's LDS holds and no compiled kernel is attached");
        const size_t lds = ((size_t)c->n_fp_slots * 4 + (size_t)c->n_mix_slots * 16) * INTERP_THREADS;
        ProfScope prof(ctx, "eval_check_interp", 4.0 * total_w * dom + 16.0 * dom);
        if (lds > 64 * 1024) {
            const hipError_t e = hipFuncSetAttribute((const void*)k_eval_check_interp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return make_err("eval_check: %zu bytes of LDS for the interpreter: %s", lds, hipGetErrorString(e));
        }
        k_eval_check_interp<<<(unsigned)((dom + INTERP_THREADS - 1) / INTERP_THREADS), INTERP_THREADS, lds, ctx->stream>>>(
            a, c->d_prog, (uint32_t)c->prog.size(), c->d_taps, c->n_fp_slots, c->ret_slot);
    }
    return last_launch_error("eval_check");
}

// ---- SYN-AIR witness ----
// ---- built-in witness generators: circuit kind 1 = SYN-AIR, kind 2 = KECCAK-F ----
static const char* keccak_check_shape(const zkh_circuit* c) {
    ZKH_REQUIRE(c->group_size[GROUP_CODE] == 15 && c->group_size[GROUP_DATA] == KF_LANES * 64 && c->group_size[GROUP_ACCUM] == 4 &&
                c->global_size[GLOBAL_OUT] == 200, "keccak witgen: the circuit does not have KECCAK-F's shape (15 / 3840 / 4 columns, 200 outputs)");
    return nullptr;
}
static const char* p2join_tables(zkh_ctx* ctx, Tmp& tab) {          // Montgomery words of the SHIPPED tables: rc[24 * 29] then diag[24]
    std::vector<uint32_t> t(PJ_T * PJ_ROUNDS + PJ_T);
    for (uint32_t i = 0; i < PJ_T * PJ_ROUNDS; i++) t[i] = fp_encode(ZKH_P2_ROUND_CONSTANTS[i]).v;
    for (uint32_t i = 0; i < PJ_T; i++) t[PJ_T * PJ_ROUNDS + i] = fp_encode(ZKH_P2_M_INT_DIAG[i]).v;
    return zkh_copy_from(ctx, "p2join_tables", t.data(), t.size(), tab.out());
}
static const char* p2join_check_shape(const zkh_circuit* c) {
    ZKH_REQUIRE(c->group_size[GROUP_CODE] == 43 && c->group_size[GROUP_DATA] == 2 * PJ_T && c->group_size[GROUP_ACCUM] == 4 &&
                c->global_size[GLOBAL_OUT] == 24, "p2join witgen: the circuit does not have P2-JOIN's shape (43 / 48 / 4 columns, 24 outputs)");
    return nullptr;
}
// inclusive prefix sum (mod P) of the first A words of a device column, in place; *last_out = the grand total (device word)
const char* zkh::prefix_sum_column(zkh_ctx* ctx, uint32_t* col, uint32_t A, uint32_t* last_out) {
    const unsigned chunks = (A + 1023) / 1024;
    Tmp totals;
    ZKH_TRY(new_buf(ctx, chunks, false, totals.out()));
    k_prefix_sum_chunks<<<chunks, 1024, 0, ctx->stream>>>(col, A, totals->ptr());
    k_prefix_sum_fp<<<1, 1024, 0, ctx->stream>>>(totals->ptr(), chunks, last_out);
    k_prefix_sum_carry<<<chunks, 1024, 0, ctx->stream>>>(col, A, totals->ptr());
    return last_launch_error("prefix_sum_column");
}
extern "C" const char* zkh_syn_code(zkh_ctx* ctx, const zkh_circuit* c, size_t po2, size_t zk_cycles, zkh_buf* code) {
    ZKH_REQUIRE(c->kind >= 1 && c->kind <= 3, "syn_code: no built-in witness generator for circuit kind %u", c->kind);
    const size_t n = (size_t)1 << po2;
    ZKH_REQUIRE(n > zk_cycles + 1, "syn_code: po2 too small for zk_cycles");
    const uint32_t wc = c->group_size[GROUP_CODE], A = (uint32_t)(n - zk_cycles);
    ZKH_REQUIRE(code->len == (size_t)wc * n, "syn_code: buffer shape mismatch");
    if (c->kind == 2) {
        ZKH_TRY(keccak_check_shape(c));
        ProfScope prof(ctx, "keccak_code", 4.0 * wc * n);
        k_keccak_code<<<dim3((unsigned)((n + 255) / 256), wc), 256, 0, ctx->stream>>>(code->ptr(), (uint32_t)n, A, A / KF_BLOCK, keccak_tables());
        return last_launch_error("keccak_code");
    }
    if (c->kind == 3) {
        ZKH_TRY(p2join_check_shape(c));
        Tmp tab;
        ZKH_TRY(p2join_tables(ctx, tab));
        ProfScope prof(ctx, "p2join_code", 4.0 * wc * n);
        k_p2join_code<<<dim3((unsigned)((n + 255) / 256), wc), 256, 0, ctx->stream>>>(code->ptr(), (uint32_t)n, A, A / PJ_BLOCK, tab->ptr());
        return last_launch_error("p2join_code");
    }
    ProfScope prof(ctx, "syn_code", 4.0 * wc * n);
    k_syn_code<<<dim3((unsigned)((n + 255) / 256), wc), 256, 0, ctx->stream>>>(code->ptr(), wc, (uint32_t)n, A, SYN_CODE_SEED);
    return last_launch_error("syn_code");
}
extern "C" const char* zkh_syn_witgen(zkh_ctx* ctx, const zkh_circuit* c, size_t po2, size_t zk_cycles, uint64_t seed,
                                      const uint32_t* noise_key, const uint32_t* pub, zkh_buf* code, zkh_buf* data, uint32_t* out_global) {
    ZKH_REQUIRE(c->kind >= 1 && c->kind <= 3, "syn_witgen: no built-in witness generator for circuit kind %u", c->kind);
    NoiseKey nk;
    ZKH_TRY(resolve_noise_key(noise_key, &nk));
    const size_t n = (size_t)1 << po2;
    ZKH_REQUIRE(n > zk_cycles + 1, "syn_witgen: po2 too small for zk_cycles");
    const uint32_t wc = c->group_size[GROUP_CODE], wd = c->group_size[GROUP_DATA], A = (uint32_t)(n - zk_cycles);
    if (c->kind == 3) {
        // P2-JOIN: `pub` = the two child claims (16 Montgomery words, required); out_global = parent ‖ left ‖ right
        ZKH_TRY(p2join_check_shape(c));
        ZKH_REQUIRE(code && code->len == (size_t)wc * n && data->len == (size_t)wd * n, "p2join witgen: buffer shape mismatch (the code group is an input of this generator)");
        ZKH_REQUIRE(pub, "p2join witgen: the two child claims (16 words) are required");
        for (uint32_t k = 0; k < 16; k++) ZKH_REQUIRE(pub[k] < P, "p2join witgen: child claim word %u is not a reduced element", k);
        const uint32_t K = A / PJ_BLOCK;
        ZKH_REQUIRE(K > 0, "p2join witgen: no room for a permutation (31 rows) in %u active rows", A);
        ZKH_TRY(zkh_syn_code(ctx, c, po2, zk_cycles, code));
        Tmp tab, kids, parent;
        ZKH_TRY(p2join_tables(ctx, tab));
        ZKH_TRY(zkh_copy_from(ctx, "children", pub, 16, kids.out()));
        ZKH_TRY(new_buf(ctx, 8, false, parent.out()));
        {
            ProfScope prof(ctx, "p2join_blocks", 4.0 * wd * n);
            k_p2join_blocks<<<1, 64, 0, ctx->stream>>>(data->ptr(), code->ptr(), (uint32_t)n, 0, 1, kids->ptr(), parent->ptr(), tab->ptr());
            if (K > 1) k_p2join_blocks<<<(K - 1 + 63) / 64, 64, 0, ctx->stream>>>(data->ptr(), code->ptr(), (uint32_t)n, 1, K - 1, kids->ptr(), parent->ptr(), tab->ptr());
            const uint32_t first = PJ_BLOCK * K;
            if (first < n) k_p2join_tail<<<dim3((unsigned)((n - first + 255) / 256), wd), 256, 0, ctx->stream>>>(data->ptr(), (uint32_t)n, A, first, nk);
        }
        ZKH_TRY(last_launch_error("p2join_witgen"));
        memcpy(out_global + 8, pub, 64);
        return zkh_read(ctx, parent, out_global, 0, 8);
    }
    if (c->kind == 2) {
        // KECCAK-F: `pub` = optional input state of the LAST permutation (25 lanes = 50 words, low word first); out_global =
        // that permutation's output state as 100 16-bit limbs, then its input state as 100 more (what the `final` / `bind` rows' constraints bind)
        ZKH_TRY(keccak_check_shape(c));
        ZKH_REQUIRE((!code || code->len == (size_t)wc * n) && data->len == (size_t)wd * n, "keccak witgen: buffer shape mismatch");
        const uint32_t K = A / KF_BLOCK;
        ZKH_REQUIRE(K > 0, "keccak witgen: no room for a permutation (25 rows) in %u active rows", A);
        if (code) ZKH_TRY(zkh_syn_code(ctx, c, po2, zk_cycles, code));
        Tmp rows, din;
        ZKH_TRY(new_buf(ctx, (size_t)K * KF_BLOCK * KF_LANES * 2, false, rows.out()));
        if (pub) ZKH_TRY(zkh_copy_from(ctx, "keccak_input", pub, 50, din.out()));
        const KeccakTables tb = keccak_tables();
        {
            ProfScope prof(ctx, "keccak_perm", 8.0 * K * KF_BLOCK * KF_LANES);
            k_keccak_perm<<<(K + 63) / 64, 64, 0, ctx->stream>>>((uint64_t*)rows->ptr(), K, seed, din ? (const uint64_t*)din->ptr() : nullptr, tb);
        }
        {
            ProfScope prof(ctx, "keccak_expand", 4.0 * wd * n);
            k_keccak_expand<<<dim3((unsigned)((n + 255) / 256), wd), 256, 0, ctx->stream>>>(data->ptr(), (const uint64_t*)rows->ptr(), (uint32_t)n, A, K, nk);
        }
        ZKH_TRY(last_launch_error("keccak_witgen"));
        // out = the last permutation's OUTPUT state, then its INPUT state (row 0 of the last block), as 16-bit limbs: the claim binds the pair
        uint32_t fin[50], first[50];
        ZKH_TRY(zkh_read(ctx, rows, fin, ((size_t)K * KF_BLOCK - 1) * KF_LANES * 2, 50));
        ZKH_TRY(zkh_read(ctx, rows, first, ((size_t)(K - 1) * KF_BLOCK) * KF_LANES * 2, 50));
        for (uint32_t l = 0; l < 25; l++)
            for (uint32_t j = 0; j < 4; j++) {
                out_global[4 * l + j] = fp_encode((fin[2 * l + (j >> 1)] >> (16 * (j & 1))) & 0xffffu).v;
                out_global[100 + 4 * l + j] = fp_encode((first[2 * l + (j >> 1)] >> (16 * (j & 1))) & 0xffffu).v;
            }
        return nullptr;
    }
    const uint32_t n_pub = c->global_size[GLOBAL_OUT] - 4;
    ZKH_REQUIRE((!code || code->len == (size_t)wc * n) && data->len == (size_t)wd * n, "syn_witgen: buffer shape mismatch");
    ZKH_REQUIRE(n_pub == 0 || pub, "syn_witgen: the circuit has %u public input words but none were given", n_pub);
    for (uint32_t k = 0; k < n_pub; k++) ZKH_REQUIRE(pub[k] < P, "syn_witgen: public input %u is not a reduced element", k);
    if (code) ZKH_TRY(zkh_syn_code(ctx, c, po2, zk_cycles, code));       // NULL: the caller holds this size's code group already (resident)
    Tmp last, dpub;
    ZKH_TRY(new_buf(ctx, 1, false, last.out()));
    if (n_pub) ZKH_TRY(zkh_copy_from(ctx, "pub", pub, n_pub, dpub.out()));
    const unsigned bx = (unsigned)((n + 255) / 256);
    k_syn_data<<<bx, 256, 0, ctx->stream>>>(data->ptr(), wd, (uint32_t)n, A, seed, nk, dpub ? dpub->ptr() : nullptr, n_pub);
    ZKH_TRY(prefix_sum_column(ctx, data->ptr() + (size_t)(wd - 1) * n, A, last->ptr()));
    ZKH_TRY(last_launch_error("syn_witgen"));
    out_global[1] = out_global[2] = out_global[3] = 0;
    for (uint32_t k = 0; k < n_pub; k++) out_global[4 + k] = pub[k];
    return zkh_read(ctx, last, out_global, 0, 1);
}
extern "C" const char* zkh_syn_chain_contributions(zkh_ctx* ctx, const zkh_circuit* c, const uint64_t* seeds, const uint32_t* po2s, size_t n,
                                                   size_t zk_cycles, uint32_t* contributions) {
    ZKH_REQUIRE(ctx && c && seeds && po2s && contributions, "syn_chain_contributions: null argument");
    ZKH_REQUIRE(circuit_has_state(c), "syn_chain_contributions: the circuit is not a SYN-C / SYN-S circuit (kind 1 whose first public input is the pre-state)");
    if (!n) return nullptr;
    for (size_t i = 0; i < n; i++) ZKH_REQUIRE(po2s[i] >= 1 && po2s[i] <= 24 && ((size_t)1 << po2s[i]) > zk_cycles + 1, "syn_chain_contributions: segment %zu: po2 out of range", i);
    Tmp dseeds, dpo2, dout;
    ZKH_TRY(zkh_copy_from(ctx, "chain_seeds", (const uint32_t*)seeds, 2 * n, dseeds.out()));
    ZKH_TRY(zkh_copy_from(ctx, "chain_po2", po2s, n, dpo2.out()));
    ZKH_TRY(new_buf(ctx, n, false, dout.out()));
    k_syn_chain_contrib<<<(unsigned)n, 1024, 0, ctx->stream>>>(dout->ptr(), (const uint64_t*)dseeds->ptr(), dpo2->ptr(), (uint32_t)zk_cycles);
    ZKH_TRY(last_launch_error("syn_chain_contrib"));
    return zkh_read(ctx, dout, contributions, 0, n);
}
extern "C" const char* zkh_syn_accum(zkh_ctx* ctx, const zkh_circuit* c, size_t po2, size_t zk_cycles, const uint32_t* noise_key,
                                     const zkh_buf* data, const uint32_t* mix_global, zkh_buf* accum) {
    ZKH_REQUIRE(c->kind >= 1 && c->kind <= 3, "syn_accum: no built-in accum witness generator for circuit kind %u", c->kind);
    NoiseKey nk;
    ZKH_TRY(resolve_noise_key(noise_key, &nk));
    const size_t n = (size_t)1 << po2;
    const uint32_t wa = c->group_size[GROUP_ACCUM], wd = c->group_size[GROUP_DATA], A = (uint32_t)(n - zk_cycles), k = wa / 4;
    ZKH_REQUIRE(accum->len == (size_t)wa * n && data->len == (size_t)wd * n, "syn_accum: buffer shape mismatch");
    Tmp mix, terms;
    ZKH_TRY(zkh_copy_from(ctx, "mix", mix_global, wa, mix.out()));
    ZKH_TRY(new_buf(ctx, 4 * (size_t)k * n, false, terms.out()));
    const unsigned bx = (unsigned)((n + 255) / 256);
    {
        ProfScope prof(ctx, "syn_accum_terms", (4.0 + 16.0) * k * n);
        k_syn_accum_terms<<<dim3(bx, k), 256, 0, ctx->stream>>>(terms->ptr(), data->ptr(), mix->ptr(), wd, (uint32_t)n, A);
    }
    ZKH_TRY(prefix_products_batched(ctx, terms->ptr(), n, k, 4 * n));      // all k running products in one set of launches
    {
        ProfScope prof(ctx, "syn_accum_store", (16.0 + 16.0) * k * n);
        k_syn_accum_store<<<dim3(bx, k), 256, 0, ctx->stream>>>(accum->ptr(), terms->ptr(), (uint32_t)n, A, nk);
    }

    return last_launch_error("syn_accum");
}
