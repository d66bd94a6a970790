// recursion.hip — the RECURSION circuit on the device: program loader, witness generator, copy-argument accumulators and the
// lift / join seal (BASELINE.json config 5).  Stands in for risc0-circuit-recursion 4.0.2
// src/prove/{mod.rs, program.rs} + its witness generator (un-vendored: /root/reference/Cargo.lock:5305), which
// risc0_zkvm::default_prover().prove (/root/reference/crates/host/src/lib.rs:137) runs once per lift and per join to turn
// a block's segment receipts into one succinct receipt.  The circuit and the program format: zeth_amd/circuits/recursion.py;
// the programs (the STARK verifier itself): zeth_amd/circuits/rec_verify.py; CPU twin: oracle/recursion.c.
//
// MI355X mapping.  A program is a DAG of ~2 x 10^5 field micro-ops and ~10^4 Poseidon2 permutations whose depth is only a
// few hundred (50 queries x ~8 Merkle paths run side by side; the transcript sponge is the long chain).  At load time the
// ops are sorted into dependency levels; a witness is one launch per level with one lane per op (ops of a level sorted by
// opcode so that waves stay uniform; a permutation is one lane running the lazily reduced form of hash.hip), values in a
// device array of Fp4.  The trace is then written by three coalesced kernels (wires; Poseidon2 blocks round by round;
// blinding rows), the copy argument by one term kernel + the batched prefix product of poly.hip.  The code group depends on
// the program only: it is committed once at load and stays resident (461 MB for a po2-19 program), as upstream keeps the
// control columns of lift / join.
#include <algorithm>
#include <memory>

#include "../../include/zkh_poseidon2_consts.h"
#include "circuit.h"
#include "poseidon2.h"
#include "poseidon2_wide.h"

using namespace zkh;

namespace zkh {
const char* prefix_products_batched(zkh_ctx* c, uint32_t* io, size_t n0, size_t count, size_t col_stride);
}

namespace {

constexpr uint32_t RC_T = 24, RC_BLOCK = 12, RC_NW = 6, RC_WD = 72, RC_WA = 12, RC_WC = 58, RC_ROW = 13, RC_ROUNDS = 29;
constexpr uint32_t RC_MAGIC = 0x5a4b5231u, RC_VERSION = 2, RC_HEADER = 16, RC_OP_WORDS = 8;
enum : uint32_t { RO_INPUT = 1, RO_GEN, RO_MUX, RO_PACK, RO_UNPACK, RO_INV, RO_BITS, RO_P2, RO_EQ, RO_ISZ };
enum : uint32_t { RG_MUX = 1, RG_BOOL = 2, RG_EMB = 4, RG_PACK0 = 8, RG_PUB = 128, RG_SWAP = 256 };

__device__ __forceinline__ Fp4 ld(const uint4* val, uint32_t v) {
    const uint4 x = val[v];
    return Fp4(Fp::raw(x.x), Fp::raw(x.y), Fp::raw(x.z), Fp::raw(x.w));
}
__device__ __forceinline__ void st(uint4* val, uint32_t v, const Fp4& x) { val[v] = make_uint4(x.c[0].v, x.c[1].v, x.c[2].v, x.c[3].v); }
__device__ __forceinline__ uint32_t comp(const uint4& x, uint32_t j) { return j == 0 ? x.x : j == 1 ? x.y : j == 2 ? x.z : x.w; }

// one op of the witness schedule; WITH_P2: a permutation on ONE lane is among them (the wide-level kernel; the persistent runs
// give permutations eight lanes and leave the single-lane form, with its ~90 live registers, out of their budget)
template <bool WITH_P2>
__device__ __forceinline__ void rec_exec_op(const uint32_t* __restrict__ o, uint4* val, const uint32_t* __restrict__ consts,
                                            const uint32_t* __restrict__ inputs, uint32_t* fail, const uint32_t* __restrict__ rc,
                                            const uint32_t* __restrict__ diag) {
    const uint32_t op = o[0] & 0xff, aux = (o[0] >> 8) & 0xff, out = o[1];
    switch (op) {
    case RO_INPUT: {
        uint32_t w[4] = {0, 0, 0, 0};
        bool ok = true;
        for (uint32_t t = 0; t < aux && t < 4; t++) { w[t] = inputs[o[2] + t]; ok &= w[t] < P; }
        if (!ok) atomicMin(fail, o[7]);
        val[out] = make_uint4(w[0], w[1], w[2], w[3]);
        break;
    }
    case RO_GEN: {
        const uint32_t* q = consts + o[5];
        const Fp4 a = ld(val, o[2]), b = ld(val, o[3]), c = ld(val, o[4]);
        Fp4 r = q[0] ? (a * b) * Fp::raw(q[0]) : Fp4::zero();
        r = r + a * Fp::raw(q[1]) + b * Fp::raw(q[2]) + c * Fp::raw(q[3]);
        r.c[0] = r.c[0] + Fp::raw(q[4]);
        st(val, out, r);
        break;
    }
    case RO_MUX: {
        const Fp4 a = ld(val, o[2]), b = ld(val, o[3]), c = ld(val, o[4]);
        st(val, out, b + (c - b) * a.c[0]);
        break;
    }
    case RO_PACK: {
        const uint32_t j = aux & 3;
        val[out] = make_uint4(comp(val[o[2]], j), comp(val[o[3]], j), comp(val[o[4]], j), comp(val[o[5]], j));
        break;
    }
    case RO_UNPACK: {
        const uint4 x = val[o[2]];
        val[out] = make_uint4(x.x, 0, 0, 0); val[out + 1] = make_uint4(x.y, 0, 0, 0);
        val[out + 2] = make_uint4(x.z, 0, 0, 0); val[out + 3] = make_uint4(x.w, 0, 0, 0);
        break;
    }
    case RO_INV: {
        const Fp4 a = ld(val, o[2]);
        if (!(a.c[0].v | a.c[1].v | a.c[2].v | a.c[3].v)) atomicMin(fail, o[7]);
        else st(val, out, fp4_inv(a));
        break;
    }
    case RO_ISZ: {
        const uint32_t a = val[o[2]].x;
        val[out] = make_uint4(a ? fp_inv(Fp::raw(a)).v : 0u, 0, 0, 0);
        break;
    }
    case RO_BITS: {
        const uint32_t x = fp_decode(Fp::raw(val[o[2]].x));
        for (uint32_t t = 0; t < 31; t++) val[out + t] = make_uint4(((x >> t) & 1) ? R1 : 0u, 0, 0, 0);
        break;
    }
    case RO_P2: if (WITH_P2) {                 // one lane, the lazily reduced form of hash.hip (levels wide enough to fill the chip)
        uint32_t s[CELLS];
        for (uint32_t w = 0; w < RC_NW; w++) { const uint4 x = val[o[2 + w]]; s[4 * w] = x.x; s[4 * w + 1] = x.y; s[4 * w + 2] = x.z; s[4 * w + 3] = x.w; }
        if (aux & 1) {                         // conditional swap: cell 16 is the bit (asserted boolean by the gate that made it)
            if (s[16]) for (uint32_t j = 0; j < 8; j++) { const uint32_t x = s[j]; s[j] = s[j + 8]; s[j + 8] = x; }
            s[16] = 0;
        }
        poseidon2_mix(s, rc, diag);
        for (uint32_t w = 0; w < RC_NW; w++) val[out + w] = make_uint4(s[4 * w], s[4 * w + 1], s[4 * w + 2], s[4 * w + 3]);
        break;
    } else { atomicMin(fail, 1u); break; }
    case RO_EQ: {
        const uint4 a = val[o[2]], b = val[o[3]];
        if (a.x != b.x || a.y != b.y || a.z != b.z || a.w != b.w) atomicMin(fail, o[7]);
        break;
    }
    default: atomicMin(fail, o[7]);
    }
}
// lane j (of eight) of a permutation op loads its wire: with the op's swap flag and the bit (wire 4, cell 0) set, lanes 0..3
// read the other digest's wires; lane 4 drops the bit from its first cell (recursion.py swap_state)
__device__ __forceinline__ void rec_p2_load(const uint32_t* o, const uint4* val, uint32_t j, uint32_t (&c)[4]) {
    c[0] = c[1] = c[2] = c[3] = 0;
    if (j >= 6) return;
    const bool swap = (o[0] >> 8) & 1;
    uint32_t src = j;
    if (swap && j < 4 && val[o[6]].x) src = j ^ 2;
    const uint4 x = val[o[2 + src]];
    c[0] = (swap && j == 4) ? 0u : x.x; c[1] = x.y; c[2] = x.z; c[3] = x.w;
}
// a wide dependency level: lane i executes op lo + i
__global__ __launch_bounds__(64) void k_rec_level(const uint32_t* __restrict__ ops, uint32_t lo, uint32_t hi, uint4* val,
                                                  const uint32_t* __restrict__ consts, const uint32_t* __restrict__ inputs, uint32_t* fail,
                                                  const uint32_t* __restrict__ rc, const uint32_t* __restrict__ diag) {
    const uint32_t i = lo + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= hi) return;
    rec_exec_op<true>(ops + (size_t)RC_OP_WORDS * i, val, consts, inputs, fail, rc, diag);
}
// A RUN of narrow levels [l0, l1) in ONE workgroup of 1024 lanes: the level chain of a verifier (the transcript sponge, the
// Horner and constraint chains) is hundreds of levels a few ops wide, and a launch per level costs ~10 us each.  Here a level is
// a barrier; its permutations run on eight lanes each (poseidon2_wide.h: ~3.5 x lower latency than one lane), in whole waves
// after the lanes of the other ops.  lv[l] = {lo, p2lo, p2hi, hi}: ops [lo, p2lo) and [p2hi, hi) are not permutations, [p2lo,
// p2hi) are (ops of a level are sorted by opcode).  The host only puts a level in a run if its lanes fit: non-permutation ops
// rounded up to a wave + 8 per permutation <= 1024.
__global__ __launch_bounds__(1024) void k_rec_run(const uint32_t* __restrict__ ops, const uint4* __restrict__ lv, uint32_t l0, uint32_t l1,
                                                  uint4* val, const uint32_t* __restrict__ consts, const uint32_t* __restrict__ inputs,
                                                  uint32_t* fail, const uint32_t* __restrict__ rc, const uint32_t* __restrict__ diag) {
    __shared__ __attribute__((aligned(16))) uint32_t rcs[ROUNDS_TOTAL * CELLS + 8];
    for (uint32_t w = threadIdx.x; w < ROUNDS_TOTAL * CELLS; w += blockDim.x) rcs[w] = rc[w];
    __syncthreads();
    const uint32_t t = threadIdx.x;
    // what this lane does in a level: nothing, op `i` on its own, or lane `j` of the permutation `i`; the op record (two uint4)
    // of the NEXT level is fetched while the current one executes - the schedule is static, only `val` carries dependencies
    auto role = [&](const uint4 r, uint32_t& i, bool& p2, bool& live) -> bool {
        const uint32_t n_a = r.y - r.x, n_other = n_a + (r.w - r.z), first_p2 = (n_other + 63) & ~63u, n_p2 = r.z - r.y;
        p2 = false; live = true;
        if (t < n_other) { i = t < n_a ? r.x + t : r.z + (t - n_a); return true; }
        if (t >= first_p2 && ((t - first_p2) >> 6) * 8 < n_p2) {                 // wave-uniform: waves with at least one permutation
            const uint32_t g = (t - first_p2) >> 3;
            p2 = true; live = g < n_p2;
            i = r.y + (live ? g : n_p2 - 1);
            return true;
        }
        return false;
    };
    const uint4* ops4 = (const uint4*)ops;
    uint32_t i = 0;
    bool p2 = false, live = false;
    bool busy = role(lv[l0], i, p2, live);
    uint4 oa = make_uint4(0, 0, 0, 0), ob = oa;
    if (busy) { oa = ops4[2 * (size_t)i]; ob = ops4[2 * (size_t)i + 1]; }
    for (uint32_t l = l0; l < l1; l++) {
        uint32_t ni = 0;
        bool np2 = false, nlive = false, nbusy = false;
        uint4 na = make_uint4(0, 0, 0, 0), nb = na;
        if (l + 1 < l1) {
            nbusy = role(lv[l + 1], ni, np2, nlive);
            if (nbusy) { na = ops4[2 * (size_t)ni]; nb = ops4[2 * (size_t)ni + 1]; }
        }
        if (busy) {
            const uint32_t o[RC_OP_WORDS] = {oa.x, oa.y, oa.z, oa.w, ob.x, ob.y, ob.z, ob.w};
            if (!p2) rec_exec_op<false>(o, val, consts, inputs, fail, rc, diag);
            else {
                const uint32_t j = t & 7;
                uint32_t c[4];
                rec_p2_load(o, val, j, c);
                wide_permute(c, j, rcs, diag);
                if (live && j < 6) val[o[1] + j] = make_uint4(c[0], c[1], c[2], c[3]);
            }
        }
        __threadfence_block();
        __syncthreads();
        busy = nbusy; i = ni; p2 = np2; live = nlive; oa = na; ob = nb;
    }
}
// the permutations of a wide level, eight lanes each
__global__ __launch_bounds__(256) void k_rec_p2_wide(const uint32_t* __restrict__ ops, uint32_t lo, uint32_t hi, uint4* val,
                                                     const uint32_t* __restrict__ rc, const uint32_t* __restrict__ diag) {
    __shared__ __attribute__((aligned(16))) uint32_t rcs[ROUNDS_TOTAL * CELLS + 8];
    for (uint32_t w = threadIdx.x; w < ROUNDS_TOTAL * CELLS; w += blockDim.x) rcs[w] = rc[w];
    __syncthreads();
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x, g = gid >> 3, j = gid & 7, n = hi - lo;
    if ((gid >> 6) * 8 >= n) return;                                             // wave-uniform
    const bool live = g < n;
    const uint32_t* o = ops + (size_t)RC_OP_WORDS * (lo + (live ? g : n - 1));
    uint32_t c[4];
    rec_p2_load(o, val, j, c);
    wide_permute(c, j, rcs, diag);
    if (live && j < 6) val[o[1] + j] = make_uint4(c[0], c[1], c[2], c[3]);
}

// wires: one lane per (row, wire); rows past A are blinding noise (all 72 data columns: blockIdx.y < 6 covers W, the rest below)
__global__ void k_rec_fill_wires(uint32_t* data, const uint32_t* __restrict__ pos, const uint4* __restrict__ val, uint32_t n, uint32_t A,
                                 NoiseKey nk) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x, w = blockIdx.y;
    if (r >= n) return;
    uint4 x = make_uint4(0, 0, 0, 0);
    if (r < A) {
        const uint32_t v = pos[(size_t)r * RC_NW + w];
        if (v) x = val[v - 1];
    } else {
        x = make_uint4(noise_cell(nk, GROUP_DATA, 4 * w, r), noise_cell(nk, GROUP_DATA, 4 * w + 1, r),
                       noise_cell(nk, GROUP_DATA, 4 * w + 2, r), noise_cell(nk, GROUP_DATA, 4 * w + 3, r));
    }
    data[(size_t)(4 * w) * n + r] = x.x; data[(size_t)(4 * w + 1) * n + r] = x.y;
    data[(size_t)(4 * w + 2) * n + r] = x.z; data[(size_t)(4 * w + 3) * n + r] = x.w;
}
// S / Q columns outside the blocks: zero while active, noise after
__global__ void k_rec_fill_tail(uint32_t* data, uint32_t n, uint32_t A, uint32_t first_row, NoiseKey nk) {
    const uint32_t r = first_row + blockIdx.x * blockDim.x + threadIdx.x, col = RC_T + blockIdx.y;
    if (r >= n) return;
    data[(size_t)col * n + r] = r < A ? 0u : noise_cell(nk, GROUP_DATA, col, r);
}
__device__ void rc_m_ext(uint32_t (&c)[RC_T]) {
    uint32_t sums[4] = {0, 0, 0, 0};
    for (uint32_t b = 0; b < RC_T; b += 4) {
        m4(c[b], c[b + 1], c[b + 2], c[b + 3]);
        for (uint32_t i = 0; i < 4; i++) sums[i] = add_mod(sums[i], c[b + i]);
    }
    for (uint32_t k = 0; k < RC_T; k++) c[k] = add_mod(c[k], sums[k & 3]);
}
// one lane per 12-row block: the permutation of the block's input row (zeth_amd/circuits/recursion.py block_rows): rows 1..4 and
// 7..10 one full round each (S, Q = cubes), rows 5 / 6 twelve / nine partial rounds ((Q_i, X_i) pairs in the Q columns), row 11
// the output and the output row's wires; tab = Montgomery words of rc[24 * 29] then diag[24]
__global__ void k_rec_blocks(uint32_t* data, uint32_t n, uint32_t K, const uint32_t* __restrict__ tab, const uint32_t* __restrict__ table) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= K) return;
    const size_t r0 = (size_t)RC_BLOCK * p;
    uint32_t* S = data + (size_t)RC_T * n;
    uint32_t* Q = data + (size_t)2 * RC_T * n;
    uint32_t s[RC_T];
    for (uint32_t j = 0; j < RC_T; j++) s[j] = data[(size_t)j * n + r0];
    if (table[r0 * RC_ROW + 6] & RG_SWAP) {            // a conditional-swap block: the wires' cell 16 is the bit
        if (s[16]) for (uint32_t j = 0; j < 8; j++) { const uint32_t x = s[j]; s[j] = s[j + 8]; s[j + 8] = x; }
        s[16] = 0;
    }
    for (uint32_t j = 0; j < RC_T; j++) { S[(size_t)j * n + r0] = s[j]; Q[(size_t)j * n + r0] = 0; }
    rc_m_ext(s);
    const uint32_t* diag = tab + RC_T * RC_ROUNDS;
    uint32_t rnd = 0;
    size_t r = r0 + 1;
    for (uint32_t half = 0; half < 2; half++) {
        for (uint32_t f = 0; f < 4; f++, rnd++, r++) {
            for (uint32_t j = 0; j < RC_T; j++) {
                S[(size_t)j * n + r] = s[j];
                const uint32_t u = add_mod(s[j], tab[rnd * RC_T + j]);
                const uint32_t q = mul_mod(mul_mod(u, u), u);
                Q[(size_t)j * n + r] = q;
                s[j] = mul_mod(mul_mod(q, q), u);
            }
            rc_m_ext(s);
        }
        if (half) break;
        for (uint32_t m = 12; m >= 9; m -= 3, r++) {
            for (uint32_t j = 0; j < RC_T; j++) S[(size_t)j * n + r] = s[j];
            for (uint32_t i = 0; i < m; i++, rnd++) {
                const uint32_t u = add_mod(s[0], tab[rnd * RC_T]);
                const uint32_t q = mul_mod(mul_mod(u, u), u), x7 = mul_mod(mul_mod(q, q), u);
                Q[(size_t)(2 * i) * n + r] = q; Q[(size_t)(2 * i + 1) * n + r] = x7;
                uint32_t tot = x7;
                for (uint32_t j = 1; j < RC_T; j++) tot = add_mod(tot, s[j]);
                s[0] = add_mod(tot, mul_mod(diag[0], x7));
                for (uint32_t j = 1; j < RC_T; j++) s[j] = add_mod(tot, mul_mod(diag[j], s[j]));
            }
            for (uint32_t j = 2 * m; j < RC_T; j++) Q[(size_t)j * n + r] = 0;
        }
    }
    for (uint32_t j = 0; j < RC_T; j++) { S[(size_t)j * n + r] = s[j]; Q[(size_t)j * n + r] = 0; data[(size_t)j * n + r] = s[j]; }
}
// the code group: one lane per (row, column)
__global__ void k_rec_code(uint32_t* code, const uint32_t* __restrict__ table, uint32_t n, uint32_t A, uint32_t K, const uint32_t* __restrict__ tab) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x, col = blockIdx.y;
    if (r >= n) return;
    uint32_t v = 0;
    if (r < A) {
        const uint32_t* row = table + (size_t)r * RC_ROW;
        const bool in_blocks = r < RC_BLOCK * K;
        const uint32_t k = r % RC_BLOCK;
        const bool full = in_blocks && ((k >= 1 && k <= 4) || (k >= 7 && k <= 10));
        const uint32_t rnd = k <= 4 ? k - 1 : k == 5 ? 4 : k == 6 ? 16 : k + 18;         // first round this row performs
        if (col == 0) v = R1;
        else if (col == 1) v = r == 0 ? R1 : 0;
        else if (col == 2) v = r > 0 ? R1 : 0;
        else if (col == 3) v = r == A - 1 ? R1 : 0;
        else if (col == 4) v = mul_mod(R2, r);
        else if (col < 11) v = mul_mod(R2, row[7 + (col - 5)]);
        else if (col < 17) v = mul_mod(R2, row[col - 11]);
        else if (col == 17) v = (row[6] & RG_MUX) ? R1 : 0;
        else if (col == 18) v = (row[6] & RG_BOOL) ? R1 : 0;
        else if (col == 19) v = (row[6] & RG_EMB) ? R1 : 0;
        else if (col < 24) v = (row[6] & (RG_PACK0 << (col - 20))) ? R1 : 0;
        else if (col == 24) v = (in_blocks && ((k == 0 && !(row[6] & RG_SWAP)) || k == RC_BLOCK - 1)) ? R1 : 0;
        else if (col == 57) v = (in_blocks && k == 0 && (row[6] & RG_SWAP)) ? R1 : 0;
        else if (col == 25) v = (row[6] & RG_PUB) ? R1 : 0;
        else if (!in_blocks) v = 0;
        else if (col == 26) v = k == 1 ? R1 : 0;
        else if (col == 27) v = full ? R1 : 0;
        else if (col == 28) v = k == 5 ? R1 : 0;
        else if (col == 29) v = k == 6 ? R1 : 0;
        else if (col == 30) v = ((k >= 2 && k <= 5) || (k >= 8 && k <= 11)) ? R1 : 0;
        else if (col == 31) v = k == 6 ? R1 : 0;
        else if (col == 32) v = k == 7 ? R1 : 0;
        else if (full) v = tab[rnd * RC_T + (col - 33)];
        else if ((k == 5 && col - 33 < 12) || (k == 6 && col - 33 < 9)) v = tab[(rnd + (col - 33)) * RC_T];
    }
    code[(size_t)col * n + r] = v;
}
// copy argument: terms[k][r] = prod_{w in {2k, 2k+1}} F(id_w, W_w) / F(sigma_w, W_w)  (active rows) or 1; AoS ExtElems
__global__ void k_rec_accum_terms(uint32_t* terms, const uint32_t* __restrict__ table, const uint32_t* __restrict__ data,
                                  const uint32_t* __restrict__ mix, uint32_t n, uint32_t A) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x, k = blockIdx.y;
    if (r >= n) return;
    Fp4 t = Fp4::one();
    if (r < A) {
        Fp4 beta[4], gamma;
        for (int i = 0; i < 4; i++) beta[i] = Fp4(Fp::raw(mix[4 * i]), Fp::raw(mix[4 * i + 1]), Fp::raw(mix[4 * i + 2]), Fp::raw(mix[4 * i + 3]));
        gamma = Fp4(Fp::raw(mix[16]), Fp::raw(mix[17]), Fp::raw(mix[18]), Fp::raw(mix[19]));
        Fp4 num = Fp4::one(), den = Fp4::one();
        for (uint32_t w = 2 * k; w < 2 * k + 2; w++) {
            Fp4 f = gamma;
            for (int i = 0; i < 4; i++) f = f + beta[i] * Fp::raw(data[(size_t)(4 * w + i) * n + r]);
            Fp4 fi = f, fs = f;
            fi.c[0] = fi.c[0] + fp_encode(RC_NW * r + w);
            fs.c[0] = fs.c[0] + fp_encode(table[(size_t)r * RC_ROW + 7 + w]);
            num = num * fi; den = den * fs;
        }
        t = num * fp4_inv(den);
    }
    ((uint4*)terms)[(size_t)k * n + r] = make_uint4(t.c[0].v, t.c[1].v, t.c[2].v, t.c[3].v);
}
__global__ void k_rec_accum_store(uint32_t* accum, const uint32_t* prods, uint32_t n, uint32_t A, NoiseKey nk) {
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
__global__ void k_rec_gather_pub(uint32_t* out, const uint32_t* __restrict__ data, uint32_t n, uint32_t row) {
    const uint32_t i = threadIdx.x;
    if (i < 16) out[i] = data[(size_t)i * n + row];
}

}  // namespace

struct zkh_rec_program {
    zkh_ctx* ctx = nullptr;
    const zkh_circuit* circuit = nullptr;
    zkh_prover* prover = nullptr;
    uint32_t po2 = 0, zk = 0, A = 0, n_vars = 0, n_consts = 0, n_ops = 0, n_inputs = 0, n_p2 = 0, n_gates = 0, pub_row = 0xffffffffu;
    zkh_buf *d_table = nullptr, *d_pos = nullptr, *d_consts = nullptr, *d_ops = nullptr, *d_tab = nullptr;
    std::vector<uint32_t> level_ptr;
    std::vector<uint32_t> lv;                      // per level: lo, p2lo, p2hi, hi
    struct Step { uint32_t l0, l1; bool run; };
    std::vector<Step> plan;                        // runs of narrow levels (one launch) and single wide levels
    zkh_buf* d_lv = nullptr;
    // the witness schedule as a hipGraph: the launches of `plan` with this program's own value / input / failure buffers as
    // constant arguments, instantiated once at load and replayed per witness (one graph launch instead of ~130 kernel launches)
    zkh_buf *d_val = nullptr, *d_in = nullptr, *d_fail = nullptr;
    hipGraphExec_t graph = nullptr;
    uint32_t root[8] = {0};
    uint64_t hash = 0;
};

// the launches of one witness schedule on the context's stream (directly, or while the stream is being captured into a graph)
static void rec_enqueue_schedule(const zkh_rec_program* p, uint4* v, const uint32_t* din, uint32_t* fail);

extern "C" void zkh_rec_program_destroy(zkh_rec_program* p) {
    if (!p) return;
    if (p->ctx) bind_thread(p->ctx);
    if (p->graph) (void)hipGraphExecDestroy(p->graph);
    if (p->prover) zkh_prover_destroy(p->prover);
    for (zkh_buf* b : {p->d_table, p->d_pos, p->d_consts, p->d_ops, p->d_tab, p->d_lv, p->d_val, p->d_in, p->d_fail}) if (b) zkh_release(b);
    delete p;
}

static const char* rec_check_shape(const zkh_circuit* c) {
    ZKH_REQUIRE(c && c->kind == 4 && c->group_size[GROUP_CODE] == RC_WC && c->group_size[GROUP_DATA] == RC_WD && c->group_size[GROUP_ACCUM] == RC_WA &&
                c->global_size[GLOBAL_OUT] == 16 && c->global_size[GLOBAL_MIX] == 20,
                "recursion: the circuit does not have RECURSION's shape (kind 4: 58 / 72 / 12 columns, 16 outputs, 20 mix words)");
    return nullptr;
}

static void rec_enqueue_schedule(const zkh_rec_program* p, uint4* v, const uint32_t* din, uint32_t* fail) {
    zkh_ctx* c = p->ctx;
    for (const auto& st : p->plan) {
        if (st.run) {
            k_rec_run<<<1, 1024, 0, c->stream>>>(p->d_ops->ptr(), (const uint4*)p->d_lv->ptr(), st.l0, st.l1, v, p->d_consts->ptr(), din, fail,
                                                 c->tab.rc, c->tab.diag);
            continue;
        }
        for (uint32_t l = st.l0; l < st.l1; l++) {
            const uint32_t lo = p->lv[4 * l], a = p->lv[4 * l + 1], b = p->lv[4 * l + 2], hi = p->lv[4 * l + 3];
            if (a > lo) k_rec_level<<<(a - lo + 63) / 64, 64, 0, c->stream>>>(p->d_ops->ptr(), lo, a, v, p->d_consts->ptr(), din, fail, c->tab.rc, c->tab.diag);
            if (hi > b) k_rec_level<<<(hi - b + 63) / 64, 64, 0, c->stream>>>(p->d_ops->ptr(), b, hi, v, p->d_consts->ptr(), din, fail, c->tab.rc, c->tab.diag);
            if (b > a) k_rec_p2_wide<<<(8 * (b - a) + 255) / 256, 256, 0, c->stream>>>(p->d_ops->ptr(), a, b, v, c->tab.rc, c->tab.diag);
        }
    }
}

extern "C" const char* zkh_rec_code(const zkh_rec_program* p, zkh_buf* code) {
    ZKH_REQUIRE(p && code, "rec_code: null argument");
    const size_t n = (size_t)1 << p->po2;
    ZKH_REQUIRE(code->len == (size_t)RC_WC * n, "rec_code: buffer shape mismatch");
    ProfScope prof(p->ctx, "rec_code", 4.0 * RC_WC * n);
    k_rec_code<<<dim3((unsigned)((n + 255) / 256), RC_WC), 256, 0, p->ctx->stream>>>(code->ptr(), p->d_table->ptr(), (uint32_t)n, p->A,
                                                                                   p->A / RC_BLOCK, p->d_tab->ptr());
    return last_launch_error("rec_code");
}

// Loads a program blob (zeth_amd/circuits/recursion.py Program.finish): validates it, sorts the witness schedule into
// dependency levels, uploads the tables, generates the code group and commits it (resident for every later seal).
extern "C" const char* zkh_rec_program_load(zkh_ctx* ctx, const zkh_circuit* circuit, const uint32_t* b, size_t words, zkh_rec_program** out) {
    ZKH_REQUIRE(ctx && circuit && b && out, "rec_program_load: null argument");
    ZKH_REQUIRE(circuit->ctx == ctx, "rec_program_load: the circuit was not loaded on this context");
    ZKH_TRY(rec_check_shape(circuit));
    ZKH_REQUIRE(words >= RC_HEADER && b[0] == RC_MAGIC && b[1] == RC_VERSION, "rec_program_load: bad header (a version-%u program blob is expected)", RC_VERSION);
    std::unique_ptr<zkh_rec_program, void (*)(zkh_rec_program*)> p(new zkh_rec_program(), zkh_rec_program_destroy);
    p->ctx = ctx; p->circuit = circuit;
    p->po2 = b[2]; p->zk = b[3]; p->A = b[4]; p->n_vars = b[5]; p->n_consts = b[6]; p->n_ops = b[7]; p->n_inputs = b[8]; p->n_p2 = b[9]; p->n_gates = b[10];
    ZKH_REQUIRE(p->po2 >= 1 && p->po2 + 2 <= (uint32_t)MAX_LOG_N && (size_t)p->A + p->zk == (size_t)1 << p->po2 && p->A > 1, "rec_program_load: bad shape");
    const size_t A = p->A, need = RC_HEADER + A * (RC_ROW + RC_NW) + p->n_consts + (size_t)RC_OP_WORDS * p->n_ops;
    ZKH_REQUIRE(words == need, "rec_program_load: %zu words, the header describes %zu", words, need);
    ZKH_REQUIRE(p->n_p2 <= A / RC_BLOCK, "rec_program_load: more permutations than blocks");
    const uint32_t* table = b + RC_HEADER;
    const uint32_t* pos = table + A * RC_ROW;
    const uint32_t* consts = pos + A * RC_NW;
    const uint32_t* ops = consts + p->n_consts;
    for (size_t i = 0; i < A * RC_NW; i++) ZKH_REQUIRE(pos[i] <= p->n_vars, "rec_program_load: a position names an unknown variable");
    for (size_t r = 0; r < A; r++) {
        const uint32_t* row = table + r * RC_ROW;
        for (int i = 0; i < 6; i++) ZKH_REQUIRE(row[i] < P, "rec_program_load: gate coefficient out of range");
        for (int w = 0; w < 6; w++) ZKH_REQUIRE(row[7 + w] < A * RC_NW, "rec_program_load: sigma out of range");
        if ((row[6] & RG_PUB) && p->pub_row == 0xffffffffu) p->pub_row = (uint32_t)r;
    }
    // dependency levels
    std::vector<uint32_t> var_level(p->n_vars ? p->n_vars : 1, 0), op_level(p->n_ops, 0);
    uint32_t n_levels = 0;
    for (uint32_t i = 0; i < p->n_ops; i++) {
        const uint32_t* o = ops + (size_t)RC_OP_WORDS * i;
        const uint32_t op = o[0] & 0xff, out_v = o[1];
        uint32_t n_in = 0, n_out = 1;
        switch (op) {
        case RO_INPUT: ZKH_REQUIRE((o[0] >> 8) >= 1 && (o[0] >> 8) <= 4 && (size_t)o[2] + (o[0] >> 8) <= p->n_inputs, "rec_program_load: op %u reads outside the inputs", i); break;
        case RO_GEN: n_in = 3; ZKH_REQUIRE((size_t)o[5] + 5 <= p->n_consts, "rec_program_load: op %u: constant index out of range", i); break;
        case RO_MUX: n_in = 3; break;
        case RO_PACK: n_in = 4; break;
        case RO_UNPACK: n_in = 1; n_out = 4; break;
        case RO_INV: case RO_ISZ: n_in = 1; break;
        case RO_BITS: n_in = 1; n_out = 31; break;
        case RO_P2: n_in = 6; n_out = 6; ZKH_REQUIRE((o[0] >> 8) <= 1, "rec_program_load: op %u: unknown permutation variant", i); break;
        case RO_EQ: n_in = 2; n_out = 0; break;
        default: return make_err("rec_program_load: op %u has unknown opcode %u", i, op);
        }
        ZKH_REQUIRE((size_t)out_v + n_out <= p->n_vars || n_out == 0, "rec_program_load: op %u writes an unknown variable", i);
        uint32_t lvl = 0;
        for (uint32_t k = 0; k < n_in; k++) {
            const uint32_t v = o[2 + k];
            ZKH_REQUIRE(v < p->n_vars, "rec_program_load: op %u reads an unknown variable", i);
            if (op == RO_GEN && v == out_v) continue;                  // a constant: its (zero-weighted) operands name itself
            lvl = std::max(lvl, var_level[v] + 1);
        }
        op_level[i] = lvl;
        for (uint32_t k = 0; k < n_out; k++) var_level[out_v + k] = lvl;
        n_levels = std::max(n_levels, lvl + 1);
    }
    std::vector<uint32_t> order(p->n_ops);
    for (uint32_t i = 0; i < p->n_ops; i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
        if (op_level[x] != op_level[y]) return op_level[x] < op_level[y];
        return (ops[(size_t)RC_OP_WORDS * x] & 0xff) < (ops[(size_t)RC_OP_WORDS * y] & 0xff);
    });
    std::vector<uint32_t> sorted((size_t)RC_OP_WORDS * p->n_ops + 1);
    p->level_ptr.assign(n_levels + 1, 0);
    for (uint32_t k = 0; k < p->n_ops; k++) {
        const uint32_t* o = ops + (size_t)RC_OP_WORDS * order[k];
        uint32_t* d = &sorted[(size_t)RC_OP_WORDS * k];
        const uint32_t op = o[0] & 0xff;
        memcpy(d, o, 4 * RC_OP_WORDS);
        // free the last word for the op's original index + 1 (what a failed assertion reports): P2 moves its sixth input out
        // of it first (no slot is spare there, so its index goes unreported - a permutation cannot fail)
        if (op != RO_P2) d[7] = order[k] + 1;
        p->level_ptr[op_level[order[k]] + 1] = k + 1;
    }
    for (uint32_t l = 1; l <= n_levels; l++) p->level_ptr[l] = std::max(p->level_ptr[l], p->level_ptr[l - 1]);
    // per level the range of its permutations (ops are sorted by opcode inside a level), and the launch plan
    p->lv.assign(4 * (size_t)n_levels + 4, 0);
    for (uint32_t l = 0; l < n_levels; l++) {
        const uint32_t lo = p->level_ptr[l], hi = p->level_ptr[l + 1];
        uint32_t a = lo, b = lo;
        while (a < hi && (sorted[(size_t)RC_OP_WORDS * a] & 0xff) < RO_P2) a++;
        b = a;
        while (b < hi && (sorted[(size_t)RC_OP_WORDS * b] & 0xff) == RO_P2) b++;
        p->lv[4 * l] = lo; p->lv[4 * l + 1] = a; p->lv[4 * l + 2] = b; p->lv[4 * l + 3] = hi;
        const uint32_t lanes = (((a - lo) + (hi - b) + 63) & ~63u) + 8 * (b - a);
        const bool narrow = lanes <= 1024;
        if (narrow && !p->plan.empty() && p->plan.back().run) p->plan.back().l1 = l + 1;
        else p->plan.push_back({l, l + 1, narrow});
    }
    bind_thread(ctx);
    std::vector<uint32_t> kk(p->n_consts ? p->n_consts : 1, 0), tab(RC_T * RC_ROUNDS + RC_T);
    for (uint32_t i = 0; i < p->n_consts; i++) { ZKH_REQUIRE(consts[i] < P, "rec_program_load: constant out of range"); kk[i] = fp_encode(consts[i]).v; }
    for (uint32_t i = 0; i < RC_T * RC_ROUNDS; i++) tab[i] = fp_encode(ZKH_P2_ROUND_CONSTANTS[i]).v;
    for (uint32_t i = 0; i < RC_T; i++) tab[RC_T * RC_ROUNDS + i] = fp_encode(ZKH_P2_M_INT_DIAG[i]).v;
    ZKH_TRY(zkh_copy_from(ctx, "rec_table", table, A * RC_ROW, &p->d_table));
    ZKH_TRY(zkh_copy_from(ctx, "rec_pos", pos, A * RC_NW, &p->d_pos));
    ZKH_TRY(zkh_copy_from(ctx, "rec_consts", kk.data(), kk.size(), &p->d_consts));
    ZKH_TRY(zkh_copy_from(ctx, "rec_ops", sorted.data(), sorted.size(), &p->d_ops));
    ZKH_TRY(zkh_copy_from(ctx, "rec_tab", tab.data(), tab.size(), &p->d_tab));
    ZKH_TRY(zkh_copy_from(ctx, "rec_lv", p->lv.data(), p->lv.size(), &p->d_lv));
    ZKH_TRY(new_buf(ctx, 4 * (size_t)(p->n_vars ? p->n_vars : 1), false, &p->d_val));
    ZKH_TRY(new_buf(ctx, p->n_inputs ? p->n_inputs : 1, true, &p->d_in));
    ZKH_TRY(new_buf(ctx, 1, true, &p->d_fail));
    if (getenv("ZKH_REC_GRAPH")) {
        // OPT-IN (measured: no gain over launching the ~130 kernels one by one, and rocprofv3 7.2 crashes on the replayed graph):
        // capture the schedule once; a failed capture (or instantiate) leaves `graph` null and the launches direct
        ZKH_HIP(hipStreamSynchronize(ctx->stream));
        hipGraph_t g = nullptr;
        if (hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
            rec_enqueue_schedule(p.get(), (uint4*)p->d_val->ptr(), p->d_in->ptr(), p->d_fail->ptr());
            if (hipStreamEndCapture(ctx->stream, &g) == hipSuccess && g) {
                if (hipGraphInstantiate(&p->graph, g, nullptr, nullptr, 0) != hipSuccess) p->graph = nullptr;
                (void)hipGraphDestroy(g);
            }
        }
        (void)hipGetLastError();
    }
    p->hash = desc_hash64(b, words);
    // the code group: generated, committed, resident
    ZKH_TRY(zkh_prover_create(ctx, circuit, &p->prover));
    {
        Tmp code;
        ZKH_TRY(new_buf(ctx, (size_t)RC_WC << p->po2, false, code.out()));
        ZKH_TRY(zkh_rec_code(p.get(), code.b));
        ZKH_TRY(zkh_prover_cache_code(p->prover, p->po2, code.b));
        ZKH_TRY(zkh_prover_cached_code_root(p->prover, p->po2, p->root));
    }
    *out = p.release();
    return nullptr;
}

extern "C" int zkh_rec_program_has_graph(const zkh_rec_program* p) { return p && p->graph ? (int)p->plan.size() : 0; }

extern "C" const char* zkh_rec_program_info(const zkh_rec_program* p, uint32_t root[8], uint32_t info[8]) {
    ZKH_REQUIRE(p, "rec_program_info: null program");
    if (root) memcpy(root, p->root, 32);
    if (info) {
        info[0] = p->po2; info[1] = p->zk; info[2] = p->n_inputs; info[3] = p->n_p2; info[4] = p->n_gates; info[5] = p->n_ops;
        info[6] = (uint32_t)(p->level_ptr.size() - 1); info[7] = p->n_vars;       // (launches per witness: plan.size() runs / wide levels)
    }
    return nullptr;
}

// The witness: runs the program on `inputs` (raw Montgomery words: the child seals and what else the program reads), fills
// data (72 x n) and out_global (16 words).  An assertion of the program that fails - the inputs are not what the program
// verifies - is an error naming the op, not a trace.
extern "C" const char* zkh_rec_witgen(const zkh_rec_program* p, const uint32_t* inputs, size_t n_inputs, const uint32_t* noise_key, zkh_buf* data,
                                      uint32_t out_global[16]) {
    ZKH_REQUIRE(p && data && out_global && (inputs || !p->n_inputs), "rec_witgen: null argument");
    ZKH_REQUIRE(n_inputs == p->n_inputs, "rec_witgen: the program reads %u input words, %zu were given", p->n_inputs, n_inputs);
    zkh_ctx* c = p->ctx;
    const size_t n = (size_t)1 << p->po2;
    ZKH_REQUIRE(data->len == (size_t)RC_WD * n, "rec_witgen: buffer shape mismatch");
    NoiseKey nk;
    ZKH_TRY(resolve_noise_key(noise_key, &nk));
    Tmp pub;
    zkh_buf *val = p->d_val, *fail = p->d_fail;
    const uint32_t none = 0xffffffffu;
    if (n_inputs) ZKH_TRY(zkh_write(c, p->d_in, inputs, 0, n_inputs));
    ZKH_TRY(zkh_write(c, p->d_fail, &none, 0, 1));
    {
        ProfScope prof(c, "rec_exec", 16.0 * p->n_vars);
        if (p->graph) {
            const hipError_t e = hipGraphLaunch(p->graph, c->stream);
            if (e != hipSuccess) return make_err("rec_witgen: hipGraphLaunch: %s", hipGetErrorString(e));
        } else {
            rec_enqueue_schedule(p, (uint4*)p->d_val->ptr(), p->d_in->ptr(), p->d_fail->ptr());
        }
    }
    ZKH_TRY(last_launch_error("rec_exec"));
    const uint32_t A = p->A, K = A / RC_BLOCK;
    {
        ProfScope prof(c, "rec_fill", 4.0 * RC_WD * n);
        const unsigned bx = (unsigned)((n + 255) / 256);
        k_rec_fill_wires<<<dim3(bx, RC_NW), 256, 0, c->stream>>>(data->ptr(), p->d_pos->ptr(), (const uint4*)val->ptr(), (uint32_t)n, A, nk);
        k_rec_blocks<<<(K + 63) / 64, 64, 0, c->stream>>>(data->ptr(), (uint32_t)n, K, p->d_tab->ptr(), p->d_table->ptr());
        const uint32_t first = RC_BLOCK * K;
        if (first < n) k_rec_fill_tail<<<dim3((unsigned)((n - first + 255) / 256), 2 * RC_T), 256, 0, c->stream>>>(data->ptr(), (uint32_t)n, A, first, nk);
    }
    ZKH_TRY(last_launch_error("rec_fill"));
    ZKH_TRY(new_buf(c, 16, true, pub.out()));
    if (p->pub_row != 0xffffffffu) k_rec_gather_pub<<<1, 64, 0, c->stream>>>(pub->ptr(), data->ptr(), (uint32_t)n, p->pub_row);
    uint32_t failed = none;
    ZKH_TRY(zkh_read(c, fail, &failed, 0, 1));
    ZKH_TRY(zkh_read(c, pub, out_global, 0, 16));
    ZKH_REQUIRE(failed == none, "rec_witgen: op %u: an assertion of the program fails on these inputs (two wires that the program ties "
                "together differ, an inverse of zero, or an unreduced input word): the input is not a valid seal", failed - 1);
    return nullptr;
}

extern "C" const char* zkh_rec_accum(const zkh_rec_program* p, const uint32_t* noise_key, const zkh_buf* data, const uint32_t* mix_global, zkh_buf* accum) {
    ZKH_REQUIRE(p && data && mix_global && accum, "rec_accum: null argument");
    NoiseKey nk;
    ZKH_TRY(resolve_noise_key(noise_key, &nk));
    zkh_ctx* c = p->ctx;
    const size_t n = (size_t)1 << p->po2;
    ZKH_REQUIRE(accum->len == (size_t)RC_WA * n && data->len == (size_t)RC_WD * n, "rec_accum: buffer shape mismatch");
    Tmp mix, terms;
    ZKH_TRY(zkh_copy_from(c, "mix", mix_global, 20, mix.out()));
    ZKH_TRY(new_buf(c, 4 * (size_t)3 * n, false, terms.out()));
    const unsigned bx = (unsigned)((n + 255) / 256);
    {
        ProfScope prof(c, "rec_accum_terms", (4.0 * RC_NW * 4 + 16.0 * 3) * n);
        k_rec_accum_terms<<<dim3(bx, 3), 256, 0, c->stream>>>(terms->ptr(), p->d_table->ptr(), data->ptr(), mix->ptr(), (uint32_t)n, p->A);
    }
    ZKH_TRY(prefix_products_batched(c, terms->ptr(), n, 3, 4 * n));
    {
        ProfScope prof(c, "rec_accum_store", 32.0 * 3 * n);
        k_rec_accum_store<<<dim3(bx, 3), 256, 0, c->stream>>>(accum->ptr(), terms->ptr(), (uint32_t)n, p->A, nk);
    }
    return last_launch_error("rec_accum");
}

// One lift / join: witness, seal.  The code group is the resident one committed at load.
extern "C" const char* zkh_rec_prove(const zkh_rec_program* p, const uint32_t* inputs, size_t n_inputs, const uint32_t* noise_key,
                                     uint32_t out_global[16], uint32_t** seal, size_t* seal_words) {
    ZKH_REQUIRE(p && seal && seal_words, "rec_prove: null argument");
    zkh_ctx* c = p->ctx;
    const size_t n = (size_t)1 << p->po2;
    Tmp data, accum;
    uint32_t out[16];
    NoiseKey nk;                                       // one key per proof (NULL / all-zero: fresh from the OS), shared by its two generators
    ZKH_TRY(resolve_noise_key(noise_key, &nk));
    ZKH_TRY(new_buf(c, (size_t)RC_WD * n, false, data.out()));
    ZKH_TRY(zkh_rec_witgen(p, inputs, n_inputs, nk.k, data.b, out));
    if (out_global) memcpy(out_global, out, sizeof out);
    zkh_seal_job* job = nullptr;
    uint32_t mix[20];
    ZKH_TRY(zkh_prove_begin(p->prover, p->po2, nullptr, data.b, out, &job, mix));
    const char* e = new_buf(c, (size_t)RC_WA * n, false, accum.out());
    if (!e) e = zkh_rec_accum(p, nk.k, data.b, mix, accum.b);
    if (e) { zkh_prove_abort(job); return e; }
    return zkh_prove_finish(job, accum.b, seal, seal_words);
}
