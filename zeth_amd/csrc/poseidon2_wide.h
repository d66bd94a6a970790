// poseidon2_wide.h — the wavefront-cooperative Poseidon2 permutation: EIGHT lanes share one state.  Lane j < 6 owns state cells
// 4j .. 4j+3 (one M4 block; lanes 6, 7 carry zeros), the column sums of M_ext and the partial-round state sum are 3-step DPP
// butterflies (quad_perm xor 1, xor 2, row_half_mirror) that never touch LDS, round constants sit in LDS (one ds_read_b128 per
// full round).  ~3.5x lower latency than one lane per permutation; used where a step has too few permutations to fill the chip:
// the narrow layers of a Merkle tree (hash.hip) and the dependency levels of a recursion witness (recursion.hip).
// Canonical Montgomery words in, canonical Montgomery words out (risc0-zkp src/core/hash/poseidon2/mod.rs poseidon2_mix).
#pragma once
#include "poseidon2.h"

namespace zkh {

__device__ __forceinline__ uint32_t dpp_xor1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true); }
__device__ __forceinline__ uint32_t dpp_xor2(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true); }
__device__ __forceinline__ uint32_t dpp_half_mirror(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true); }
// sum over the 8 lanes of a group (lanes 6, 7 hold zeros), result in every lane
__device__ __forceinline__ uint32_t group_sum(uint32_t v) {
    v = add_mod(v, dpp_xor1(v));
    v = add_mod(v, dpp_xor2(v));
    return add_mod(v, dpp_half_mirror(v));
}
__device__ __forceinline__ void wide_m_ext(uint32_t (&c)[4]) {
    m4(c[0], c[1], c[2], c[3]);
#pragma unroll
    for (int k = 0; k < 4; k++) c[k] = add_mod(c[k], group_sum(c[k]));
}
// The permutation on the group's registers: c = the four cells this lane owns (zeros in lanes 6, 7), j = lane index inside the
// group; every lane of the WAVE must call it (the butterflies read neighbours).  rcs = the round constants (rc - P form of
// sbox7_rc) in LDS, pc = the partial-round table (its first 24 words: the internal diagonal, Montgomery).
__device__ __forceinline__ void wide_permute(uint32_t (&c)[4], uint32_t j, const uint32_t* rcs, const uint32_t* __restrict__ pc) {
    const bool owner = j < 6;
    const uint32_t jj = owner ? j : 5;
    uint32_t d[4];
#pragma unroll
    for (int k = 0; k < 4; k++) d[k] = pc[4 * jj + k];
    auto full_round = [&](int round) {
        const uint4 r = *(const uint4*)(rcs + round * CELLS + 4 * jj);
        c[0] = sbox7_rc(c[0], r.x); c[1] = sbox7_rc(c[1], r.y); c[2] = sbox7_rc(c[2], r.z); c[3] = sbox7_rc(c[3], r.w);
        if (!owner) { c[0] = c[1] = c[2] = c[3] = 0; }
        wide_m_ext(c);
        if (!owner) { c[0] = c[1] = c[2] = c[3] = 0; }
    };
    wide_m_ext(c);
    if (!owner) { c[0] = c[1] = c[2] = c[3] = 0; }
    int round = 0;
#pragma unroll
    for (int r = 0; r < HALF_FULL; r++, round++) full_round(round);
#pragma unroll 3
    for (int r = 0; r < PARTIAL; r++, round++) {
        const uint32_t z = sbox7_rc(c[0], rcs[round * CELLS]);
        c[0] = j == 0 ? z : c[0];
        const uint32_t sum = group_sum(add_mod(add_mod(c[0], c[1]), add_mod(c[2], c[3])));
#pragma unroll
        for (int k = 0; k < 4; k++) c[k] = owner ? mont_reduce_wide(((uint64_t)sum << 32) + (uint64_t)d[k] * c[k]) : 0u;
    }
#pragma unroll
    for (int r = 0; r < HALF_FULL; r++, round++) full_round(round);
}

}  // namespace zkh
