// poseidon2.h — the Poseidon2 permutation over BabyBear (t = 24: 4 + 21 + 4 rounds, x^7 s-box), shared by the
// gfx950 kernels (hash.hip) and the host-side Fiat-Shamir sponge (prover.hip).
// Stands in for risc0-zkp 3.0.2 src/core/hash/poseidon2/mod.rs (un-vendored; /root/reference/Cargo.lock:5393).
// The 24-word state stays in registers: every cell loop is fully unrolled, round loops are not (they only index
// the constant tables, which arrive through scalar loads on the device).
#pragma once
#include "fp.h"

namespace zkh {

constexpr int CELLS = 24, RATE = 16, OUT = 8, HALF_FULL = 4, PARTIAL = 21;

// (x + rc)^7.  The round-constant tables hold rc - P (in [-P, 0)), so x + rcs is already a valid signed operand in
// [-P, P): one plain add instead of a modular add.  Four signed Montgomery products (no per-product correction) and
// one canonicalisation.
ZKH_HD uint32_t sbox7_rc(uint32_t x, uint32_t rcs) {
    const int32_t sx = (int32_t)(x + rcs);
    const int32_t x2 = smont(sx, sx), x3 = smont(x2, sx), x4 = smont(x2, x2);
    return canon(smont(x3, x4));
}
// 2x mod P as x + x (v_add_u32 is full rate on gfx950, v_lshlrev_b32 is not); the empty asm keeps hipcc from
// canonicalising the add back into a shift.
ZKH_HD uint32_t dbl_mod(uint32_t x) {
    uint32_t y = x;
#if defined(__HIP_DEVICE_COMPILE__)
    asm("" : "+v"(y));
#endif
    return reduce_once(x + y);
}
// 4x4 block [[5,7,1,3],[4,6,1,1],[1,3,5,7],[1,1,4,6]] with 8 additions (Poseidon2 paper, appendix B)
ZKH_HD void m4(uint32_t& x0, uint32_t& x1, uint32_t& x2, uint32_t& x3) {
    const uint32_t t0 = add_mod(x0, x1), t1 = add_mod(x2, x3);
    const uint32_t t2 = add_mod(dbl_mod(x1), t1), t3 = add_mod(dbl_mod(x3), t0);
    const uint32_t t4 = add_mod(dbl_mod(dbl_mod(t1)), t3), t5 = add_mod(dbl_mod(dbl_mod(t0)), t2);
    x0 = add_mod(t3, t5); x1 = t5; x2 = add_mod(t2, t4); x3 = t4;
}
ZKH_HD void m_ext(uint32_t (&s)[CELLS]) {
    uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0;
#pragma unroll
    for (int i = 0; i < CELLS; i += 4) {
        m4(s[i], s[i + 1], s[i + 2], s[i + 3]);
        c0 = add_mod(c0, s[i]); c1 = add_mod(c1, s[i + 1]); c2 = add_mod(c2, s[i + 2]); c3 = add_mod(c3, s[i + 3]);
    }
#pragma unroll
    for (int i = 0; i < CELLS; i += 4) {
        s[i] = add_mod(s[i], c0); s[i + 1] = add_mod(s[i + 1], c1);
        s[i + 2] = add_mod(s[i + 2], c2); s[i + 3] = add_mod(s[i + 3], c3);
    }
}
// rc: round constants stored as rc - P (two's complement words); diag: Montgomery form.
ZKH_HD void poseidon2_mix(uint32_t (&s)[CELLS], const uint32_t* __restrict__ rc,
                                              const uint32_t* __restrict__ diag) {
    m_ext(s);
    int round = 0;
#pragma unroll 1
    for (int r = 0; r < HALF_FULL; r++, round++) {
#pragma unroll
        for (int i = 0; i < CELLS; i++) s[i] = sbox7_rc(s[i], rc[round * CELLS + i]);
        m_ext(s);
    }
#pragma unroll 1
    for (int r = 0; r < PARTIAL; r++, round++) {
        s[0] = sbox7_rc(s[0], rc[round * CELLS]);
        // tree-shaped sum keeps the dependency chain short
        uint32_t p[12];
#pragma unroll
        for (int i = 0; i < 12; i++) p[i] = add_mod(s[2 * i], s[2 * i + 1]);
#pragma unroll
        for (int i = 0; i < 6; i++) p[i] = add_mod(p[2 * i], p[2 * i + 1]);
        const uint32_t sum = add_mod(add_mod(add_mod(p[0], p[1]), add_mod(p[2], p[3])), add_mod(p[4], p[5]));
#pragma unroll
        for (int i = 0; i < CELLS; i++)   // sum + diag*s in ONE reduction: (sum * 2^32 + diag*s) * 2^-32
            s[i] = mont_reduce_wide(((uint64_t)sum << 32) + (uint64_t)diag[i] * s[i]);
    }
#pragma unroll 1
    for (int r = 0; r < HALF_FULL; r++, round++) {
#pragma unroll
        for (int i = 0; i < CELLS; i++) s[i] = sbox7_rc(s[i], rc[round * CELLS + i]);
        m_ext(s);
    }
}


}  // namespace zkh
