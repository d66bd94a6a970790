// poseidon2.h — the Poseidon2 permutation over BabyBear (t = 24: 4 + 21 + 4 rounds, x^7 s-box), shared by the
// gfx950 kernels (hash.hip) and the host-side Fiat-Shamir sponge (prover.hip).
// Stands in for risc0-zkp 3.0.2 src/core/hash/poseidon2/mod.rs (un-vendored; /root/reference/Cargo.lock:5393).
// The 24-word state stays in registers: every cell loop is fully unrolled, round loops are not (they only index
// the constant tables, which arrive through scalar loads on the device).
#pragma once
#include "fp.h"

namespace zkh {

constexpr int CELLS = 24, RATE = 16, OUT = 8, HALF_FULL = 4, PARTIAL = 21, ROUNDS_TOTAL = 2 * HALF_FULL + PARTIAL;
// Partial-round table (Montgomery form): [0,24) d, [24,48) d^2, [48,72) d^3, [72] c1 = sum_{i>=1} d_i, [73] 23,
// then the three rows again as centred two's-complement words in [74, 146); then what the lane-per-permutation
// form below needs (scaled representations, see poseidon2_mix_raw): [146, 338) the eight full rounds' constants,
// [338, 410) the first partial group's centred rows, [410, 413) three scale corrections.
constexpr int P2_TAB_SIGNED = 74, P2_TAB_FULL = P2_TAB_SIGNED + 3 * CELLS, P2_TAB_GROUP0 = P2_TAB_FULL + 2 * HALF_FULL * CELLS,
              P2_TAB_KAPPA = P2_TAB_GROUP0 + 3 * CELLS, P2_TAB_WORDS = P2_TAB_KAPPA + 4;

// (x + rc)^7.  The round-constant tables hold rc - P (in [-P, 0)), so x + rcs is already a valid signed operand in
// [-P, P): one plain add instead of a modular add.  Four signed Montgomery products (no per-product correction) and
// one canonicalisation.
ZKH_HD uint32_t sbox7_rc(uint32_t x, uint32_t rcs) {
    const int32_t sx = (int32_t)(x + rcs);
    const int32_t x2 = smont(sx, sx), x3 = smont(x2, sx), x4 = smont(x2, x2);
    return canon(smont(x3, x4));
}
// 2x mod P as x + x: v_add_u32 is full rate on gfx950 and v_lshlrev_b32 is not, but hipcc canonicalises x + x into
// the shift, and hiding x behind an empty asm costs a register copy (36 v_mov per round) — so the add is spelled out.
ZKH_HD uint32_t dbl_mod(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t d;
    asm("v_add_u32_e32 %0, %1, %1" : "=v"(d) : "v"(x));
    return reduce_once(d);
#else
    return reduce_once(x + x);
#endif
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
// ---------------------------------------------------------------------------------------------------------
// Lane-per-permutation form.  M_ext is the fattest linear slice of the permutation (132 modular adds = 396
// instructions per application, nine applications), and BabyBear's P > 2^30 leaves a 32-bit word no room for lazy
// sums.  A double has room: the s-box outputs are exact integers in (-P, P), every M_ext output is an exact integer
// |T| < 112 P < 2^38, so the whole layer is 96 v_add_f64 / v_fma_f64 with NO modular correction.  Coming back,
// T + 1.5 * 2^52 has the bit pattern 0x43380000'00000000 + T, which IS a valid 64-bit operand of the signed
// Montgomery step: hi32(T + m P) comes out with the constant F64_OFF added, and because |T| is small the result
// is nearly centred, |r| <= P/2 + 64.  Two consequences: (1) the next round's  x + rc  needs no canonical x when rc
// is stored centred (|x + rc| <= P + 64 < 2^31, and F64_OFF folds into the stored constant), so the s-box loses its
// canonicalisation; (2) every M_ext divides the representation by 2^32.  The state is therefore carried as
// v = lambda_r * x with a known per-round scale (exact: M_ext is linear, x^7 turns lambda into lambda^7 / R^6), the
// stored round constants are lambda_r * rc, and the scale is brought back to R where the partial rounds start (one
// product for cell 0, folded into the first group's table rows for cells 1..23) and where a caller consumes cells
// (p2_finish: one product per consumed cell).  Scales, R = 2^32:
//   in R | F1: 1, F2: R^-7, F3: R^-56, F4: R^-399 | partial rounds: R (entry from R^-2800) | F5: R, F6: 1,
//   F7: R^-7, F8: R^-56 | out: R^-399 (+ F64_OFF).
// Exact field identities throughout: digests equal the literal 29-round oracle's bit for bit (tests).
// ---------------------------------------------------------------------------------------------------------
constexpr uint32_t F64_OFF = 0x43380000u;                  // high word of 1.5 * 2^52
constexpr double F64_MAGIC = 6755399441055744.0;           // 1.5 * 2^52

ZKH_HD int64_t f64_bits(double d) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __double_as_longlong(d);
#else
    int64_t b; __builtin_memcpy(&b, &d, 8); return b;
#endif
}
ZKH_HD double fmad(double a, double b, double c) {           // exact here either way; pinned to v_fma_f64 on the device
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_fma(a, b, c);
#else
    return a * b + c;
#endif
}
ZKH_HD void m4d(double& x0, double& x1, double& x2, double& x3) {
    const double t0 = x0 + x1, t1 = x2 + x3;
    const double t2 = fmad(x1, 2.0, t1), t3 = fmad(x3, 2.0, t0);
    const double t4 = fmad(t1, 4.0, t3), t5 = fmad(t0, 4.0, t2);
    x0 = t3 + t5; x1 = t5; x2 = t2 + t4; x3 = t4;
}
// d: exact integers (|d_i| < P) -> s_i = smont_reduce(M_ext(d)_i) + F64_OFF (wrapping), |smont_reduce| <= P/2 + 64
ZKH_HD void m_ext_f64(uint32_t (&s)[CELLS], double (&d)[CELLS]) {
    double c0 = F64_MAGIC, c1 = F64_MAGIC, c2 = F64_MAGIC, c3 = F64_MAGIC;
#pragma unroll
    for (int i = 0; i < CELLS; i += 4) {
        m4d(d[i], d[i + 1], d[i + 2], d[i + 3]);
        c0 += d[i]; c1 += d[i + 1]; c2 += d[i + 2]; c3 += d[i + 3];
    }
#pragma unroll
    for (int i = 0; i < CELLS; i++) {
        const int64_t t = f64_bits(d[i] + (i % 4 == 0 ? c0 : i % 4 == 1 ? c1 : i % 4 == 2 ? c2 : c3));
        const int32_t m = (int32_t)((uint32_t)t * NEG_PINV);
        s[i] = (uint32_t)((uint64_t)mad_i64_k(m, (int32_t)P, t) >> 32);
    }
}
// t + (a mod P) for a 64-bit accumulator a = hi 2^32 + lo:  hi R + lo  (one multiply-add, one add)
ZKH_HD uint64_t fold64(uint64_t a, uint64_t t) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint64_t d, carry;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(carry) : "v"((uint32_t)(a >> 32)), "s"(R1), "v"(t));
    asm("v_mad_u64_u32 %0, %1, %2, 1, %3" : "=v"(t), "=s"(carry) : "v"((uint32_t)a), "v"(d));
    return t;
#else
    return t + (a >> 32) * R1 + (uint32_t)a;
#endif
}
// (v + rcf)^7 / R^6 in (-P, P): v + rcf is a signed operand with |.| <= P + 64 (see above); no canonicalisation.
ZKH_HD int32_t sbox7_lazy(uint32_t v, uint32_t rcf) {
    const int32_t sx = (int32_t)(v + rcf);
    const int32_t x2 = smont(sx, sx), x3 = smont(x2, sx), x4 = smont(x2, x2);
    return smont(x3, x4);
}

// Host-side: all tables of the permutation from the canonical round constants and internal diagonal (consts.rs as data).
inline uint32_t p2_mulm(uint32_t a, uint32_t b) { return (uint32_t)((uint64_t)a * b % P); }
inline uint32_t p2_powm(uint32_t a, uint64_t e) { uint32_t r = 1; while (e) { if (e & 1) r = p2_mulm(r, a); a = p2_mulm(a, a); e >>= 1; } return r; }
inline void poseidon2_partial_table(uint32_t* t, const uint32_t* rc_canonical, const uint32_t* diag_canonical) {
    Fp c1 = Fp::zero();
    for (int i = 0; i < CELLS; i++) {
        const Fp d = fp_encode(diag_canonical[i]);
        t[i] = d.v; t[CELLS + i] = (d * d).v; t[2 * CELLS + i] = (d * d * d).v;
        if (i) c1 = c1 + d;
    }
    t[3 * CELLS] = c1.v; t[3 * CELLS + 1] = fp_encode(23).v;
    for (int i = 0; i < 3 * CELLS; i++) t[P2_TAB_SIGNED + i] = (uint32_t)center(t[i]);
    // scales as plain residues: lambda(k) = R^-k
    const uint32_t rinv = p2_powm(R1, P - 2);
    auto rneg = [&](uint64_t k) { return p2_powm(rinv, k); };
    const uint32_t lam[2 * HALF_FULL] = {1u, rneg(7), rneg(56), rneg(399), R1, 1u, rneg(7), rneg(56)};
    for (int f = 0; f < 2 * HALF_FULL; f++) {
        const int round = f < HALF_FULL ? f : HALF_FULL + PARTIAL + (f - HALF_FULL);
        for (int i = 0; i < CELLS; i++) {
            const uint32_t v = p2_mulm(lam[f], rc_canonical[round * CELLS + i] % P);
            // F5 meets canonical cells with no offset (rc - P as in sbox7_rc); the others meet r + F64_OFF, r nearly centred
            t[P2_TAB_FULL + f * CELLS + i] = f == HALF_FULL ? v - P : (uint32_t)center(v) - F64_OFF;
        }
    }
    const uint32_t rho = p2_powm(R1, 2801);                   // R / lambda at the partial-round entry (lambda = R^-2800)
    for (int i = 0; i < 3 * CELLS; i++) t[P2_TAB_GROUP0 + i] = (uint32_t)center(p2_mulm(t[i], rho));
    t[P2_TAB_KAPPA] = (uint32_t)center(p2_mulm(rho, R1));     // cell 0 at the entry: smont(v, .) = v rho
    t[P2_TAB_KAPPA + 1] = p2_mulm(rho, R1);                   // group 0's A: mul_mod(sum v, .) = rho sum v
    t[P2_TAB_KAPPA + 2] = (uint32_t)center(p2_powm(R1, 401)); // p2_finish: smont(v, .) = v R^400 = x R   (v = x R^-399)
    t[P2_TAB_KAPPA + 3] = 0;
}
// A cell of poseidon2_mix_raw's output -> canonical Montgomery word
ZKH_HD uint32_t p2_finish(uint32_t v, const uint32_t* __restrict__ diag) {
    return canon(smont((int32_t)(v - F64_OFF), (int32_t)diag[P2_TAB_KAPPA + 2]));
}
// rc: round constants stored as rc - P (two's complement words; the partial rounds read cell 0's); diag: the table above.
// In: canonical Montgomery words.  Out: every cell as (x R^-399 nearly centred) + F64_OFF: p2_finish the ones you read.
ZKH_HD void poseidon2_mix_raw(uint32_t (&s)[CELLS], const uint32_t* __restrict__ rc,
                                                  const uint32_t* __restrict__ diag) {
    const uint32_t* __restrict__ rcf = diag + P2_TAB_FULL;
    double d[CELLS];
#pragma unroll
    for (int i = 0; i < CELLS; i++) d[i] = (double)s[i];
    m_ext_f64(s, d);
#pragma unroll 1
    for (int r = 0; r < HALF_FULL; r++) {
#pragma unroll
        for (int i = 0; i < CELLS; i++) d[i] = (double)sbox7_lazy(s[i], rcf[r * CELLS + i]);
        m_ext_f64(s, d);
    }
    // back to scale R for the partial rounds: cell 0 by one product, cells 1..23 through group 0's table rows
    s[0] = canon(smont((int32_t)(s[0] - F64_OFF), (int32_t)diag[P2_TAB_KAPPA]));
#pragma unroll
    for (int i = 1; i < CELLS; i++) s[i] -= F64_OFF;
    int round = HALF_FULL;
    // ---- 21 partial rounds, three at a time ----
    // Only cell 0 meets the s-box; cells 1..23 evolve linearly (s_i <- S + d_i s_i with S the round's state sum), so
    // three rounds collapse to  s_i <- S2 + d_i S1 + d_i^2 S0 + d_i^3 s_i  (ONE reduction for three products instead
    // of three).  The sums the cell-0 chain needs in between come from weighted sums taken once per group:
    //   A = sum s_i, D1 = sum d_i s_i, D2 = sum d_i^2 s_i  (i >= 1);  A' = 23 S0 + D1;  A'' = 23 S1 + c1 S0 + D2.
    // Inside this section cells 1..23 are SIGNED representatives in (-P, P) and the table rows used with them are
    // centred (|d^k| <= (P-1)/2), which makes the per-cell update one signed reduction with no correction and no
    // separate add (bound at the update below); sums of signed products start from the bias P 2^32 (= 0 mod P), which
    // keeps every accumulator an unsigned word pair.
    // Group 0 reads cells at the entry scale: its rows for the terms in s_i
    // are d^k rho instead of d^k and its plain sum is multiplied by rho once (same bounds: |s_i| <= P/2 + 64 there).
    const uint32_t* __restrict__ pc = diag;                                   // unsigned rows + c1 + 23
    const int32_t* __restrict__ pcs = (const int32_t*)(diag + P2_TAB_SIGNED); // centred d, d^2, d^3
    constexpr int64_t BIAS = (int64_t)((uint64_t)P << 32);                    // = 0 mod P; sums wrap as unsigned
    constexpr int32_t R1S = (int32_t)R1;                                      // 2^32 mod P = 268435454 < 2^28
#pragma unroll 1
    for (int grp = 0; grp < PARTIAL / 3; grp++, round += 3) {
        const int32_t* __restrict__ pcg = grp == 0 ? (const int32_t*)(diag + P2_TAB_GROUP0) : pcs;   // rows for terms in s_i
        // A: 12 + 11 terms s_i * R (|sum| <= 12 P 2^28 = 0.75 P 2^32 < bias)
        int64_t ta = BIAS, tb = BIAS;
#pragma unroll
        for (int i = 1; i <= 12; i++) ta = mad_i64_k((int32_t)s[i], R1S, ta);
#pragma unroll
        for (int i = 13; i < CELLS; i++) tb = mad_i64_k((int32_t)s[i], R1S, tb);
        // D1 = sum d_i s_i, D2 = sum d_i^2 s_i as six accumulators each (<= 4 signed products per accumulator,
        // |sum| <= 2 P^2 < bias, so every accumulator is an unsigned word pair below 1.94 P 2^32).  None of these sums
        // is reduced on its own: hi 2^32 + lo = hi R + lo (mod P) folds an accumulator into the 64-bit sum that
        // produces the next state sum with one multiply-add and one add (fold64), so S0, S1 and S2 cost one reduction each.
        uint64_t a1[6], a2[6];
#pragma unroll
        for (int c = 0; c < 6; c++) {
            int64_t x1 = BIAS, x2 = BIAS;
#pragma unroll
            for (int i = 1 + 4 * c; i < 5 + 4 * c && i < CELLS; i++) {
                x1 = mad_i64_k((int32_t)s[i], pcg[i], x1);
                x2 = mad_i64_k((int32_t)s[i], pcg[CELLS + i], x2);
            }
            a1[c] = (uint64_t)x1; a2[c] = (uint64_t)x2;
        }
        const uint32_t d0 = pc[0], c1 = pc[3 * CELLS], m23 = pc[3 * CELLS + 1];
        const uint32_t z0 = sbox7_rc(s[0], rc[round * CELLS]);
        // S0 = z0 + A:  z0 R + 2 (1.75 P 2^28 + 2^32)  <  P 2^32
        uint32_t S0 = mont_reduce(fold64((uint64_t)tb, fold64((uint64_t)ta, (uint64_t)z0 * R1)));
        if (grp == 0)      // the cells came in at the entry scale: S0 = z0 + rho sum
            S0 = add_mod(z0, mul_mod(add_mod(mont_reduce_wide((uint64_t)ta), mont_reduce_wide((uint64_t)tb)), diag[P2_TAB_KAPPA + 1]));
        const uint32_t s0a = mont_reduce_wide(((uint64_t)S0 << 32) + (uint64_t)d0 * z0);
        const uint32_t z1 = sbox7_rc(s0a, rc[(round + 1) * CELLS]);
        // S1 = z1 + 23 S0 + D1:  z1 R + m23 S0 + 6 (1.94 P 2^28 + 2^32)  <  (0.0625 + 0.47 + 0.73) P 2^32  <  2 P 2^32
        uint64_t t1 = (uint64_t)m23 * S0 + (uint64_t)z1 * R1;
#pragma unroll
        for (int c = 0; c < 6; c++) t1 = fold64(a1[c], t1);
        const uint32_t S1 = mont_reduce_wide(t1);
        const uint32_t s0b = mont_reduce_wide(((uint64_t)S1 << 32) + (uint64_t)d0 * z1);
        const uint32_t z2 = sbox7_rc(s0b, rc[(round + 2) * CELLS]);
        // S2 = z2 + 23 S1 + c1 S0 + D2:  < (0.0625 + 0.47 + 0.47 + 0.73) P 2^32  <  2 P 2^32
        uint64_t t2 = (uint64_t)m23 * S1 + (uint64_t)c1 * S0 + (uint64_t)z2 * R1;
#pragma unroll
        for (int c = 0; c < 6; c++) t2 = fold64(a2[c], t2);
        const uint32_t S2 = mont_reduce_wide(t2);
        s[0] = mont_reduce_wide(((uint64_t)S2 << 32) + (uint64_t)d0 * z2);
        // cells 1..23: S2 rides in the accumulator as S2 R, so the signed reduction's output IS the new cell:
        // |S2 R + d S1 + d^2 S0 + d^3 s| <= (P-1)/2 R + 2 ((P-1)/2)^2 + (P-1)(P-1)/2 = (P-1)(P + 2^27 - 2) < P 2^31.
        const int32_t S0c = center(S0), S1c = center(S1);
        const int64_t acc0 = mad_i64_k(center(S2), R1S, 0);
#pragma unroll
        for (int i = 1; i < CELLS; i++)
            s[i] = (uint32_t)smont_reduce(mad_i64_k((int32_t)s[i], pcg[2 * CELLS + i],
                                                    mad_i64_k(S0c, pcs[CELLS + i], mad_i64_k(S1c, pcs[i], acc0))));   // in (-P, P)
    }
#pragma unroll
    for (int i = 1; i < CELLS; i++) s[i] = canon((int32_t)s[i]);
#pragma unroll 1
    for (int r = HALF_FULL; r < 2 * HALF_FULL; r++) {
#pragma unroll
        for (int i = 0; i < CELLS; i++) d[i] = (double)sbox7_lazy(s[i], rcf[r * CELLS + i]);
        m_ext_f64(s, d);
    }
}
// The permutation on canonical Montgomery words, in and out (host sponges; kernels that read few cells use the raw form).
ZKH_HD void poseidon2_mix(uint32_t (&s)[CELLS], const uint32_t* __restrict__ rc, const uint32_t* __restrict__ diag) {
    poseidon2_mix_raw(s, rc, diag);
#pragma unroll
    for (int i = 0; i < CELLS; i++) s[i] = p2_finish(s[i], diag);
}

}  // namespace zkh
