/*
 * noise.h — CPU restatement of the blinding-row generator (csrc/noise.h is the device twin; TEST INFRASTRUCTURE like the rest of
 * oracle/).  Upstream fills the last ZK_CYCLES rows of every data / accum column with `Elem::random(&mut rng)` draws from the OS RNG
 * (risc0-zkp 3.0.2 prove/, risc0-core 3.0.0 field/baby_bear.rs `random`: six next_u32() words folded mod P; un-vendored:
 * /root/reference/Cargo.lock:5393,5338).  Here the randomness is a 256-bit key and cell (group, column, row) is
 *     fold_mod_P( first six words of ChaCha12(key; counter = (row, column), nonce = (group, "ZKN1")) ),
 * ChaCha = RFC 8439 section 2.3's block function with 6 double rounds.  Written from the RFC as a state ARRAY and a round loop —
 * deliberately not the unrolled register form of the device twin — and pinned by the RFC's ChaCha20 block test vector
 * (tests/test_noise.py) through zko_chacha_block.
 */
#ifndef ZKO_NOISE_H
#define ZKO_NOISE_H
#include <stdint.h>
#include <string.h>

#include "field.h"

#define ZKO_NOISE_TAG 0x314e4b5au
#define ZKO_NOISE_DOUBLE_ROUNDS 6

static inline void zko_quarter_round(uint32_t* s, int a, int b, int c, int d) {
    s[a] += s[b]; s[d] ^= s[a]; s[d] = (s[d] << 16) | (s[d] >> 16);
    s[c] += s[d]; s[b] ^= s[c]; s[b] = (s[b] << 12) | (s[b] >> 20);
    s[a] += s[b]; s[d] ^= s[a]; s[d] = (s[d] << 8) | (s[d] >> 24);
    s[c] += s[d]; s[b] ^= s[c]; s[b] = (s[b] << 7) | (s[b] >> 25);
}
/* state = "expand 32-byte k" | key | tail[0..4) (counter / nonce words 12..15); out = working state + state */
static inline void zko_chacha_block_inline(const uint32_t key[8], const uint32_t tail[4], int double_rounds, uint32_t out[16]) {
    static const uint32_t sigma[4] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u};
    uint32_t st[16], w[16];
    memcpy(st, sigma, 16); memcpy(st + 4, key, 32); memcpy(st + 12, tail, 16);
    memcpy(w, st, 64);
    for (int r = 0; r < double_rounds; r++) {
        for (int c = 0; c < 4; c++) zko_quarter_round(w, c, 4 + c, 8 + c, 12 + c);                          /* columns */
        for (int c = 0; c < 4; c++) zko_quarter_round(w, c, 4 + (c + 1) % 4, 8 + (c + 2) % 4, 12 + (c + 3) % 4);   /* diagonals */
    }
    for (int i = 0; i < 16; i++) out[i] = w[i] + st[i];
}
static inline uint32_t zko_noise_cell_inline(const uint32_t key[8], uint32_t group, uint32_t col, uint32_t row) {
    const uint32_t tail[4] = {row, col, group, ZKO_NOISE_TAG};
    uint32_t blk[16];
    zko_chacha_block_inline(key, tail, ZKO_NOISE_DOUBLE_ROUNDS, blk);
    uint64_t v = 0;
    for (int i = 0; i < 6; i++) v = ((v << 32) + blk[i]) % FP_P;
    return fp_from_u32((uint32_t)v);
}
#endif
