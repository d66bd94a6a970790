// noise.h — the zero-knowledge blinding rows (the last ZK_CYCLES rows of every data / accum column) as a keyed CSPRNG stream.
//
// Upstream fills those rows from the OS RNG, one `Elem::random(&mut rng)` per cell (risc0-zkp 3.0.2 prove/ + risc0-core 3.0.0
// field/baby_bear.rs `random`, un-vendored: /root/reference/Cargo.lock:5393,5338; reached from
// /root/reference/crates/host/src/lib.rs:137).  A GPU cannot read /dev/urandom per cell, so the randomness enters ONCE per segment
// as a 256-bit key (getrandom on the host: session.hip / prover.hip) and every cell is a function of (key, group, column, row)
// evaluated where the row is written:
//
//     block = ChaCha12(key; counter = (row, column), nonce = (group, "ZKN1"))          -- RFC 8439's block function with 6 double
//     value = (((((w0 * 2^32 + w1) mod P) * 2^32 + w2) mod P ...) * 2^32 + w5) mod P      rounds (the generator Rust's StdRng is);
//                                                                                         six 32-bit words folded mod P exactly
// as upstream's `Elem::random` folds six `next_u32()` draws (a uniform 192-bit integer mod P: bias < 2^-160).  Distinct cells never
// share a (counter, nonce) pair, so the rows are independent uniform field elements to anyone who does not hold the key — which
// the 64-bit splitmix-style hash of rounds 1-4 (a non-cryptographic generator over a 64-bit seed) could not claim.
// A fixed key gives reproducible seals: what tests, bench.py and the golden fixtures pass; the product default is a fresh OS key
// per segment (an all-zero / NULL key at the C ABI means exactly that).  CPU twin: oracle/noise.h.
#pragma once
#include <stdint.h>

#include "fp.h"

namespace zkh {

struct NoiseKey { uint32_t k[8]; };

constexpr uint32_t NOISE_TAG = 0x314e4b5au;          // "ZKN1"
constexpr int NOISE_DOUBLE_ROUNDS = 6;               // ChaCha12

ZKH_HD uint32_t rotl32(uint32_t v, int c) { return (v << c) | (v >> (32 - c)); }
#define ZKH_CHACHA_QR(a, b, c, d) \
    a += b; d ^= a; d = rotl32(d, 16); c += d; b ^= c; b = rotl32(b, 12); a += b; d ^= a; d = rotl32(d, 8); c += d; b ^= c; b = rotl32(b, 7)

// RFC 8439 §2.3 block function with `double_rounds` double rounds (10 = ChaCha20: pinned by the RFC's test vector in
// tests/test_noise.py); state words 12..15 = (c0, c1, n0, n1).  Only the first `n_out` (<= 16) output words are produced.
ZKH_HD void chacha_block(const uint32_t key[8], uint32_t c0, uint32_t c1, uint32_t n0, uint32_t n1, int double_rounds, uint32_t* out, int n_out) {
    const uint32_t in[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key[0], key[1], key[2], key[3],
                             key[4], key[5], key[6], key[7], c0, c1, n0, n1};
    uint32_t x0 = in[0], x1 = in[1], x2 = in[2], x3 = in[3], x4 = in[4], x5 = in[5], x6 = in[6], x7 = in[7];
    uint32_t x8 = in[8], x9 = in[9], x10 = in[10], x11 = in[11], x12 = in[12], x13 = in[13], x14 = in[14], x15 = in[15];
    for (int r = 0; r < double_rounds; r++) {
        ZKH_CHACHA_QR(x0, x4, x8, x12); ZKH_CHACHA_QR(x1, x5, x9, x13); ZKH_CHACHA_QR(x2, x6, x10, x14); ZKH_CHACHA_QR(x3, x7, x11, x15);
        ZKH_CHACHA_QR(x0, x5, x10, x15); ZKH_CHACHA_QR(x1, x6, x11, x12); ZKH_CHACHA_QR(x2, x7, x8, x13); ZKH_CHACHA_QR(x3, x4, x9, x14);
    }
    const uint32_t x[16] = {x0, x1, x2, x3, x4, x5, x6, x7, x8, x9, x10, x11, x12, x13, x14, x15};
    for (int i = 0; i < n_out; i++) out[i] = x[i] + in[i];
}

// one blinding cell: a uniform field element, Montgomery form
ZKH_HD uint32_t noise_cell(const NoiseKey& key, uint32_t group, uint32_t col, uint32_t row) {
    uint32_t w[6];
    chacha_block(key.k, row, col, group, NOISE_TAG, NOISE_DOUBLE_ROUNDS, w, 6);
    uint64_t v = 0;
    for (int i = 0; i < 6; i++) v = ((v << 32) + w[i]) % P;          // v < P < 2^31: (v << 32) + w < 2^63
    return mont_reduce((uint64_t)(uint32_t)v * R2);                   // to Montgomery form
}

}  // namespace zkh
