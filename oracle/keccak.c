/* oracle/keccak.c — CPU witness generator of the KECCAK-F circuit (TEST INFRASTRUCTURE ONLY; the checker, never the
 * product).  Restates FIPS 202 keccak-f[1600] row by row in the trace layout of zeth_amd/circuits/keccak_f.py: the circuit
 * that stands in for risc0-circuit-keccak 4.0.2 (un-vendored: /root/reference/Cargo.lock:5289; zeth reaches it through the
 * patched tiny-keccak, /root/reference/guests/stateless-client/Cargo.toml:39).  Mirrored by the HIP kernels
 * k_keccak_* in zeth_amd/csrc/circuit.hip; pinned by hashlib.sha3_256 in tests/test_keccak_circuit.py. */
#include <stdlib.h>
#include <string.h>

#include "circuit.h"
#include "field.h"

#define KF_ROUNDS 24
#define KF_BLOCK 25
#define KF_LANES 60           /* A 0..24, T 25..29, C 30..34, B 35..59 */
static const int KF_RC_POS[7] = {0, 1, 3, 7, 15, 31, 63};

static uint64_t rotl64(uint64_t v, unsigned k) { k &= 63; return k ? (v << k) | (v >> (64 - k)) : v; }

static void kf_tables(unsigned rho[5][5], uint64_t rc[KF_ROUNDS]) {
    memset(rho, 0, sizeof(unsigned) * 25);
    unsigned x = 1, y = 0;
    for (unsigned t = 0; t < 24; t++) {
        rho[x][y] = ((t + 1) * (t + 2) / 2) % 64;
        unsigned nx = y, ny = (2 * x + 3 * y) % 5;
        x = nx; y = ny;
    }
    unsigned reg = 1;                                    /* LFSR x^8 + x^6 + x^5 + x^4 + 1: bit t of the rc sequence */
    for (unsigned i = 0; i < KF_ROUNDS; i++) {
        rc[i] = 0;
        for (unsigned j = 0; j < 7; j++) {
            if (reg & 1) rc[i] |= 1ull << ((1u << j) - 1);
            reg <<= 1;
            if (reg & 0x100) reg ^= 0x171;
        }
    }
}

uint64_t zko_keccak_lane(uint64_t seed, uint64_t perm, uint32_t lane) {
    uint64_t z = seed ^ 0x4B454343414B5F46ull;
    z += perm * 0xBF58476D1CE4E5B9ull;
    z += (uint64_t)(lane + 1) * 0x94D049BB133111EBull;
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return z;
}

/* one permutation: rows[r][lane] for r < 25 (row 24: output state in lanes 0..24, the rest zero) */
static void kf_rows(const uint64_t in[25], uint64_t rows[KF_BLOCK][KF_LANES], unsigned rho[5][5], const uint64_t rc[KF_ROUNDS]) {
    uint64_t a[25];
    memcpy(a, in, sizeof a);
    for (unsigned r = 0; r < KF_ROUNDS; r++) {
        uint64_t* row = rows[r];
        uint64_t c[5], d[5];
        for (unsigned x = 0; x < 5; x++) {
            row[25 + x] = a[x] ^ a[x + 5] ^ a[x + 10];
            c[x] = row[30 + x] = row[25 + x] ^ a[x + 15] ^ a[x + 20];
        }
        for (unsigned x = 0; x < 5; x++) d[x] = c[(x + 4) % 5] ^ rotl64(c[(x + 1) % 5], 1);
        for (unsigned i = 0; i < 25; i++) row[i] = a[i];
        for (unsigned x = 0; x < 5; x++)
            for (unsigned y = 0; y < 5; y++)
                row[35 + y + 5 * ((2 * x + 3 * y) % 5)] = rotl64(a[x + 5 * y] ^ d[x], rho[x][y]);
        const uint64_t* b = row + 35;
        for (unsigned y = 0; y < 5; y++)
            for (unsigned x = 0; x < 5; x++)
                a[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
        a[0] ^= rc[r];
    }
    memset(rows[KF_ROUNDS], 0, sizeof(uint64_t) * KF_LANES);
    memcpy(rows[KF_ROUNDS], a, sizeof a);
}

void zko_keccak_code(const zko_circuit* c, unsigned po2, unsigned zk, uint32_t* code) {
    size_t n = (size_t)1 << po2, A = n - zk, K = A / KF_BLOCK;
    size_t wc = c->group_size[ZKC_GROUP_CODE];
    unsigned rho[5][5]; uint64_t rc[KF_ROUNDS];
    kf_tables(rho, rc);
    fp one = fp_from_u32(1);
    memset(code, 0, 4 * wc * n);
    for (size_t r = 0; r < n; r++) {
        const int in_blocks = r < KF_BLOCK * K;
        const unsigned k = (unsigned)(r % KF_BLOCK);
        code[0 * n + r] = r < A ? one : 0;
        code[1 * n + r] = r == 0 ? one : 0;
        code[2 * n + r] = (r > 0 && r < A) ? one : 0;
        code[3 * n + r] = (in_blocks && k < KF_ROUNDS) ? one : 0;
        code[4 * n + r] = (in_blocks && k >= 1) ? one : 0;
        code[5 * n + r] = (in_blocks && k == 0) ? one : 0;
        code[6 * n + r] = (K > 0 && r == KF_BLOCK * K - 1) ? one : 0;
        if (wc > 14) code[14 * n + r] = (K > 0 && r == KF_BLOCK * (K - 1)) ? one : 0;       /* bind: row 0 of the last block */
        if (in_blocks && k >= 1)
            for (int j = 0; j < 7; j++) code[(7 + j) * n + r] = ((rc[k - 1] >> KF_RC_POS[j]) & 1) ? one : 0;
    }
}

/* last_input: 50 words (25 lanes, low word first) = the input state of the LAST permutation, or NULL (seeded like the
 * others).  out_global: 200 words — the output state of the last permutation as 16-bit limbs (lane l, limb j at 4 l + j), then its
 * input state at 100 + 4 l + j (the claim binds the PAIR: an output alone always has a preimage). */
void zko_keccak_witgen(const zko_circuit* c, unsigned po2, unsigned zk, uint64_t seed, const uint32_t* noise_key,
                       const uint32_t* last_input, uint32_t* code, uint32_t* data, uint32_t* out_global) {
    size_t n = (size_t)1 << po2, A = n - zk, K = A / KF_BLOCK;
    size_t wd = c->group_size[ZKC_GROUP_DATA];
    unsigned rho[5][5]; uint64_t rc[KF_ROUNDS];
    kf_tables(rho, rc);
    zko_keccak_code(c, po2, zk, code);
    fp one = fp_from_u32(1);
    memset(data, 0, 4 * wd * n);
    memset(out_global, 0, 4 * c->global_size[ZKC_GLOBAL_OUT]);
    uint64_t (*rows)[KF_LANES] = malloc(sizeof(uint64_t) * KF_BLOCK * KF_LANES);
    for (size_t p = 0; p < K; p++) {
        uint64_t in[25];
        for (uint32_t l = 0; l < 25; l++) in[l] = zko_keccak_lane(seed, p, l);
        if (last_input && p + 1 == K)
            for (uint32_t l = 0; l < 25; l++) in[l] = (uint64_t)last_input[2 * l] | ((uint64_t)last_input[2 * l + 1] << 32);
        kf_rows(in, rows, rho, rc);
        for (size_t k = 0; k < KF_BLOCK; k++)
            for (size_t col = 0; col < wd; col++)
                data[col * n + KF_BLOCK * p + k] = ((rows[k][col >> 6] >> (col & 63)) & 1) ? one : 0;
        if (p + 1 == K)
            for (uint32_t l = 0; l < 25; l++)
                for (uint32_t j = 0; j < 4; j++) {
                    out_global[4 * l + j] = fp_from_u32((uint32_t)((rows[KF_ROUNDS][l] >> (16 * j)) & 0xFFFF));
                    if (c->global_size[ZKC_GLOBAL_OUT] >= 200) out_global[100 + 4 * l + j] = fp_from_u32((uint32_t)((in[l] >> (16 * j)) & 0xFFFF));
                }
    }
    free(rows);
    for (size_t col = 0; col < wd; col++)
        for (size_t r = A; r < n; r++)
            data[col * n + r] = zko_noise_cell(noise_key, ZKC_GROUP_DATA, (uint32_t)col, (uint32_t)r);
}
