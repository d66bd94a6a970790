// Host-side check of the lane-per-permutation Poseidon2 form (zeth_amd/csrc/poseidon2.h) against a literal 29-round
// permutation written here with plain 64-bit modular arithmetic (src/core/hash/poseidon2/mod.rs semantics), on the
// inputs and constant tables that sit at the edges of the representation bounds the fast form relies on: states and
// round constants of all 0 / all P-1 / mixed, internal diagonals at the centring boundary (+-(P-1)/2), and seeded
// random ones.  The same header compiles for the device; only instruction selection differs there.
// Build: g++ -O2 -std=c++17 -I zeth_amd/csrc tests/cpp/poseidon2_bounds.cpp -o <out>;  exit code 0 = all equal.
#include "poseidon2.h"
#include "zkh_poseidon2_consts.h"
#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace zkh;

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t next64() { uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
                           z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
static uint32_t mulp(uint32_t a, uint32_t b) { return (uint32_t)((uint64_t)a * b % P); }
static uint32_t addp(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a + b) % P); }
static uint32_t pow7(uint32_t x) { const uint32_t x2 = mulp(x, x), x4 = mulp(x2, x2); return mulp(mulp(x4, x2), x); }

static void lit_m_ext(uint32_t* x) {
    static const uint32_t M4[4][4] = {{5, 7, 1, 3}, {4, 6, 1, 1}, {1, 3, 5, 7}, {1, 1, 4, 6}};
    uint32_t y[CELLS], col[4] = {0, 0, 0, 0};
    for (int b = 0; b < CELLS; b += 4)
        for (int r = 0; r < 4; r++) {
            uint32_t acc = 0;
            for (int c = 0; c < 4; c++) acc = addp(acc, mulp(M4[r][c], x[b + c]));
            y[b + r] = acc; col[r] = addp(col[r], acc);
        }
    for (int i = 0; i < CELLS; i++) x[i] = addp(y[i], col[i % 4]);
}
// plain residues in, plain residues out
static void literal(uint32_t* x, const uint32_t* rc, const uint32_t* diag) {
    lit_m_ext(x);
    int round = 0;
    for (int r = 0; r < HALF_FULL; r++, round++) {
        for (int i = 0; i < CELLS; i++) x[i] = pow7(addp(x[i], rc[round * CELLS + i] % P));
        lit_m_ext(x);
    }
    for (int r = 0; r < PARTIAL; r++, round++) {
        x[0] = pow7(addp(x[0], rc[round * CELLS] % P));
        uint32_t s = 0;
        for (int i = 0; i < CELLS; i++) s = addp(s, x[i]);
        for (int i = 0; i < CELLS; i++) x[i] = addp(s, mulp(diag[i] % P, x[i]));
    }
    for (int r = 0; r < HALF_FULL; r++, round++) {
        for (int i = 0; i < CELLS; i++) x[i] = pow7(addp(x[i], rc[round * CELLS + i] % P));
        lit_m_ext(x);
    }
}

static long checked = 0;
static bool run_case(const char* what, const uint32_t* state_plain, const uint32_t* rc, const uint32_t* diag) {
    std::vector<uint32_t> rcs(ROUNDS_TOTAL * CELLS), tab(P2_TAB_WORDS);
    for (int i = 0; i < ROUNDS_TOTAL * CELLS; i++) rcs[i] = fp_encode(rc[i]).v - P;
    poseidon2_partial_table(tab.data(), rc, diag);
    uint32_t want[CELLS], s[CELLS];
    for (int i = 0; i < CELLS; i++) { want[i] = state_plain[i] % P; s[i] = fp_encode(state_plain[i]).v; }
    literal(want, rc, diag);
    poseidon2_mix(s, rcs.data(), tab.data());
    checked++;
    for (int i = 0; i < CELLS; i++)
        if (s[i] >= P || fp_decode(Fp::raw(s[i])) != want[i]) {
            fprintf(stderr, "MISMATCH (%s) cell %d: got %u (plain %u) want %u\n", what, i, s[i], fp_decode(Fp::raw(s[i])), want[i]);
            return false;
        }
    return true;
}

int main(int argc, char** argv) {
    const long random_cases = argc > 1 ? atol(argv[1]) : 20000;
    std::vector<uint32_t> rc(ROUNDS_TOTAL * CELLS), diag(CELLS), st(CELLS);
    const uint32_t edge[] = {0u, 1u, P - 1, (P - 1) / 2, (P + 1) / 2, P - 2, 2u, R1, P - R1};
    const int n_edge = sizeof edge / sizeof edge[0];
    // every combination of constant fills from the edge set, with edge and random states
    for (int a = 0; a < n_edge; a++)
        for (int b = 0; b < n_edge; b++)
            for (int c = 0; c < n_edge; c++) {
                for (auto& v : rc) v = edge[a];
                for (auto& v : diag) v = edge[b];
                for (auto& v : st) v = edge[c];
                if (!run_case("edge fills", st.data(), rc.data(), diag.data())) return 1;
                for (auto& v : st) v = (uint32_t)(next64() % P);
                if (!run_case("edge constants, random state", st.data(), rc.data(), diag.data())) return 1;
            }
    // per-word mixtures of edge values and random values
    for (long t = 0; t < random_cases; t++) {
        const int mode = (int)(t % 4);
        for (auto& v : rc) v = (mode & 1) ? edge[next64() % n_edge] : (uint32_t)(next64() % P);
        for (auto& v : diag) v = (mode & 2) ? edge[next64() % n_edge] : (uint32_t)(next64() % P);
        for (auto& v : st) v = (t % 3 == 0) ? edge[next64() % n_edge] : (uint32_t)(next64() % P);
        if (!run_case("mixtures", st.data(), rc.data(), diag.data())) return 1;
    }
    // the building blocks at their stated operand bounds
    {   // M_ext on doubles: every cell +-(P-1)
        for (int pat = 0; pat < 64; pat++) {
            double d[CELLS]; uint32_t s[CELLS], lit[CELLS];
            for (int i = 0; i < CELLS; i++) {
                const bool neg = ((pat >> (i % 6)) ^ (i / 6)) & 1;
                d[i] = neg ? -(double)(P - 1) : (double)(P - 1);
                lit[i] = neg ? 1u : P - 1;                       // -(P-1) = 1 (mod P)
            }
            m_ext_f64(s, d);
            lit_m_ext(lit);
            for (int i = 0; i < CELLS; i++) {
                const int32_t r = (int32_t)(s[i] - F64_OFF);
                if (r > (int32_t)(P / 2 + 64) || r < -(int32_t)(P / 2 + 64)) { fprintf(stderr, "m_ext_f64 range: %d\n", r); return 1; }
                const uint32_t canon_r = r < 0 ? (uint32_t)(r + (int32_t)P) : (uint32_t)r;
                if (mulp(canon_r, R1) != lit[i]) { fprintf(stderr, "m_ext_f64 value at pattern %d cell %d\n", pat, i); return 1; }
            }
        }
        // the s-box at |v + rc| = P + 64
        const int32_t ext[] = {(int32_t)(P + 64), -(int32_t)(P + 64), (int32_t)P, -(int32_t)P, 0, 1, -1};
        const uint32_t r6 = mulp(mulp(mulp(R1, R1), mulp(R1, R1)), mulp(R1, R1));
        for (int32_t sx : ext) {
            const int32_t y = sbox7_lazy((uint32_t)sx, 0u);
            if (y <= -(int32_t)P || y >= (int32_t)P) { fprintf(stderr, "sbox7_lazy range\n"); return 1; }
            const uint32_t xs = (uint32_t)(((int64_t)sx % (int64_t)P + P) % P);
            const uint32_t yc = y < 0 ? (uint32_t)(y + (int32_t)P) : (uint32_t)y;
            if (mulp(yc, r6) != pow7(xs)) { fprintf(stderr, "sbox7_lazy value at %d\n", sx); return 1; }
        }
    }
    {   // the published known-answer vector of the instance (tests/golden/poseidon2_kat.json) with the shipped tables:
        // checked on the literal permutation itself, then run_case compares the fast form with it
        static const uint32_t kat[CELLS] = {0x2ed3e23d, 0x12921fb0, 0x0e659e79, 0x61d81dc9, 0x32bae33b, 0x62486ae3, 0x1e681b60, 0x24b91325,
                                            0x2a2ef5b9, 0x50e8593e, 0x5bc818ec, 0x10691997, 0x35a14520, 0x2ba6a3c5, 0x279d47ec, 0x55014e81,
                                            0x5953a67f, 0x2f403111, 0x6b8828ff, 0x1801301f, 0x2749207a, 0x3dc9cf21, 0x3c985ba2, 0x57a99864};
        uint32_t x[CELLS];
        for (int i = 0; i < CELLS; i++) { x[i] = (uint32_t)i; st[i] = (uint32_t)i; }
        literal(x, ZKH_P2_ROUND_CONSTANTS, ZKH_P2_M_INT_DIAG);
        for (int i = 0; i < CELLS; i++)
            if (x[i] != kat[i]) { fprintf(stderr, "known-answer vector: cell %d is %08x, published %08x\n", i, x[i], kat[i]); return 1; }
        if (!run_case("known-answer input, shipped tables", st.data(), ZKH_P2_ROUND_CONSTANTS, ZKH_P2_M_INT_DIAG)) return 1;
    }
    printf("poseidon2 fast form == literal permutation on %ld cases (incl. the published known-answer vector)\n", checked);
    return 0;
}
