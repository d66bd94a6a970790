// poly.hip — polynomial-side Hal ops on gfx950: batch_evaluate_any, mix_poly_coeffs, combos_divide,
// prefix_products (risc0-zkp 3.0.2 src/hal/mod.rs, semantics src/hal/cpu.rs + src/core/poly.rs; un-vendored:
// /root/reference/Cargo.lock:5393).  Issued by Prover::finalize (src/prove/prover.rs), reached from
// /root/reference/crates/host/src/lib.rs:137.
//
// All of these stream W x n (or combos x n) words once: HBM-bound.  The two linear recurrences (synthetic
// division, running product) are done as three-level 256-wide block scans (up-sweep of block totals, down-sweep
// with carries) instead of upstream's chunked sequential loops.
#include <algorithm>

#include "common.h"

using namespace zkh;

namespace {

constexpr int TB = 256;

__device__ __forceinline__ Fp4 ld_ext(const uint32_t* p) {
    const uint4 v = *(const uint4*)p;
    return Fp4(Fp::raw(v.x), Fp::raw(v.y), Fp::raw(v.z), Fp::raw(v.w));
}
__device__ __forceinline__ void st_ext(uint32_t* p, Fp4 v) { *(uint4*)p = make_uint4(v.c[0].v, v.c[1].v, v.c[2].v, v.c[3].v); }

// out[i] = start * base^i, i < n   (small tables: mix powers, x^t, ...)
__global__ void k_ext_powers(uint32_t* out, Fp4 start, Fp4 base, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) st_ext(out + 4 * i, start * fp4_pow(base, i));
}

// out[i] = base^exps[i], i < n   (eval_check's mix powers gathered into the order a generated kernel reads them)
// (bit 31 of an exponent word = the slot is read CENTRED by its kernel: see k_ext_center_at)
__global__ void k_ext_powers_at(uint32_t* out, Fp4 base, const uint32_t* __restrict__ exps, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) st_ext(out + 4 * i, fp4_pow(base, exps[i] & 0x7fffffffu));
}
// slots flagged in bit 31 of their exponent word hold the CENTRED representative of every component, x or x - P in
// [-(P-1)/2, (P-1)/2] as a two's-complement word (the signed constraint sums of the generated kernels: fp.h fold_acc_s); last step of
// the table build, after the constants of k_ext_scale_at
__global__ void k_ext_center_at(uint32_t* out, const uint32_t* __restrict__ exps, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !(exps[i] >> 31)) return;
    uint4 v = *(uint4*)(out + 4 * (size_t)i);
    v.x = (uint32_t)center(v.x); v.y = (uint32_t)center(v.y); v.z = (uint32_t)center(v.z); v.w = (uint32_t)center(v.w);
    *(uint4*)(out + 4 * (size_t)i) = v;
}

// out[slot] *= C for the slots a generated kernel wants scaled by an Fp4 constant: recs = (slot, c0, c1, c2, c3) x n
__global__ void k_ext_scale_at(uint32_t* out, const uint32_t* __restrict__ recs, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t* r = recs + 5 * (size_t)i;
    const Fp4 c(Fp::raw(r[1]), Fp::raw(r[2]), Fp::raw(r[3]), Fp::raw(r[4]));
    uint32_t* o = out + 4 * (size_t)r[0];
    st_ext(o, ld_ext(o) * c);
}

// ---------------------------------------------------------------------------------------------------------
// batch_evaluate_any.  Block (chunk, k): partial[k][chunk] = sum_{j in chunk} coeffs[which[k]][j] * x_k^j.
// Lane t owns coefficients j = chunk*CH + i*256 + t (coalesced); term = c * X^i (X = x^256, table in LDS),
// the lane total is multiplied once by x^t and the block total once by x^(chunk*CH).
//
// A real tap set evaluates the same column at several points (taps at backs 0..4 of one register): consecutive entries
// with the same which[] form a run, and ONE block streams the column once for up to EV_MAXP points of the run (the other
// blocks of the run exit at once), so DEEP evaluation reads W x n words instead of #taps x n.  (EV_MAXP is a register
// budget: 8 accumulating points need 256 VGPRs = one wave per SIMD, which costs more bandwidth than the re-reads save.)
// ---------------------------------------------------------------------------------------------------------
constexpr int EV_PER = 64, EV_CH = TB * EV_PER, EV_MAXP = 5, EV_SCAN = 256;   // 5 = the longest back-set of a register (backs 0..4)

template <int NP>
__device__ __forceinline__ void ev_accumulate(Fp4 (&acc)[EV_MAXP], const uint32_t* __restrict__ c, size_t j0, size_t po,
                                              uint4 (*xp)[EV_PER]) {
    // Fp x Fp4 multiply-accumulate, lazily: four products per 64-bit accumulator (4 P^2 < 2 P 2^32), then ONE reduction
    // and one modular add per component instead of four of each.  Four coefficients are loaded once and then used for all
    // NP points, one point at a time (only one point's 64-bit accumulators are live: the register budget stays small).
    for (int i = 0; i < EV_PER; i += 4) {
        uint64_t cj[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const size_t j = j0 + (size_t)(i + u) * TB;
            cj[u] = j < po ? c[j] : 0u;
        }
#pragma unroll
        for (int p = 0; p < NP; p++) {
            uint64_t a0 = 0, a1 = 0, a2 = 0, a3 = 0;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint4 pw = xp[p][i + u];
                a0 += cj[u] * pw.x; a1 += cj[u] * pw.y; a2 += cj[u] * pw.z; a3 += cj[u] * pw.w;
            }
            acc[p].c[0] = Fp::raw(add_mod(acc[p].c[0].v, mont_reduce_wide(a0))); acc[p].c[1] = Fp::raw(add_mod(acc[p].c[1].v, mont_reduce_wide(a1)));
            acc[p].c[2] = Fp::raw(add_mod(acc[p].c[2].v, mont_reduce_wide(a2))); acc[p].c[3] = Fp::raw(add_mod(acc[p].c[3].v, mont_reduce_wide(a3)));
        }
    }
}

// Power tables of one evaluation point, built ONCE per point (not once per block: a 2^20-coefficient column is 64 blocks,
// and the ~30 dependent Fp4 products + 14 barriers of the table build cost more than streaming a block's 64 KiB):
//   [0, 256)    x^t            | BITREV: x^(bitrev8(t) << (k-8))
//   [256, 320)  X^i, X = x^256 | BITREV: x^(bitrev6(i) << (k-14))
//   [320, 352)  [320] = X^64   | BITREV: x^(2^m), m < k
// BITREV: the column holds its coefficients in the bit-reversed order batch_interpolate_ntt leaves them in (position p
// holds coefficient bitrev_k(p)); x^bitrev(p) still factors over the bit fields of p = chunk*CH + i*256 + t, so only the
// tables change and PolyGroup never has to bit-reverse W x n coefficient words.
constexpr int EV_TAB = TB + EV_PER + 32;
template <bool BITREV>
__global__ __launch_bounds__(TB) void k_eval_tables(uint4* __restrict__ tab, const uint32_t* __restrict__ xs, uint32_t log_n) {
    __shared__ uint4 xt[TB], xp[EV_PER], sq[32];
    const uint32_t t = threadIdx.x;
    const Fp4 x = ld_ext(xs + 4 * blockIdx.x);
    if (t < 32) st_ext((uint32_t*)&sq[t], Fp4::zero());
    __syncthreads();
    if (BITREV) {
        if (t == 0) {
            Fp4 y = x;
            for (uint32_t m = 0; m < log_n; m++) { st_ext((uint32_t*)&sq[m], y); y = y * y; }
            st_ext((uint32_t*)&xt[0], Fp4::one());
            st_ext((uint32_t*)&xp[0], Fp4::one());
        }
        __syncthreads();
        for (uint32_t b = 0; b < 8; b++) {                    // position bit b of t  ->  exponent bit k-1-b
            const uint32_t s = 1u << b;
            if (t < s) st_ext((uint32_t*)&xt[s + t], ld_ext((const uint32_t*)&xt[t]) * ld_ext((const uint32_t*)&sq[log_n - 1 - b]));
            __syncthreads();
        }
        for (uint32_t b = 0; b < 6; b++) {                    // position bit 8+b  ->  exponent bit k-9-b
            const uint32_t s = 1u << b;
            if (t < s) st_ext((uint32_t*)&xp[s + t], ld_ext((const uint32_t*)&xp[t]) * ld_ext((const uint32_t*)&sq[log_n - 9 - b]));
            __syncthreads();
        }
    } else {
        // doubling: tab[s + i] = tab[i] * x^s
        if (t == 0) { st_ext((uint32_t*)&xt[0], Fp4::one()); st_ext((uint32_t*)&xp[0], Fp4::one()); }
        __syncthreads();
        Fp4 xs_pow = x;    // x^s
        for (uint32_t s = 1; s < TB; s <<= 1) {
            if (t < s) st_ext((uint32_t*)&xt[s + t], ld_ext((const uint32_t*)&xt[t]) * xs_pow);
            xs_pow = xs_pow * xs_pow;
            __syncthreads();
        }
        Fp4 Xs = xs_pow;   // X = x^256
        for (uint32_t s = 1; s < EV_PER; s <<= 1) {
            if (t < s) st_ext((uint32_t*)&xp[s + t], ld_ext((const uint32_t*)&xp[t]) * Xs);
            Xs = Xs * Xs;
            __syncthreads();
        }
        if (t == 0) st_ext((uint32_t*)&sq[0], Xs);            // X^64 = x^(chunk length)
        __syncthreads();
    }
    uint4* out = tab + (size_t)blockIdx.x * EV_TAB;
    out[t] = xt[t];
    if (t < EV_PER) out[TB + t] = xp[t];
    if (t < 32) out[TB + EV_PER + t] = sq[t];
}

template <bool BITREV>
__global__ __launch_bounds__(TB) void k_eval_partial(uint32_t* __restrict__ partial, const uint32_t* __restrict__ coeffs,
                                                     size_t po, const uint32_t* __restrict__ which,
                                                     const uint4* __restrict__ tab, uint32_t n_chunks, uint32_t log_n,
                                                     uint32_t n_eval) {
    __shared__ uint4 xp[EV_MAXP][EV_PER];
    __shared__ uint4 red[TB];
    // run of this block: leader = first entry of the run, or every EV_MAXP-th entry of a longer run
    const uint32_t k0 = blockIdx.y, chunk = blockIdx.x, t = threadIdx.x;
    const uint32_t col = which[k0];
    uint32_t start = k0;
    for (uint32_t back = 0; back < EV_SCAN && start > 0 && which[start - 1] == col; back++) start--;
    if ((k0 - start) % EV_MAXP != 0) return;                  // another block of the run covers this entry
    uint32_t np = 1;
    while (np < EV_MAXP && k0 + np < n_eval && which[k0 + np] == col) np++;
    for (uint32_t e = t; e < np * EV_PER; e += TB) xp[e / EV_PER][e % EV_PER] = tab[(size_t)(k0 + e / EV_PER) * EV_TAB + TB + e % EV_PER];
    __syncthreads();
    const uint32_t* c = coeffs + (size_t)col * po;
    const size_t j0 = (size_t)chunk * EV_CH + t;
    Fp4 acc[EV_MAXP];
#pragma unroll
    for (int p = 0; p < EV_MAXP; p++) acc[p] = Fp4::zero();
    switch (np) {
    case 1: ev_accumulate<1>(acc, c, j0, po, xp); break;
    case 2: ev_accumulate<2>(acc, c, j0, po, xp); break;
    case 3: ev_accumulate<3>(acc, c, j0, po, xp); break;
    case 4: ev_accumulate<4>(acc, c, j0, po, xp); break;
    default: ev_accumulate<5>(acc, c, j0, po, xp); break;
    }
    // ---- per point: lane total * x^t, block reduction, * x^(chunk*CH) ----
#pragma unroll
    for (int p = 0; p < EV_MAXP; p++) {
        if ((uint32_t)p >= np) break;
        const uint4* pt = tab + (size_t)(k0 + p) * EV_TAB;
        const uint4 xtv = pt[t];
        __syncthreads();                                      // red of the previous point is no longer read
        st_ext((uint32_t*)&red[t], acc[p] * ld_ext((const uint32_t*)&xtv));
        __syncthreads();
        for (uint32_t s = TB / 2; s >= 1; s >>= 1) {
            if (t < s) st_ext((uint32_t*)&red[t], ld_ext((const uint32_t*)&red[t]) + ld_ext((const uint32_t*)&red[t + s]));
            __syncthreads();
        }
        if (t == 0) {
            Fp4 chunk_pow = Fp4::one();
            if (BITREV) {
                for (uint32_t cb = 0; cb + 14 < log_n; cb++)       // position bit 14+c -> exponent bit k-15-c
                    if ((chunk >> cb) & 1) { const uint4 q = pt[TB + EV_PER + log_n - 15 - cb]; chunk_pow = chunk_pow * ld_ext((const uint32_t*)&q); }
            } else {
                const uint4 q = pt[TB + EV_PER];                  // X^64 = x^(chunk length)
                chunk_pow = fp4_pow(ld_ext((const uint32_t*)&q), chunk);
            }
            st_ext(partial + 4 * ((size_t)(k0 + p) * n_chunks + chunk), ld_ext((const uint32_t*)&red[0]) * chunk_pow);
        }
    }
}
// in-place bit reversal of `count` polynomials of n ExtElems (AoS)
__global__ void k_bit_reverse_ext(uint4* io, uint32_t log_n, size_t total) {
    size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total) return;
    const size_t n = (size_t)1 << log_n;
    const uint32_t i = (uint32_t)(g & (n - 1));
    const uint32_t r = log_n ? __brev(i) >> (32 - log_n) : 0;
    if (i < r) {
        uint4* col = io + (g - i);
        const uint4 a = col[i], b = col[r];
        col[i] = b; col[r] = a;
    }
}
__global__ void k_eval_final(uint32_t* out, const uint32_t* partial, uint32_t n_chunks, uint32_t n_eval) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_eval) return;
    Fp4 acc = Fp4::zero();
    for (uint32_t c = 0; c < n_chunks; c++) acc = acc + ld_ext(partial + 4 * ((size_t)k * n_chunks + c));
    st_ext(out + 4 * k, acc);
}

// ---------------------------------------------------------------------------------------------------------
// mix_poly_coeffs: out[combos[c]][idx] += (mix_start * mix^c) * in[c][idx].  One lane per idx, columns streamed
// with lanes on consecutive words; the running combo accumulator is flushed whenever the (wave-uniform)
// combo id changes.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TB) void k_mix_poly_coeffs(uint32_t* __restrict__ out, const uint32_t* __restrict__ in,
                                                        const uint32_t* __restrict__ combos, const uint32_t* __restrict__ pw,
                                                        uint32_t input_size, size_t count) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= count) return;
    // lazy Fp x Fp4 multiply-accumulate: up to four products per 64-bit accumulator, one reduction per component
    Fp4 acc = Fp4::zero();
    uint64_t a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    uint32_t pending = 0;
    auto fold = [&]() {
        acc.c[0] = Fp::raw(add_mod(acc.c[0].v, mont_reduce_wide(a0))); acc.c[1] = Fp::raw(add_mod(acc.c[1].v, mont_reduce_wide(a1)));
        acc.c[2] = Fp::raw(add_mod(acc.c[2].v, mont_reduce_wide(a2))); acc.c[3] = Fp::raw(add_mod(acc.c[3].v, mont_reduce_wide(a3)));
        a0 = a1 = a2 = a3 = 0; pending = 0;
    };
    uint32_t cur_combo = combos[0];
    for (uint32_t c = 0; c < input_size; c++) {
        const uint32_t cb = combos[c];                       // wave-uniform: every lane walks the same columns
        if (cb != cur_combo) {
            fold();
            uint32_t* o = out + 4 * ((size_t)cur_combo * count + idx);
            st_ext(o, ld_ext(o) + acc);
            acc = Fp4::zero();
            cur_combo = cb;
        }
        const uint64_t v = in[(size_t)c * count + idx];
        const uint4 m = *(const uint4*)(pw + 4 * c);
        a0 += v * m.x; a1 += v * m.y; a2 += v * m.z; a3 += v * m.w;
        if (++pending == 4) fold();
    }
    fold();
    uint32_t* o = out + 4 * ((size_t)cur_combo * count + idx);
    st_ext(o, ld_ext(o) + acc);
}

// ---------------------------------------------------------------------------------------------------------
// Weighted suffix scan over blocks of BL = 256 * E ExtElems:
//     S[t] = sum_{t' >= t} v[t'] * w^(t'-t)  +  w^(BL-t) * carry          (carry = S of the next block's start)
// TOTAL_ONLY: write S[0] per block (up-sweep).  Otherwise write S shifted by `shift` positions
// (shift = 1 turns suffix sums into synthetic-division quotients: q_i = S_{i+1}).
// Work-efficient form: a lane owns E consecutive elements — a sequential Horner run for its total (E-1 products), a
// 256-wide log-step scan over the lane totals with weight w^E (9 products), the run again seeded with the carry
// (E products): (2E + 8) / E Fp4 products per element (3 at E = 8) instead of the 9 of a plain log-step scan.  Loads and
// stores stay lane-strided (coalesced); the blocked view goes through a padded LDS stage.
// blockIdx.y selects one of several independent polynomials handled by the same launch (combos_divide_all): polynomial y
// lives at in/out + offs[y] words, its carries at carries + y * carry_stride words, its weight is ws[y] (Fp4, device).
// ---------------------------------------------------------------------------------------------------------
constexpr int SC_E = 8;          // elements per lane of the level-0 scans (levels 1, 2 are tiny: E = 1)
template <bool TOTAL_ONLY, int E>
__global__ __launch_bounds__(TB) void k_suffix_scan(uint32_t* __restrict__ out, const uint32_t* __restrict__ in, size_t n,
                                                    const uint32_t* __restrict__ ws /* per y: Fp4 weight of this level */,
                                                    const uint32_t* __restrict__ carries /* per block b: S of block b+1 start; may be null */,
                                                    size_t n_carries, uint32_t shift, uint32_t* __restrict__ rem_out,
                                                    const uint32_t* __restrict__ in_offs, const uint32_t* __restrict__ out_offs,
                                                    size_t carry_stride, const uint32_t* __restrict__ rem_idx) {
    constexpr int BL = TB * E;
    __shared__ uint4 stage[E > 1 ? BL + TB : 1];       // element j of the block at j + j / E (one pad slot per lane run)
    __shared__ uint4 buf[2][TB + 1];
    const uint32_t t = threadIdx.x, y = blockIdx.y;
    in += in_offs ? (size_t)in_offs[y] : 0;
    out += out_offs ? (size_t)out_offs[y] : 0;
    if (carries) carries += (size_t)y * carry_stride;
    const Fp4 w = ld_ext(ws + 4 * y);
    const size_t b = blockIdx.x, base = b * BL;
    Fp4 e[E];
    if (E == 1) {
        const size_t i = base + t;
        e[0] = i < n ? ld_ext(in + 4 * i) : Fp4::zero();
    } else {
#pragma unroll
        for (int k = 0; k < E; k++) {
            const uint32_t j = k * TB + t;
            const size_t gi = base + j;
            st_ext((uint32_t*)&stage[j + j / E], gi < n ? ld_ext(in + 4 * gi) : Fp4::zero());
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < E; k++) e[k] = ld_ext((const uint32_t*)&stage[t * E + k + t]);
    }
    Fp4 tot = e[E - 1];
#pragma unroll
    for (int k = E - 2; k >= 0; k--) tot = e[k] + tot * w;
    Fp4 carry = Fp4::zero();
    if (carries && b + 1 < n_carries) carry = ld_ext(carries + 4 * (b + 1));
    st_ext((uint32_t*)&buf[0][t], tot);
    if (t == 0) st_ext((uint32_t*)&buf[0][TB], carry);
    __syncthreads();
    int cur = 0;
    Fp4 wd = w;
#pragma unroll
    for (int q = 1; q < E; q <<= 1) wd = wd * wd;         // w^E: weight between neighbouring lane totals
    for (uint32_t d = 1; d <= TB; d <<= 1) {       // 9 steps cover offsets up to 256 (the carry slot)
        // each slot t in [0, 256]: S_t += w^d * S_{t+d}
        for (uint32_t s = t; s <= TB; s += TB) {
            Fp4 a = ld_ext((const uint32_t*)&buf[cur][s]);
            if (s + d <= TB) a = a + ld_ext((const uint32_t*)&buf[cur][s + d]) * wd;
            st_ext((uint32_t*)&buf[cur ^ 1][s], a);
        }
        wd = wd * wd;
        cur ^= 1;
        __syncthreads();
    }
    if (TOTAL_ONLY) {
        if (t == 0) st_ext(out + 4 * b, ld_ext((const uint32_t*)&buf[cur][0]));
        return;
    }
    // S at the start of the next lane's run (slot TB = the block's carry), then this lane's run seeded with it
    Fp4 r[E + 1];
    r[E] = ld_ext((const uint32_t*)&buf[cur][t + 1]);
#pragma unroll
    for (int k = E - 1; k >= 0; k--) r[k] = e[k] + r[k + 1] * w;
    if (rem_out && b == 0 && t == 0) st_ext(rem_out + 4 * (rem_idx ? rem_idx[y] : 0), r[0]);
    if (E == 1) {
        const size_t i = base + t;
        if (i < n) st_ext(out + 4 * i, shift ? r[1] : r[0]);
    } else {
        __syncthreads();                                  // every lane has read its run from the stage
#pragma unroll
        for (int k = 0; k < E; k++) st_ext((uint32_t*)&stage[t * E + k + t], shift ? r[k + 1] : r[k]);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < E; k++) {
            const uint32_t j = k * TB + t;
            const size_t gi = base + j;
            if (gi < n) st_ext(out + 4 * gi, ld_ext((const uint32_t*)&stage[j + j / E]));
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// prefix_products: inclusive running product over ExtElems, same structure in the prefix direction (a lane's run,
// a log-step scan over the lane totals, the run again seeded with the product of everything before it).
// blockIdx.y = one of `count` independent columns at io + y * col_stride (carries at carries + y * carry_stride).
// ---------------------------------------------------------------------------------------------------------
template <bool TOTAL_ONLY, int E>
__global__ __launch_bounds__(TB) void k_prefix_prod(uint32_t* __restrict__ out, const uint32_t* __restrict__ in, size_t n,
                                                    const uint32_t* __restrict__ carries /* per block b: product of everything before block b at [b-1] */,
                                                    size_t col_stride_in, size_t col_stride_out, size_t carry_stride) {
    constexpr int BL = TB * E;
    __shared__ uint4 stage[E > 1 ? BL + TB : 1];
    __shared__ uint4 buf[2][TB];
    const uint32_t t = threadIdx.x, y = blockIdx.y;
    in += (size_t)y * col_stride_in;
    out += (size_t)y * col_stride_out;
    if (carries) carries += (size_t)y * carry_stride;
    const size_t b = blockIdx.x, base = b * BL;
    Fp4 e[E];
    if (E == 1) {
        const size_t i = base + t;
        e[0] = i < n ? ld_ext(in + 4 * i) : Fp4::one();
    } else {
#pragma unroll
        for (int k = 0; k < E; k++) {
            const uint32_t j = k * TB + t;
            const size_t gi = base + j;
            st_ext((uint32_t*)&stage[j + j / E], gi < n ? ld_ext(in + 4 * gi) : Fp4::one());
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < E; k++) e[k] = ld_ext((const uint32_t*)&stage[t * E + k + t]);
    }
    if (carries && t == 0 && b > 0) e[0] = e[0] * ld_ext(carries + 4 * (b - 1));
    Fp4 tot = e[0];
#pragma unroll
    for (int k = 1; k < E; k++) tot = tot * e[k];
    st_ext((uint32_t*)&buf[0][t], tot);
    __syncthreads();
    int cur = 0;
    for (uint32_t d = 1; d < TB; d <<= 1) {
        Fp4 a = ld_ext((const uint32_t*)&buf[cur][t]);
        if (t >= d) a = a * ld_ext((const uint32_t*)&buf[cur][t - d]);
        st_ext((uint32_t*)&buf[cur ^ 1][t], a);
        cur ^= 1;
        __syncthreads();
    }
    if (TOTAL_ONLY) {
        if (t == TB - 1) st_ext(out + 4 * b, ld_ext((const uint32_t*)&buf[cur][t]));
        return;
    }
    Fp4 r = t > 0 ? ld_ext((const uint32_t*)&buf[cur][t - 1]) : Fp4::one();     // product of everything before this lane's run
    if (E == 1) {
        const size_t i = base + t;
        if (i < n) st_ext(out + 4 * i, r * e[0]);
    } else {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < E; k++) { r = r * e[k]; st_ext((uint32_t*)&stage[t * E + k + t], r); }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < E; k++) {
            const uint32_t j = k * TB + t;
            const size_t gi = base + j;
            if (gi < n) st_ext(out + 4 * gi, ld_ext((const uint32_t*)&stage[j + j / E]));
        }
    }
}

// combos_divide_all, partial-fraction form: combo polynomial y := sum_p A[p] * Q[p] over its points p (Q[p] = quotient of the
// ORIGINAL polynomial by (x - z_p) alone, A[p] = 1 / prod_{j != p} (z_p - z_j)).  desc per combo: {first pair, count}.
__global__ __launch_bounds__(TB) void k_combine_quotients(uint32_t* __restrict__ combos, const uint32_t* __restrict__ q, size_t cycles,
                                                          const uint32_t* __restrict__ combo_off, const uint32_t* __restrict__ first_pair,
                                                          const uint32_t* __restrict__ n_pairs, const uint32_t* __restrict__ weights) {
    const size_t i = (size_t)blockIdx.x * TB + threadIdx.x;
    const uint32_t y = blockIdx.y;
    if (i >= cycles) return;
    Fp4 acc = Fp4::zero();
    const uint32_t p0 = first_pair[y], np = n_pairs[y];
    for (uint32_t p = 0; p < np; p++) acc = acc + ld_ext(q + 4 * ((size_t)(p0 + p) * cycles + i)) * ld_ext(weights + 4 * (p0 + p));
    st_ext(combos + combo_off[y] + 4 * i, acc);
}
// The remainders of dividing successively by (x - z_1), (x - z_2), ... are the Newton divided differences of the
// polynomial's values at those points: r_1 = c(z_1), r_2 = (c(z_2) - c(z_1)) / (z_2 - z_1), ...  One lane per combo.
__global__ void k_divided_differences(uint32_t* __restrict__ rem_out, const uint32_t* __restrict__ vals, const uint32_t* __restrict__ pts,
                                      const uint32_t* __restrict__ first_pair, const uint32_t* __restrict__ n_pairs, uint32_t n_combos) {
    const uint32_t y = blockIdx.x * blockDim.x + threadIdx.x;
    if (y >= n_combos) return;
    const uint32_t p0 = first_pair[y], np = n_pairs[y];
    Fp4 d[8];
    for (uint32_t p = 0; p < np; p++) d[p] = ld_ext(vals + 4 * (p0 + p));
    for (uint32_t lvl = 1; lvl < np; lvl++)
        for (uint32_t p = np - 1; p >= lvl; p--)
            d[p] = (d[p] - d[p - 1]) * fp4_inv(ld_ext(pts + 4 * (p0 + p)) - ld_ext(pts + 4 * (p0 + p - lvl)));
    for (uint32_t p = 0; p < np; p++) st_ext(rem_out + 4 * (p0 + p), d[p]);
}

inline size_t ceil_div(size_t a, size_t b) { return (a + b - 1) / b; }
inline Fp4 to_fp4(const uint32_t* w) { return Fp4(Fp::raw(w[0]), Fp::raw(w[1]), Fp::raw(w[2]), Fp::raw(w[3])); }

}  // namespace

namespace zkh {
const char* launch_ext_powers(zkh_ctx* c, uint32_t* out, const uint32_t start[4], const uint32_t base[4], uint32_t n) {
    if (!n) return nullptr;
    k_ext_powers<<<(unsigned)ceil_div(n, TB), TB, 0, c->stream>>>(out, to_fp4(start), to_fp4(base), n);
    return last_launch_error("ext_powers");
}
const char* launch_ext_powers_at(zkh_ctx* c, uint32_t* out, const uint32_t base[4], const uint32_t* d_exps, uint32_t n) {
    if (!n) return nullptr;
    k_ext_powers_at<<<(unsigned)ceil_div(n, TB), TB, 0, c->stream>>>(out, to_fp4(base), d_exps, n);
    return last_launch_error("ext_powers_at");
}
const char* launch_ext_center_at(zkh_ctx* c, uint32_t* out, const uint32_t* d_exps, uint32_t n) {
    if (!n) return nullptr;
    k_ext_center_at<<<(unsigned)ceil_div(n, TB), TB, 0, c->stream>>>(out, d_exps, n);
    return last_launch_error("ext_center_at");
}
const char* launch_ext_scale_at(zkh_ctx* c, uint32_t* out, const uint32_t* d_recs, uint32_t n) {
    if (!n) return nullptr;
    k_ext_scale_at<<<(unsigned)ceil_div(n, TB), TB, 0, c->stream>>>(out, d_recs, n);
    return last_launch_error("ext_scale_at");
}
}  // namespace zkh

static const char* evaluate_any_impl(zkh_ctx* c, const zkh_buf* coeffs, size_t poly_count, const zkh_buf* which,
                                     const zkh_buf* xs, zkh_buf* out, bool bitrev) {
    ZKH_REQUIRE(poly_count && coeffs->len % poly_count == 0, "batch_evaluate_any: coeffs size not a multiple of poly_count");
    const size_t n_eval = which->len;
    ZKH_REQUIRE(xs->len == 4 * n_eval && out->len == 4 * n_eval, "batch_evaluate_any: which/xs/out size mismatch");
    if (!n_eval) return nullptr;
    ZKH_REQUIRE(n_eval <= 65535, "batch_evaluate_any: too many evaluation points");
    const size_t po = coeffs->len / poly_count;
    const uint32_t log_n = log2_ceil(po);
    if (bitrev) ZKH_REQUIRE(((size_t)1 << log_n) == po && log_n >= 14 && log_n <= 31,
                            "batch_evaluate_any_bitrev: column length must be a power of two >= 2^14");
    const uint32_t n_chunks = (uint32_t)ceil_div(po, EV_CH);
    Tmp partial, tab;
    ZKH_TRY(new_buf(c, 4 * n_eval * n_chunks, false, partial.out()));
    ZKH_TRY(new_buf(c, 4 * n_eval * (size_t)EV_TAB, false, tab.out()));
    {
        // §8d: each coefficient column is streamed once for all the points it is evaluated at
        ProfScope prof(c, "batch_evaluate_any", 4.0 * po * (double)(n_eval < poly_count ? n_eval : poly_count));
        if (bitrev) {
            k_eval_tables<true><<<(unsigned)n_eval, TB, 0, c->stream>>>((uint4*)tab->ptr(), xs->ptr(), log_n);
            k_eval_partial<true><<<dim3(n_chunks, (unsigned)n_eval), TB, 0, c->stream>>>(partial->ptr(), coeffs->ptr(), po, which->ptr(),
                                                                                      (const uint4*)tab->ptr(), n_chunks, log_n, (uint32_t)n_eval);
        } else {
            k_eval_tables<false><<<(unsigned)n_eval, TB, 0, c->stream>>>((uint4*)tab->ptr(), xs->ptr(), log_n);
            k_eval_partial<false><<<dim3(n_chunks, (unsigned)n_eval), TB, 0, c->stream>>>(partial->ptr(), coeffs->ptr(), po, which->ptr(),
                                                                                       (const uint4*)tab->ptr(), n_chunks, log_n, (uint32_t)n_eval);
        }
        k_eval_final<<<(unsigned)ceil_div(n_eval, TB), TB, 0, c->stream>>>(out->ptr(), partial->ptr(), n_chunks, (uint32_t)n_eval);
    }
    return last_launch_error("batch_evaluate_any");
}
extern "C" const char* zkh_batch_evaluate_any(zkh_ctx* c, const zkh_buf* coeffs, size_t poly_count, const zkh_buf* which,
                                              const zkh_buf* xs, zkh_buf* out) {
    return evaluate_any_impl(c, coeffs, poly_count, which, xs, out, false);
}
extern "C" const char* zkh_batch_evaluate_any_bitrev(zkh_ctx* c, const zkh_buf* coeffs, size_t poly_count, const zkh_buf* which,
                                                     const zkh_buf* xs, zkh_buf* out) {
    return evaluate_any_impl(c, coeffs, poly_count, which, xs, out, true);
}
extern "C" const char* zkh_batch_bit_reverse_extelem(zkh_ctx* c, zkh_buf* io, size_t count) {
    ZKH_REQUIRE(count && io->len % (4 * count) == 0, "batch_bit_reverse_extelem: size not a multiple of count ExtElem columns");
    const size_t n = io->len / 4 / count;
    const uint32_t log_n = log2_ceil(n);
    ZKH_REQUIRE(((size_t)1 << log_n) == n, "batch_bit_reverse_extelem: column length %zu is not a power of two", n);
    ProfScope prof(c, "batch_bit_reverse_extelem", 8.0 * io->len);
    k_bit_reverse_ext<<<(unsigned)ceil_div(n * count, TB), TB, 0, c->stream>>>((uint4*)io->ptr(), log_n, n * count);
    return last_launch_error("batch_bit_reverse_extelem");
}
extern "C" const char* zkh_mix_poly_coeffs(zkh_ctx* c, zkh_buf* out, const uint32_t mix_start[4], const uint32_t mix[4],
                                           const zkh_buf* in, const zkh_buf* combos, size_t input_size, size_t count) {
    ZKH_REQUIRE(in->len == input_size * count && combos->len >= input_size, "mix_poly_coeffs: input shape mismatch");
    ZKH_REQUIRE(out->len % (4 * count) == 0, "mix_poly_coeffs: output is not a whole number of ExtElem columns");
    if (!input_size || !count) return nullptr;
    Tmp pw;
    ZKH_TRY(new_buf(c, 4 * input_size, false, pw.out()));
    {
        ProfScope prof(c, "mix_poly_coeffs", 4.0 * in->len + 32.0 * count * 2);
        k_ext_powers<<<(unsigned)ceil_div(input_size, TB), TB, 0, c->stream>>>(pw->ptr(), to_fp4(mix_start), to_fp4(mix), (uint32_t)input_size);
        k_mix_poly_coeffs<<<(unsigned)ceil_div(count, TB), TB, 0, c->stream>>>(out->ptr(), in->ptr(), combos->ptr(), pw->ptr(),
                                                                              (uint32_t)input_size, count);
    }
    return last_launch_error("mix_poly_coeffs");
}

namespace zkh { const char* h2d(zkh_ctx* c, uint32_t* dst, const uint32_t* host, size_t n); }

// One round of synthetic division for `ny` independent polynomials of `cycles` ExtElems each (polynomial y at
// combos + poly_off[y] words, divided by (x - pts[y]); remainder to rem_out[rem_idx[y]]): the five scan launches of the
// three-level weighted suffix scan, each covering all ny polynomials through blockIdx.y.
static const char* divide_round(zkh_ctx* c, zkh_buf* combos, size_t cycles, size_t ny, const uint32_t* poly_off, const Fp4* pts,
                                const uint32_t* rem_idx, zkh_buf* rem_out, zkh_buf* quot_out = nullptr, const uint32_t* quot_off = nullptr) {
    const size_t BL0 = (size_t)TB * SC_E;
    const size_t n0 = cycles, n1 = ceil_div(n0, BL0), n2 = ceil_div(n1, TB);
    ZKH_REQUIRE(n2 <= TB, "combos_divide: polynomial too long");
    Tmp t0, t1, meta;
    ZKH_TRY(new_buf(c, 4 * n1 * ny, false, t0.out()));    // level-0 block totals, then S at level-0 block starts
    ZKH_TRY(new_buf(c, 4 * n2 * ny, false, t1.out()));    // level-1 block totals, then S at level-1 block starts
    // per-launch metadata: weights of the three levels (z, z^256, z^65536), polynomial offsets, per-y offsets into t0 / t1
    std::vector<uint32_t> m(12 * ny + 5 * ny);
    for (size_t y = 0; y < ny; y++) {
        const Fp4 z = pts[y], z256 = fp4_pow(z, BL0), z64k = fp4_pow(z256, TB);      // weights of levels 0, 1, 2
        memcpy(&m[4 * y], &z, 16); memcpy(&m[4 * ny + 4 * y], &z256, 16); memcpy(&m[8 * ny + 4 * y], &z64k, 16);
        m[12 * ny + y] = poly_off[y];
        m[13 * ny + y] = (uint32_t)(4 * n1 * y);
        m[14 * ny + y] = (uint32_t)(4 * n2 * y);
        m[15 * ny + y] = rem_idx[y];
        m[16 * ny + y] = quot_off ? quot_off[y] : poly_off[y];
    }
    ZKH_TRY(new_buf(c, m.size(), false, meta.out()));
    ZKH_TRY(h2d(c, meta->ptr(), m.data(), m.size()));
    const uint32_t *w0 = meta->ptr(), *w1 = w0 + 4 * ny, *w2 = w0 + 8 * ny, *poff = w0 + 12 * ny, *t0off = w0 + 13 * ny,
                   *t1off = w0 + 14 * ny, *ridx = w0 + 15 * ny, *qoff = w0 + 16 * ny;
    uint32_t* qbase = quot_out ? quot_out->ptr() : combos->ptr();      // where the quotients go (in place by default)
    {
        ProfScope prof(c, "combos_divide", 32.0 * cycles * ny);
        // up-sweep
        k_suffix_scan<true, SC_E><<<dim3((unsigned)n1, (unsigned)ny), TB, 0, c->stream>>>(t0->ptr(), combos->ptr(), n0, w0, nullptr, 0, 0, nullptr, poff, t0off, 0, nullptr);
        k_suffix_scan<true, 1><<<dim3((unsigned)n2, (unsigned)ny), TB, 0, c->stream>>>(t1->ptr(), t0->ptr(), n1, w1, nullptr, 0, 0, nullptr, t0off, t1off, 0, nullptr);
        // top level (n2 <= 256): S at level-1 block starts
        k_suffix_scan<false, 1><<<dim3(1, (unsigned)ny), TB, 0, c->stream>>>(t1->ptr(), t1->ptr(), n2, w2, nullptr, 0, 0, nullptr, t1off, t1off, 0, nullptr);
        // down-sweep: S at level-0 block starts, then the quotient itself (shift 1), remainder = S_0
        k_suffix_scan<false, 1><<<dim3((unsigned)n2, (unsigned)ny), TB, 0, c->stream>>>(t0->ptr(), t0->ptr(), n1, w1, t1->ptr(), n2, 0, nullptr, t0off, t0off, 4 * n2, nullptr);
        k_suffix_scan<false, SC_E><<<dim3((unsigned)n1, (unsigned)ny), TB, 0, c->stream>>>(qbase, combos->ptr(), n0, w0, t0->ptr(), n1, 1, rem_out->ptr(), poff, qoff, 4 * n1, ridx);
    }
    return last_launch_error("combos_divide");
}

extern "C" const char* zkh_combos_divide(zkh_ctx* c, zkh_buf* combos, size_t combo, size_t cycles, const uint32_t* pts,
                                         size_t n_pts, zkh_buf* rem_out) {
    ZKH_REQUIRE((combo + 1) * cycles * 4 <= combos->len, "combos_divide: combo %zu out of range", combo);
    ZKH_REQUIRE(rem_out->len >= 4 * n_pts, "combos_divide: remainder buffer too small");
    ZKH_REQUIRE(cycles <= ((size_t)1 << 24) && combos->len < ((size_t)1 << 32), "combos_divide: polynomial too long");
    const uint32_t off = (uint32_t)(4 * combo * cycles);
    for (size_t k = 0; k < n_pts; k++) {
        const Fp4 z = to_fp4(pts + 4 * k);
        const uint32_t ridx = (uint32_t)k;
        ZKH_TRY(divide_round(c, combos, cycles, 1, &off, &z, &ridx, rem_out));
    }
    return nullptr;
}
// Hal::combos_divide for the whole combo buffer at once: combo i is divided by (x - pt) for each of its points
// pts[pts_begin[i] .. pts_begin[i+1]); remainders land in rem_out in the same order.  The divisions of one combo are
// sequential, different combos are independent: round r divides, in ONE set of launches, every combo that has an r-th point.
extern "C" const char* zkh_combos_divide_all(zkh_ctx* c, zkh_buf* combos, size_t cycles, size_t n_combos, const uint32_t* pts,
                                             const uint32_t* pts_begin, zkh_buf* rem_out) {
    ZKH_REQUIRE(n_combos * cycles * 4 <= combos->len, "combos_divide_all: %zu combos of %zu cycles exceed the buffer", n_combos, cycles);
    ZKH_REQUIRE(cycles <= ((size_t)1 << 24) && combos->len < ((size_t)1 << 32), "combos_divide_all: polynomial too long");
    ZKH_REQUIRE(rem_out->len >= 4 * (size_t)pts_begin[n_combos], "combos_divide_all: remainder buffer too small");
    size_t max_pts = 0;
    bool distinct = true;
    for (size_t i = 0; i < n_combos; i++) {
        ZKH_REQUIRE(pts_begin[i + 1] >= pts_begin[i], "combos_divide_all: pts_begin must be non-decreasing");
        max_pts = std::max(max_pts, (size_t)(pts_begin[i + 1] - pts_begin[i]));
        for (uint32_t a = pts_begin[i]; a < pts_begin[i + 1]; a++)
            for (uint32_t b = a + 1; b < pts_begin[i + 1]; b++) distinct &= memcmp(pts + 4 * a, pts + 4 * b, 16) != 0;
    }
    const size_t n_pairs = pts_begin[n_combos];
    if (!n_pairs) return nullptr;
    if (!distinct || max_pts > 8 || n_pairs * cycles * 4 >= ((size_t)1 << 32)) {
        // generic fallback: one round of launches per division step (round r divides every combo that has an r-th point)
        for (size_t r = 0; r < max_pts; r++) {
            std::vector<uint32_t> off, ridx;
            std::vector<Fp4> zs;
            for (size_t i = 0; i < n_combos; i++) {
                if (pts_begin[i] + r >= pts_begin[i + 1]) continue;
                off.push_back((uint32_t)(4 * i * cycles));
                ridx.push_back((uint32_t)(pts_begin[i] + r));
                zs.push_back(to_fp4(pts + 4 * (pts_begin[i] + r)));
            }
            ZKH_TRY(divide_round(c, combos, cycles, off.size(), off.data(), zs.data(), ridx.data(), rem_out));
        }
        return nullptr;
    }
    // Partial fractions: floor(c / prod_p (x - z_p)) = sum_p A_p * floor(c / (x - z_p)),  A_p = 1 / prod_{j != p} (z_p - z_j)
    // (the difference of the two sides is (c mod D - Lagrange interpolant of c at the z_p) / D = 0), so ALL single-point
    // divisions of ALL combos are independent and share one set of five scan launches; a sixth kernel forms the weighted
    // sums.  The sequential remainders upstream checks are the divided differences of c(z_p).
    std::vector<uint32_t> poly_off(n_pairs), quot_off(n_pairs), ridx(n_pairs), meta(4 * n_combos + 4 * n_pairs + 4 * n_pairs);
    std::vector<Fp4> zs(n_pairs);
    uint32_t* m_off = meta.data();                // combo -> word offset of its polynomial
    uint32_t* m_first = m_off + n_combos;         // combo -> first pair
    uint32_t* m_cnt = m_first + n_combos;         // combo -> number of pairs
    uint32_t* m_w = m_cnt + 2 * n_combos;         // pair -> A_p (Fp4)      (one spare block keeps the arrays 16-byte aligned)
    uint32_t* m_pts = m_w + 4 * n_pairs;          // pair -> z_p (Fp4)
    for (size_t i = 0; i < n_combos; i++) {
        m_off[i] = (uint32_t)(4 * i * cycles); m_first[i] = pts_begin[i]; m_cnt[i] = pts_begin[i + 1] - pts_begin[i];
        for (uint32_t p = pts_begin[i]; p < pts_begin[i + 1]; p++) {
            poly_off[p] = (uint32_t)(4 * i * cycles); quot_off[p] = (uint32_t)(4 * (size_t)p * cycles); ridx[p] = p;
            zs[p] = to_fp4(pts + 4 * p);
            Fp4 den = Fp4::one();
            for (uint32_t j = pts_begin[i]; j < pts_begin[i + 1]; j++) if (j != p) den = den * (zs[p] - to_fp4(pts + 4 * j));
            const Fp4 w = fp4_inv(den);
            memcpy(m_w + 4 * p, &w, 16); memcpy(m_pts + 4 * p, pts + 4 * p, 16);
        }
    }
    Tmp quot, vals, dmeta;
    ZKH_TRY(new_buf(c, 4 * n_pairs * cycles, false, quot.out()));
    ZKH_TRY(new_buf(c, 4 * n_pairs, false, vals.out()));
    ZKH_TRY(new_buf(c, meta.size(), false, dmeta.out()));
    ZKH_TRY(h2d(c, dmeta->ptr(), meta.data(), meta.size()));
    ZKH_TRY(divide_round(c, combos, cycles, n_pairs, poly_off.data(), zs.data(), ridx.data(), vals, quot, quot_off.data()));
    {
        ProfScope prof(c, "combos_divide", 16.0 * cycles * (n_pairs + n_combos));
        const uint32_t* d = dmeta->ptr();
        k_combine_quotients<<<dim3((unsigned)ceil_div(cycles, TB), (unsigned)n_combos), TB, 0, c->stream>>>(
            combos->ptr(), quot->ptr(), cycles, d, d + n_combos, d + 2 * n_combos, d + 4 * n_combos);
        k_divided_differences<<<(unsigned)ceil_div(n_combos, 64), 64, 0, c->stream>>>(rem_out->ptr(), vals->ptr(), d + 4 * n_combos + 4 * n_pairs,
                                                                                      d + n_combos, d + 2 * n_combos, (uint32_t)n_combos);
    }
    return last_launch_error("combos_divide_all");
}

namespace zkh {
// `count` independent running products over columns of n0 ExtElems at io + y * col_stride words (one set of launches)
const char* prefix_products_batched(zkh_ctx* c, uint32_t* io, size_t n0, size_t count, size_t col_stride) {
    if (n0 <= 1 || !count) return nullptr;
    ZKH_REQUIRE(n0 <= ((size_t)1 << 24) && count <= 65535, "prefix_products: buffer too long");
    const size_t n1 = ceil_div(n0, (size_t)TB * SC_E), n2 = ceil_div(n1, TB);
    ZKH_REQUIRE(n2 <= TB, "prefix_products: buffer too long");
    Tmp t0, t1;
    ZKH_TRY(new_buf(c, 4 * n1 * count, false, t0.out()));
    ZKH_TRY(new_buf(c, 4 * n2 * count, false, t1.out()));
    {
        ProfScope prof(c, "prefix_products", 32.0 * n0 * count);
        const unsigned ny = (unsigned)count;
        k_prefix_prod<true, SC_E><<<dim3((unsigned)n1, ny), TB, 0, c->stream>>>(t0->ptr(), io, n0, nullptr, col_stride, 4 * n1, 0);
        k_prefix_prod<true, 1><<<dim3((unsigned)n2, ny), TB, 0, c->stream>>>(t1->ptr(), t0->ptr(), n1, nullptr, 4 * n1, 4 * n2, 0);
        k_prefix_prod<false, 1><<<dim3(1, ny), TB, 0, c->stream>>>(t1->ptr(), t1->ptr(), n2, nullptr, 4 * n2, 4 * n2, 0);                   // inclusive over level-1 totals
        k_prefix_prod<false, 1><<<dim3((unsigned)n2, ny), TB, 0, c->stream>>>(t0->ptr(), t0->ptr(), n1, t1->ptr(), 4 * n1, 4 * n1, 4 * n2);  // inclusive over level-0 totals
        k_prefix_prod<false, SC_E><<<dim3((unsigned)n1, ny), TB, 0, c->stream>>>(io, io, n0, t0->ptr(), col_stride, col_stride, 4 * n1);
    }
    return last_launch_error("prefix_products");
}
}  // namespace zkh

extern "C" const char* zkh_prefix_products(zkh_ctx* c, zkh_buf* io) {
    ZKH_REQUIRE(io->len % 4 == 0, "prefix_products: not an ExtElem buffer");
    return prefix_products_batched(c, io->ptr(), io->len / 4, 1, 0);
}
