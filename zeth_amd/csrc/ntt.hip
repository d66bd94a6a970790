// ntt.hip — batched BabyBear NTTs for gfx950: Hal::{batch_interpolate_ntt, batch_expand_into_evaluate_ntt,
// zk_shift, batch_bit_reverse} (risc0-zkp 3.0.2 src/hal/mod.rs; semantics src/core/ntt.rs + src/hal/cpu.rs,
// un-vendored: /root/reference/Cargo.lock:5393).  Reached from /root/reference/crates/host/src/lib.rs:137.
//
// Structure (MI355X-first, not sppark's): a size-2^k transform is cut into <= 3 HBM passes.  One pass owns R
// index bits [L, L+R): a workgroup stages a tile of 2^R x T words in LDS (T consecutive low indices per row so
// every global access is a contiguous run), does all R radix-2 layers there, and applies the four-step
// inter-pass twiddle w_{L+R}^(l * bitrev_R(m)) from two 4096-entry tables (w^lo, w^(4096 hi)) on the way in
// (forward, DIT) or out (inverse, DIF).  expand (x4 replication, first two layers skipped), n^-1 scaling and
// the zk coset shift 3^bitrev(i) are fused into the first/last pass, so neither the zero-padded input nor the
// unshifted coefficients ever touch HBM.
//   HBM traffic: 8 B per element per pass (one read + one write), i.e. 16 B/elem for k <= 22, 24 B/elem up to 2^26.
#include "common.h"

using namespace zkh;

namespace {

constexpr int NTT_THREADS = 256;

struct PassParams {
    const uint32_t* in;
    uint32_t* out;
    size_t in_col_stride;    // words between columns of `in`
    size_t out_col_stride;
    uint32_t log_n;          // transform size 2^log_n (output size)
    uint32_t L, R;           // this pass owns index bits [L, L+R)
    uint32_t log_t;          // tile width (consecutive low indices) = 2^log_t, log_t <= L
    uint32_t expand_bits;    // forward first pass: in[i >> expand_bits], first `expand_bits` layers skipped
    uint32_t first_layer;    // forward: first sub-layer to run in this tile (1-based), normally 1
    uint32_t twiddle;        // 1: apply the inter-pass twiddle (pre for DIT, post for DIF)
    uint32_t scale;          // inverse last pass: multiply by n^-1 (Montgomery word) ...
    uint32_t zk_shift;       // ... and by 3^bitrev(i)
    const uint32_t* tile_tw; // w_{2^12}^j
    const uint32_t* layer_tw; // per-layer tables: [2^(j-1) + e] = w_j^e
    const uint32_t* tw_lo;   // w_{2^26}^lo
    const uint32_t* tw_hi;   // w_{2^26}^(4096 hi), hi < 2^14
    const uint32_t* sh_lo;
    const uint32_t* sh_hi;
    uint32_t tiles_per_col;
    uint32_t lazy;           // forward register-radix pair (low12 + high<8|10>): see radix_layers<..., LAZY>
    uint32_t lazy_comp;      // R^(number of lazy layers) as a Montgomery word: folded into the four-step twiddle
    uint32_t shift16[16];    // inverse last pass: n^-1 * 3^(bitrev4(k) << (log_n - 4)), k < 16 (Montgomery; 0 = unused)
};

// global index of tile element (m, t):  a*2^(L+R) + m*2^L + l0 + t
template <bool INVERSE>
__global__ __launch_bounds__(NTT_THREADS) void k_ntt_pass(PassParams p) {
    extern __shared__ __attribute__((aligned(16))) uint32_t tile[];
    const uint32_t T = 1u << p.log_t, rows = 1u << p.R, elems = rows << p.log_t;
    const uint32_t tid = threadIdx.x;
    const uint32_t tile_id = xcd_remap(blockIdx.x, p.tiles_per_col);
    const uint32_t col = blockIdx.y;
    // tile_id -> (a, l-tile)
    const uint32_t ltiles = 1u << (p.L - p.log_t);
    const uint32_t a = tile_id >> (p.L - p.log_t), lt = tile_id & (ltiles - 1);
    const size_t base = ((size_t)a << (p.L + p.R)) + ((size_t)lt << p.log_t);
    const uint32_t l0 = lt << p.log_t;
    const uint32_t* in = p.in + (size_t)col * p.in_col_stride;
    uint32_t* out = p.out + (size_t)col * p.out_col_stride;
    const uint32_t tw_shift = MAX_LOG_N - (p.L + p.R);   // w_{L+R}^e = w_26^(e << tw_shift)

    // ---- load (+ expand, + DIT pre-twiddle) ----
    for (uint32_t e = tid; e < elems; e += NTT_THREADS) {
        const uint32_t m = e >> p.log_t, t = e & (T - 1);
        const size_t gi = base + ((size_t)m << p.L) + t;
        uint32_t v = in[gi >> p.expand_bits];
        if (!INVERSE && p.twiddle) {
            const uint32_t r = __brev(m) >> (32 - p.R);
            const uint32_t ex = ((l0 + t) * r) << tw_shift;          // < 2^26
            const uint32_t w = mul_mod(p.tw_lo[ex & (TW_SIZE - 1)], p.tw_hi[ex >> TW_BITS]);
            v = mul_mod(v, w);
        }
        tile[e] = v;
    }
    __syncthreads();

    // ---- R radix-2 layers in LDS ----
    const uint32_t half_elems = elems >> 1;
    if (INVERSE) {
        for (uint32_t j = p.R; j >= 1; j--) {
            const uint32_t hb = j - 1;                                 // partner distance 2^hb rows
            for (uint32_t q = tid; q < half_elems; q += NTT_THREADS) {
                const uint32_t t = q & (T - 1), mm = q >> p.log_t;
                const uint32_t low = mm & ((1u << hb) - 1), m = ((mm >> hb) << j) | low;
                const uint32_t i0 = (m << p.log_t) | t, i1 = i0 + (1u << (hb + p.log_t));
                const uint32_t x = tile[i0], y = tile[i1];
                const uint32_t w = p.tile_tw[low << (LDS_TW_LOG - j)];
                tile[i0] = add_mod(x, y);
                tile[i1] = mul_mod(x - y + P, w);          // lazy difference in (0, 2P): valid Montgomery operand
            }
            __syncthreads();
        }
    } else {
        for (uint32_t j = p.first_layer; j <= p.R; j++) {
            const uint32_t hb = j - 1;
            for (uint32_t q = tid; q < half_elems; q += NTT_THREADS) {
                const uint32_t t = q & (T - 1), mm = q >> p.log_t;
                const uint32_t low = mm & ((1u << hb) - 1), m = ((mm >> hb) << j) | low;
                const uint32_t i0 = (m << p.log_t) | t, i1 = i0 + (1u << (hb + p.log_t));
                const uint32_t w = p.tile_tw[low << (LDS_TW_LOG - j)];
                const uint32_t x = tile[i0], y = mul_mod(tile[i1], w);
                tile[i0] = add_mod(x, y);
                tile[i1] = sub_mod(x, y);
            }
            __syncthreads();
        }
    }

    // ---- store (+ DIF post-twiddle, + n^-1, + zk shift) ----
    for (uint32_t e = tid; e < elems; e += NTT_THREADS) {
        const uint32_t m = e >> p.log_t, t = e & (T - 1);
        const size_t gi = base + ((size_t)m << p.L) + t;
        uint32_t v = tile[e];
        if (INVERSE) {
            if (p.twiddle) {
                const uint32_t r = __brev(m) >> (32 - p.R);
                const uint32_t ex = ((l0 + t) * r) << tw_shift;
                const uint32_t w = mul_mod(p.tw_lo[ex & (TW_SIZE - 1)], p.tw_hi[ex >> TW_BITS]);
                v = mul_mod(v, w);
            }
            if (p.scale) v = mul_mod(v, p.scale);
            if (p.zk_shift) {
                const uint32_t pos = (uint32_t)gi;                     // position within the column
                const uint32_t ex = __brev(pos) >> (32 - p.log_n);
                v = mul_mod(v, mul_mod(p.sh_lo[ex & (TW_SIZE - 1)], p.sh_hi[ex >> TW_BITS]));
            }
        }
        out[gi] = v;
    }
}

// Top pass, index bits [L, L + R) with L + R = log_n, L >= 12 and R in 1..4 — what is left above the register-radix passes at 2^21 and
// 2^23 .. 2^26 (po2 19, 21 .. 24).  One lane = 4 consecutive low indices x ALL 2^R rows, in registers: rows are 2^L words apart, so every
// load and store is a 16-byte piece of a contiguous run (the pass streams the columns once, at HBM speed), the R butterfly layers use
// wave-uniform twiddles (w_{2^j}^low: scalar loads), and the four-step twiddle of row m at low index j is (w^j)^bitrev(m): ONE two-level
// table root per lane, the neighbours and the powers by products.  Same arithmetic as k_ntt_pass on the same tile (canonical words in,
// canonical words out): results are identical.  (round 5: the generic LDS kernel spent 8.8 ms per po2-21 seal here, 15.3 at po2 22.)
template <int R, bool INVERSE>
__global__ __launch_bounds__(256) void k_ntt_top(PassParams p) {
    constexpr int N = 1 << R;
    const uint32_t j0 = (blockIdx.x * 256u + threadIdx.x) << 2;             // first of this lane's 4 low indices, < 2^L
    const uint32_t* in = p.in + (size_t)blockIdx.y * p.in_col_stride + j0;
    uint32_t* out = p.out + (size_t)blockIdx.y * p.out_col_stride + j0;
    const uint32_t tw_shift = MAX_LOG_N - p.log_n;
    uint32_t v[N][4];
#pragma unroll
    for (int m = 0; m < N; m++) {
        const uint4 w = *(const uint4*)(in + ((size_t)m << p.L));
        v[m][0] = w.x; v[m][1] = w.y; v[m][2] = w.z; v[m][3] = w.w;
    }
    auto root = [&](uint32_t e) -> uint32_t {                                // w_{log_n}^e, e < 2^log_n
        const uint32_t ex = e << tw_shift;
        return mul_mod(p.tw_lo[ex & (TW_SIZE - 1)], p.tw_hi[ex >> TW_BITS]);
    };
    // v[m][c] *= (w^(j0 + c))^bitrev_R(m): forward before the layers (DIT pre-twiddle), inverse after them (DIF post-twiddle)
    auto four_step = [&]() {
        const uint32_t step = root(1u);                                      // (wave-uniform)
        uint32_t g = root(j0);
#pragma unroll
        for (int c = 0; c < 4; c++) {
            uint32_t pw[N];
            pw[1] = g;
#pragma unroll
            for (int r = 2; r < N; r++) pw[r] = mul_mod(pw[r - 1], g);
#pragma unroll
            for (int m = 1; m < N; m++) {
                const int r = (int)(__builtin_bitreverse32((uint32_t)m) >> (32 - R));
                v[m][c] = mul_mod(v[m][c], pw[r]);
            }
            if (c < 3) g = mul_mod(g, step);
        }
    };
    if (!INVERSE) {
        four_step();
#pragma unroll
        for (int j = 1; j <= R; j++) {
            const int hb = j - 1;
#pragma unroll
            for (int i0 = 0; i0 < N; i0++) {
                if (i0 & (1 << hb)) continue;
                const int low = i0 & ((1 << hb) - 1), i1 = i0 + (1 << hb);
                const uint32_t w = low ? p.tile_tw[(uint32_t)low << (LDS_TW_LOG - j)] : 0u;     // low == 0: the unit twiddle, no product
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const uint32_t x = v[i0][c], y = low ? mul_mod(v[i1][c], w) : v[i1][c];
                    v[i0][c] = add_mod(x, y);
                    v[i1][c] = sub_mod(x, y);
                }
            }
        }
    } else {
#pragma unroll
        for (int j = R; j >= 1; j--) {
            const int hb = j - 1;
#pragma unroll
            for (int i0 = 0; i0 < N; i0++) {
                if (i0 & (1 << hb)) continue;
                const int low = i0 & ((1 << hb) - 1), i1 = i0 + (1 << hb);
                const uint32_t w = low ? p.tile_tw[(uint32_t)low << (LDS_TW_LOG - j)] : 0u;
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const uint32_t x = v[i0][c], y = v[i1][c];
                    v[i0][c] = add_mod(x, y);
                    v[i1][c] = low ? mul_mod(x - y + P, w) : sub_mod(x, y);    // lazy difference in (0, 2P): a valid Montgomery operand
                }
            }
        }
        four_step();
    }
#pragma unroll
    for (int m = 0; m < N; m++) *(uint4*)(out + ((size_t)m << p.L)) = make_uint4(v[m][0], v[m][1], v[m][2], v[m][3]);
}

// zk_shift alone (Hal::zk_shift): io[c][i] *= 3^bitrev(i)
__global__ void k_zk_shift(uint32_t* io, size_t total, uint32_t log_n, const uint32_t* sh_lo, const uint32_t* sh_hi) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const uint32_t pos = (uint32_t)(i & (((size_t)1 << log_n) - 1));
    const uint32_t ex = log_n ? __brev(pos) >> (32 - log_n) : 0;
    io[i] = mul_mod(io[i], mul_mod(sh_lo[ex & (TW_SIZE - 1)], sh_hi[ex >> TW_BITS]));
}

// In-place bit reversal of each column through LDS tiles so that both the read and the write side are contiguous
// runs of 2^H words: index = (hi:H | mid | lo:H)  ->  (rev lo | rev mid | rev hi).  Tile `mid` swaps with rev(mid).
constexpr int BR_H = 5;
__global__ __launch_bounds__(256) void k_bit_reverse_tiled(uint32_t* io, uint32_t log_n, size_t col_stride) {
    __shared__ uint32_t ta[1 << (2 * BR_H)], tb[1 << (2 * BR_H)];
    const uint32_t mid_bits = log_n - 2 * BR_H;
    const uint32_t mid = blockIdx.x;
    const uint32_t rmid = mid_bits ? __brev(mid) >> (32 - mid_bits) : 0;
    if (rmid < mid) return;                      // the pair is handled by the block with the smaller index
    uint32_t* col = io + (size_t)blockIdx.y * col_stride;
    const uint32_t H = 1u << BR_H;
    for (uint32_t e = threadIdx.x; e < H * H; e += blockDim.x) {
        const uint32_t hi = e >> BR_H, lo = e & (H - 1);
        ta[e] = col[((size_t)hi << (log_n - BR_H)) | ((size_t)mid << BR_H) | lo];
        if (rmid != mid) tb[e] = col[((size_t)hi << (log_n - BR_H)) | ((size_t)rmid << BR_H) | lo];
    }
    __syncthreads();
    for (uint32_t e = threadIdx.x; e < H * H; e += blockDim.x) {
        // destination (hi', lo') inside tile rmid receives source (hi = rev(lo'), lo = rev(hi')) of tile mid
        const uint32_t hi2 = e >> BR_H, lo2 = e & (H - 1);
        const uint32_t shi = __brev(lo2) >> (32 - BR_H), slo = __brev(hi2) >> (32 - BR_H);
        const uint32_t src = (shi << BR_H) | slo;
        col[((size_t)hi2 << (log_n - BR_H)) | ((size_t)rmid << BR_H) | lo2] = ta[src];
        if (rmid != mid) col[((size_t)hi2 << (log_n - BR_H)) | ((size_t)mid << BR_H) | lo2] = tb[src];
    }
}
// small columns: direct swap
__global__ void k_bit_reverse_small(uint32_t* io, uint32_t log_n, size_t count) {
    size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t n = (size_t)1 << log_n;
    if (g >= n * count) return;
    uint32_t i = (uint32_t)(g & (n - 1));
    uint32_t r = log_n ? __brev(i) >> (32 - log_n) : 0;
    if (i < r) {
        uint32_t* col = io + (g - i);
        uint32_t x = col[i], y = col[r];
        col[i] = y; col[r] = x;
    }
}


// =====================================================================================================
// Register-radix fast paths (the shapes a po2 >= 16 seal actually uses).  Each lane keeps 16 elements in
// VGPRs and runs 4 radix-2 layers there; LDS is only the exchange medium between rounds (2 exchanges per
// 4096-point tile instead of 12 read-modify-write sweeps), twiddles come from per-layer tables so that
// consecutive lanes read consecutive words.
// =====================================================================================================
//
// LAZY (forward only): values are SIGNED representatives in (-P, P) and every layer divides by R = 2^32:
//   a = (x + w y) / R,  b = (x - w y) / R   as   smont_reduce(sext(x) +- w*y)   (|x + w y| <= P + P^2 < P 2^31),
// with w read from the PLAIN (non-Montgomery) twiddle table, so that x and the product pick up the same factor 1/R,
// i.e. one sign extension, two v_mad_i64_i32 and two uncorrected reductions per butterfly (11.9 add-slots instead of
// 14.1 for product + modular add + modular subtract).  The accumulated R^-layers is cancelled by one constant folded
// into the four-step twiddle of the strided pass, whose outputs are canonicalised once.  Exact field arithmetic:
// results are identical to the plain path.
template <int LOGR, bool INVERSE, bool BASE0, int J_LO, bool LAZY = false>
__device__ __forceinline__ void radix_layers(uint32_t (&v)[1 << LOGR], const uint32_t* __restrict__ ltab,
                                             const uint32_t base_low, const int first_b) {
    constexpr int N = 1 << LOGR;
    if (!INVERSE && LAZY) {
#pragma unroll
        for (int b = 0; b < LOGR; b++) {
            if (b < first_b) continue;
            const uint32_t* tw = ltab + (1u << (J_LO + b - 1));
#pragma unroll
            for (int kk = 0; kk < (1 << b); kk++) {
                const bool unit = BASE0 && kk == 0;
                const int32_t w = (int32_t)(unit ? 1u : tw[base_low + ((uint32_t)kk << (J_LO - 1))]), nw = -w;   // plain residue
#pragma unroll
                for (int hi = 0; hi < (N >> (b + 1)); hi++) {
                    const int k = (hi << (b + 1)) | kk;
                    const int64_t x = (int64_t)(int32_t)v[k];
                    const int32_t y = (int32_t)v[k + (1 << b)];
                    // BASE0 call sites pass base_low = 0: the twiddle is wave-uniform and stays in an SGPR (the "v"
                    // operand of mad_i64 would cost a v_mov per product)
                    v[k] = (uint32_t)smont_reduce(BASE0 ? mad_i64_k(y, w, x) : mad_i64(y, w, x));
                    v[k + (1 << b)] = (uint32_t)smont_reduce(BASE0 ? mad_i64_k(y, nw, x) : mad_i64(y, nw, x));
                }
            }
        }
    } else if (INVERSE) {
#pragma unroll
        for (int b = LOGR - 1; b >= 0; b--) {
            const uint32_t* tw = ltab + (1u << (J_LO + b - 1));
#pragma unroll
            for (int kk = 0; kk < (1 << b); kk++) {
                const bool unit = BASE0 && kk == 0;
                const uint32_t w = unit ? 0u : tw[base_low + ((uint32_t)kk << (J_LO - 1))];
#pragma unroll
                for (int hi = 0; hi < (N >> (b + 1)); hi++) {
                    const int k = (hi << (b + 1)) | kk;
                    const uint32_t x = v[k], y = v[k + (1 << b)];
                    v[k] = add_mod(x, y);
                    // the difference feeds a Montgomery product, which accepts a lazy operand: x - y + P in (0, 2P)
                    // (2 P^2 < P 2^32) costs two plain adds instead of a modular subtract
                    v[k + (1 << b)] = unit ? sub_mod(x, y) : mul_mod(x - y + P, w);
                }
            }
        }
    } else {
#pragma unroll
        for (int b = 0; b < LOGR; b++) {
            if (b < first_b) continue;
            const uint32_t* tw = ltab + (1u << (J_LO + b - 1));
#pragma unroll
            for (int kk = 0; kk < (1 << b); kk++) {
                const bool unit = BASE0 && kk == 0;
                const uint32_t w = unit ? 0u : tw[base_low + ((uint32_t)kk << (J_LO - 1))];
#pragma unroll
                for (int hi = 0; hi < (N >> (b + 1)); hi++) {
                    const int k = (hi << (b + 1)) | kk;
                    const uint32_t x = v[k], y = unit ? v[k + (1 << b)] : mul_mod(v[k + (1 << b)], w);
                    v[k] = add_mod(x, y);
                    v[k + (1 << b)] = sub_mod(x, y);
                }
            }
        }
    }
}

// Lowest pass, index bits [0, 12): one workgroup = 4096 contiguous words of one column.
template <bool INVERSE, bool LAZY = false>
__global__ __launch_bounds__(256) void k_ntt_low12(PassParams p) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[4096];
    const uint32_t tid = threadIdx.x;
    const uint32_t tile = xcd_remap(blockIdx.x, p.tiles_per_col);
    const size_t base = (size_t)tile << 12;
    const uint32_t* in = p.in + (size_t)blockIdx.y * p.in_col_stride;
    uint32_t* out = p.out + (size_t)blockIdx.y * p.out_col_stride;
    const uint32_t* __restrict__ ltab = p.layer_tw;
    const uint32_t hi = tid >> 4, low = tid & 15;
    uint32_t v[16];
    if (INVERSE) {
#pragma unroll
        for (int k = 0; k < 16; k++) v[k] = in[base + k * 256 + tid];
        radix_layers<4, true, false, 9>(v, ltab, tid, 0);                       // layers 12..9
#pragma unroll
        for (int k = 0; k < 16; k++) lds[k * 256 + tid] = v[k];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; k++) v[k] = lds[hi * 256 + k * 16 + low];
        radix_layers<4, true, false, 5>(v, ltab, low, 0);                       // layers 8..5
#pragma unroll
        for (int k = 0; k < 16; k++) lds[hi * 256 + k * 16 + low] = v[k];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint4 w = ((const uint4*)lds)[tid * 4 + q];
            v[4 * q] = w.x; v[4 * q + 1] = w.y; v[4 * q + 2] = w.z; v[4 * q + 3] = w.w;
        }
        radix_layers<4, true, true, 1>(v, ltab, 0, 0);                          // layers 4..1
        const size_t pos0 = base + (size_t)tid * 16;
        if (p.zk_shift) {
            // 3^bitrev(pos0 + k) = 3^bitrev(pos0) * 3^(bitrev4(k) << (log_n - 4)): one two-level lookup per lane, the 16
            // k-dependent factors (already times n^-1) are kernel arguments
            const uint32_t ex = __brev((uint32_t)pos0) >> (32 - p.log_n);
            const uint32_t b = mul_mod(p.sh_lo[ex & (TW_SIZE - 1)], p.sh_hi[ex >> TW_BITS]);
#pragma unroll
            for (int k = 0; k < 16; k++) v[k] = mul_mod(mul_mod(v[k], b), p.shift16[k]);
        } else if (p.scale) {
#pragma unroll
            for (int k = 0; k < 16; k++) v[k] = mul_mod(v[k], p.scale);
        }
#pragma unroll
        for (int q = 0; q < 4; q++) ((uint4*)(out + pos0))[q] = make_uint4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
    } else {
        const size_t pos0 = base + (size_t)tid * 16;
        if (p.expand_bits == 0) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint4 w = ((const uint4*)(in + pos0))[q];
                v[4 * q] = w.x; v[4 * q + 1] = w.y; v[4 * q + 2] = w.z; v[4 * q + 3] = w.w;
            }
        } else if (p.expand_bits == 2) {
            // 16 outputs of a lane replicate 4 consecutive inputs: one 16-byte load instead of 16 (4x redundant) dword loads
            const uint4 w = *(const uint4*)(in + (pos0 >> 2));
            v[0] = v[1] = v[2] = v[3] = w.x; v[4] = v[5] = v[6] = v[7] = w.y;
            v[8] = v[9] = v[10] = v[11] = w.z; v[12] = v[13] = v[14] = v[15] = w.w;
        } else {
#pragma unroll
            for (int k = 0; k < 16; k++) v[k] = in[(pos0 + k) >> p.expand_bits];
        }
        radix_layers<4, false, true, 1, LAZY>(v, ltab, 0, (int)p.expand_bits);  // layers 1..4 (first expand_bits skipped)
#pragma unroll
        for (int q = 0; q < 4; q++) ((uint4*)lds)[tid * 4 + q] = make_uint4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; k++) v[k] = lds[hi * 256 + k * 16 + low];
        radix_layers<4, false, false, 5, LAZY>(v, ltab, low, 0);                // layers 5..8
#pragma unroll
        for (int k = 0; k < 16; k++) lds[hi * 256 + k * 16 + low] = v[k];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; k++) v[k] = lds[k * 256 + tid];
        radix_layers<4, false, false, 9, LAZY>(v, ltab, tid, 0);                // layers 9..12 (LAZY: signed words scaled by R^-(12 - expand_bits))
#pragma unroll
        for (int k = 0; k < 16; k++) out[base + k * 256 + tid] = v[k];
    }
}

// Strided pass, index bits [L, L+RH) with RH in {8, 10}: tile = 2^RH rows x 16 consecutive words, one lane per
// (row group, column), 16 rows per lane.  Inverse: first pass (reads the witness, DIF, post-twiddle).  Forward:
// last pass (pre-twiddle, DIT, in place).
template <int RH, bool INVERSE, bool LAZY = false>
__global__ __launch_bounds__(1 << RH) void k_ntt_high(PassParams p) {
    constexpr int LT = 4;                 // tile width 2^LT words: 64-byte runs, 2^RH lanes
    constexpr uint32_t TW = 1u << LT;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];     // [2^RH][16]
    const uint32_t tid = threadIdx.x, t = tid & (TW - 1), g = tid >> LT;
    const uint32_t tile_id = xcd_remap(blockIdx.x, p.tiles_per_col), col = blockIdx.y;
    const uint32_t lt_bits = p.L - LT;
    const uint32_t a = tile_id >> lt_bits, lt = tile_id & ((1u << lt_bits) - 1);
    const size_t base = ((size_t)a << (p.L + RH)) + ((size_t)lt << LT);      // wave-uniform: first word of the tile
    const uint32_t lcol = (lt << LT) + t;
    const uint32_t tw_shift = MAX_LOG_N - (p.L + RH);
    // Tile accesses as uniform base (SGPR pair) + 32-bit per-lane byte offset — global_load's native addressing form; a
    // 64-bit VGPR address per access costs three extra VALU instructions each, ~10 % of this VALU-bound kernel.
    // (row < 2^RH, L <= 14: the offset stays below 2^RH+L+2 <= 2^26 bytes.)
    const char* in = (const char*)(p.in + (size_t)col * p.in_col_stride + base);
    char* out = (char*)(p.out + (size_t)col * p.out_col_stride + base);
    const uint32_t tb = t * 4u;
#define TILE_IN(row) (*(const uint32_t*)(in + ((((uint32_t)(row)) << (p.L + 2)) + tb)))
#define TILE_OUT(row) (*(uint32_t*)(out + ((((uint32_t)(row)) << (p.L + 2)) + tb)))
    const uint32_t* __restrict__ ltab = p.layer_tw;
    // twiddle products: the lazy forward pass only ever uses them as signed operands of smont, so they may stay signed
    // representatives in (-P, P) themselves (3 instructions per product instead of 5)
    auto tmul = [](uint32_t x, uint32_t y) -> uint32_t {
        return (LAZY && !INVERSE) ? (uint32_t)smont((int32_t)x, (int32_t)y) : mul_mod(x, y);
    };
    auto root = [&](uint32_t e) -> uint32_t {                  // w_{L+RH}^e, e < 2^(L+RH)
        const uint32_t ex = e << tw_shift;
        return tmul(p.tw_lo[ex & (TW_SIZE - 1)], p.tw_hi[ex >> TW_BITS]);
    };
    // Four-step twiddles w^(lcol * bitrev(m)) of this lane's 16 rows, factored so that only 3 (RH = 10) or 2 (RH = 8)
    // table gathers are needed instead of 16: bitrev splits over the bit fields of m.
    uint32_t tw[16];
    if (RH == 10) {         // m = g*16 + i*4 + k: bitrev10(m) = br2(k)*256 + br2(i)*64 + br6(g);  slot = 4*i + k
        uint32_t w0 = root(lcol * (__brev(g) >> 26));
        if (LAZY) w0 = tmul(w0, p.lazy_comp);                // cancels the R^-layers of both lazy passes
        const uint32_t u1 = root(lcol * 64u), v1 = root(lcol * 256u);
        const uint32_t u2 = tmul(u1, u1), u3 = tmul(u2, u1), v2 = tmul(v1, v1), v3 = tmul(v2, v1);
        const uint32_t wi[4] = {w0, tmul(w0, u2), tmul(w0, u1), tmul(w0, u3)};      // U^br2(i)
#pragma unroll
        for (int i = 0; i < 4; i++) {
            tw[4 * i] = wi[i]; tw[4 * i + 1] = tmul(wi[i], v2); tw[4 * i + 2] = tmul(wi[i], v1); tw[4 * i + 3] = tmul(wi[i], v3);
        }
    } else {                // m = g*16 + k: bitrev8(m) = br4(k)*16 + br4(g);  slot = k
        uint32_t w0 = root(lcol * (__brev(g) >> 28));
        if (LAZY) w0 = tmul(w0, p.lazy_comp);
        const uint32_t q1 = root(lcol * 16u);
        uint32_t qp[16];
        qp[0] = w0;
#pragma unroll
        for (int e = 1; e < 16; e++) qp[e] = tmul(qp[e - 1], q1);     // w0 * Q^e
#pragma unroll
        for (int k = 0; k < 16; k++) tw[k] = qp[__builtin_bitreverse8((unsigned char)k) >> 4];
    }
    uint32_t v[16];
    if (RH == 10) {
        const uint32_t hi = g >> 2, low = g & 3;
        if (INVERSE) {
#pragma unroll
            for (int k = 0; k < 16; k++) v[k] = TILE_IN(k * 64 + g);
            radix_layers<4, true, false, 7>(v, ltab, g, 0);                     // sub-layers 10..7
#pragma unroll
            for (int k = 0; k < 16; k++) lds[(k * 64 + g) * TW + t] = v[k];
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 16; k++) v[k] = lds[(hi * 64 + k * 4 + low) * TW + t];
            radix_layers<4, true, false, 3>(v, ltab, low, 0);                   // 6..3
#pragma unroll
            for (int k = 0; k < 16; k++) lds[(hi * 64 + k * 4 + low) * TW + t] = v[k];
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 4; i++) {
                uint32_t u[4];
                const uint32_t m0 = (g * 4 + i) * 4;
#pragma unroll
                for (int k = 0; k < 4; k++) u[k] = lds[(m0 + k) * TW + t];
                radix_layers<2, true, true, 1>(u, ltab, 0, 0);                  // 2..1
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    uint32_t x = mul_mod(u[k], tw[4 * i + k]);
                    if (p.scale) x = mul_mod(x, p.scale);
                    TILE_OUT(m0 + k) = x;
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                uint32_t u[4];
                const uint32_t m0 = (g * 4 + i) * 4;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t x = TILE_IN(m0 + k);
                    u[k] = LAZY ? (uint32_t)smont((int32_t)x, (int32_t)tw[4 * i + k]) : mul_mod(x, tw[4 * i + k]);
                }
                radix_layers<2, false, true, 1, LAZY>(u, ltab, 0, 0);
#pragma unroll
                for (int k = 0; k < 4; k++) lds[(m0 + k) * TW + t] = u[k];
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 16; k++) v[k] = lds[(hi * 64 + k * 4 + low) * TW + t];
            radix_layers<4, false, false, 3, LAZY>(v, ltab, low, 0);
#pragma unroll
            for (int k = 0; k < 16; k++) lds[(hi * 64 + k * 4 + low) * TW + t] = v[k];
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 16; k++) v[k] = lds[(k * 64 + g) * TW + t];
            radix_layers<4, false, false, 7, LAZY>(v, ltab, g, 0);
#pragma unroll
            for (int k = 0; k < 16; k++) TILE_OUT(k * 64 + g) = LAZY ? canon((int32_t)v[k]) : v[k];
        }
    } else {   // RH == 8: 256 rows, g < 16
        if (INVERSE) {
#pragma unroll
            for (int k = 0; k < 16; k++) v[k] = TILE_IN(k * 16 + g);
            radix_layers<4, true, false, 5>(v, ltab, g, 0);                     // 8..5
#pragma unroll
            for (int k = 0; k < 16; k++) lds[(k * 16 + g) * TW + t] = v[k];
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 16; k++) v[k] = lds[(g * 16 + k) * TW + t];
            radix_layers<4, true, true, 1>(v, ltab, 0, 0);                      // 4..1
#pragma unroll
            for (int k = 0; k < 16; k++) {
                uint32_t x = mul_mod(v[k], tw[k]);
                if (p.scale) x = mul_mod(x, p.scale);
                TILE_OUT(g * 16 + k) = x;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const uint32_t x = TILE_IN(g * 16 + k);
                v[k] = LAZY ? (uint32_t)smont((int32_t)x, (int32_t)tw[k]) : mul_mod(x, tw[k]);
            }
            radix_layers<4, false, true, 1, LAZY>(v, ltab, 0, 0);
#pragma unroll
            for (int k = 0; k < 16; k++) lds[(g * 16 + k) * TW + t] = v[k];
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 16; k++) v[k] = lds[(k * 16 + g) * TW + t];
            radix_layers<4, false, false, 5, LAZY>(v, ltab, g, 0);
#pragma unroll
            for (int k = 0; k < 16; k++) TILE_OUT(k * 16 + g) = LAZY ? canon((int32_t)v[k]) : v[k];
        }
    }
}
#undef TILE_IN
#undef TILE_OUT

struct Pass { uint32_t L, R; };
// Cut log_n index bits into HBM passes of <= 12 bits.  The lowest pass works on contiguous runs (cheapest), the
// strided passes above it are kept <= 10 bits so a 2^R x 16-word tile stays <= 64 KiB.
std::vector<Pass> plan_passes(uint32_t log_n) {
    std::vector<Pass> v;
    if (log_n <= 12) { v.push_back({0, log_n}); return v; }
    if (log_n < 18) {                       // two balanced LDS passes
        const uint32_t low = (log_n + 1) / 2;
        v.push_back({0, low});
        v.push_back({low, log_n - low});
        return v;
    }
    // >= 2^18: the contiguous 4096-word register-radix pass, then strided passes.  2^8 and 2^10 rows have
    // register-radix kernels; what is left over (1..4 bits at 2^21, 2^23 .. 2^26: po2 19, 21 .. 24) goes to a last streaming pass in
    // registers (k_ntt_top) instead of a 9..12-layer LDS sweep.
    const uint32_t rem = log_n - 12;
    v.push_back({0, 12});
    if (rem == 8 || rem == 10 || rem <= 7) v.push_back({12, rem});
    else if (rem == 9) { v.push_back({12, 8}); v.push_back({20, 1}); }
    else if (rem == 11 || rem == 12) { v.push_back({12, 8}); v.push_back({20, rem - 8}); }     // the top pass streams 1..4 bits at the same cost: the cheaper strided pass below it
    else { v.push_back({12, 10}); v.push_back({22, rem - 10}); }
    return v;
}

const char* run_transform(zkh_ctx* c, bool inverse, const uint32_t* in, size_t in_col_stride, uint32_t* out,
                          size_t out_col_stride, uint32_t log_n, size_t count, uint32_t expand_bits, bool zk, const char* name) {
    ZKH_REQUIRE(log_n <= (uint32_t)MAX_LOG_N, "%s: transform size 2^%u exceeds the supported 2^%d", name, log_n, MAX_LOG_N);
    ZKH_REQUIRE(count <= 65535, "%s: too many columns (%zu)", name, count);
    std::vector<Pass> passes = plan_passes(log_n);
    const size_t n = (size_t)1 << log_n;
    Fp ninv = fp_inv(fp_encode((uint32_t)(n % P)));
    const size_t npass = passes.size();
    // forward transforms whose two passes are both register-radix kernels (2^20 and 2^22: what po2-18/20 seals
    // expand into) run the lazy signed butterflies; the constant R^(layers run) rides on the four-step twiddle
    // (letting the lazy pair also run under a third pass - 2^21 / 2^23 / 2^24 - is bit-exact and was measured flat twice; the
    // per-shape twiddle matrix, 32-byte tiles and the column-fast grid were measured slower: profiles/README.md, r03_ntt_matrix.txt,
    // r03_ntt_ab*.jsonl.  None of those variants is in the library any more.)
    const bool lazy = !inverse && npass == 2 && passes[0].R == 12 && (passes[1].R == 8 || passes[1].R == 10) && expand_bits <= 4;
    const uint32_t lazy_comp = lazy ? fp_pow(Fp::raw(R2), passes[0].R - expand_bits + passes[1].R).v : 0;
    for (size_t pi = 0; pi < npass; pi++) {
        // inverse: high bits first; forward: low bits first
        const Pass ps = inverse ? passes[npass - 1 - pi] : passes[pi];
        PassParams p{};
        const bool first = pi == 0, last = pi + 1 == npass;
        p.in = first ? in : out;
        p.in_col_stride = first ? in_col_stride : out_col_stride;
        p.out = out; p.out_col_stride = out_col_stride;
        p.log_n = log_n; p.L = ps.L; p.R = ps.R;
        // tile width: 16 words for the register-radix strided kernels; the generic kernel takes as wide a run as a
        // 4096-element tile allows (a 1..3-bit top pass then moves 2..8 KiB contiguous runs)
        const bool reg_high = ps.L >= 4 && (ps.R == 8 || ps.R == 10);
        p.log_t = ps.L == 0 ? 0 : (ps.L < 4 ? ps.L : 4);
        // the generic kernel on a strided pass: 4096-element tiles, i.e. runs of 2^(12 - R) consecutive words (a 1..3-bit top pass
        // at 2^21 / 2^23 / 2^24 then streams 2..8 KiB runs; with 16-word runs it moved 64 elements per workgroup: 0.6 TB/s)
        if (ps.L != 0 && !reg_high && ps.R < 12) p.log_t = std::max<uint32_t>(p.log_t, std::min<uint32_t>(ps.L, 12 - ps.R));
        // keep the tile <= 64 KiB
        while (p.R + p.log_t > 14 && p.log_t > 0) p.log_t--;
        p.expand_bits = (!inverse && first) ? expand_bits : 0;
        p.first_layer = (!inverse && first) ? expand_bits + 1 : 1;
        p.twiddle = ps.L != 0;
        p.scale = (inverse && last) ? ninv.v : 0;
        p.zk_shift = (inverse && last && zk) ? 1 : 0;
        p.lazy = lazy; p.lazy_comp = lazy_comp;
        p.tile_tw = inverse ? c->tab.tile_rev : c->tab.tile_fwd;
        p.layer_tw = inverse ? c->tab.layer_rev : (lazy ? c->tab.layer_fwd_plain : c->tab.layer_fwd);
        p.tw_lo = inverse ? c->tab.tw_rev_lo : c->tab.tw_fwd_lo;
        p.tw_hi = inverse ? c->tab.tw_rev_hi : c->tab.tw_fwd_hi;
        p.sh_lo = c->tab.shift_lo; p.sh_hi = c->tab.shift_hi;
        if (p.zk_shift && log_n >= 4) {
            const Fp three = fp_encode(3);
            for (uint32_t k = 0; k < 16; k++)
                p.shift16[k] = (ninv * fp_pow(three, (uint64_t)(bitrev32(k) >> 28) << (log_n - 4))).v;
        }
        p.tiles_per_col = (uint32_t)(n >> (p.R + p.log_t));
        const size_t lds = ((size_t)4 << (p.R + p.log_t));
        dim3 grid(p.tiles_per_col, (unsigned)count);
        // algorithmic bytes in SURVEY.md §8d's sense (operands read once + written once, whatever the number of HBM passes):
        // charged to the first pass, the other passes of the same transform add time only
        const double alg_bytes = pi == 0 ? 4.0 * count * (double)(((size_t)1 << log_n) >> expand_bits) + 4.0 * count * (double)n : 0.0;
        const bool scale_here = p.scale != 0;
        // profiler record = "<Hal op>:<kernel>", so both the op totals and the per-kernel (per-pass) times can be read off
        const bool k_low = ps.L == 0 && ps.R == 12 && p.expand_bits <= 4;
        const bool k_h10 = !k_low && ps.L >= 4 && p.log_t == 4 && ps.R == 10 && !(scale_here && p.zk_shift);
        const bool k_h8 = !k_low && !k_h10 && ps.L >= 4 && p.log_t == 4 && ps.R == 8 && !(scale_here && p.zk_shift);
        const bool k_top = !k_low && !k_h10 && !k_h8 && ps.L >= 12 && ps.L + ps.R == log_n && ps.R >= 1 && ps.R <= 4 && p.twiddle &&
                           p.expand_bits == 0 && p.first_layer == 1 && !scale_here && !p.zk_shift;
        const std::string pname = std::string(name) + (k_low ? ":k_ntt_low12" : k_h10 ? ":k_ntt_high10" : k_h8 ? ":k_ntt_high8" : k_top ? ":k_ntt_top" : ":k_ntt_pass");
        ProfScope prof(c, pname.c_str(), alg_bytes);
        if (ps.L == 0 && ps.R == 12 && p.expand_bits <= 4) {
            if (inverse) k_ntt_low12<true><<<grid, 256, 0, c->stream>>>(p);
            else if (lazy) k_ntt_low12<false, true><<<grid, 256, 0, c->stream>>>(p);
            else k_ntt_low12<false><<<grid, 256, 0, c->stream>>>(p);
        } else if (ps.L >= 4 && p.log_t == 4 && ps.R == 10 && !(scale_here && p.zk_shift)) {
            if (inverse) k_ntt_high<10, true><<<grid, 1024, lds, c->stream>>>(p);
            else if (lazy) k_ntt_high<10, false, true><<<grid, 1024, lds, c->stream>>>(p);
            else k_ntt_high<10, false><<<grid, 1024, lds, c->stream>>>(p);
        } else if (ps.L >= 4 && p.log_t == 4 && ps.R == 8 && !(scale_here && p.zk_shift)) {
            if (inverse) k_ntt_high<8, true><<<grid, 256, lds, c->stream>>>(p);
            else if (lazy) k_ntt_high<8, false, true><<<grid, 256, lds, c->stream>>>(p);
            else k_ntt_high<8, false><<<grid, 256, lds, c->stream>>>(p);
        } else if (k_top) {
            const dim3 tg((unsigned)(((size_t)1 << ps.L) >> 10), (unsigned)count);          // 256 lanes x 4 low indices per workgroup
            switch (ps.R * 2 + (inverse ? 1 : 0)) {
            case 2: k_ntt_top<1, false><<<tg, 256, 0, c->stream>>>(p); break;
            case 3: k_ntt_top<1, true><<<tg, 256, 0, c->stream>>>(p); break;
            case 4: k_ntt_top<2, false><<<tg, 256, 0, c->stream>>>(p); break;
            case 5: k_ntt_top<2, true><<<tg, 256, 0, c->stream>>>(p); break;
            case 6: k_ntt_top<3, false><<<tg, 256, 0, c->stream>>>(p); break;
            case 7: k_ntt_top<3, true><<<tg, 256, 0, c->stream>>>(p); break;
            case 8: k_ntt_top<4, false><<<tg, 256, 0, c->stream>>>(p); break;
            default: k_ntt_top<4, true><<<tg, 256, 0, c->stream>>>(p); break;
            }
        } else if (inverse) k_ntt_pass<true><<<grid, NTT_THREADS, lds, c->stream>>>(p);
        else k_ntt_pass<false><<<grid, NTT_THREADS, lds, c->stream>>>(p);
        ZKH_TRY(last_launch_error(name));
    }
    return nullptr;
}

}  // namespace

// Dynamic-LDS limits are per (function, device): set them for the device of every context that is created, and fail the
// context creation if the device refuses (hal.hip calls this from zkh_ctx_create).
const char* zkh::ntt_device_init(zkh_ctx* c) {
    bind_thread(c);
    ZKH_HIP(hipFuncSetAttribute((const void*)k_ntt_high<10, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    ZKH_HIP(hipFuncSetAttribute((const void*)k_ntt_high<10, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    ZKH_HIP(hipFuncSetAttribute((const void*)k_ntt_high<10, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    return nullptr;
}

static const char* interpolate_impl(zkh_ctx* c, zkh_buf* io, size_t count, bool zk, const char* name) {
    ZKH_REQUIRE(count && io->len % count == 0, "%s: size %zu not a multiple of count %zu", name, io->len, count);
    const size_t n = io->len / count;
    const uint32_t log_n = log2_ceil(n);
    ZKH_REQUIRE(((size_t)1 << log_n) == n, "%s: column length %zu is not a power of two", name, n);
    return run_transform(c, true, io->ptr(), n, io->ptr(), n, log_n, count, 0, zk, name);
}
extern "C" const char* zkh_batch_interpolate_ntt(zkh_ctx* c, zkh_buf* io, size_t count) {
    return interpolate_impl(c, io, count, false, "batch_interpolate_ntt");
}
extern "C" const char* zkh_batch_interpolate_ntt_zk_shift(zkh_ctx* c, zkh_buf* io, size_t count) {
    return interpolate_impl(c, io, count, true, "batch_interpolate_ntt_zk_shift");
}
extern "C" const char* zkh_batch_interpolate_ntt_from(zkh_ctx* c, zkh_buf* out, const zkh_buf* in, size_t count, int zk_shift) {
    ZKH_REQUIRE(count && in->len % count == 0 && out->len == in->len, "batch_interpolate_ntt_from: shape mismatch");
    const size_t n = in->len / count;
    const uint32_t log_n = log2_ceil(n);
    ZKH_REQUIRE(((size_t)1 << log_n) == n, "batch_interpolate_ntt_from: column length %zu is not a power of two", n);
    return run_transform(c, true, in->ptr(), n, out->ptr(), n, log_n, count, 0, zk_shift != 0, "batch_interpolate_ntt_from");
}
extern "C" const char* zkh_batch_expand_into_evaluate_ntt(zkh_ctx* c, zkh_buf* out, const zkh_buf* in, size_t count,
                                                          size_t expand_bits) {
    ZKH_REQUIRE(count && out->len % count == 0 && in->len % count == 0, "batch_expand_into_evaluate_ntt: sizes not multiples of count");
    const size_t n_out = out->len / count, n_in = in->len / count;
    const uint32_t log_n = log2_ceil(n_out);
    ZKH_REQUIRE(((size_t)1 << log_n) == n_out && (n_in << expand_bits) == n_out,
                "batch_expand_into_evaluate_ntt: out column %zu != in column %zu << %zu", n_out, n_in, expand_bits);
    ZKH_REQUIRE(expand_bits <= log_n, "batch_expand_into_evaluate_ntt: expand_bits too large");
    if (expand_bits == log_n) {   // degenerate: pure replication
        return make_err("batch_expand_into_evaluate_ntt: expand_bits == log2(size) unsupported");
    }
    return run_transform(c, false, in->ptr(), n_in, out->ptr(), n_out, log_n, count, (uint32_t)expand_bits, false,
                         "batch_expand_into_evaluate_ntt");
}
extern "C" const char* zkh_zk_shift(zkh_ctx* c, zkh_buf* io, size_t count) {
    ZKH_REQUIRE(count && io->len % count == 0, "zk_shift: size not a multiple of count");
    const size_t n = io->len / count;
    const uint32_t log_n = log2_ceil(n);
    ZKH_REQUIRE(((size_t)1 << log_n) == n && log_n <= (uint32_t)MAX_LOG_N, "zk_shift: bad column length %zu", n);
    ProfScope prof(c, "zk_shift", 8.0 * io->len);
    k_zk_shift<<<(unsigned)((io->len + 255) / 256), 256, 0, c->stream>>>(io->ptr(), io->len, log_n, c->tab.shift_lo, c->tab.shift_hi);
    return last_launch_error("zk_shift");
}
extern "C" const char* zkh_batch_bit_reverse(zkh_ctx* c, zkh_buf* io, size_t count) {
    ZKH_REQUIRE(count && io->len % count == 0, "batch_bit_reverse: size not a multiple of count");
    const size_t n = io->len / count;
    const uint32_t log_n = log2_ceil(n);
    ZKH_REQUIRE(((size_t)1 << log_n) == n, "batch_bit_reverse: column length %zu is not a power of two", n);
    ProfScope prof(c, "batch_bit_reverse", 8.0 * io->len);
    if (log_n >= 2 * BR_H + 1 && count <= 65535) {
        dim3 grid(1u << (log_n - 2 * BR_H), (unsigned)count);
        k_bit_reverse_tiled<<<grid, 256, 0, c->stream>>>(io->ptr(), log_n, n);
    } else {
        k_bit_reverse_small<<<(unsigned)((io->len + 255) / 256), 256, 0, c->stream>>>(io->ptr(), log_n, count);
    }
    return last_launch_error("batch_bit_reverse");
}
