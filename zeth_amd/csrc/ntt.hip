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
//   HBM traffic: 8 B per element per pass (one read + one write), i.e. 16 B/elem for k <= 24.
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
    const uint32_t* tw_lo;   // w_{2^24}^lo
    const uint32_t* tw_hi;   // w_{2^24}^(4096 hi)
    const uint32_t* sh_lo;
    const uint32_t* sh_hi;
    uint32_t tiles_per_col;
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
    const uint32_t tw_shift = MAX_LOG_N - (p.L + p.R);   // w_{L+R}^e = w_24^(e << tw_shift)

    // ---- load (+ expand, + DIT pre-twiddle) ----
    for (uint32_t e = tid; e < elems; e += NTT_THREADS) {
        const uint32_t m = e >> p.log_t, t = e & (T - 1);
        const size_t gi = base + ((size_t)m << p.L) + t;
        uint32_t v = in[gi >> p.expand_bits];
        if (!INVERSE && p.twiddle) {
            const uint32_t r = __brev(m) >> (32 - p.R);
            const uint32_t ex = ((l0 + t) * r) << tw_shift;          // < 2^24
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
                tile[i1] = mul_mod(sub_mod(x, y), w);
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

struct Pass { uint32_t L, R; };
// Cut log_n index bits into HBM passes of <= 12 bits.  The lowest pass works on contiguous runs (cheapest), the
// strided passes above it are kept <= 10 bits so a 2^R x 16-word tile stays <= 64 KiB.
std::vector<Pass> plan_passes(uint32_t log_n) {
    std::vector<Pass> v;
    if (log_n <= 12) { v.push_back({0, log_n}); return v; }
    if (log_n <= 22) {
        uint32_t low = (log_n + 1) / 2;
        if (log_n - low > 10) low = log_n - 10;
        v.push_back({0, low});
        v.push_back({low, log_n - low});
        return v;
    }
    const uint32_t rem = log_n - 12, r1 = (rem + 1) / 2;
    v.push_back({0, 12});
    v.push_back({12, r1});
    v.push_back({12 + r1, rem - r1});
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
    for (size_t pi = 0; pi < npass; pi++) {
        // inverse: high bits first; forward: low bits first
        const Pass ps = inverse ? passes[npass - 1 - pi] : passes[pi];
        PassParams p{};
        const bool first = pi == 0, last = pi + 1 == npass;
        p.in = first ? in : out;
        p.in_col_stride = first ? in_col_stride : out_col_stride;
        p.out = out; p.out_col_stride = out_col_stride;
        p.log_n = log_n; p.L = ps.L; p.R = ps.R;
        p.log_t = ps.L == 0 ? 0 : (ps.L < 4 ? ps.L : 4);
        // keep the tile <= 64 KiB
        while (p.R + p.log_t > 14 && p.log_t > 0) p.log_t--;
        p.expand_bits = (!inverse && first) ? expand_bits : 0;
        p.first_layer = (!inverse && first) ? expand_bits + 1 : 1;
        p.twiddle = ps.L != 0;
        p.scale = (inverse && last) ? ninv.v : 0;
        p.zk_shift = (inverse && last && zk) ? 1 : 0;
        p.tile_tw = inverse ? c->tab.tile_rev : c->tab.tile_fwd;
        p.tw_lo = inverse ? c->tab.tw_rev_lo : c->tab.tw_fwd_lo;
        p.tw_hi = inverse ? c->tab.tw_rev_hi : c->tab.tw_fwd_hi;
        p.sh_lo = c->tab.shift_lo; p.sh_hi = c->tab.shift_hi;
        p.tiles_per_col = (uint32_t)(n >> (p.R + p.log_t));
        const size_t lds = ((size_t)4 << (p.R + p.log_t));
        dim3 grid(p.tiles_per_col, (unsigned)count);
        ProfScope prof(c, name, 8.0 * n * count);
        if (inverse) k_ntt_pass<true><<<grid, NTT_THREADS, lds, c->stream>>>(p);
        else k_ntt_pass<false><<<grid, NTT_THREADS, lds, c->stream>>>(p);
        ZKH_TRY(last_launch_error(name));
    }
    return nullptr;
}

}  // namespace

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
