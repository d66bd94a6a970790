// hal.hip — context, Buffer<T> plumbing, stream-ordered pool, profiling, and the small element-wise Hal ops.
// Stands in for risc0-zkp 3.0.2 src/hal/{mod.rs,cuda.rs} (un-vendored; /root/reference/Cargo.lock:5393),
// reached from /root/reference/crates/host/src/lib.rs:137.
#include <algorithm>

#include "common.h"

#include <sys/random.h>
#include "poseidon2.h"
#include "../../include/zkh_poseidon2_consts.h"

using namespace zkh;

extern "C" void zkh_free_error(const char* e) { free((void*)e); }
// The version string says where the Poseidon2 tables come from: "placeholder" (filler: digests cannot
// match upstream's), "derived" (produced by the published parameter-generation procedure, tools/gen_poseidon2_consts.py;
// not yet compared with upstream's consts.rs word for word) or "upstream".
#if ZKH_P2_CONSTS_ARE_PLACEHOLDER
extern "C" const char* zkh_version(void) { return "zkhal-mi355x 0.2.0 (gfx950; poseidon2_consts=placeholder)"; }
#elif ZKH_P2_CONSTS_ARE_DERIVED
extern "C" const char* zkh_version(void) { return "zkhal-mi355x 0.2.0 (gfx950; poseidon2_consts=derived)"; }
#else
extern "C" const char* zkh_version(void) { return "zkhal-mi355x 0.2.0 (gfx950; poseidon2_consts=upstream)"; }
#endif

namespace zkh {

const char* pool_alloc(zkh_ctx* c, size_t bytes, void** out) {
    bytes = (bytes + 255) & ~(size_t)255;
    if (bytes == 0) bytes = 256;
    auto it = c->pool.find(bytes);
    if (it != c->pool.end()) {
        *out = it->second;
        c->pool.erase(it);
        c->pool_bytes -= bytes;
    } else {
        hipError_t e = hipMalloc(out, bytes);
        if (e != hipSuccess) {
            // release the cached blocks and retry once
            (void)hipStreamSynchronize(c->stream);
            for (auto& kv : c->pool) (void)hipFree(kv.second);
            c->pool.clear(); c->pool_bytes = 0;
            e = hipMalloc(out, bytes);
            if (e != hipSuccess) return make_err("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        }
    }
    c->live_bytes += bytes;
    if (c->live_bytes > c->peak_bytes) c->peak_bytes = c->live_bytes;
    return nullptr;
}
void pool_free(zkh_ctx* c, void* p, size_t bytes) {
    bytes = (bytes + 255) & ~(size_t)255;
    if (bytes == 0) bytes = 256;
    c->live_bytes -= bytes;
    // all work is in-order on one stream, so a block may be handed out again without a sync
    c->pool.emplace(bytes, p);
    c->pool_bytes += bytes;
}
const char* new_buf(zkh_ctx* c, size_t n_words, bool zero, zkh_buf** out) {
    bind_thread(c);
    void* p = nullptr;
    ZKH_TRY(pool_alloc(c, n_words * 4, &p));
    if (zero && n_words) {
        hipError_t e = hipMemsetAsync(p, 0, n_words * 4, c->stream);
        if (e != hipSuccess) return make_err("hipMemsetAsync: %s", hipGetErrorString(e));
    }
    zkh_alloc_t* a = new zkh_alloc_t(p, n_words * 4, 1, true, c);
    *out = new zkh_buf(a, 0, n_words, 1);
    return nullptr;
}
const char* ensure_pinned(zkh_ctx* c, size_t words) {
    if (c->pinned_words >= words) return nullptr;
    if (c->pinned) (void)hipHostFree(c->pinned);
    size_t w = words < 65536 ? 65536 : words;
    ZKH_HIP(hipHostMalloc((void**)&c->pinned, w * 4, hipHostMallocDefault));
    c->pinned_words = w;
    return nullptr;
}
static hipEvent_t get_event(zkh_ctx* c) {
    if (!c->event_pool.empty()) { hipEvent_t e = c->event_pool.back(); c->event_pool.pop_back(); return e; }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}
void prof_begin(zkh_ctx* c, const char* name, double bytes) {
    ProfPending p{name, get_event(c), get_event(c), bytes};
    (void)hipEventRecord(p.a, c->stream);
    c->pending.push_back(p);
}
void prof_end(zkh_ctx* c) { (void)hipEventRecord(c->pending.back().b, c->stream); }
static void prof_flush(zkh_ctx* c) {
    if (c->pending.empty()) return;
    (void)hipStreamSynchronize(c->stream);
    for (auto& p : c->pending) {
        float ms = 0;
        (void)hipEventElapsedTime(&ms, p.a, p.b);
        auto& g = c->agg[p.name];
        g.calls++; g.ms += ms; g.bytes += p.bytes;
        c->event_pool.push_back(p.a); c->event_pool.push_back(p.b);
    }
    c->pending.clear();
}

static const char* upload(uint32_t** dst, const std::vector<uint32_t>& v) {
    ZKH_HIP(hipMalloc((void**)dst, v.size() * 4));
    ZKH_HIP(hipMemcpy(*dst, v.data(), v.size() * 4, hipMemcpyHostToDevice));
    return nullptr;
}
const char* resolve_noise_key(const uint32_t* key, NoiseKey* out) {
    bool given = false;
    if (key) for (int i = 0; i < 8; i++) given = given || key[i] != 0;
    if (given) { memcpy(out->k, key, sizeof out->k); return nullptr; }
    bool ok = false;
    for (int tries = 0; tries < 4 && !ok; tries++) {
        ok = getrandom(out->k, sizeof out->k, 0) == (ssize_t)sizeof out->k;
        bool nonzero = false;
        for (int i = 0; i < 8; i++) nonzero = nonzero || out->k[i] != 0;
        ok = ok && nonzero;
    }
    ZKH_REQUIRE(ok, "OS randomness (getrandom) is unavailable: refusing to seal with a predictable blinding key");
    return nullptr;
}
extern "C" uint32_t zkh_noise_cell_host(const uint32_t noise_key[8], uint32_t group, uint32_t column, uint32_t row) {
    NoiseKey k;
    memcpy(k.k, noise_key, sizeof k.k);
    return noise_cell(k, group, column, row);
}
extern "C" void zkh_chacha_block_host(const uint32_t key[8], const uint32_t counter_nonce[4], int double_rounds, uint32_t out[16]) {
    chacha_block(key, counter_nonce[0], counter_nonce[1], counter_nonce[2], counter_nonce[3], double_rounds, out, 16);
}
static std::vector<uint32_t> powers(Fp base, size_t n) {
    std::vector<uint32_t> v(n);
    Fp cur = Fp::one();
    for (size_t i = 0; i < n; i++) { v[i] = cur.v; cur = cur * base; }
    return v;
}

}  // namespace zkh

extern "C" const char* zkh_poseidon2_set_constants(zkh_ctx* c, const uint32_t* rc, const uint32_t* diag) {
    bind_thread(c);
    std::vector<uint32_t> r(24 * 29), d(ZKH_P2_PTAB);
    for (int i = 0; i < 24 * 29; i++) r[i] = fp_encode(rc[i]).v - P;   // stored as rc - P: see poseidon2.h sbox7_rc
    poseidon2_partial_table(d.data(), rc, diag);
    memcpy(c->h_rc, r.data(), sizeof c->h_rc);
    memcpy(c->h_diag, d.data(), sizeof c->h_diag);
    ZKH_HIP(hipStreamSynchronize(c->stream));
    ZKH_HIP(hipMemcpy(c->tab.rc, r.data(), r.size() * 4, hipMemcpyHostToDevice));
    ZKH_HIP(hipMemcpy(c->tab.diag, d.data(), d.size() * 4, hipMemcpyHostToDevice));
    return nullptr;
}

static const char* ctx_init_tables(zkh_ctx* c);
extern "C" const char* zkh_ctx_create(int device, const char* suite, zkh_ctx** out) {
    ZKH_REQUIRE(suite && strcmp(suite, "poseidon2") == 0, "unsupported hash suite '%s' (only poseidon2)", suite ? suite : "(null)");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    ZKH_REQUIRE(e == hipSuccess && ndev > 0, "no HIP device available (%s): libzkhal_mi355x has no CPU fallback",
                e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    ZKH_REQUIRE(device >= 0 && device < ndev, "device ordinal %d out of range (%d devices)", device, ndev);
    ZKH_HIP(hipSetDevice(device));
    zkh_ctx* c = new zkh_ctx();
    c->device = device;
    if (hipError_t se = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking); se != hipSuccess) {
        delete c;
        return make_err("hipStreamCreateWithFlags: %s", hipGetErrorString(se));
    }
    if (const char* err = ctx_init_tables(c)) {     // a half-built context (stream, tables, pool) is not leaked
        zkh_ctx_destroy(c);
        return err;
    }
    *out = c;
    return nullptr;
}
static const char* ctx_init_tables(zkh_ctx* c) {
    Fp g = fp_encode(137);
    for (int k = 0; k <= 27; k++) {
        Fp w = fp_pow(g, 1ull << (27 - k));
        c->rou_fwd[k] = w.v;
        c->rou_rev[k] = fp_inv(w).v;
    }
    Fp wf = Fp::raw(c->rou_fwd[MAX_LOG_N]), wr = Fp::raw(c->rou_rev[MAX_LOG_N]);
    ZKH_TRY(upload(&c->tab.tw_fwd_lo, powers(wf, TW_SIZE)));
    ZKH_TRY(upload(&c->tab.tw_fwd_hi, powers(fp_pow(wf, TW_SIZE), TW_HI_SIZE)));
    ZKH_TRY(upload(&c->tab.tw_rev_lo, powers(wr, TW_SIZE)));
    ZKH_TRY(upload(&c->tab.tw_rev_hi, powers(fp_pow(wr, TW_SIZE), TW_HI_SIZE)));
    ZKH_TRY(upload(&c->tab.tile_fwd, powers(Fp::raw(c->rou_fwd[LDS_TW_LOG]), 1 << (LDS_TW_LOG - 1))));
    ZKH_TRY(upload(&c->tab.tile_rev, powers(Fp::raw(c->rou_rev[LDS_TW_LOG]), 1 << (LDS_TW_LOG - 1))));
    {
        std::vector<uint32_t> lf(1 << LDS_TW_LOG, R1), lr(1 << LDS_TW_LOG, R1);
        for (int j = 1; j <= LDS_TW_LOG; j++) {
            auto pf = powers(Fp::raw(c->rou_fwd[j]), (size_t)1 << (j - 1)), pr = powers(Fp::raw(c->rou_rev[j]), (size_t)1 << (j - 1));
            for (size_t e = 0; e < pf.size(); e++) { lf[((size_t)1 << (j - 1)) + e] = pf[e]; lr[((size_t)1 << (j - 1)) + e] = pr[e]; }
        }
        ZKH_TRY(upload(&c->tab.layer_fwd, lf));
        ZKH_TRY(upload(&c->tab.layer_rev, lr));
        for (auto& w : lf) w = mont_reduce((uint64_t)w);          // word / R: the plain residue
        ZKH_TRY(upload(&c->tab.layer_fwd_plain, lf));
    }
    Fp three = fp_encode(3);
    ZKH_TRY(upload(&c->tab.shift_lo, powers(three, TW_SIZE)));
    ZKH_TRY(upload(&c->tab.shift_hi, powers(fp_pow(three, TW_SIZE), TW_HI_SIZE)));
    ZKH_TRY(upload(&c->tab.rc, std::vector<uint32_t>(24 * 29)));
    ZKH_TRY(upload(&c->tab.diag, std::vector<uint32_t>(ZKH_P2_PTAB)));
    ZKH_TRY(zkh_poseidon2_set_constants(c, ZKH_P2_ROUND_CONSTANTS, ZKH_P2_M_INT_DIAG));
    ZKH_TRY(ntt_device_init(c));
    ZKH_HIP(hipMalloc((void**)&c->d_fail, 8));
    ZKH_HIP(hipMemsetAsync(c->d_fail, 0, 8, c->stream));
    ZKH_HIP(hipHostMalloc((void**)&c->h_fail, 8, hipHostMallocDefault));
    c->h_fail[0] = c->h_fail[1] = 0;
    return nullptr;
}
extern "C" const char* zkh_ctx_trim(zkh_ctx* c) {
    bind_thread(c);
    ZKH_HIP(hipStreamSynchronize(c->stream));           // cached blocks may still be read by queued work
    for (auto& kv : c->pool) (void)hipFree(kv.second);
    c->pool.clear(); c->pool_bytes = 0;
    return nullptr;
}
extern "C" void zkh_ctx_memory(const zkh_ctx* c, size_t* live, size_t* cached, size_t* peak) {
    if (live) *live = c->live_bytes;
    if (cached) *cached = c->pool_bytes;
    if (peak) *peak = c->peak_bytes;
}
extern "C" void zkh_ctx_destroy(zkh_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (auto& kv : c->pool) (void)hipFree(kv.second);
    uint32_t* t[] = {c->tab.rc, c->tab.diag, c->tab.tw_fwd_lo, c->tab.tw_fwd_hi, c->tab.tw_rev_lo, c->tab.tw_rev_hi,
                     c->tab.tile_fwd, c->tab.tile_rev, c->tab.shift_lo, c->tab.shift_hi, c->tab.layer_fwd, c->tab.layer_rev, c->tab.layer_fwd_plain};
    for (auto p : t) (void)hipFree(p);
    for (auto& p : c->pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    for (auto ev : c->event_pool) (void)hipEventDestroy(ev);
    if (c->pinned) (void)hipHostFree(c->pinned);
    if (c->d_fail) (void)hipFree(c->d_fail);
    if (c->h_fail) (void)hipHostFree(c->h_fail);
    for (auto& kv : c->host_blocks) (void)hipHostFree(kv.first);
    (void)hipStreamDestroy(c->stream);
    delete c;
}
// the sticky device-side failure word (common.h): fetched with the sync the caller asked for anyway
static const char* check_device_fail(zkh_ctx* c) {
    if (!c->fail_armed) return nullptr;
    c->fail_armed = false;
    const uint32_t code = c->h_fail[0], detail = c->h_fail[1];
    if (!code) return nullptr;
    (void)hipMemsetAsync(c->d_fail, 0, 8, c->stream);
    switch (code) {
    case 1: return make_err("combos_prepare_regs: register %u has size 0 or more than `cycles` words (found on the device; the call's combos are unchanged)", detail);
    case 2: return make_err("combos_prepare_regs: register %u names a combo beyond combo_count (found on the device; the call's combos are unchanged)", detail);
    case 3: return make_err("combos_prepare_regs: the register sizes add up to %u U coefficients (+ %d of the check polynomial), more than coeff_u holds (found on the device; the call's combos are unchanged)", detail, ZKH_CHECK_SIZE);
    default: return make_err("a kernel reported failure code %u (%u)", code, detail);
    }
}
// every host-visible synchronisation of the library's stream goes through here: the sticky word is fetched with it, so a
// device-side failure (zkh_combos_prepare_regs) surfaces at the FIRST sync after it, whichever call that is (zkh_sync, zkh_read,
// a large zkh_write / zkh_copy_from, a full staging ring).  Callers that synchronise zkh_ctx_stream() themselves must call zkh_sync.
static const char* sync_checked(zkh_ctx* c) {
    if (c->fail_armed) ZKH_HIP(hipMemcpyAsync(c->h_fail, c->d_fail, 8, hipMemcpyDeviceToHost, c->stream));
    ZKH_HIP(hipStreamSynchronize(c->stream));
    c->stage_used = 0;
    return check_device_fail(c);
}
extern "C" const char* zkh_sync(zkh_ctx* c) {
    bind_thread(c);
    return sync_checked(c);
}
extern "C" void* zkh_ctx_stream(zkh_ctx* c) { return (void*)c->stream; }

// ---- buffers ----
extern "C" const char* zkh_alloc(zkh_ctx* c, const char*, size_t n, int zero, zkh_buf** out) {
    return new_buf(c, n, zero != 0, out);
}
// Small host->device uploads (indices, challenges, tables: a few KB) go through a ring of pinned staging slots so that
// the borrowed host pointer is consumed by a memcpy and NO stream synchronisation is needed; a slot is only reused after
// the stream has been synchronised at least once since it was filled (every zkh_read does that).
namespace zkh {
constexpr size_t STAGE_SLOT_WORDS = 16384, STAGE_SLOTS = 64;      // 64 x 64 KiB pinned
const char* h2d(zkh_ctx* c, uint32_t* dst, const uint32_t* host, size_t n) {
    if (!n) return nullptr;
    bind_thread(c);
    if (n <= STAGE_SLOT_WORDS) {
        ZKH_TRY(ensure_pinned(c, STAGE_SLOT_WORDS * STAGE_SLOTS));
        if (c->stage_used == STAGE_SLOTS) ZKH_TRY(sync_checked(c));
        uint32_t* slot = c->pinned + (size_t)c->stage_next * STAGE_SLOT_WORDS;
        memcpy(slot, host, n * 4);
        ZKH_HIP(hipMemcpyAsync(dst, slot, n * 4, hipMemcpyHostToDevice, c->stream));
        c->stage_next = (c->stage_next + 1) % STAGE_SLOTS;
        c->stage_used++;
        return nullptr;
    }
    ZKH_HIP(hipMemcpyAsync(dst, host, n * 4, hipMemcpyHostToDevice, c->stream));
    return sync_checked(c);                         // the host pointer is only borrowed for the call
}
}  // namespace zkh
extern "C" const char* zkh_copy_from(zkh_ctx* c, const char*, const uint32_t* host, size_t n, zkh_buf** out) {
    ZKH_TRY(new_buf(c, n, false, out));
    return h2d(c, (*out)->ptr(), host, n);
}
extern "C" const char* zkh_wrap(zkh_ctx* c, void* dptr, size_t n, zkh_buf** out) {
    zkh_alloc_t* a = new zkh_alloc_t(dptr, n * 4, 1, false, c);
    *out = new zkh_buf(a, 0, n, 1);
    return nullptr;
}
extern "C" const char* zkh_slice(zkh_buf* b, size_t off, size_t n, zkh_buf** out) {
    ZKH_REQUIRE(n <= b->len && off <= b->len - n, "slice [%zu, +%zu) out of range (size %zu)", off, n, b->len);   // no wrap-around
    b->a->refs.fetch_add(1, std::memory_order_relaxed);
    *out = new zkh_buf(b->a, b->off + off, n, 1);
    return nullptr;
}
extern "C" void zkh_retain(zkh_buf* b) { b->refs.fetch_add(1, std::memory_order_relaxed); }
extern "C" void zkh_release(zkh_buf* b) {
    if (!b || b->refs.fetch_sub(1, std::memory_order_acq_rel) > 1) return;
    zkh_alloc_t* a = b->a;
    if (a->refs.fetch_sub(1, std::memory_order_acq_rel) == 1) {
        if (a->owned) pool_free(a->ctx, a->ptr, a->bytes);
        delete a;
    }
    delete b;
}
extern "C" size_t zkh_size(const zkh_buf* b) { return b->len; }
extern "C" void* zkh_device_ptr(const zkh_buf* b) { return (void*)b->ptr(); }
extern "C" const char* zkh_read(zkh_ctx* c, const zkh_buf* b, uint32_t* host, size_t off, size_t n) {
    bind_thread(c);
    ZKH_REQUIRE(n <= b->len && off <= b->len - n, "read [%zu, +%zu) out of range (size %zu)", off, n, b->len);   // no wrap-around
    if (n) ZKH_HIP(hipMemcpyAsync(host, b->ptr() + off, n * 4, hipMemcpyDeviceToHost, c->stream));
    return sync_checked(c);
}
extern "C" const char* zkh_write(zkh_ctx* c, zkh_buf* b, const uint32_t* host, size_t off, size_t n) {
    bind_thread(c);
    ZKH_REQUIRE(n <= b->len && off <= b->len - n, "write [%zu, +%zu) out of range (size %zu)", off, n, b->len);   // no wrap-around
    return h2d(c, b->ptr() + off, host, n);
}

// ---- witness ingress: pinned host memory + asynchronous upload ----
// A host that produces the witness on the CPU (upstream: preflight + witgen) writes it straight
// into pinned memory and enqueues the upload; the copy is a DMA on this context's stream, so it overlaps the kernels of the
// OTHER contexts sealing on the same GPU (bench.py keeps three seals in flight) and returns without a host sync.
extern "C" const char* zkh_host_alloc(zkh_ctx* c, size_t n_words, uint32_t** host) {
    ZKH_REQUIRE(c && host && n_words, "host_alloc: bad argument");
    bind_thread(c);
    void* p = nullptr;
    ZKH_HIP(hipHostMalloc(&p, n_words * 4, hipHostMallocDefault));
    c->host_blocks[p] = n_words * 4;
    *host = (uint32_t*)p;
    return nullptr;
}
extern "C" void zkh_host_free(zkh_ctx* c, uint32_t* host) {
    if (!c || !host) return;
    auto it = c->host_blocks.find((void*)host);
    if (it == c->host_blocks.end()) return;
    bind_thread(c);
    (void)hipStreamSynchronize(c->stream);           // an upload from this block may still be in flight
    (void)hipHostFree(it->first);
    c->host_blocks.erase(it);
}
extern "C" const char* zkh_write_async(zkh_ctx* c, zkh_buf* b, const uint32_t* pinned_host, size_t off, size_t n) {
    bind_thread(c);
    ZKH_REQUIRE(n <= b->len && off <= b->len - n, "write_async [%zu, +%zu) out of range (size %zu)", off, n, b->len);   // no wrap-around
    // the source must lie inside a block from zkh_host_alloc: only then is the copy a true asynchronous DMA, and only then
    // does the library know the memory stays mapped until zkh_host_free (which drains the stream first)
    auto it = c->host_blocks.upper_bound((void*)pinned_host);
    bool inside = false;
    if (it != c->host_blocks.begin()) {
        --it;
        const char* base = (const char*)it->first;
        inside = (const char*)pinned_host >= base && (const char*)(pinned_host + n) <= base + it->second;
    }
    ZKH_REQUIRE(inside, "write_async: the source is not inside a zkh_host_alloc block of this context");
    if (n) ZKH_HIP(hipMemcpyAsync(b->ptr() + off, pinned_host, n * 4, hipMemcpyHostToDevice, c->stream));
    return nullptr;
}

// ---- profiling ----
extern "C" const char* zkh_prof_enable(zkh_ctx* c, int on) { prof_flush(c); c->prof = on != 0; return nullptr; }
extern "C" const char* zkh_prof_reset(zkh_ctx* c) { prof_flush(c); c->agg.clear(); return nullptr; }
extern "C" const char* zkh_prof_get(zkh_ctx* c, zkh_prof_rec* recs, size_t cap, size_t* n) {
    prof_flush(c);
    size_t i = 0;
    for (auto& kv : c->agg) {
        if (i >= cap) break;
        memset(&recs[i], 0, sizeof recs[i]);
        strncpy(recs[i].name, kv.first.c_str(), sizeof(recs[i].name) - 1);
        recs[i].calls = kv.second.calls; recs[i].total_ms = kv.second.ms; recs[i].alg_bytes = kv.second.bytes;
        i++;
    }
    *n = i;
    return nullptr;
}

// =====================================================================================================
// element-wise / gather ops.  All HBM-bound: one pass, lanes on consecutive words.
// =====================================================================================================
namespace {

constexpr int TB = 256;
inline unsigned blocks_for(size_t n, int per_block = TB) { return (unsigned)((n + per_block - 1) / per_block); }

__global__ void k_add(uint32_t* out, const uint32_t* a, const uint32_t* b, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = add_mod(a[i], b[i]);
}
__global__ void k_copy(uint32_t* out, const uint32_t* in, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i];
}
__global__ void k_zeroize(uint32_t* io, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && io[i] == INVALID) io[i] = 0;
}
// in: k ExtElems (AoS) per idx at in[(j*count + idx)*4 ..]; out: 4 planes of count
__global__ void k_sum_ext(uint32_t* out, const uint4* in, size_t count, size_t k) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= count) return;
    uint32_t s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    for (size_t j = 0; j < k; j++) {
        uint4 v = in[j * count + idx];
        s0 = add_mod(s0, v.x); s1 = add_mod(s1, v.y); s2 = add_mod(s2, v.z); s3 = add_mod(s3, v.w);
    }
    out[idx] = s0; out[count + idx] = s1; out[2 * count + idx] = s2; out[3 * count + idx] = s3;
}
// fri_fold: in = 4 planes x (16*count), out = 4 planes x count; slice order bit-reversed (cpu.rs fri_fold)
__global__ void k_fri_fold(uint32_t* out, const uint32_t* in, size_t count, Fp4 mix) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= count) return;
    Fp4 tot = Fp4::zero(), cur = Fp4::one();
#pragma unroll
    for (unsigned i = 0; i < 16; i++) {
        unsigned r = __brev(i) >> 28;
        Fp4 f;
#pragma unroll
        for (int p = 0; p < 4; p++) f.c[p] = Fp::raw(in[(size_t)p * count * 16 + r * count + idx]);
        tot = tot + cur * f;
        cur = cur * mix;
    }
#pragma unroll
    for (int p = 0; p < 4; p++) out[p * count + idx] = tot.c[p].v;
}
__global__ void k_gather(uint32_t* dst, const uint32_t* src, size_t idx, size_t size, size_t stride) {
    size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g < size) dst[g] = src[g * stride + idx];
}
__global__ void k_scatter(uint32_t* into, const uint32_t* index, const uint32_t* values, size_t n) {
    size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) into[index[j]] = values[j];
}
__global__ void k_combos_prepare(uint4* combos, const uint32_t* pos, const uint4* vals, size_t n) {
    size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    uint4 c = combos[pos[k]], v = vals[k];
    c.x = sub_mod(c.x, v.x); c.y = sub_mod(c.y, v.y); c.z = sub_mod(c.z, v.z); c.w = sub_mod(c.w, v.w);
    combos[pos[k]] = c;
}
// Hal::combos_prepare with upstream's own argument list, entirely on the device (no host round trip, like CudaHal):
//   cur = 1; for each register r: combos[cycles * combo_id[r] + i] -= cur * coeff_u[pos + i] (i < size[r]); cur *= mix; pos += size[r]
//   then the CHECK_SIZE check columns: combos[cycles * combo_count] -= cur * coeff_u[pos]; pos += 1; cur *= mix.
// One lane per TARGET position (combo c, offset i) walks the register list and sums its own contributions (several
// registers hit the same position, so the walk is per target, not per register); exact field arithmetic, so the order of
// the subtractions does not matter.  A few dozen lanes x regs_count Fp4 products: ~0.1 ms for a thousand registers.
// Lane (c, i) owns the target offsets i, i + 32, i + 64, ... of combo c (registers of any size: `max_size` is the largest one, from the
// host's validation of the register list, which also bounds every coeff_u index this kernel forms).
constexpr uint32_t PREP_LANES_PER_COMBO = 32;
// The register list is the CALLER'S device data.  One wave walks it before the main kernel: every size in 1 .. cycles, every combo id
// below combo_count, the sizes (+ CHECK_SIZE) within coeff_u — meta[0] = the largest register (0: the list is inconsistent and the
// main kernel does nothing), and on an inconsistency the context's sticky failure word is raised (reported by the next zkh_sync /
// zkh_read: no host round trip here, one host-visible sync per commit stays one).
__global__ void k_combos_regs_check(uint32_t* meta, uint32_t* fail, const uint32_t* reg_sizes, const uint32_t* reg_combo_ids, uint32_t regs_count,
                                    uint32_t combo_count, size_t cycles, size_t coeff_ext) {
    __shared__ uint32_t s_max, s_bad, s_bad_at;
    __shared__ unsigned long long s_total;
    if (threadIdx.x == 0) { s_max = 1; s_bad = 0; s_bad_at = 0xffffffffu; s_total = 0; }
    __syncthreads();
    unsigned long long part = 0;
    for (uint32_t r = threadIdx.x; r < regs_count; r += blockDim.x) {
        const uint32_t sz = reg_sizes[r], id = reg_combo_ids[r];
        const uint32_t bad = (sz < 1 || sz > cycles) ? 1u : (id >= combo_count ? 2u : 0u);
        if (bad && atomicMin(&s_bad_at, r) > r) atomicExch(&s_bad, bad);       // (the code of SOME bad register; the index is the lowest)
        atomicMax(&s_max, sz);
        part += sz;
    }
    atomicAdd(&s_total, part);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t code = 0, detail = 0;
        if (s_bad_at != 0xffffffffu) {
            const uint32_t sz = reg_sizes[s_bad_at];
            code = (sz < 1 || sz > cycles) ? 1u : 2u; detail = s_bad_at;
        } else if (cycles < (size_t)ZKH_CHECK_SIZE || s_total + ZKH_CHECK_SIZE > coeff_ext) {
            code = 3; detail = (uint32_t)s_total;
        }
        meta[0] = code ? 0u : s_max;
        if (code && atomicCAS(&fail[0], 0u, code) == 0u) fail[1] = detail;
    }
}
__global__ void k_combos_prepare_regs(uint4* combos, size_t combos_ext, const uint4* coeff_u, uint32_t combo_count, size_t cycles,
                                      uint32_t regs_count, const uint32_t* reg_sizes, const uint32_t* reg_combo_ids, Fp4 mix, const uint32_t* meta) {
    const uint32_t max_size = meta[0];
    if (max_size == 0) return;                               // the check kernel refused the list: nothing is read, nothing is written
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t c = tid / PREP_LANES_PER_COMBO, i0 = tid % PREP_LANES_PER_COMBO;
    if (c > combo_count || (c == combo_count && i0 != 0)) return;
    auto ld = [&](size_t k) { const uint4 v = coeff_u[k]; return Fp4(Fp::raw(v.x), Fp::raw(v.y), Fp::raw(v.z), Fp::raw(v.w)); };
    for (uint32_t i = i0; i < (c == combo_count ? 1u : max_size); i += PREP_LANES_PER_COMBO) {
        Fp4 cur = Fp4::one(), acc = Fp4::zero();
        size_t pos = 0;
        bool any = false;
        for (uint32_t r = 0; r < regs_count; r++) {
            const uint32_t sz = reg_sizes[r];
            if (reg_combo_ids[r] == c && i < sz) { acc = acc + cur * ld(pos + i); any = true; }
            cur = cur * mix;
            pos += sz;
        }
        if (c == combo_count) {
            for (int k = 0; k < ZKH_CHECK_SIZE; k++) { acc = acc + cur * ld(pos + k); cur = cur * mix; }
            any = true;
        }
        const size_t at = cycles * c + i;
        if (!any || at >= combos_ext) continue;
        uint4 v = combos[at];
        v.x = sub_mod(v.x, acc.c[0].v); v.y = sub_mod(v.y, acc.c[1].v); v.z = sub_mod(v.z, acc.c[2].v); v.w = sub_mod(v.w, acc.c[3].v);
        combos[at] = v;
    }
}
// Merkle opening for many query indices: block q handles idx[q]; first the column words then the sibling path.
__global__ void k_merkle_open(uint32_t* out, const uint32_t* matrix, const uint32_t* nodes, const uint32_t* idxs,
                              size_t rows, size_t cols, size_t top_size, size_t words_per_query) {
    const uint32_t q = blockIdx.x;
    const size_t idx = idxs[q];
    uint32_t* o = out + (size_t)q * words_per_query;
    for (size_t c = threadIdx.x; c < cols; c += blockDim.x) o[c] = matrix[c * rows + idx];
    o += cols;
    // path: level t (t = 0..) sibling of (idx + rows) >> t, while node index >= 2*top_size
    size_t node = idx + rows;
    unsigned levels = 0;
    for (size_t j = node; j >= 2 * top_size; j >>= 1) levels++;
    for (unsigned w = threadIdx.x; w < levels * 8; w += blockDim.x) {
        unsigned lvl = w >> 3;
        size_t sib = (node >> lvl) ^ 1;
        o[w] = nodes[sib * 8 + (w & 7)];
    }
}

}  // namespace

extern "C" const char* zkh_eltwise_add_elem(zkh_ctx* c, zkh_buf* out, const zkh_buf* a, const zkh_buf* b) {
    ZKH_REQUIRE(out->len == a->len && a->len == b->len, "eltwise_add_elem: size mismatch");
    if (!out->len) return nullptr;
    ProfScope ps(c, "eltwise_add_elem", 12.0 * out->len);
    k_add<<<blocks_for(out->len), TB, 0, c->stream>>>(out->ptr(), a->ptr(), b->ptr(), out->len);
    return last_launch_error("eltwise_add_elem");
}
extern "C" const char* zkh_eltwise_copy_elem(zkh_ctx* c, zkh_buf* out, const zkh_buf* in) {
    ZKH_REQUIRE(out->len == in->len, "eltwise_copy_elem: size mismatch");
    if (!out->len) return nullptr;
    ProfScope ps(c, "eltwise_copy_elem", 8.0 * out->len);
    k_copy<<<blocks_for(out->len), TB, 0, c->stream>>>(out->ptr(), in->ptr(), out->len);
    return last_launch_error("eltwise_copy_elem");
}
extern "C" const char* zkh_eltwise_zeroize_elem(zkh_ctx* c, zkh_buf* io) {
    if (!io->len) return nullptr;
    ProfScope ps(c, "eltwise_zeroize_elem", 8.0 * io->len);
    k_zeroize<<<blocks_for(io->len), TB, 0, c->stream>>>(io->ptr(), io->len);
    return last_launch_error("eltwise_zeroize_elem");
}
extern "C" const char* zkh_eltwise_sum_extelem(zkh_ctx* c, zkh_buf* out, const zkh_buf* in) {
    ZKH_REQUIRE(out->len % 4 == 0 && in->len % 4 == 0, "eltwise_sum_extelem: sizes must be multiples of 4");
    size_t count = out->len / 4;
    ZKH_REQUIRE(count && (in->len / 4) % count == 0, "eltwise_sum_extelem: input not a multiple of output");
    size_t k = in->len / 4 / count;
    ProfScope ps(c, "eltwise_sum_extelem", 4.0 * (in->len + out->len));
    k_sum_ext<<<blocks_for(count), TB, 0, c->stream>>>(out->ptr(), (const uint4*)in->ptr(), count, k);
    return last_launch_error("eltwise_sum_extelem");
}
extern "C" const char* zkh_fri_fold(zkh_ctx* c, zkh_buf* out, const zkh_buf* in, const uint32_t mix[4]) {
    ZKH_REQUIRE(out->len % 4 == 0 && in->len == out->len * 16, "fri_fold: input must be 16x output");
    size_t count = out->len / 4;
    Fp4 m(Fp::raw(mix[0]), Fp::raw(mix[1]), Fp::raw(mix[2]), Fp::raw(mix[3]));
    ProfScope ps(c, "fri_fold", 4.0 * (in->len + out->len));
    k_fri_fold<<<blocks_for(count), TB, 0, c->stream>>>(out->ptr(), in->ptr(), count, m);
    return last_launch_error("fri_fold");
}
extern "C" const char* zkh_gather_sample(zkh_ctx* c, zkh_buf* dst, const zkh_buf* src, size_t idx, size_t size,
                                         size_t stride) {
    ZKH_REQUIRE(dst->len >= size && (size == 0 || (size - 1) * stride + idx < src->len), "gather_sample: out of range");
    if (!size) return nullptr;
    ProfScope ps(c, "gather_sample", 8.0 * size);
    k_gather<<<blocks_for(size), TB, 0, c->stream>>>(dst->ptr(), src->ptr(), idx, size, stride);
    return last_launch_error("gather_sample");
}
extern "C" const char* zkh_scatter(zkh_ctx* c, zkh_buf* into, const uint32_t* index, const uint32_t* offsets,
                                   const uint32_t* values, size_t n_idx, size_t n_val) {
    // cpu.rs scatter walks offsets[i]..offsets[i+1] per cycle; the union of those ranges is [offsets[0], offsets[n_idx])
    (void)n_idx;
    if (!n_val) return nullptr;
    for (size_t j = 0; j < n_val; j++) ZKH_REQUIRE(index[j] < into->len, "scatter: index %u out of range", index[j]);
    (void)offsets;
    Tmp di, dv;
    ZKH_TRY(zkh_copy_from(c, "scatter_index", index, n_val, di.out()));
    ZKH_TRY(zkh_copy_from(c, "scatter_values", values, n_val, dv.out()));
    {
        ProfScope ps(c, "scatter", 12.0 * n_val);
        k_scatter<<<blocks_for(n_val), TB, 0, c->stream>>>(into->ptr(), di->ptr(), dv->ptr(), n_val);
    }

    return last_launch_error("scatter");
}
extern "C" const char* zkh_combos_prepare(zkh_ctx* c, zkh_buf* combos, const uint32_t* pos, const uint32_t* vals,
                                          size_t n) {
    if (!n) return nullptr;
    for (size_t k = 0; k < n; k++) ZKH_REQUIRE((size_t)pos[k] * 4 + 4 <= combos->len, "combos_prepare: position out of range");
    Tmp dp, dv;
    ZKH_TRY(zkh_copy_from(c, "prep_pos", pos, n, dp.out()));
    ZKH_TRY(zkh_copy_from(c, "prep_val", vals, 4 * n, dv.out()));
    {
        ProfScope ps(c, "combos_prepare", 36.0 * n);
        k_combos_prepare<<<blocks_for(n), TB, 0, c->stream>>>((uint4*)combos->ptr(), dp->ptr(), (const uint4*)dv->ptr(), n);
    }

    return last_launch_error("combos_prepare");
}
extern "C" const char* zkh_combos_prepare_regs(zkh_ctx* c, zkh_buf* combos, const zkh_buf* coeff_u, size_t combo_count, size_t cycles,
                                               size_t regs_count, const zkh_buf* reg_sizes, const zkh_buf* reg_combo_ids, const uint32_t mix[4]) {
    ZKH_REQUIRE(c && combos && coeff_u && reg_sizes && reg_combo_ids && mix, "combos_prepare_regs: null argument");
    ZKH_REQUIRE(combos->len == (combo_count + 1) * cycles * 4, "combos_prepare_regs: combos has %zu words, expected (combo_count + 1) x cycles ExtElems", combos->len);
    ZKH_REQUIRE(reg_sizes->len == regs_count && reg_combo_ids->len == regs_count, "combos_prepare_regs: reg_sizes / reg_combo_ids must hold regs_count words");
    ZKH_REQUIRE(coeff_u->len % 4 == 0 && coeff_u->len >= 4 * (regs_count + ZKH_CHECK_SIZE), "combos_prepare_regs: coeff_u too short");
    for (int i = 0; i < 4; i++) ZKH_REQUIRE(mix[i] < P, "combos_prepare_regs: mix is not a reduced element");
    // The register list is the caller's device data and this is a public entry point — and the Rust shim's per-seal call: NOTHING is
    // read back here.  k_combos_regs_check validates the list on the device (sizes, combo ids, that coeff_u holds what the sizes add up
    // to); an inconsistent list makes the main kernel a no-op and raises the context's sticky failure word, which the next zkh_sync /
    // zkh_read reports.  Never an out-of-bounds read, never a host round trip of its own.
    const Fp4 m(Fp::raw(mix[0]), Fp::raw(mix[1]), Fp::raw(mix[2]), Fp::raw(mix[3]));
    const size_t lanes = (combo_count + 1) * PREP_LANES_PER_COMBO;
    Tmp meta;
    ZKH_TRY(new_buf(c, 1, false, meta.out()));
    ProfScope ps(c, "combos_prepare", 36.0 * (double)(coeff_u->len / 4));
    k_combos_regs_check<<<1, 256, 0, c->stream>>>(meta->ptr(), c->d_fail, reg_sizes->ptr(), reg_combo_ids->ptr(), (uint32_t)regs_count,
                                                 (uint32_t)combo_count, cycles, coeff_u->len / 4);
    c->fail_armed = true;
    k_combos_prepare_regs<<<(unsigned)((lanes + 63) / 64), 64, 0, c->stream>>>((uint4*)combos->ptr(), combos->len / 4, (const uint4*)coeff_u->ptr(),
                                                                               (uint32_t)combo_count, cycles, (uint32_t)regs_count, reg_sizes->ptr(),
                                                                               reg_combo_ids->ptr(), m, meta->ptr());
    return last_launch_error("combos_prepare_regs");
}
extern "C" const char* zkh_merkle_open(zkh_ctx* c, const zkh_buf* matrix, const zkh_buf* nodes, size_t rows, size_t cols,
                                       const uint32_t* idx, size_t n_idx, zkh_buf* out) {
    ZKH_REQUIRE(matrix->len == rows * cols && nodes->len == rows * 16, "merkle_open: shape mismatch");
    unsigned layers = log2_ceil(rows), top_layer = 0;
    for (unsigned i = 1; i < layers; i++) { if ((1u << i) > ZKH_QUERIES) break; top_layer = i; }
    size_t top_size = (size_t)1 << top_layer;
    size_t wpq = cols + 8 * (size_t)(layers - top_layer);
    ZKH_REQUIRE(out->len >= wpq * n_idx, "merkle_open: output too small (%zu < %zu)", out->len, wpq * n_idx);
    for (size_t i = 0; i < n_idx; i++) ZKH_REQUIRE(idx[i] < rows, "merkle_open: index out of range");
    if (!n_idx) return nullptr;
    Tmp di;
    ZKH_TRY(zkh_copy_from(c, "open_idx", idx, n_idx, di.out()));
    {
        ProfScope ps(c, "merkle_open", 8.0 * wpq * n_idx);
        k_merkle_open<<<(unsigned)n_idx, TB, 0, c->stream>>>(out->ptr(), matrix->ptr(), nodes->ptr(), di->ptr(), rows, cols,
                                                            top_size, wpq);
    }

    return last_launch_error("merkle_open");
}
