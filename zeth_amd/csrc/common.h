// common.h — internal structures of libzkhal_mi355x: context, buffers, stream-ordered pool, launch + profiling.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/zkhal.h"
#include "fp.h"
#include "noise.h"

namespace zkh {

// Error strings follow risc0-sys's convention: NULL = ok, heap string otherwise.
inline const char* make_err(const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    return strdup(buf);
}
#define ZKH_HIP(expr)                                                                               \
    do {                                                                                            \
        hipError_t _e = (expr);                                                                     \
        if (_e != hipSuccess) return zkh::make_err("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)
#define ZKH_TRY(expr)                    \
    do {                                 \
        const char* _err = (expr);       \
        if (_err) return _err;           \
    } while (0)
#define ZKH_REQUIRE(cond, ...)                         \
    do {                                               \
        if (!(cond)) return zkh::make_err(__VA_ARGS__); \
    } while (0)

constexpr int ZKH_P2_PTAB = 414;            // partial-round table, layout in poseidon2.h (P2_TAB_WORDS)
constexpr int TW_BITS = 12;                 // two-level twiddle tables: w_{2^26}^(hi*4096 + lo), lo < 2^12, hi < 2^14
constexpr int TW_SIZE = 1 << TW_BITS;
constexpr int TW_HI_BITS = 14;              // (64 KiB per hi table: L2-resident; the index arithmetic of every kernel is unchanged)
constexpr int TW_HI_SIZE = 1 << TW_HI_BITS;
// largest NTT / coset-shift domain: 2^26 = the evaluation domain (INV_RATE 4) of a po2-24 segment, upstream's MAX_CYCLES_PO2
// (--segment-po2 at /root/reference/crates/host/src/bin/cli.rs:61-66 accepts it); BabyBear's 2-adicity is 27
constexpr int MAX_LOG_N = TW_BITS + TW_HI_BITS;
constexpr int LDS_TW_LOG = 12;              // in-tile twiddles: w_{2^12}^j, j < 2^11

struct DeviceTables {
    // Poseidon2 (Montgomery form)
    uint32_t* rc;        // 24*29
    uint32_t* diag;      // ZKH_P2_PTAB words: partial-round table
    // twiddles, Montgomery form
    uint32_t* tw_fwd_lo; uint32_t* tw_fwd_hi;   // w^lo, w^(hi*4096), w = ROU_FWD[MAX_LOG_N]
    uint32_t* tw_rev_lo; uint32_t* tw_rev_hi;   // same for ROU_REV[MAX_LOG_N]
    uint32_t* tile_fwd;  uint32_t* tile_rev;    // ROU_FWD[12]^j / ROU_REV[12]^j, j < 2048
    uint32_t* layer_fwd; uint32_t* layer_rev;   // per-layer: [2^(j-1) + e] = ROU[j]^e, e < 2^(j-1), j <= 12 (4096 words)
    uint32_t* layer_fwd_plain;                  // layer_fwd out of Montgomery form (plain residues): lazy butterflies (ntt.hip)
    uint32_t* shift_lo;  uint32_t* shift_hi;    // 3^lo, 3^(hi*4096), hi < 2^14
};

struct ProfAgg { uint64_t calls = 0; double ms = 0; double bytes = 0; };
struct ProfPending { std::string name; hipEvent_t a, b; double bytes; };

}  // namespace zkh

// Reference counts are atomic so that handles may be retained / released from any host thread; everything that touches
// the context itself (pool, stream, staging ring) stays single-threaded per context, as include/zkhal.h states.
struct zkh_alloc_t {
    void* ptr;
    size_t bytes;
    std::atomic<int> refs;
    bool owned;
    zkh_ctx* ctx;
    zkh_alloc_t(void* p, size_t b, int r, bool o, zkh_ctx* c) : ptr(p), bytes(b), refs(r), owned(o), ctx(c) {}
};

struct zkh_buf {
    zkh_alloc_t* a;
    size_t off, len;   // words
    std::atomic<int> refs;
    zkh_buf(zkh_alloc_t* al, size_t o, size_t l, int r) : a(al), off(o), len(l), refs(r) {}
    uint32_t* ptr() const { return (uint32_t*)a->ptr + off; }
};

struct zkh_ctx {
    int device;
    hipStream_t stream;
    zkh::DeviceTables tab;
    uint32_t rou_fwd[28], rou_rev[28];          // host copies (Montgomery)
    uint32_t h_rc[24 * 29], h_diag[zkh::ZKH_P2_PTAB]; // host copies of the Poseidon2 tables (rc - P; partial-round table)
    std::multimap<size_t, void*> pool;          // stream-ordered free list (single in-order stream)
    size_t pool_bytes = 0, live_bytes = 0, peak_bytes = 0;
    bool prof = false;
    std::vector<zkh::ProfPending> pending;
    std::map<std::string, zkh::ProfAgg> agg;
    std::vector<hipEvent_t> event_pool;
    uint32_t* pinned = nullptr;                  // host staging
    size_t pinned_words = 0;
    size_t stage_next = 0, stage_used = 0;       // pinned staging ring for small uploads (hal.hip h2d)
    // Device-side argument checks that must not cost a host round trip (zkh_combos_prepare_regs: the register list is DEVICE data):
    // the kernel refuses to index out of bounds and raises this sticky word instead; the next host-visible sync point of the context
    // (zkh_sync / zkh_read — one per commit) reads it back with the data it was reading anyway and reports the failure there.
    uint32_t* d_fail = nullptr;                  // 2 words: code, detail
    uint32_t* h_fail = nullptr;                  // pinned mirror
    bool fail_armed = false;                     // an op that may raise it was enqueued since the last check
    std::map<void*, size_t> host_blocks;         // pinned host memory handed to the caller (zkh_host_alloc): ptr -> bytes
};

namespace zkh {

const char* pool_alloc(zkh_ctx* c, size_t bytes, void** out);
void pool_free(zkh_ctx* c, void* p, size_t bytes);
const char* new_buf(zkh_ctx* c, size_t n_words, bool zero, zkh_buf** out);
void prof_begin(zkh_ctx* c, const char* name, double bytes);
void prof_end(zkh_ctx* c);
const char* ensure_pinned(zkh_ctx* c, size_t words);
// The blinding key of one seal / witness: the caller's 8 words, or — NULL or all-zero — 256 fresh bits from the OS (getrandom).  If
// the OS cannot deliver, the call FAILS: predictable blinding rows would silently lose zero-knowledge.  (hal.hip)
const char* resolve_noise_key(const uint32_t* key_or_null, NoiseKey* out);

// The HIP "current device" is per host thread: every entry point binds the calling thread to the context's GPU
// (several host threads may each drive their own context on the same or on different GPUs).  No caching: other code
// on this thread (a torch synchronize, the embedding host) may have selected another device since the last call.
inline void bind_thread(const zkh_ctx* c) { (void)hipSetDevice(c->device); }
// per-device kernel attributes (dynamic LDS sizes); set when a context is created on the device (ntt.hip)
const char* ntt_device_init(zkh_ctx* c);

// Scoped temporary buffer of an op: released on every return path (ZKH_TRY / ZKH_REQUIRE leave early on errors).
struct Tmp {
    zkh_buf* b = nullptr;
    Tmp() {}
    Tmp(const Tmp&) = delete;
    Tmp& operator=(const Tmp&) = delete;
    ~Tmp() { if (b) zkh_release(b); }
    zkh_buf** out() { if (b) { zkh_release(b); b = nullptr; } return &b; }
    operator zkh_buf*() const { return b; }
    zkh_buf* operator->() const { return b; }
    explicit operator bool() const { return b != nullptr; }
};

// Launch helper: optional HIP-event bracket on the ctx stream (what bench.py's roofline uses).
struct ProfScope {
    zkh_ctx* c;
    ProfScope(zkh_ctx* ctx, const char* name, double bytes) : c(ctx) { bind_thread(c); if (c->prof) prof_begin(c, name, bytes); }
    ~ProfScope() { if (c->prof) prof_end(c); }
};

inline const char* last_launch_error(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return make_err("launch %s: %s", what, hipGetErrorString(e));
    return nullptr;
}

// XCD-aware block remap: hardware places block b on XCD b % 8; make consecutive logical tiles share an XCD
// (and its L2) by giving XCD x the x-th contiguous slice of the tile range.  Bijective for any nblocks.
__device__ __forceinline__ uint32_t xcd_remap(uint32_t b, uint32_t nblocks) {
    const uint32_t xcd = b & 7u, q = nblocks >> 3, r = nblocks & 7u, idx = b >> 3;
    const uint32_t base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

}  // namespace zkh
