// prover.hip — one segment seal driven entirely through the C ABI in include/zkhal.h.
// Host-side restatement (for this library's HAL) of risc0-zkp 3.0.2 src/prove/{prover.rs, poly_group.rs,
// merkle.rs, fri.rs, write_iop.rs} and src/core/hash/poseidon2/rng.rs, sequenced the way
// risc0-circuit-rv32im 4.0.2 src/prove SegmentProver::prove does (un-vendored: /root/reference/Cargo.lock:5393,
// :5320).  This is what /root/reference/crates/host/src/lib.rs:137 ends up running once per segment.
//
// Differences from upstream that change no result:
//   * commit_group's copy + iNTT + zk_shift are one out-of-place transform (zkh_batch_interpolate_ntt_from);
//   * the Merkle tree is built by zkh_merkle_build (hash_rows, then a lane-per-parent kernel for wide layers and an 8-lane cooperative
//     kernel below 2^15 parents);
//   * the 50 query indices only depend on the Fiat-Shamir state after the last commit, so they are drawn first and
//     each tree is opened for all of them with ONE gather kernel + ONE D2H instead of 50 x (gather + view).
#include <memory>

#include "circuit.h"
#include "poseidon2.h"
#include <map>

using namespace zkh;

namespace {

// ---------------- host Poseidon2 sponge + Fiat-Shamir RNG (rng.rs) ----------------
struct HostHash {
    const uint32_t* rc; const uint32_t* diag;
    void mix(uint32_t (&s)[CELLS]) const { poseidon2_mix(s, rc, diag); }
    void hash_elems(const uint32_t* in, size_t n, uint32_t out[8]) const {
        uint32_t s[CELLS] = {0};
        size_t unmixed = 0;
        for (size_t i = 0; i < n; i++) {
            s[unmixed++] = in[i];
            if (unmixed == RATE) { mix(s); unmixed = 0; }
        }
        if (unmixed != 0 || n == 0) {
            for (size_t i = unmixed; i < RATE; i++) s[i] = 0;
            mix(s);
        }
        memcpy(out, s, 32);
    }
};
struct Rng {
    const HostHash* h;
    uint32_t cells[CELLS] = {0};
    uint32_t pool_used = 0;
    void mix(const uint32_t d[8]) {
        if (pool_used != 0) { h->mix(cells); pool_used = 0; }
        for (int i = 0; i < OUT; i++) cells[i] = add_mod(cells[i], d[i]);
        h->mix(cells);
    }
    uint32_t random_elem() {
        if (pool_used == RATE) { h->mix(cells); pool_used = 0; }
        return cells[pool_used++];
    }
    uint32_t random_bits(unsigned bits) {
        uint32_t val = fp_decode(Fp::raw(random_elem()));
        for (int i = 0; i < 3; i++) {
            uint32_t nv = fp_decode(Fp::raw(random_elem()));
            if (val == 0) val = nv;
        }
        return (bits >= 32 ? 0xffffffffu : ((1u << bits) - 1)) & val;
    }
    Fp4 random_ext() {
        Fp4 r;
        for (int i = 0; i < 4; i++) r.c[i] = Fp::raw(random_elem());
        return r;
    }
};
struct Iop {
    std::vector<uint32_t> proof;
    Rng rng;
    void write(const uint32_t* p, size_t n) { proof.insert(proof.end(), p, p + n); }
    void commit(const uint32_t d[8]) { rng.mix(d); }
};

// RAII handle for zkh_buf
struct Buf {
    zkh_buf* b = nullptr;
    Buf() {}
    Buf(const Buf&) = delete;
    Buf& operator=(const Buf&) = delete;
    Buf(Buf&& o) : b(o.b) { o.b = nullptr; }
    Buf& operator=(Buf&& o) { reset(); b = o.b; o.b = nullptr; return *this; }
    ~Buf() { reset(); }
    void reset() { if (b) zkh_release(b); b = nullptr; }
    zkh_buf** out() { reset(); return &b; }
    operator zkh_buf*() const { return b; }
};

// ---------------- MerkleTreeProver (prove/merkle.rs) ----------------
struct Merkle {
    Buf nodes;
    zkh_buf* matrix = nullptr;   // borrowed
    size_t rows = 0, cols = 0, layers = 0, top_layer = 0, top_size = 0;
    std::vector<uint32_t> top;   // nodes[1 .. 2*top_size) as words; root = top[0..8)

    const char* enqueue(zkh_ctx* c, zkh_buf* mat, size_t rows_, size_t cols_) {
        rows = rows_; cols = cols_; matrix = mat;
        layers = log2_ceil(rows); top_layer = 0;
        for (size_t i = 1; i < layers; i++) { if (((size_t)1 << i) > ZKH_QUERIES) break; top_layer = i; }
        top_size = (size_t)1 << top_layer;
        ZKH_TRY(zkh_alloc(c, "nodes", rows * 2 * 8, 0, nodes.out()));
        return zkh_merkle_build(c, nodes, mat, rows);         // hash_rows + every hash_fold layer
    }
    // one D2H: root + everything down to the top layer (the only host-visible sync of a commit)
    const char* fetch_top(zkh_ctx* c) {
        top.resize((2 * top_size - 1) * 8);
        return zkh_read(c, nodes, top.data(), 8, top.size());
    }
    const uint32_t* root() const { return top.data(); }
    void commit(Iop& iop) const {
        iop.write(top.data() + (top_size - 1) * 8, top_size * 8);  // nodes[top_size .. 2*top_size)
        iop.commit(root());
    }
    size_t words_per_query() const { return cols + 8 * (layers - top_layer); }
    // open all query rows at once: result[q] = column words ++ path digests
    // (enqueue and fetch are split so that every tree's gather is in flight before the first read-back)
    const char* open_enqueue(zkh_ctx* c, const std::vector<uint32_t>& idx, Buf& out) const {
        ZKH_TRY(zkh_alloc(c, "open", words_per_query() * idx.size(), 0, out.out()));
        return zkh_merkle_open(c, matrix, nodes, rows, cols, idx.data(), idx.size(), out);
    }
    const char* open_fetch(zkh_ctx* c, const Buf& out, size_t n_idx, std::vector<uint32_t>& result) const {
        result.resize(words_per_query() * n_idx);
        return zkh_read(c, out, result.data(), 0, result.size());
    }
};

// ---------------- PolyGroup (prove/poly_group.rs) ----------------
struct PolyGroup {
    Buf coeffs, evaluated;
    size_t count = 0, n = 0;
    bool bitrev = false;         // coeffs left in the iNTT's bit-reversed order (n >= 2^14): consumers index accordingly
    Merkle merkle;
    // takes ownership of bit-reversed coefficient columns; `enqueue` only queues GPU work, `merkle.fetch_top` syncs
    const char* enqueue(zkh_ctx* c, Buf&& co, size_t count_, size_t n_) {
        coeffs = std::move(co); count = count_; n = n_;
        const size_t dom = n * ZKH_INV_RATE;
        ZKH_TRY(zkh_alloc(c, "evaluated", count * dom, 0, evaluated.out()));
        ZKH_TRY(zkh_batch_expand_into_evaluate_ntt(c, evaluated, coeffs, count, 2));
        // upstream bit-reverses the coefficients here for batch_evaluate_any / mix_poly_coeffs; both are order-agnostic
        // once they know the layout, so for n >= 2^14 the W x n words stay put and only the few combo polynomials are
        // bit-reversed later (zkh_batch_bit_reverse_extelem)
        bitrev = n >= ((size_t)1 << 14);
        if (!bitrev) ZKH_TRY(zkh_batch_bit_reverse(c, coeffs, count));
        return merkle.enqueue(c, evaluated, dom, count);
    }
    const char* build(zkh_ctx* c, Buf&& co, size_t count_, size_t n_) {
        ZKH_TRY(enqueue(c, std::move(co), count_, n_));
        return merkle.fetch_top(c);
    }
    // a second, read-only owner of a committed group (new handles on the same device allocations)
    const char* share_from(const PolyGroup& s) {
        count = s.count; n = s.n; bitrev = s.bitrev;
        ZKH_TRY(zkh_slice(s.coeffs, 0, s.coeffs.b->len, coeffs.out()));
        ZKH_TRY(zkh_slice(s.evaluated, 0, s.evaluated.b->len, evaluated.out()));
        ZKH_TRY(zkh_slice(s.merkle.nodes, 0, s.merkle.nodes.b->len, merkle.nodes.out()));
        merkle.matrix = evaluated;
        merkle.rows = s.merkle.rows; merkle.cols = s.merkle.cols; merkle.layers = s.merkle.layers;
        merkle.top_layer = s.merkle.top_layer; merkle.top_size = s.merkle.top_size; merkle.top = s.merkle.top;
        return nullptr;
    }
};

struct FriRound {
    size_t domain = 0;
    Buf coeffs, evaluated;
    Merkle merkle;
};

Fp4 lagrange_eval_basis(const std::vector<Fp4>& xs, size_t skip, std::vector<Fp4>& poly) {
    // poly = prod_{j != skip} (x - xs[j]); returns poly(xs[skip])
    poly.assign(1, Fp4::one());
    for (size_t j = 0; j < xs.size(); j++) {
        if (j == skip) continue;
        poly.push_back(Fp4::zero());
        for (size_t k = poly.size() - 1; k >= 1; k--) poly[k] = poly[k - 1] - poly[k] * xs[j];
        poly[0] = Fp4::zero() - poly[0] * xs[j];
    }
    Fp4 acc = Fp4::zero();
    for (size_t k = poly.size(); k-- > 0;) acc = acc * xs[skip] + poly[k];
    return acc;
}
// core/poly.rs poly_interpolate: coefficients of the unique degree < size polynomial through (xs, fx)
void poly_interpolate(Fp4* out, const Fp4* xs, const Fp4* fx, size_t size) {
    if (size == 1) { out[0] = fx[0]; return; }
    std::vector<Fp4> x(xs, xs + size), basis;
    for (size_t i = 0; i < size; i++) out[i] = Fp4::zero();
    for (size_t i = 0; i < size; i++) {
        const Fp4 d = lagrange_eval_basis(x, i, basis);
        const Fp4 m = fx[i] * fp4_inv(d);
        for (size_t k = 0; k < size; k++) out[k] = out[k] + basis[k] * m;
    }
}

}  // namespace

struct zkh_prover {
    zkh_ctx* ctx;
    const zkh_circuit* circuit;
    HostHash hash;
    std::map<size_t, std::unique_ptr<PolyGroup>> code_cache;   // po2 -> the committed code group, resident (zkh_prover_cache_code)
};

extern "C" const char* zkh_prover_create(zkh_ctx* ctx, const zkh_circuit* circuit, zkh_prover** out) {
    ZKH_REQUIRE(ctx && circuit, "prover_create: null argument");
    *out = new zkh_prover{ctx, circuit, HostHash{ctx->h_rc, ctx->h_diag}, {}};
    return nullptr;
}
extern "C" void zkh_prover_destroy(zkh_prover* p) { delete p; }
extern "C" void zkh_free_seal(uint32_t* s) { free(s); }

static const char* commit_group_enqueue(zkh_ctx* c, PolyGroup& pg, const zkh_buf* trace, size_t count, size_t n) {
    ZKH_REQUIRE(trace->len == count * n, "commit_group: trace has %zu words, expected %zu x %zu", trace->len, count, n);
    Buf coeffs;
    ZKH_TRY(zkh_alloc(c, "coeffs", count * n, 0, coeffs.out()));
    ZKH_TRY(zkh_batch_interpolate_ntt_from(c, coeffs, trace, count, 1));
    return pg.enqueue(c, std::move(coeffs), count, n);
}
static const char* commit_group_finish(zkh_ctx* c, Iop& iop, PolyGroup& pg) {
    ZKH_TRY(pg.merkle.fetch_top(c));
    pg.merkle.commit(iop);
    return nullptr;
}
static const char* commit_group(zkh_ctx* c, Iop& iop, PolyGroup& pg, const zkh_buf* trace, size_t count, size_t n) {
    ZKH_TRY(commit_group_enqueue(c, pg, trace, count, n));
    return commit_group_finish(c, iop, pg);
}

// An in-progress seal: everything Prover::commit_group has produced so far.  Upstream's SegmentProver drives
// `Prover` in exactly these two halves — commit code and data, draw the accum mix from the transcript, let the
// circuit's accum witness generator run (CircuitHal::accumulate, circuit-specific), then commit accum and finalize.
struct zkh_seal_job {
    zkh_prover* pr;
    size_t po2;
    Iop iop;
    PolyGroup groups[3];
    std::vector<uint32_t> out_global, mix_global;
};

extern "C" void zkh_prove_abort(zkh_seal_job* job) { delete job; }

extern "C" const char* zkh_prover_cache_code(zkh_prover* pr, size_t po2, const zkh_buf* code) {
    ZKH_REQUIRE(pr && code, "prover_cache_code: null argument");
    ZKH_REQUIRE(po2 >= 1 && po2 + 2 <= (size_t)MAX_LOG_N, "prover_cache_code: po2 %zu out of range", po2);
    ZKH_REQUIRE(code->len == ((size_t)pr->circuit->group_size[GROUP_CODE] << po2), "prover_cache_code: code trace has %zu words, expected %zu (W_code x 2^po2)",
                code->len, (size_t)pr->circuit->group_size[GROUP_CODE] << po2);
    std::unique_ptr<PolyGroup> pg(new PolyGroup());
    ZKH_TRY(commit_group_enqueue(pr->ctx, *pg, code, pr->circuit->group_size[GROUP_CODE], (size_t)1 << po2));
    ZKH_TRY(pg->merkle.fetch_top(pr->ctx));
    pr->code_cache[po2] = std::move(pg);
    return nullptr;
}
extern "C" void zkh_prover_drop_code_cache(zkh_prover* pr) { if (pr) pr->code_cache.clear(); }
// The resident entry is keyed by po2 alone, while a code trace also depends on zk_cycles: a caller that changes zk_cycles
// must re-cache.  This getter lets it check what is resident against the control root it expects before sealing with it.
extern "C" const char* zkh_prover_cached_code_root(zkh_prover* pr, size_t po2, uint32_t root[8]) {
    ZKH_REQUIRE(pr && root, "prover_cached_code_root: null argument");
    auto it = pr->code_cache.find(po2);
    ZKH_REQUIRE(it != pr->code_cache.end(), "prover_cached_code_root: no resident code group for po2 %zu", po2);
    memcpy(root, it->second->merkle.root(), 32);
    return nullptr;
}

extern "C" const char* zkh_prove_begin(zkh_prover* pr, size_t po2, const zkh_buf* code, const zkh_buf* data,
                                       const uint32_t* out_global, zkh_seal_job** out_job, uint32_t* mix_out) {
    ZKH_REQUIRE(pr && data && out_global && out_job, "prove_begin: null argument");
    ZKH_REQUIRE(code || pr->code_cache.count(po2), "prove_begin: no code trace and no resident code group for po2 %zu (zkh_prover_cache_code)", po2);
    zkh_ctx* c = pr->ctx;
    const zkh_circuit* cir = pr->circuit;
    ZKH_REQUIRE(po2 >= 1 && po2 + 2 <= (size_t)MAX_LOG_N, "prove_begin: po2 %zu too large (max %d)", po2, MAX_LOG_N - 2);
    const size_t n = (size_t)1 << po2;
    const size_t wc = cir->group_size[GROUP_CODE], wd = cir->group_size[GROUP_DATA];
    const size_t out_size = cir->global_size[GLOBAL_OUT];
    std::unique_ptr<zkh_seal_job> job(new zkh_seal_job());
    job->pr = pr; job->po2 = po2;
    Iop& iop = job->iop;
    iop.rng.h = &pr->hash;
    // ---- header: out globals + po2, both as field elements (write_field_elem_slice), bound into the transcript ----
    job->out_global.assign(out_global, out_global + out_size);
    for (size_t i = 0; i < out_size; i++) ZKH_REQUIRE(out_global[i] < P, "prove_begin: output global %zu is not a reduced element", i);
    {
        std::vector<uint32_t> hdr(job->out_global);
        hdr.push_back(fp_encode((uint32_t)po2).v);
        iop.write(hdr.data(), hdr.size());
        uint32_t dg[8];
        pr->hash.hash_elems(hdr.data(), hdr.size(), dg);
        iop.commit(dg);
    }
    // ---- commit code, data: neither commitment depends on a challenge, so both groups are queued before the first
    // root is read back; the transcript still absorbs them in upstream's order (code, then data)
    if (code) ZKH_TRY(commit_group_enqueue(c, job->groups[GROUP_CODE], code, wc, n));
    else ZKH_TRY(job->groups[GROUP_CODE].share_from(*pr->code_cache[po2]));      // resident: nothing to compute
    ZKH_TRY(commit_group_enqueue(c, job->groups[GROUP_DATA], data, wd, n));
    if (code) ZKH_TRY(commit_group_finish(c, iop, job->groups[GROUP_CODE]));
    else job->groups[GROUP_CODE].merkle.commit(iop);
    ZKH_TRY(commit_group_finish(c, iop, job->groups[GROUP_DATA]));
    // ---- accum mix challenges ----
    job->mix_global.resize(cir->global_size[GLOBAL_MIX] ? cir->global_size[GLOBAL_MIX] : 1);
    for (size_t i = 0; i < cir->global_size[GLOBAL_MIX]; i++) job->mix_global[i] = iop.rng.random_elem();
    if (mix_out) memcpy(mix_out, job->mix_global.data(), 4 * cir->global_size[GLOBAL_MIX]);
    *out_job = job.release();
    return nullptr;
}

static const char* prove_finish_impl(zkh_seal_job* job, const zkh_buf* accum, uint32_t** seal, size_t* seal_words);

extern "C" const char* zkh_prove_finish(zkh_seal_job* job_raw, const zkh_buf* accum, uint32_t** seal, size_t* seal_words) {
    ZKH_REQUIRE(job_raw && accum && seal && seal_words, "prove_finish: null argument");
    std::unique_ptr<zkh_seal_job> job(job_raw);      // consumed whether or not the seal succeeds
    return prove_finish_impl(job.get(), accum, seal, seal_words);
}

// SegmentProver::prove for the SYN-AIR family (kind 1), whose accum witness generator lives in this library.
extern "C" const char* zkh_prove_segment(zkh_prover* pr, size_t po2, size_t zk_cycles, const uint32_t* noise_key, const zkh_buf* code,
                                         const zkh_buf* data, const uint32_t* out_global, uint32_t** seal, size_t* seal_words) {
    ZKH_REQUIRE(pr && data && out_global && seal && seal_words, "prove_segment: null argument");
    zkh_ctx* c = pr->ctx;
    const zkh_circuit* cir = pr->circuit;
    ZKH_REQUIRE(cir->kind >= 1 && cir->kind <= 3, "prove_segment: no built-in accum witness generator for circuit kind %u (use zkh_prove_begin / zkh_prove_finish)", cir->kind);
    ZKH_REQUIRE(po2 + 2 <= (size_t)MAX_LOG_N, "prove_segment: po2 %zu too large (max %d)", po2, MAX_LOG_N - 2);
    const size_t n = (size_t)1 << po2;
    ZKH_REQUIRE(n > zk_cycles + 1, "prove_segment: po2 too small for zk_cycles");
    zkh_seal_job* job_raw = nullptr;
    ZKH_TRY(zkh_prove_begin(pr, po2, code, data, out_global, &job_raw, nullptr));
    std::unique_ptr<zkh_seal_job> job(job_raw);
    Buf accum;
    ZKH_TRY(zkh_alloc(c, "accum", (size_t)cir->group_size[GROUP_ACCUM] * n, 0, accum.out()));
    ZKH_TRY(zkh_syn_accum(c, cir, po2, zk_cycles, noise_key, data, job->mix_global.data(), accum));
    return prove_finish_impl(job.get(), accum, seal, seal_words);
}

// Merkle root of the committed code group — the control-ID analogue that zkh_verify_segment checks (verify/mod.rs
// check_code).  Same commit_group as the seal's, so the root equals what a seal of this code trace carries.
extern "C" const char* zkh_code_root(zkh_prover* pr, const zkh_buf* code, size_t po2, uint32_t root[8]) {
    ZKH_REQUIRE(pr && code && root, "code_root: null argument");
    ZKH_REQUIRE(po2 >= 1 && po2 + 2 <= (size_t)MAX_LOG_N, "code_root: po2 %zu out of range", po2);
    PolyGroup pg;
    ZKH_TRY(commit_group_enqueue(pr->ctx, pg, code, pr->circuit->group_size[GROUP_CODE], (size_t)1 << po2));
    ZKH_TRY(pg.merkle.fetch_top(pr->ctx));
    memcpy(root, pg.merkle.root(), 32);
    return nullptr;
}
extern "C" const char* zkh_syn_control_root(zkh_prover* pr, size_t po2, size_t zk_cycles, uint32_t root[8]) {
    ZKH_REQUIRE(pr && root, "syn_control_root: null argument");
    ZKH_REQUIRE(po2 >= 1 && po2 + 2 <= (size_t)MAX_LOG_N && ((size_t)1 << po2) > zk_cycles + 1, "syn_control_root: po2 %zu out of range", po2);
    Buf code;
    ZKH_TRY(zkh_alloc(pr->ctx, "code", (size_t)pr->circuit->group_size[GROUP_CODE] << po2, 0, code.out()));
    ZKH_TRY(zkh_syn_code(pr->ctx, pr->circuit, po2, zk_cycles, code));
    return zkh_code_root(pr, code, po2, root);
}

static const char* prove_finish_impl(zkh_seal_job* job, const zkh_buf* accum_trace, uint32_t** seal, size_t* seal_words) {
    zkh_prover* pr = job->pr;
    zkh_ctx* c = pr->ctx;
    const zkh_circuit* cir = pr->circuit;
    const size_t po2 = job->po2;
    const size_t n = (size_t)1 << po2, dom = n * ZKH_INV_RATE;
    const size_t wa = cir->group_size[GROUP_ACCUM];
    Iop& iop = job->iop;
    PolyGroup* groups = job->groups;
    const std::vector<uint32_t>& mix_global = job->mix_global;
    const uint32_t* out_global = job->out_global.data();
    ZKH_TRY(commit_group(c, iop, groups[GROUP_ACCUM], accum_trace, wa, n));
    // ---- finalize: constraint polynomial ----
    const Fp4 poly_mix = iop.rng.random_ext();
    PolyGroup check_group;
    {
        Buf check, g_out, g_mix;
        ZKH_TRY(zkh_alloc(c, "check_poly", ZKH_EXT_SIZE * dom, 0, check.out()));
        ZKH_TRY(zkh_copy_from(c, "out", out_global, job->out_global.size(), g_out.out()));
        ZKH_TRY(zkh_copy_from(c, "mix", mix_global.data(), mix_global.size(), g_mix.out()));
        const zkh_buf* gev[3] = {groups[0].evaluated, groups[1].evaluated, groups[2].evaluated};
        const zkh_buf* globals[2] = {g_out, g_mix};
        ZKH_TRY(zkh_eval_check(c, cir, check, gev, 3, globals, 2, (const uint32_t*)&poly_mix, po2, n, 0));
        ZKH_TRY(zkh_batch_interpolate_ntt(c, check, ZKH_EXT_SIZE));
        // 4 polys of degree 4n reinterpreted as 16 of degree n (bit-reversed layout makes them contiguous quarters)
        ZKH_TRY(check_group.build(c, std::move(check), ZKH_CHECK_SIZE, n));
        check_group.merkle.commit(iop);
    }
    // ---- DEEP: evaluate every tap at z * w^-back ----
    const Fp4 z = iop.rng.random_ext();
    const Fp back_one = Fp::raw(c->rou_rev[po2]);
    const size_t n_taps = cir->taps.size();
    std::vector<Fp4> all_xs(n_taps), eval_u(n_taps);
    const size_t n_u = n_taps + ZKH_CHECK_SIZE;
    std::vector<Fp4> coeff_u(n_u);
    const Fp4 z_pow = fp4_pow(z, ZKH_EXT_SIZE);
    {
        // all four evaluations (three trace groups at z*w^-back, the check group at z^4) are enqueued before the first
        // read-back, so the stream never idles on a host round trip between them
        Buf dout[4];
        size_t counts[4] = {0, 0, 0, 0}, pos = 0;
        for (uint32_t g = 0; g < 3; g++) {
            std::vector<uint32_t> which;
            for (size_t t = 0; t < n_taps; t++) {
                if (cir->taps[t].group != g) continue;
                which.push_back(cir->taps[t].offset);
                all_xs[pos + which.size() - 1] = z * fp_pow(back_one, cir->taps[t].back);
            }
            counts[g] = which.size();
            if (which.empty()) continue;
            Buf dw, dx;
            ZKH_TRY(zkh_copy_from(c, "which", which.data(), which.size(), dw.out()));
            ZKH_TRY(zkh_copy_from(c, "xs", (const uint32_t*)&all_xs[pos], 4 * which.size(), dx.out()));
            ZKH_TRY(zkh_alloc(c, "out", 4 * which.size(), 0, dout[g].out()));
            if (groups[g].bitrev) ZKH_TRY(zkh_batch_evaluate_any_bitrev(c, groups[g].coeffs, groups[g].count, dw, dx, dout[g]));
            else ZKH_TRY(zkh_batch_evaluate_any(c, groups[g].coeffs, groups[g].count, dw, dx, dout[g]));
            pos += which.size();
        }
        {
            uint32_t which[ZKH_CHECK_SIZE];
            Fp4 xs[ZKH_CHECK_SIZE];
            for (int i = 0; i < ZKH_CHECK_SIZE; i++) { which[i] = i; xs[i] = z_pow; }
            Buf dw, dx;
            ZKH_TRY(zkh_copy_from(c, "which", which, ZKH_CHECK_SIZE, dw.out()));
            ZKH_TRY(zkh_copy_from(c, "xs", (const uint32_t*)xs, 4 * ZKH_CHECK_SIZE, dx.out()));
            ZKH_TRY(zkh_alloc(c, "out", 4 * ZKH_CHECK_SIZE, 0, dout[3].out()));
            if (check_group.bitrev) ZKH_TRY(zkh_batch_evaluate_any_bitrev(c, check_group.coeffs, ZKH_CHECK_SIZE, dw, dx, dout[3]));
            else ZKH_TRY(zkh_batch_evaluate_any(c, check_group.coeffs, ZKH_CHECK_SIZE, dw, dx, dout[3]));
        }
        pos = 0;
        for (uint32_t g = 0; g < 3; g++) {
            if (counts[g]) ZKH_TRY(zkh_read(c, dout[g], (uint32_t*)&eval_u[pos], 0, 4 * counts[g]));
            pos += counts[g];
        }
        ZKH_TRY(zkh_read(c, dout[3], (uint32_t*)&coeff_u[n_taps], 0, 4 * ZKH_CHECK_SIZE));
        pos = 0;
        for (const Reg& r : cir->regs) {
            poly_interpolate(&coeff_u[pos], &all_xs[pos], &eval_u[pos], r.size);
            pos += r.size;
        }
    }
    iop.write((const uint32_t*)coeff_u.data(), 4 * n_u);
    {
        uint32_t dg[8];
        pr->hash.hash_elems((const uint32_t*)coeff_u.data(), 4 * n_u, dg);   // hash_ext_elem_slice
        iop.commit(dg);
    }
    // ---- FRI batching: mix all columns into one polynomial per combo ----
    const Fp4 mix = iop.rng.random_ext();
    const size_t combo_count = cir->combos.size();
    Buf combos;
    ZKH_TRY(zkh_alloc(c, "combos", n * (combo_count + 1) * 4, 1, combos.out()));
    {
        Fp4 cur_mix = Fp4::one();
        for (uint32_t g = 0; g < 3; g++) {
            std::vector<uint32_t> which;
            for (const Reg& r : cir->regs) if (r.group == g) which.push_back(r.combo_id);
            Buf dw;
            ZKH_TRY(zkh_copy_from(c, "which", which.data(), which.size(), dw.out()));
            ZKH_TRY(zkh_mix_poly_coeffs(c, combos, (const uint32_t*)&cur_mix, (const uint32_t*)&mix, groups[g].coeffs, dw,
                                        which.size(), n));
            cur_mix = cur_mix * fp4_pow(mix, which.size());
        }
        std::vector<uint32_t> which(ZKH_CHECK_SIZE, (uint32_t)combo_count);
        Buf dw;
        ZKH_TRY(zkh_copy_from(c, "which", which.data(), which.size(), dw.out()));
        ZKH_TRY(zkh_mix_poly_coeffs(c, combos, (const uint32_t*)&cur_mix, (const uint32_t*)&mix, check_group.coeffs, dw,
                                    ZKH_CHECK_SIZE, n));
    }
    if (check_group.bitrev) ZKH_TRY(zkh_batch_bit_reverse_extelem(c, combos, combo_count + 1));   // combos -> natural order
    // combos_prepare: subtract the interpolated U polynomials (aggregated per position on the host)
    {
        std::map<uint32_t, Fp4> sub;
        auto acc = [&](uint32_t pos, Fp4 v) {
            auto it = sub.find(pos);
            if (it == sub.end()) sub.emplace(pos, v); else it->second = it->second + v;
        };
        size_t cur_pos = 0;
        Fp4 cur = Fp4::one();
        for (const Reg& r : cir->regs) {
            for (uint32_t i = 0; i < r.size; i++) acc((uint32_t)(n * r.combo_id + i), cur * coeff_u[cur_pos + i]);
            cur = cur * mix; cur_pos += r.size;
        }
        for (int i = 0; i < ZKH_CHECK_SIZE; i++) {
            acc((uint32_t)(n * combo_count), cur * coeff_u[cur_pos]);
            cur_pos++; cur = cur * mix;
        }
        std::vector<uint32_t> pos; std::vector<Fp4> vals;
        for (auto& kv : sub) { pos.push_back(kv.first); vals.push_back(kv.second); }
        ZKH_TRY(zkh_combos_prepare(c, combos, pos.data(), (const uint32_t*)vals.data(), pos.size()));
    }
    // combos_divide: by prod (x - z w^-back) per combo, the check combo by (x - z^4); remainders must vanish.  One call:
    // the r-th divisions of all combos share their launches.
    {
        std::vector<Fp4> pts;
        std::vector<uint32_t> begin(1, 0);
        for (size_t i = 0; i <= combo_count; i++) {
            if (i == combo_count) pts.push_back(z_pow);
            else for (uint32_t b : cir->combos[i]) pts.push_back(z * fp_pow(back_one, b));
            begin.push_back((uint32_t)pts.size());
        }
        Buf rems;
        ZKH_TRY(zkh_alloc(c, "rems", 4 * pts.size(), 1, rems.out()));
        ZKH_TRY(zkh_combos_divide_all(c, combos, n, combo_count + 1, (const uint32_t*)pts.data(), begin.data(), rems));
        std::vector<uint32_t> r(4 * pts.size());
        ZKH_TRY(zkh_read(c, rems, r.data(), 0, r.size()));
        for (uint32_t w : r) ZKH_REQUIRE(w == 0, "prove_segment: DEEP quotient has a non-zero remainder (witness does not satisfy the constraints)");
    }
    Buf final_coeffs;
    ZKH_TRY(zkh_alloc(c, "final_poly_coeffs", n * ZKH_EXT_SIZE, 0, final_coeffs.out()));
    ZKH_TRY(zkh_eltwise_sum_extelem(c, final_coeffs, combos));
    combos.reset();
    ZKH_TRY(zkh_batch_bit_reverse(c, final_coeffs, ZKH_EXT_SIZE));

    // ---- fri_prove ----
    std::vector<std::unique_ptr<FriRound>> rounds;
    {
        zkh_buf* cur = final_coeffs;
        while (cur->len / ZKH_EXT_SIZE > ZKH_FRI_MIN_DEGREE) {
            std::unique_ptr<FriRound> r(new FriRound());
            const size_t size = cur->len / ZKH_EXT_SIZE;
            r->domain = size * ZKH_INV_RATE;
            ZKH_TRY(zkh_alloc(c, "evaluated", r->domain * ZKH_EXT_SIZE, 0, r->evaluated.out()));
            ZKH_TRY(zkh_batch_expand_into_evaluate_ntt(c, r->evaluated, cur, ZKH_EXT_SIZE, 2));
            ZKH_TRY(r->merkle.enqueue(c, r->evaluated, r->domain / ZKH_FRI_FOLD, ZKH_FRI_FOLD * ZKH_EXT_SIZE));
            ZKH_TRY(r->merkle.fetch_top(c));
            r->merkle.commit(iop);
            const Fp4 fold_mix = iop.rng.random_ext();
            ZKH_TRY(zkh_alloc(c, "out_coeffs", size / ZKH_FRI_FOLD * ZKH_EXT_SIZE, 0, r->coeffs.out()));
            ZKH_TRY(zkh_fri_fold(c, r->coeffs, cur, (const uint32_t*)&fold_mix));
            cur = r->coeffs;
            rounds.push_back(std::move(r));
        }
        Buf fin;
        ZKH_TRY(zkh_alloc(c, "final_coeffs", cur->len, 0, fin.out()));
        ZKH_TRY(zkh_eltwise_copy_elem(c, fin, cur));
        ZKH_TRY(zkh_batch_bit_reverse(c, fin, ZKH_EXT_SIZE));
        std::vector<uint32_t> fw(cur->len);
        ZKH_TRY(zkh_read(c, fin, fw.data(), 0, fw.size()));
        iop.write(fw.data(), fw.size());
        uint32_t dg[8];
        pr->hash.hash_elems(fw.data(), fw.size(), dg);
        iop.commit(dg);
    }
    // ---- queries: draw all indices, open every tree once, then serialise in upstream order ----
    {
        const size_t orig_domain = dom;
        std::vector<uint32_t> pos0(ZKH_QUERIES);
        for (int q = 0; q < ZKH_QUERIES; q++) pos0[q] = iop.rng.random_bits(log2_ceil(orig_domain)) % (uint32_t)orig_domain;
        std::vector<std::vector<uint32_t>> opened(4 + rounds.size());
        const Merkle* trees[4] = {&groups[0].merkle, &groups[1].merkle, &groups[2].merkle, &check_group.merkle};
        std::vector<Buf> dev(4 + rounds.size());
        for (int t = 0; t < 4; t++) ZKH_TRY(trees[t]->open_enqueue(c, pos0, dev[t]));
        std::vector<uint32_t> pos = pos0;
        for (size_t r = 0; r < rounds.size(); r++) {
            for (auto& p : pos) p %= (uint32_t)(rounds[r]->domain / ZKH_FRI_FOLD);
            ZKH_TRY(rounds[r]->merkle.open_enqueue(c, pos, dev[4 + r]));
        }
        for (int t = 0; t < 4; t++) ZKH_TRY(trees[t]->open_fetch(c, dev[t], ZKH_QUERIES, opened[t]));
        for (size_t r = 0; r < rounds.size(); r++) ZKH_TRY(rounds[r]->merkle.open_fetch(c, dev[4 + r], ZKH_QUERIES, opened[4 + r]));
        for (int q = 0; q < ZKH_QUERIES; q++) {
            for (int t = 0; t < 4; t++) {
                const size_t w = trees[t]->words_per_query();
                iop.write(opened[t].data() + w * q, w);
            }
            for (size_t r = 0; r < rounds.size(); r++) {
                const size_t w = rounds[r]->merkle.words_per_query();
                iop.write(opened[4 + r].data() + w * q, w);
            }
        }
    }
    *seal_words = iop.proof.size();
    *seal = (uint32_t*)malloc(iop.proof.size() * 4 + 4);
    memcpy(*seal, iop.proof.data(), iop.proof.size() * 4);
    return nullptr;
}
