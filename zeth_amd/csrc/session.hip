// session.hip — the session executor: every segment of a session sealed on G devices x K lanes, receipts in index order,
// optionally folded through the P2-JOIN tree to one root receipt, and the verification of what comes out.
//
// Stands in for risc0-zkvm 3.0.3 `ProverServer::prove_session` / `ProverImpl::{prove_segment, lift, join}` (un-vendored:
// /root/reference/Cargo.lock:5418) — what `default_prover().prove(env, elf)` (/root/reference/crates/host/src/lib.rs:137) runs
// after the executor has cut the guest's execution into segments — and for `receipt.verify(image_id)`
// (/root/reference/crates/host/src/bin/cli.rs:103).  Upstream proves the segments of a session in a plain loop on one
// device; segments are independent, so here segment i goes to whichever lane is free next (one shared work
// index over all lanes of all devices: round-robin with work stealing for the short tail), with NO exchange between devices.
// Host code only (threads + the C ABI of this library); one lane = one zkh_ctx (device + stream) + circuit + prover.
#include <atomic>
#include <chrono>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <sys/random.h>

#include "circuit.h"

using namespace zkh;

namespace {

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

uint64_t os_random64() {
    uint64_t v = 0;
    if (getrandom(&v, sizeof v, 0) != (ssize_t)sizeof v) v = (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count() * 0x9E3779B97F4A7C15ull;
    return v ? v : 1;
}

struct Lane {
    int device = 0;
    zkh_ctx* ctx = nullptr;
    zkh_circuit *circuit = nullptr, *join_circuit = nullptr, *rec_circuit = nullptr;
    zkh_prover *prover = nullptr, *join_prover = nullptr;
    std::vector<zkh_rec_program*> programs;
    void close() {
        for (auto p : programs) zkh_rec_program_destroy(p);
        programs.clear();
        if (rec_circuit) zkh_circuit_destroy(rec_circuit);
        rec_circuit = nullptr;
        if (join_prover) zkh_prover_destroy(join_prover);
        if (prover) zkh_prover_destroy(prover);
        if (join_circuit) zkh_circuit_destroy(join_circuit);
        if (circuit) zkh_circuit_destroy(circuit);
        if (ctx) zkh_ctx_destroy(ctx);
        join_prover = prover = nullptr; join_circuit = circuit = nullptr; ctx = nullptr;
    }
};

struct ErrorSlot {
    std::mutex lock;
    std::string first;
    bool set(const char* err, const char* what) {          // takes ownership of err; returns true if there was an error
        if (!err) return false;
        {
            std::lock_guard<std::mutex> lk(lock);
            if (first.empty()) first = std::string(what) + ": " + err;
        }
        zkh_free_error(err);
        return true;
    }
    bool any() { std::lock_guard<std::mutex> lk(lock); return !first.empty(); }
};

}  // namespace

struct RecKind { uint32_t join, a, b; };
struct zkh_session {
    std::vector<uint32_t> desc, join_desc, rec_desc;
    std::vector<RecKind> rec_kinds;
    std::vector<std::vector<uint32_t>> rec_roots;                 // control root of program i
    std::vector<std::vector<std::vector<uint32_t>>> allowed;       // levels of the allowed-programs tree (8 leaves)
    std::vector<Lane> lanes;
    std::vector<Lane> fold_lanes;        // extra contexts that only lift / join (zkh_session_set_recursion): the fold packs the GPU with more lanes than the seals need
    std::vector<Lane*> rec_lanes;        // lanes + fold_lanes
    size_t lanes_per_device = 0;
    zkh_accumulate_fn accumulate = nullptr;
    void* accumulate_user = nullptr;
    ~zkh_session() { for (auto& l : fold_lanes) l.close(); for (auto& l : lanes) l.close(); }
};

extern "C" const char* zkh_session_create(const int* devices, size_t n_devices, size_t lanes_per_device, const uint32_t* desc, size_t desc_words,
                                          const uint32_t* join_desc, size_t join_desc_words, zkh_session** out) {
    ZKH_REQUIRE(devices && n_devices && lanes_per_device && desc && desc_words >= 16 && out, "session_create: bad argument");
    ZKH_REQUIRE(!join_desc || (join_desc_words >= 16 && join_desc[13] == 3), "session_create: the join circuit must be a P2-JOIN description (kind 3)");
    std::unique_ptr<zkh_session> s(new zkh_session());
    s->desc.assign(desc, desc + desc_words);
    if (join_desc) s->join_desc.assign(join_desc, join_desc + join_desc_words);
    s->lanes.resize(n_devices * lanes_per_device);
    s->lanes_per_device = lanes_per_device;
    for (size_t i = 0; i < s->lanes.size(); i++) {
        Lane& l = s->lanes[i];
        l.device = devices[i / lanes_per_device];
        ZKH_TRY(zkh_ctx_create(l.device, "poseidon2", &l.ctx));
        ZKH_TRY(zkh_circuit_load(l.ctx, s->desc.data(), s->desc.size(), &l.circuit));
        ZKH_TRY(zkh_prover_create(l.ctx, l.circuit, &l.prover));
        if (join_desc) {
            ZKH_TRY(zkh_circuit_load(l.ctx, s->join_desc.data(), s->join_desc.size(), &l.join_circuit));
            ZKH_TRY(zkh_prover_create(l.ctx, l.join_circuit, &l.join_prover));
        }
    }
    *out = s.release();
    return nullptr;
}
extern "C" void zkh_session_destroy(zkh_session* s) { delete s; }
extern "C" size_t zkh_session_lanes(const zkh_session* s) { return s ? s->lanes.size() : 0; }
extern "C" zkh_circuit* zkh_session_circuit(zkh_session* s, size_t lane, int join) {
    if (!s || lane >= s->lanes.size()) return nullptr;
    return join ? s->lanes[lane].join_circuit : s->lanes[lane].circuit;
}
extern "C" void zkh_session_set_accumulate(zkh_session* s, zkh_accumulate_fn fn, void* user) {
    if (s) { s->accumulate = fn; s->accumulate_user = user; }
}

static const char* hash_pair_host(const uint32_t* a, const uint32_t* b, uint32_t out[8]) {
    uint32_t st[24] = {0};
    memcpy(st, a, 32); memcpy(st + 8, b, 32);
    ZKH_TRY(zkh_poseidon2_mix_host(nullptr, nullptr, st, 1));
    memcpy(out, st, 32);
    return nullptr;
}
constexpr size_t REC_ALLOWED = 16, REC_DEPTH = 4;                   // zeth_amd/circuits/rec_verify.py ALLOWED_DEPTH

extern "C" const char* zkh_session_set_recursion(zkh_session* s, const uint32_t* rec_desc, size_t rec_desc_words, const uint32_t* const* blobs,
                                                 const size_t* words, const uint32_t* kinds, size_t n_programs) {
    ZKH_REQUIRE(s && rec_desc && rec_desc_words >= 16 && blobs && words && kinds && n_programs && n_programs <= REC_ALLOWED, "session_set_recursion: bad argument");
    ZKH_REQUIRE(s->rec_kinds.empty(), "session_set_recursion: the session already has its programs");
    s->rec_desc.assign(rec_desc, rec_desc + rec_desc_words);
    // a lift's / join's witness schedule is a chain of ~300 dependency levels (latency): the fold runs on ZKH_FOLD_LANES lanes
    // per device (default 6: the measured knee, profiles/r03_recursion_fold_lanes.txt), the sealing lanes plus extra contexts
    const char* env = getenv("ZKH_FOLD_LANES");
    const size_t want = env ? (size_t)strtoul(env, nullptr, 10) : 6;
    const size_t n_devices = s->lanes.size() / s->lanes_per_device;
    if (want > s->lanes_per_device) {
        s->fold_lanes.resize(n_devices * (want - s->lanes_per_device));
        for (size_t i = 0; i < s->fold_lanes.size(); i++) {
            Lane& l = s->fold_lanes[i];
            l.device = s->lanes[(i / (want - s->lanes_per_device)) * s->lanes_per_device].device;
            ZKH_TRY(zkh_ctx_create(l.device, "poseidon2", &l.ctx));
        }
    }
    for (auto& l : s->lanes) s->rec_lanes.push_back(&l);
    for (auto& l : s->fold_lanes) s->rec_lanes.push_back(&l);
    for (Lane* lp : s->rec_lanes) {
        Lane& l = *lp;
        ZKH_TRY(zkh_circuit_load(l.ctx, s->rec_desc.data(), s->rec_desc.size(), &l.rec_circuit));
        for (size_t i = 0; i < n_programs; i++) {
            zkh_rec_program* p = nullptr;
            ZKH_TRY(zkh_rec_program_load(l.ctx, l.rec_circuit, blobs[i], words[i], &p));
            l.programs.push_back(p);
        }
    }
    for (size_t i = 0; i < n_programs; i++) {
        s->rec_kinds.push_back(RecKind{kinds[3 * i], kinds[3 * i + 1], kinds[3 * i + 2]});
        std::vector<uint32_t> root(8);
        ZKH_TRY(zkh_rec_program_info(s->lanes[0].programs[i], root.data(), nullptr));
        s->rec_roots.push_back(root);
    }
    std::vector<std::vector<uint32_t>> level(s->rec_roots);
    level.resize(REC_ALLOWED, std::vector<uint32_t>(8, 0));
    s->allowed.assign(1, level);
    while (level.size() > 1) {
        std::vector<std::vector<uint32_t>> up(level.size() / 2, std::vector<uint32_t>(8));
        for (size_t k = 0; k < up.size(); k++) ZKH_TRY(hash_pair_host(level[2 * k].data(), level[2 * k + 1].data(), up[k].data()));
        s->allowed.push_back(up);
        level.swap(up);
    }
    return nullptr;
}

extern "C" void zkh_prove_info_free(zkh_prove_info* info) {
    if (!info) return;
    for (size_t i = 0; i < info->n_segments; i++) if (info->seals) zkh_free_seal(info->seals[i]);
    free(info->seals); free(info->seal_words);
    zkh_free_seal(info->root_seal);
    memset(info, 0, sizeof *info);
}

// control root of the leaf circuit at one segment's size: generated for the built-in circuits, committed from the caller's
// code trace otherwise (a deployment ships these per (circuit, po2): upstream's control IDs)
static const char* leaf_control_root(zkh_session* s, const zkh_segment& seg, uint32_t root[8]) {
    Lane& l = s->lanes[0];
    if (l.circuit->kind >= 1 && l.circuit->kind <= 3) return zkh_syn_control_root(l.prover, seg.po2, ZKH_ZK_CYCLES, root);
    ZKH_REQUIRE(seg.host_code, "session: circuit kind %u has no built-in code generator and the segment carries no code trace", l.circuit->kind);
    Tmp code;
    ZKH_TRY(zkh_copy_from(l.ctx, "code", seg.host_code, (size_t)l.circuit->group_size[GROUP_CODE] << seg.po2, code.out()));
    return zkh_code_root(l.prover, code, seg.po2, root);
}

// one segment on one lane: built-in witness generator, or the caller's traces through prove_begin / accumulate / prove_finish
static const char* seal_one(zkh_session* s, Lane& l, const zkh_segment& seg, uint32_t** seal, size_t* words, double* witgen_s) {
    const zkh_circuit* cir = l.circuit;
    const size_t n = (size_t)1 << seg.po2;
    const uint64_t noise = seg.noise_seed ? seg.noise_seed : os_random64();
    Tmp code, data;
    ZKH_TRY(zkh_alloc(l.ctx, "code", (size_t)cir->group_size[GROUP_CODE] * n, 0, code.out()));
    ZKH_TRY(zkh_alloc(l.ctx, "data", (size_t)cir->group_size[GROUP_DATA] * n, 0, data.out()));
    std::vector<uint32_t> out_global(cir->global_size[GLOBAL_OUT]);
    const double t0 = now_s();
    if (seg.host_code || seg.host_data) {
        ZKH_REQUIRE(seg.host_code && seg.host_data && seg.out_global, "session: a segment with host traces needs host_code, host_data and out_global");
        ZKH_TRY(zkh_write(l.ctx, code, seg.host_code, 0, code->len));
        ZKH_TRY(zkh_write(l.ctx, data, seg.host_data, 0, data->len));
        out_global.assign(seg.out_global, seg.out_global + out_global.size());
        *witgen_s = now_s() - t0;
        zkh_seal_job* job = nullptr;
        std::vector<uint32_t> mix(cir->global_size[GLOBAL_MIX] ? cir->global_size[GLOBAL_MIX] : 1);
        ZKH_TRY(zkh_prove_begin(l.prover, seg.po2, code, data, out_global.data(), &job, mix.data()));
        Tmp accum;
        const char* err = zkh_alloc(l.ctx, "accum", (size_t)cir->group_size[GROUP_ACCUM] * n, 0, accum.out());
        if (!err) {
            if (s->accumulate) err = s->accumulate(s->accumulate_user, l.ctx, cir, seg.po2, data, mix.data(), accum);
            else if (cir->kind >= 1 && cir->kind <= 3) err = zkh_syn_accum(l.ctx, cir, seg.po2, ZKH_ZK_CYCLES, noise, data, mix.data(), accum);
            else err = make_err("session: circuit kind %u has no built-in accum witness generator and no accumulate callback was set", cir->kind);
        }
        if (err) { zkh_prove_abort(job); return err; }
        return zkh_prove_finish(job, accum, seal, words);
    }
    ZKH_REQUIRE(cir->kind >= 1 && cir->kind <= 3, "session: circuit kind %u has no built-in witness generator: supply host traces", cir->kind);
    ZKH_TRY(zkh_syn_witgen(l.ctx, cir, seg.po2, ZKH_ZK_CYCLES, seg.seed, noise, seg.n_pub ? seg.pub : nullptr, code, data, out_global.data()));
    *witgen_s = now_s() - t0;
    return zkh_prove_segment(l.prover, seg.po2, ZKH_ZK_CYCLES, noise, code, data, out_global.data(), seal, words);
}

extern "C" const char* zkh_session_prove(zkh_session* s, const zkh_segment* segs, size_t n, int join_tree, size_t join_po2, uint64_t join_noise_seed,
                                         zkh_prove_info* info) {
    ZKH_REQUIRE(s && segs && n && info, "session_prove: bad argument");
    ZKH_REQUIRE(join_tree != 1 || !s->join_desc.empty(), "session_prove: the session was created without a join circuit");
    ZKH_REQUIRE(join_tree != 2 || !s->rec_kinds.empty(), "session_prove: join_tree 2 needs zkh_session_set_recursion");
    memset(info, 0, sizeof *info);
    info->n_segments = n;
    info->seals = (uint32_t**)calloc(n, sizeof(uint32_t*));
    info->seal_words = (size_t*)calloc(n, sizeof(size_t));
    ErrorSlot errs;
    std::atomic<size_t> next{0};
    std::mutex stat_lock;
    const double t0 = now_s();
    {
        std::vector<std::thread> th;
        for (auto& lane : s->lanes)
            th.emplace_back([&, l = &lane] {
                double wit = 0, seal_t = 0;
                for (;;) {
                    const size_t i = next.fetch_add(1);
                    if (i >= n || errs.any()) break;
                    double w = 0;
                    const double ts = now_s();
                    if (errs.set(seal_one(s, *l, segs[i], &info->seals[i], &info->seal_words[i], &w), "segment")) break;
                    wit += w; seal_t += now_s() - ts - w;
                }
                (void)zkh_sync(l->ctx);
                std::lock_guard<std::mutex> lk(stat_lock);
                info->witgen_s_sum += wit; info->seal_s_sum += seal_t;
            });
        for (auto& t : th) t.join();
    }
    info->leaves_s = now_s() - t0;
    // ---- the join tree: level l pairs nodes (2k, 2k+1) of level l-1, an unpaired last node is carried up; the joins of a
    // level are independent and pulled from one index by the lanes.  A leaf's claim is zkh_receipt_claim (needs the leaf's
    // control root: computed per size by the first lane), a join's claim is the parent digest it constrains (out[0..8)). ----
    if (join_tree == 2 && !errs.any()) {
        // ---- lift every receipt, then join level by level: every node verifies its child seal(s) in-circuit ----
        struct Node { uint32_t* seal = nullptr; size_t words = 0; uint32_t po2 = 0, program = 0; };
        auto program_of = [&](uint32_t join, uint32_t a, uint32_t b) -> int {
            for (size_t i = 0; i < s->rec_kinds.size(); i++)
                if (s->rec_kinds[i].join == join && s->rec_kinds[i].a == a && s->rec_kinds[i].b == b) return (int)i;     // lifts: b = circuit family, 0 = the session's
            return -1;
        };
        auto po2_of = [&](uint32_t program) { uint32_t inf[8]; (void)zkh_rec_program_info(s->lanes[0].programs[program], nullptr, inf); return inf[0]; };
        auto path_of = [&](uint32_t program, std::vector<uint32_t>& out) {      // per level: the direction bit as an element, the sibling
            size_t idx = program;
            for (size_t l = 0; l < REC_DEPTH; l++) {
                out.push_back(fp_encode((uint32_t)(idx & 1)).v);
                out.insert(out.end(), s->allowed[l][idx ^ 1].begin(), s->allowed[l][idx ^ 1].end());
                idx >>= 1;
            }
        };
        const std::vector<uint32_t>& A = s->allowed.back()[0];
        // bottom level: a pair of segments is ONE proof where the program set has lift2(po2_l, po2_r) (lift + lift + join fused),
        // else lift, lift (and the pair is joined with the level above); an unpaired last segment is lifted
        const size_t n_pairs = n / 2;
        std::vector<char> fused(n_pairs, 0);
        size_t n_fused = 0;
        for (size_t k = 0; k < n_pairs; k++)
            if (program_of(2, segs[2 * k].po2, segs[2 * k + 1].po2) >= 0) { fused[k] = 1; n_fused++; }
        const bool all_fused = n_fused == n_pairs && n > 1;
        std::vector<Node> level(all_fused ? n_pairs + n % 2 : n);
        const double tl = now_s();
        {
            // jobs: all pairs fused -> one job per pair (+ the odd tail); otherwise one lift per segment
            const size_t n_jobs = level.size();
            std::atomic<size_t> idx{0};
            std::vector<std::thread> th;
            for (Lane* lane : s->rec_lanes)
                th.emplace_back([&, l = lane] {
                    std::vector<uint32_t> in;
                    for (;;) {
                        const size_t k = idx.fetch_add(1);
                        if (k >= n_jobs || errs.any()) break;
                        int p;
                        if (all_fused && k < n_pairs) {
                            p = program_of(2, segs[2 * k].po2, segs[2 * k + 1].po2);
                            in.assign(info->seals[2 * k], info->seals[2 * k] + info->seal_words[2 * k]);
                            in.insert(in.end(), info->seals[2 * k + 1], info->seals[2 * k + 1] + info->seal_words[2 * k + 1]);
                        } else {
                            const size_t i = all_fused ? n - 1 : k;
                            p = program_of(0, segs[i].po2, 0);
                            if (p < 0) { errs.set(make_err("no lift program for po2-%u segments", segs[i].po2), "lift"); break; }
                            in.assign(info->seals[i], info->seals[i] + info->seal_words[i]);
                        }
                        in.insert(in.end(), A.begin(), A.end());
                        Node& nd = level[k];
                        nd.program = (uint32_t)p; nd.po2 = po2_of((uint32_t)p);
                        if (errs.set(zkh_rec_prove(l->programs[p], in.data(), in.size(), join_noise_seed ? join_noise_seed : os_random64(), nullptr, &nd.seal, &nd.words), "lift")) break;
                    }
                    (void)zkh_sync(l->ctx);
                });
            for (auto& t : th) t.join();
        }
        info->n_joins = 0;
        info->n_lifts = all_fused ? n_pairs + n % 2 : n;      // proofs of the bottom level (lift2 counts once)
        info->lift_s = now_s() - tl;
        const double tj = now_s();
        while (level.size() > 1 && !errs.any()) {
            const size_t pairs = level.size() / 2;
            std::vector<Node> up(pairs);
            std::atomic<size_t> idx{0};
            std::vector<std::thread> th;
            for (Lane* lane : s->rec_lanes)
                th.emplace_back([&, l = lane] {
                    std::vector<uint32_t> in;
                    for (;;) {
                        const size_t k = idx.fetch_add(1);
                        if (k >= pairs || errs.any()) break;
                        const Node &a = level[2 * k], &b = level[2 * k + 1];
                        const int p = program_of(1, a.po2, b.po2);
                        if (p < 0) { errs.set(make_err("no join program for children of po2 %u and %u", a.po2, b.po2), "join"); break; }
                        in.assign(a.seal, a.seal + a.words);
                        path_of(a.program, in);
                        in.insert(in.end(), b.seal, b.seal + b.words);
                        path_of(b.program, in);
                        Node& nd = up[k];
                        nd.program = (uint32_t)p; nd.po2 = po2_of((uint32_t)p);
                        if (errs.set(zkh_rec_prove(l->programs[p], in.data(), in.size(), join_noise_seed ? join_noise_seed : os_random64(), nullptr, &nd.seal, &nd.words), "join")) break;
                    }
                    (void)zkh_sync(l->ctx);
                });
            for (auto& t : th) t.join();
            info->n_joins += pairs;
            for (size_t k = 0; k < 2 * pairs; k++) zkh_free_seal(level[k].seal);          // children are not kept
            if (level.size() % 2) up.push_back(level.back());
            level.swap(up);
        }
        info->join_s = now_s() - tj;
        if (!errs.any()) { info->root_seal = level[0].seal; info->root_seal_words = level[0].words; info->root_program = level[0].program; level[0].seal = nullptr; }
        for (auto& nd : level) zkh_free_seal(nd.seal);
    }
    if (join_tree == 1 && n > 1 && !errs.any()) {
        const double tj = now_s();
        std::vector<std::vector<uint32_t>> claims(n, std::vector<uint32_t>(8));
        std::vector<std::pair<uint32_t, std::vector<uint32_t>>> roots;          // (po2, control root) of the leaf circuit
        for (size_t i = 0; i < n && !errs.any(); i++) {
            const uint32_t po2 = segs[i].po2;
            const std::vector<uint32_t>* root = nullptr;
            for (auto& r : roots) if (r.first == po2) root = &r.second;
            if (!root) {
                std::vector<uint32_t> cr(8);
                if (errs.set(leaf_control_root(s, segs[i], cr.data()), "control root")) break;
                roots.emplace_back(po2, cr);
                root = &roots.back().second;
            }
            errs.set(zkh_receipt_claim(s->lanes[0].circuit, info->seals[i], info->seal_words[i], root->data(), nullptr, nullptr, claims[i].data()), "receipt_claim");
        }
        while (claims.size() > 1 && !errs.any()) {
            const size_t pairs = claims.size() / 2;
            std::vector<std::vector<uint32_t>> up(pairs, std::vector<uint32_t>(8));
            std::vector<uint32_t*> seals(pairs, nullptr);
            std::vector<size_t> words(pairs, 0);
            std::atomic<size_t> idx{0};
            std::vector<std::thread> th;
            for (auto& lane : s->lanes)
                th.emplace_back([&, l = &lane] {
                    const zkh_circuit* jc = l->join_circuit;
                    const size_t jn = (size_t)1 << join_po2;
                    Tmp code, data;
                    if (errs.set(zkh_alloc(l->ctx, "code", (size_t)jc->group_size[GROUP_CODE] * jn, 0, code.out()), "join alloc") ||
                        errs.set(zkh_alloc(l->ctx, "data", (size_t)jc->group_size[GROUP_DATA] * jn, 0, data.out()), "join alloc")) return;
                    uint32_t pub[16], outg[24];
                    for (;;) {
                        const size_t k = idx.fetch_add(1);
                        if (k >= pairs || errs.any()) break;
                        memcpy(pub, claims[2 * k].data(), 32);
                        memcpy(pub + 8, claims[2 * k + 1].data(), 32);
                        const uint64_t noise = join_noise_seed ? join_noise_seed : os_random64();
                        if (errs.set(zkh_syn_witgen(l->ctx, jc, join_po2, ZKH_ZK_CYCLES, 0, noise, pub, code, data, outg), "join witgen") ||
                            errs.set(zkh_prove_segment(l->join_prover, join_po2, ZKH_ZK_CYCLES, noise, code, data, outg, &seals[k], &words[k]), "join seal")) break;
                        memcpy(up[k].data(), seals[k], 32);
                    }
                    (void)zkh_sync(l->ctx);
                });
            for (auto& t : th) t.join();
            if (!errs.any()) {
                info->n_joins += pairs;
                if (claims.size() == 2) { info->root_seal = seals[0]; info->root_seal_words = words[0]; seals[0] = nullptr; }
            }
            for (auto p : seals) zkh_free_seal(p);            // joins below the root are not kept: the verifier does not need them
            if (claims.size() % 2) up.push_back(claims.back());
            claims.swap(up);
        }
        info->join_s = now_s() - tj;
    }
    info->wall_s = now_s() - t0;
    if (errs.any()) {
        zkh_prove_info_free(info);
        return make_err("session_prove: %s", errs.first.c_str());
    }
    return nullptr;
}

// receipt.verify for what zkh_session_prove returned: every leaf seal against the control root of its size, and — when a root
// receipt is present — the root seal against the join circuit's control root plus the claim tree recomputed on the host.
// Host arithmetic only, except that the control roots of the built-in circuits are computed on lane 0 (a deployment ships them).
extern "C" const char* zkh_session_verify(zkh_session* s, const zkh_segment* segs, const zkh_prove_info* info, size_t join_po2) {
    ZKH_REQUIRE(s && segs && info && info->n_segments, "session_verify: bad argument");
    zkh_circuit* hc = nullptr;
    ZKH_TRY(zkh_circuit_load(nullptr, s->desc.data(), s->desc.size(), &hc));
    std::unique_ptr<zkh_circuit, void (*)(zkh_circuit*)> hold(hc, zkh_circuit_destroy);
    std::vector<std::pair<uint32_t, std::vector<uint32_t>>> roots;
    std::vector<std::vector<uint32_t>> claims(info->n_segments, std::vector<uint32_t>(8));
    for (size_t i = 0; i < info->n_segments; i++) {
        const uint32_t po2 = segs[i].po2;
        const std::vector<uint32_t>* root = nullptr;
        for (auto& r : roots) if (r.first == po2) root = &r.second;
        if (!root) {
            std::vector<uint32_t> cr(8);
            ZKH_TRY(leaf_control_root(s, segs[i], cr.data()));
            roots.emplace_back(po2, cr);
            root = &roots.back().second;
        }
        if (const char* e = zkh_verify_segment(hc, info->seals[i], info->seal_words[i], root->data(), nullptr, nullptr)) {
            const char* out = make_err("session_verify: segment %zu: %s", i, e);
            zkh_free_error(e);
            return out;
        }
        ZKH_TRY(zkh_receipt_claim(hc, info->seals[i], info->seal_words[i], root->data(), nullptr, nullptr, claims[i].data()));
    }
    if (!info->root_seal) return nullptr;
    if (info->n_lifts) {
        // a RECURSION root: ONE seal under a program of the allowed set, out = claim tree root ‖ allowed-programs root
        ZKH_REQUIRE(!s->rec_kinds.empty() && info->root_program < s->rec_roots.size(), "session_verify: a recursion root without the session's programs");
        zkh_circuit* rc = nullptr;
        ZKH_TRY(zkh_circuit_load(nullptr, s->rec_desc.data(), s->rec_desc.size(), &rc));
        std::unique_ptr<zkh_circuit, void (*)(zkh_circuit*)> rhold(rc, zkh_circuit_destroy);
        if (const char* e = zkh_verify_segment(rc, info->root_seal, info->root_seal_words, s->rec_roots[info->root_program].data(), nullptr, nullptr)) {
            const char* out = make_err("session_verify: root receipt: %s", e);
            zkh_free_error(e);
            return out;
        }
        while (claims.size() > 1) {
            std::vector<std::vector<uint32_t>> up(claims.size() / 2, std::vector<uint32_t>(8));
            for (size_t k = 0; k < up.size(); k++) ZKH_TRY(hash_pair_host(claims[2 * k].data(), claims[2 * k + 1].data(), up[k].data()));
            if (claims.size() % 2) up.push_back(claims.back());
            claims.swap(up);
        }
        ZKH_REQUIRE(info->root_seal_words > 16 && memcmp(info->root_seal, claims[0].data(), 32) == 0,
                    "session_verify: the root receipt does not commit to the claim tree of these segments");
        ZKH_REQUIRE(memcmp(info->root_seal + 8, s->allowed.back()[0].data(), 32) == 0, "session_verify: the root receipt was produced under another allowed-programs root");
        return nullptr;
    }
    ZKH_REQUIRE(!s->join_desc.empty() && info->n_segments > 1, "session_verify: a root receipt without a join circuit");
    zkh_circuit* jc = nullptr;
    ZKH_TRY(zkh_circuit_load(nullptr, s->join_desc.data(), s->join_desc.size(), &jc));
    std::unique_ptr<zkh_circuit, void (*)(zkh_circuit*)> jhold(jc, zkh_circuit_destroy);
    uint32_t jroot[8];
    ZKH_TRY(zkh_syn_control_root(s->lanes[0].join_prover, join_po2, ZKH_ZK_CYCLES, jroot));
    if (const char* e = zkh_verify_segment(jc, info->root_seal, info->root_seal_words, jroot, nullptr, nullptr)) {
        const char* out = make_err("session_verify: root receipt: %s", e);
        zkh_free_error(e);
        return out;
    }
    while (claims.size() > 2) {
        std::vector<std::vector<uint32_t>> up(claims.size() / 2, std::vector<uint32_t>(8));
        for (size_t k = 0; k < up.size(); k++) {
            uint32_t st[24] = {0};
            memcpy(st, claims[2 * k].data(), 32); memcpy(st + 8, claims[2 * k + 1].data(), 32);
            ZKH_TRY(zkh_poseidon2_mix_host(nullptr, nullptr, st, 1));
            memcpy(up[k].data(), st, 32);
        }
        if (claims.size() % 2) up.push_back(claims.back());
        claims.swap(up);
    }
    ZKH_REQUIRE(info->root_seal_words > 24 && memcmp(info->root_seal + 8, claims[0].data(), 32) == 0 && memcmp(info->root_seal + 16, claims[1].data(), 32) == 0,
                "session_verify: the root receipt does not commit to the claim tree of these segments");
    return nullptr;
}
