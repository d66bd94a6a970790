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
#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <sys/random.h>

#include "circuit.h"
#include "scheduler.h"

using namespace zkh;

namespace {

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Lane {
    int device = 0;
    zkh_ctx* ctx = nullptr;
    zkh_circuit *circuit = nullptr, *join_circuit = nullptr, *rec_circuit = nullptr;
    zkh_prover *prover = nullptr, *join_prover = nullptr;
    std::vector<zkh_rec_program*> programs;
    std::vector<uint32_t> resident_po2;       // sizes whose committed code group this lane's prover keeps in HBM
    // witness source 1 (host preflight): pinned record slots of this lane (zkh_host_alloc), 4 x 2^po2 words each, recycled
    std::vector<uint32_t*> record_slots;
    size_t record_slot_words = 0;
    void close() {
        for (auto p : record_slots) if (ctx) zkh_host_free(ctx, p);
        record_slots.clear(); record_slot_words = 0;
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
    bool resident_code = true;           // built-in circuits: the committed code group of each size stays in HBM per lane
    bool streamed_fold = true;           // join_tree 2: lift2 / join a node the moment its children exist, concurrently with the sealing lanes
    bool chained = false;                // SYN-C sessions: segment i's pre-state = initial + contributions of segments 0 .. i-1 (continuity)
    uint32_t initial_state = 0;          // ... as a canonical residue
    bool has_journal = false;            // SYN-S: the bytes the guest commits (zkh_session_set_journal); unset: the final state word
    std::vector<uint8_t> journal;
    const uint8_t* journal_ptr() const {     // non-NULL also for an EMPTY journal (NULL means "the final state word" downstream)
        static const uint8_t none = 0;
        return !has_journal ? nullptr : journal.empty() ? &none : journal.data();
    }
    int witness_source = 0;              // 0: closed-form generators on the device; 1: sequential host preflight -> compact records -> row fill
    size_t producers_per_lane = 2;       // ... host threads per sealing lane that run the preflight ahead of the seals
    // assumption receipts of the session (zkh_session_set_assumptions): seals of ANOTHER circuit (keccak batches) proven beforehand;
    // join_tree 2 lifts them (family 1), unites them pairwise and resolves the session's root against the union root
    std::vector<uint32_t> assum_desc;
    std::vector<std::vector<uint32_t>> assum_seals;
    std::vector<uint32_t> assum_po2;
    std::vector<std::vector<uint32_t>> assum_roots;    // the assumption circuit's control root at assum_po2[i]
    std::vector<uint32_t> assum_claims;                // 8 words each: the receipts' claim digests (zkh_receipt_claim), in the caller's order
    // the digest a chained SYN-S session's last seal binds beside its journal: Assumptions([Assumption{claim, control root}, ..]) of the
    // receipts above, NULL when the session assumes nothing (verifier.hip assumptions_digest)
    const uint32_t* assumptions_digest(uint32_t out[8]) const {
        if (assum_seals.empty()) return nullptr;
        std::vector<uint32_t> roots;
        for (const auto& r : assum_roots) roots.insert(roots.end(), r.begin(), r.end());
        zkh::assumptions_digest(assum_claims.data(), roots.data(), assum_seals.size(), out);
        return out;
    }
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
extern "C" void zkh_session_set_resident_code(zkh_session* s, int on) {
    if (!s) return;
    s->resident_code = on != 0;
    if (!on) for (auto& l : s->lanes) { if (l.prover) zkh_prover_drop_code_cache(l.prover); l.resident_po2.clear(); }
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

// claim' = hash_pair(core, (pre, post, 0, 0, 0, 0, 0, 0)): what every recursion receipt publishes (circuits/rec_verify.py _wrap) — its
// core claim bound to the state range [pre, post] it covers.  A join opens both children's claim' in-circuit and asserts
// post(left) = pre(right); the opening (core, pre, post) travels next to a receipt as witness for its parent, never trusted.
struct NodeClaim { uint32_t core[8] = {0}; uint32_t pre = 0, post = 0; };
static const char* wrap_claim(const NodeClaim& c, uint32_t out[8]) {
    const uint32_t st[8] = {c.pre, c.post, 0, 0, 0, 0, 0, 0};
    return hash_pair_host(c.core, st, out);
}
// (receipt claim, pre, post) of a segment seal: SYN-C circuits (kind 1, five outputs) carry their state in out[4] / out[0]
static const char* leaf_claim(const zkh_circuit* c, const uint32_t* seal, size_t words, const uint32_t* control_root, NodeClaim* out) {
    ZKH_TRY(zkh_receipt_claim(c, seal, words, control_root, nullptr, nullptr, out->core));
    const bool chained = circuit_has_state(c);
    out->pre = chained ? seal[4] : 0; out->post = chained ? seal[0] : 0;
    return nullptr;
}
// parent of two nodes: core = hash_pair(claim'_l, claim'_r), state range = [pre_l, post_r]; the children must chain
static const char* parent_claim(const NodeClaim& l, const NodeClaim& r, NodeClaim* out) {
    ZKH_REQUIRE(l.post == r.pre, "claim tree: two neighbouring nodes do not chain (post(l) != pre(r)): no join has a witness for them");
    uint32_t cl[8], cr[8];
    ZKH_TRY(wrap_claim(l, cl)); ZKH_TRY(wrap_claim(r, cr));
    ZKH_TRY(hash_pair_host(cl, cr, out->core));
    out->pre = l.pre; out->post = r.post;
    return nullptr;
}

// the fold plan's shape over claim nodes (zeth_amd/recursion.py fold_plan): the first level pairs, every level above takes three at
// a time (a group of three is join(join(a, b), c)), a remainder of two is a join, of one moves up
static const char* fold_claim_nodes(std::vector<NodeClaim> nodes, NodeClaim* root) {
    ZKH_REQUIRE(!nodes.empty(), "claim tree: no leaves");
    for (size_t group = 2; nodes.size() > 1; group = 3) {
        std::vector<NodeClaim> up;
        size_t k = 0;
        for (; k + group <= nodes.size(); k += group) {
            NodeClaim nd;
            ZKH_TRY(parent_claim(nodes[k], nodes[k + 1], &nd));
            if (group == 3) { const NodeClaim ab = nd; ZKH_TRY(parent_claim(ab, nodes[k + 2], &nd)); }
            up.push_back(nd);
        }
        if (nodes.size() - k == 2) { NodeClaim nd; ZKH_TRY(parent_claim(nodes[k], nodes[k + 1], &nd)); up.push_back(nd); }
        else if (nodes.size() - k == 1) up.push_back(nodes[k]);
        nodes.swap(up);
    }
    *root = nodes[0];
    return nullptr;
}

// `receipt.verify` for a SUCCINCT receipt, on the host alone (no GPU, no session): what zeth_amd/recursion.py RecReceipt.verify does.
// union(l, r): wrap(hash_pair(lo, hi), 0, 0) with (lo, hi) the two claim' sorted by their canonical words, lexicographically
// (zeth_amd/recursion.py union_node; ProverServer::union, risc0-zkvm 3.0.3: the UnionClaim keeps left <= right)
static const char* union_claim(const uint32_t* l, const uint32_t* r, uint32_t out[8]) {
    bool swap = false;
    for (int i = 0; i < 8; i++) {
        const uint32_t a = fp_decode(Fp::raw(l[i])), b = fp_decode(Fp::raw(r[i]));
        if (a != b) { swap = b < a; break; }
    }
    NodeClaim nd;
    ZKH_TRY(hash_pair_host(swap ? r : l, swap ? l : r, nd.core));
    return wrap_claim(nd, out);
}
// the union tree over assumption claim' (zeth_amd/recursion.py union_claims): neighbours pairwise, level by level, an odd one moves up
static const char* union_tree(std::vector<std::array<uint32_t, 8>> level, uint32_t out[8]) {
    ZKH_REQUIRE(!level.empty(), "union tree: no assumption claims");
    while (level.size() > 1) {
        std::vector<std::array<uint32_t, 8>> up;
        for (size_t k = 0; k + 1 < level.size(); k += 2) {
            std::array<uint32_t, 8> u;
            ZKH_TRY(union_claim(level[k].data(), level[k + 1].data(), u.data()));
            up.push_back(u);
        }
        if (level.size() % 2) up.push_back(level.back());
        level.swap(up);
    }
    memcpy(out, level[0].data(), 32);
    return nullptr;
}

static const char* succinct_verify_impl(const uint32_t* root_seal, size_t root_words, const uint32_t* allowed_roots, size_t n_allowed,
                                        size_t root_program, const uint32_t* leaves, size_t n_leaves, size_t ranks,
                                        const uint32_t* assumptions, size_t n_assumptions) {
    ZKH_REQUIRE(root_seal && allowed_roots && (leaves || !n_leaves) && (assumptions || !n_assumptions), "succinct_verify: null argument");
    ZKH_REQUIRE(n_allowed >= 1 && n_allowed <= REC_ALLOWED && root_program < n_allowed, "succinct_verify: the receipt's program is not in the allowed set");
    ZKH_REQUIRE(n_leaves + n_assumptions >= 1, "succinct_verify: neither leaves nor assumptions");
    ZKH_REQUIRE(ranks >= 1 && n_leaves % ranks == 0 && (n_leaves || ranks == 1), "succinct_verify: %zu leaves do not split into %zu equal ranges", n_leaves, ranks);
    ZKH_REQUIRE(root_words > 16, "succinct_verify: not a recursion seal");
    // 1) ONE seal, under the control root of an allowed program
    const uint32_t* rdesc = nullptr;
    size_t rdesc_words = 0;
    ZKH_TRY(zkh_shipped_circuit_desc("recursion", &rdesc, &rdesc_words));
    zkh_circuit* rc = nullptr;
    ZKH_TRY(zkh_circuit_load(nullptr, rdesc, rdesc_words, &rc));
    std::unique_ptr<zkh_circuit, void (*)(zkh_circuit*)> hold(rc, zkh_circuit_destroy);
    if (const char* e = zkh_verify_segment(rc, root_seal, root_words, allowed_roots + 8 * root_program, nullptr, nullptr)) {
        const char* out = make_err("succinct_verify: root receipt: %s", e);
        zkh_free_error(e);
        return out;
    }
    // 2) the allowed-programs root the receipt carries is the root of THIS set (16 leaves, zero padded)
    std::vector<std::vector<uint32_t>> level(REC_ALLOWED, std::vector<uint32_t>(8, 0));
    for (size_t i = 0; i < n_allowed; i++) level[i].assign(allowed_roots + 8 * i, allowed_roots + 8 * i + 8);
    while (level.size() > 1) {
        std::vector<std::vector<uint32_t>> up(level.size() / 2, std::vector<uint32_t>(8));
        for (size_t k = 0; k < up.size(); k++) ZKH_TRY(hash_pair_host(level[2 * k].data(), level[2 * k + 1].data(), up[k].data()));
        level.swap(up);
    }
    ZKH_REQUIRE(memcmp(root_seal + 8, level[0].data(), 32) == 0, "succinct_verify: the receipt was produced under another allowed-programs root");
    // 3) its claim is the root of the leaves' claim tree: `ranks` contiguous equal ranges folded on their own, their roots folded again
    std::vector<NodeClaim> nodes(n_leaves);
    for (size_t i = 0; i < n_leaves; i++) { memcpy(nodes[i].core, leaves + 10 * i, 32); nodes[i].pre = leaves[10 * i + 8]; nodes[i].post = leaves[10 * i + 9]; }
    if (ranks > 1) {
        const size_t per = n_leaves / ranks;
        std::vector<NodeClaim> tops(ranks);
        for (size_t r = 0; r < ranks; r++) ZKH_TRY(fold_claim_nodes(std::vector<NodeClaim>(nodes.begin() + r * per, nodes.begin() + (r + 1) * per), &tops[r]));
        nodes.swap(tops);
    }
    uint32_t want[8], assumed[8];
    if (n_assumptions) {                  // a lifted assumption receipt publishes wrap(receipt claim, 0, 0)
        std::vector<std::array<uint32_t, 8>> lifted(n_assumptions);
        for (size_t i = 0; i < n_assumptions; i++) {
            NodeClaim nd;
            memcpy(nd.core, assumptions + 8 * i, 32);
            ZKH_TRY(wrap_claim(nd, lifted[i].data()));
        }
        ZKH_TRY(union_tree(lifted, assumed));
    }
    if (n_leaves) {
        NodeClaim top;
        ZKH_TRY(fold_claim_nodes(nodes, &top));
        ZKH_TRY(wrap_claim(top, want));
        if (n_assumptions) {              // resolve(cond, assum): the session's state range over hash_pair(claim'_cond, claim'_assum)
            NodeClaim res;
            ZKH_TRY(hash_pair_host(want, assumed, res.core));
            res.pre = top.pre; res.post = top.post;
            ZKH_TRY(wrap_claim(res, want));
        }
    } else {
        memcpy(want, assumed, 32);
    }
    ZKH_REQUIRE(memcmp(root_seal, want, 32) == 0, n_assumptions ? "succinct_verify: the receipt's claim is not the resolved root of the leaves' claim tree and the assumptions' union tree"
                                                                : "succinct_verify: the receipt's claim is not the root of the leaves' claim tree");
    return nullptr;
}
extern "C" const char* zkh_succinct_verify(const uint32_t* root_seal, size_t root_words, const uint32_t* allowed_roots, size_t n_allowed,
                                           size_t root_program, const uint32_t* leaves, size_t n_leaves, size_t ranks) {
    ZKH_REQUIRE(leaves && n_leaves >= 1, "succinct_verify: null argument");
    return succinct_verify_impl(root_seal, root_words, allowed_roots, n_allowed, root_program, leaves, n_leaves, ranks, nullptr, 0);
}
extern "C" const char* zkh_succinct_verify_resolved(const uint32_t* root_seal, size_t root_words, const uint32_t* allowed_roots, size_t n_allowed,
                                                    size_t root_program, const uint32_t* leaves, size_t n_leaves, size_t ranks,
                                                    const uint32_t* assumption_claims, size_t n_assumptions) {
    ZKH_REQUIRE(n_assumptions >= 1, "succinct_verify_resolved: no assumption claims (zkh_succinct_verify checks a receipt without assumptions)");
    return succinct_verify_impl(root_seal, root_words, allowed_roots, n_allowed, root_program, leaves, n_leaves, ranks, assumption_claims, n_assumptions);
}

extern "C" const char* zkh_session_set_recursion(zkh_session* s, const uint32_t* rec_desc, size_t rec_desc_words, const uint32_t* const* blobs,
                                                 const size_t* words, const uint32_t* kinds, size_t n_programs) {
    ZKH_REQUIRE(s && rec_desc && rec_desc_words >= 16 && blobs && words && kinds && n_programs && n_programs <= REC_ALLOWED, "session_set_recursion: bad argument");
    ZKH_REQUIRE(s->rec_kinds.empty() && s->rec_lanes.empty() && s->fold_lanes.empty(), "session_set_recursion: the session already has its programs");
    // Everything is built in locals and committed to the session only when every context, circuit and program has loaded: a
    // failure half way leaves the session exactly as it was (no lane holds a program, no fold lane exists, a retry starts clean).
    std::vector<Lane> fold_lanes;
    struct Loaded { zkh_circuit* circuit = nullptr; std::vector<zkh_rec_program*> programs; };
    std::vector<Loaded> loaded;
    auto undo = [&] {
        for (auto& ld : loaded) { for (auto p : ld.programs) zkh_rec_program_destroy(p); if (ld.circuit) zkh_circuit_destroy(ld.circuit); }
        for (auto& l : fold_lanes) l.close();
    };
#define REC_TRY(expr) do { const char* _e = (expr); if (_e) { undo(); return _e; } } while (0)
    // a lift's / join's witness schedule is a chain of ~300 dependency levels (latency): the fold runs on ZKH_FOLD_LANES lanes
    // per device (default 6: the measured knee, profiles/r03_recursion_fold_lanes.txt), the sealing lanes plus extra contexts
    const char* env = getenv("ZKH_FOLD_LANES");
    const size_t want = env ? (size_t)strtoul(env, nullptr, 10) : 6;
    const size_t n_devices = s->lanes.size() / s->lanes_per_device;
    if (want > s->lanes_per_device) {
        fold_lanes.resize(n_devices * (want - s->lanes_per_device));
        for (size_t i = 0; i < fold_lanes.size(); i++) {
            Lane& l = fold_lanes[i];
            l.device = s->lanes[(i / (want - s->lanes_per_device)) * s->lanes_per_device].device;
            REC_TRY(zkh_ctx_create(l.device, "poseidon2", &l.ctx));
        }
    }
    std::vector<Lane*> rec_lanes;
    for (auto& l : s->lanes) rec_lanes.push_back(&l);
    for (auto& l : fold_lanes) rec_lanes.push_back(&l);
    loaded.resize(rec_lanes.size());
    for (size_t k = 0; k < rec_lanes.size(); k++) {
        REC_TRY(zkh_circuit_load(rec_lanes[k]->ctx, rec_desc, rec_desc_words, &loaded[k].circuit));
        for (size_t i = 0; i < n_programs; i++) {
            zkh_rec_program* p = nullptr;
            REC_TRY(zkh_rec_program_load(rec_lanes[k]->ctx, loaded[k].circuit, blobs[i], words[i], &p));
            loaded[k].programs.push_back(p);
        }
    }
    std::vector<RecKind> rec_kinds;
    std::vector<std::vector<uint32_t>> rec_roots;
    for (size_t i = 0; i < n_programs; i++) {
        rec_kinds.push_back(RecKind{kinds[3 * i], kinds[3 * i + 1], kinds[3 * i + 2]});
        std::vector<uint32_t> root(8);
        REC_TRY(zkh_rec_program_info(loaded[0].programs[i], root.data(), nullptr));
        rec_roots.push_back(root);
    }
    std::vector<std::vector<uint32_t>> level(rec_roots);
    level.resize(REC_ALLOWED, std::vector<uint32_t>(8, 0));
    std::vector<std::vector<std::vector<uint32_t>>> allowed(1, level);
    while (level.size() > 1) {
        std::vector<std::vector<uint32_t>> up(level.size() / 2, std::vector<uint32_t>(8));
        for (size_t k = 0; k < up.size(); k++) REC_TRY(hash_pair_host(level[2 * k].data(), level[2 * k + 1].data(), up[k].data()));
        allowed.push_back(up);
        level.swap(up);
    }
#undef REC_TRY
    // ---- commit (nothing below can fail) ----
    s->rec_desc.assign(rec_desc, rec_desc + rec_desc_words);
    s->fold_lanes = std::move(fold_lanes);
    s->rec_lanes.clear();
    for (auto& l : s->lanes) s->rec_lanes.push_back(&l);
    for (auto& l : s->fold_lanes) s->rec_lanes.push_back(&l);
    for (size_t k = 0; k < s->rec_lanes.size(); k++) { s->rec_lanes[k]->rec_circuit = loaded[k].circuit; s->rec_lanes[k]->programs = std::move(loaded[k].programs); }
    s->rec_kinds = std::move(rec_kinds);
    s->rec_roots = std::move(rec_roots);
    s->allowed = std::move(allowed);
    return nullptr;
}
extern "C" const char* zkh_session_set_assumptions(zkh_session* s, const uint32_t* desc, size_t desc_words, const uint32_t* const* seals,
                                                   const size_t* seal_words, const uint32_t* po2s, const uint32_t* control_roots, size_t n) {
    ZKH_REQUIRE(s, "session_set_assumptions: null session");
    if (!n) { s->assum_desc.clear(); s->assum_seals.clear(); s->assum_po2.clear(); s->assum_roots.clear(); s->assum_claims.clear(); return nullptr; }
    ZKH_REQUIRE(desc && desc_words >= 16 && seals && seal_words && po2s && control_roots, "session_set_assumptions: null argument");
    // every receipt is verified HERE, on the host, against the control root it is handed with: what the lifts then prove in-circuit
    zkh_circuit* hc = nullptr;
    ZKH_TRY(zkh_circuit_load(nullptr, desc, desc_words, &hc));
    std::unique_ptr<zkh_circuit, void (*)(zkh_circuit*)> hold(hc, zkh_circuit_destroy);
    ZKH_REQUIRE(!circuit_has_state(hc), "session_set_assumptions: an assumption circuit has no state words (its claim' is wrap(claim, 0, 0))");
    std::vector<uint32_t> claims(8 * n);
    for (size_t i = 0; i < n; i++) {
        ZKH_REQUIRE(seals[i] && po2s[i] >= 4 && po2s[i] <= 24, "session_set_assumptions: assumption %zu: bad seal / po2", i);
        if (const char* e = zkh_verify_segment(hc, seals[i], seal_words[i], control_roots + 8 * i, nullptr, nullptr)) {
            const char* out = make_err("session_set_assumptions: assumption receipt %zu: %s", i, e);
            zkh_free_error(e);
            return out;
        }
        ZKH_TRY(zkh_receipt_claim(hc, seals[i], seal_words[i], control_roots + 8 * i, nullptr, nullptr, &claims[8 * i]));
    }
    s->assum_claims = std::move(claims);
    s->assum_desc.assign(desc, desc + desc_words);
    s->assum_seals.clear(); s->assum_po2.assign(po2s, po2s + n); s->assum_roots.clear();
    for (size_t i = 0; i < n; i++) {
        s->assum_seals.emplace_back(seals[i], seals[i] + seal_words[i]);
        s->assum_roots.emplace_back(control_roots + 8 * i, control_roots + 8 * i + 8);
    }
    return nullptr;
}
extern "C" void zkh_session_set_streamed_fold(zkh_session* s, int on) { if (s) s->streamed_fold = on != 0; }
extern "C" const char* zkh_session_set_chained(zkh_session* s, int on, uint32_t initial_state) {
    ZKH_REQUIRE(s, "session_set_chained: null session");
    ZKH_REQUIRE(!on || circuit_has_state(s->lanes[0].circuit),
                "session_set_chained: continuity needs a SYN-C or SYN-S circuit (kind 1 whose first public input is the pre-state)");
    ZKH_REQUIRE(initial_state < P, "session_set_chained: the initial state is not a reduced element");
    s->chained = on != 0; s->initial_state = initial_state;
    return nullptr;
}
extern "C" const char* zkh_session_set_journal(zkh_session* s, const uint8_t* journal, size_t journal_len) {
    ZKH_REQUIRE(s, "session_set_journal: null session");
    ZKH_REQUIRE(!journal || circuit_is_session(s->lanes[0].circuit), "session_set_journal: only a SYN-S circuit binds an output digest");
    ZKH_REQUIRE(journal || !journal_len, "session_set_journal: a length without bytes");
    s->has_journal = journal != nullptr;
    s->journal.assign(journal, journal + (journal ? journal_len : 0));
    return nullptr;
}
extern "C" const char* zkh_session_set_witness_source(zkh_session* s, int source, size_t producers_per_lane) {
    ZKH_REQUIRE(s && (source == 0 || source == 1), "session_set_witness_source: source must be 0 (closed form on the device) or 1 (host preflight)");
    ZKH_REQUIRE(source == 0 || (s->lanes[0].circuit->kind == 1 && s->lanes[0].circuit->global_size[GLOBAL_OUT] == 4),
                "session_set_witness_source: the host preflight drives SYN-AIR circuits (kind 1) without public inputs only");
    s->witness_source = source;
    s->producers_per_lane = producers_per_lane ? producers_per_lane : 2;
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

// The program set of a block whose segments have the sizes `po2s` (largest first), built HERE (rec_builder.hip: no Python, no
// files) and loaded on every lane: a lift per size, a lift2 per pair (a >= b), joins for every pair of program sizes until the set
// closes, the join3 of the largest size if three such children fit that size again (with_join3) — the set and the ORDER of
// zeth_amd/recursion.py build_programs, hence the same allowed-programs root.  Built-in circuits (kinds 1..3): the control roots
// come from their own code generators.
extern "C" const char* zkh_session_build_recursion(zkh_session* s, const uint32_t* po2s, size_t n_po2s, int with_join3) {
    ZKH_REQUIRE(s && po2s && n_po2s >= 1 && n_po2s <= 4, "session_build_recursion: 1..4 segment sizes");
    ZKH_REQUIRE(s->lanes[0].circuit->kind >= 1 && s->lanes[0].circuit->kind <= 3, "session_build_recursion: the segment circuit has no built-in code generator "
                "(build the programs with zkh_rec_build_program from its control roots and use zkh_session_set_recursion)");
    for (size_t k = 1; k < n_po2s; k++) ZKH_REQUIRE(po2s[k] < po2s[k - 1], "session_build_recursion: sizes go largest first, each once");
    std::vector<std::vector<uint32_t>> roots(n_po2s, std::vector<uint32_t>(8));
    for (size_t k = 0; k < n_po2s; k++) {
        zkh_segment seg;
        memset(&seg, 0, sizeof seg);
        seg.po2 = po2s[k];
        ZKH_TRY(leaf_control_root(s, seg, roots[k].data()));
    }
    const uint32_t* rdesc = nullptr;
    size_t rdesc_words = 0;
    ZKH_TRY(zkh_shipped_circuit_desc("recursion", &rdesc, &rdesc_words));
    struct Built { uint32_t kind, a, b; uint32_t* blob; size_t words; };
    std::vector<Built> built;
    struct Free { std::vector<Built>& b; ~Free() { for (auto& x : b) zkh_free_seal(x.blob); } } free_blobs{built};
    std::vector<uint32_t> sizes;
    auto add = [&](uint32_t kind, const uint32_t* d, size_t dw, std::vector<uint32_t> ps, const uint32_t* rts, uint32_t ka, uint32_t kb, uint32_t only_po2 = 0) -> const char* {
        ps.resize(3, 0);
        uint32_t* blob = nullptr;
        size_t words = 0;
        ZKH_TRY(zkh_rec_build_program(kind, d, dw, ps.data(), rts, ZKH_ZK_CYCLES, &blob, &words));
        if (only_po2 && blob[2] != only_po2) { zkh_free_seal(blob); return nullptr; }
        built.push_back({kind, ka, kb, blob, words});
        if (std::find(sizes.begin(), sizes.end(), blob[2]) == sizes.end()) sizes.push_back(blob[2]);
        return nullptr;
    };
    const std::vector<uint32_t>& desc = s->desc;
    for (size_t k = 0; k < n_po2s; k++) ZKH_TRY(add(0, desc.data(), desc.size(), {po2s[k]}, roots[k].data(), po2s[k], 0));
    // the session's assumption receipts (zkh_session_set_assumptions before this call): a lift per size of THEIR circuit (family 1),
    // kept out of the join closure: they meet the session's tree in one resolve
    std::vector<uint32_t> leaf_sizes;                           // program sizes of the assumption lifts
    {
        std::vector<std::pair<uint32_t, const uint32_t*>> asz;  // distinct assumption po2s, largest first, with their control root
        for (size_t i = 0; i < s->assum_po2.size(); i++) {
            bool have = false;
            for (auto& a : asz) {
                if (a.first != s->assum_po2[i]) continue;
                have = true;
                ZKH_REQUIRE(memcmp(a.second, s->assum_roots[i].data(), 32) == 0, "session_build_recursion: two assumption receipts of po2 %u under different control roots", a.first);
            }
            if (!have) asz.emplace_back(s->assum_po2[i], s->assum_roots[i].data());
        }
        std::sort(asz.begin(), asz.end(), [](auto& x, auto& y) { return x.first > y.first; });
        const std::vector<uint32_t> session_sizes(sizes);
        for (auto& a : asz) {
            ZKH_TRY(add(0, s->assum_desc.data(), s->assum_desc.size(), {a.first}, a.second, a.first, 1));
            leaf_sizes.push_back(built.back().blob[2]);
        }
        sizes = session_sizes;                                  // (add() recorded the assumption lifts' sizes: not part of the join closure)
    }
    for (size_t i = 0; i < n_po2s; i++)
        for (size_t j = i; j < n_po2s; j++) {
            std::vector<uint32_t> two(roots[i]);
            two.insert(two.end(), roots[j].begin(), roots[j].end());
            ZKH_TRY(add(2, desc.data(), desc.size(), {po2s[i], po2s[j]}, two.data(), po2s[i], po2s[j]));
        }
    std::vector<std::pair<uint32_t, uint32_t>> done;
    for (bool more = true; more;) {
        more = false;
        std::vector<uint32_t> cur(sizes);
        std::sort(cur.begin(), cur.end());
        for (uint32_t a : cur)
            for (uint32_t b : cur) {
                if (std::find(done.begin(), done.end(), std::make_pair(a, b)) != done.end()) continue;
                done.push_back({a, b});
                ZKH_TRY(add(1, rdesc, rdesc_words, {a, b}, nullptr, a, b));
                more = true;
            }
    }
    if (with_join3 && built.size() < REC_ALLOWED) {
        const uint32_t m = *std::max_element(sizes.begin(), sizes.end());
        ZKH_TRY(add(3, rdesc, rdesc_words, {m, m, m}, nullptr, m, m, m));
    }
    if (!leaf_sizes.empty()) {
        // unions for the pairs the union tree can meet — two leaves, or a union result on the LEFT of anything (an odd node moves up at
        // the END of its level) — until the sizes close; then resolves, the largest session sizes first, as far as the set has room
        const std::vector<uint32_t> session_sizes(sizes);
        std::sort(leaf_sizes.begin(), leaf_sizes.end());
        leaf_sizes.erase(std::unique(leaf_sizes.begin(), leaf_sizes.end()), leaf_sizes.end());
        std::vector<uint32_t> usizes;
        std::vector<std::pair<uint32_t, uint32_t>> udone;
        auto in = [](const std::vector<uint32_t>& v, uint32_t x) { return std::find(v.begin(), v.end(), x) != v.end(); };
        for (bool more = true; more;) {
            more = false;
            std::vector<uint32_t> all(leaf_sizes);
            for (uint32_t u : usizes) if (!in(all, u)) all.push_back(u);
            std::sort(all.begin(), all.end());
            std::vector<std::pair<uint32_t, uint32_t>> todo;      // (decided on the sizes known when the round starts: build_programs' order)
            for (uint32_t a : all)
                for (uint32_t b : all)
                    if (std::find(udone.begin(), udone.end(), std::make_pair(a, b)) == udone.end() && (in(usizes, a) || (in(leaf_sizes, a) && in(leaf_sizes, b))))
                        todo.push_back({a, b});
            for (auto& ab : todo) {
                udone.push_back(ab);
                ZKH_TRY(add(4, rdesc, rdesc_words, {ab.first, ab.second}, nullptr, ab.first, ab.second));
                if (!in(usizes, built.back().blob[2])) usizes.push_back(built.back().blob[2]);
                more = true;
            }
        }
        std::vector<uint32_t> all(leaf_sizes);
        for (uint32_t u : usizes) if (!in(all, u)) all.push_back(u);
        std::sort(all.begin(), all.end());
        std::vector<uint32_t> ss(session_sizes);
        std::sort(ss.begin(), ss.end(), std::greater<uint32_t>());
        for (uint32_t a : ss)
            for (uint32_t b : all) {
                if (built.size() < REC_ALLOWED) ZKH_TRY(add(5, rdesc, rdesc_words, {a, b}, nullptr, a, b));
                else ZKH_REQUIRE(a != ss[0], "session_build_recursion: the allowed set (%zu programs) has no room for resolve(%u, %u): this circuit's lift / lift2 / join "
                                 "programs come in %zu sizes (their join closure alone is %zu programs)%s", REC_ALLOWED, a, b, ss.size(), ss.size() * ss.size(),
                                 with_join3 ? "; pass with_join3 = 0" : ": a circuit this narrow cannot resolve assumptions within one allowed set (the BASELINE widths give two sizes)");
            }
    }
    ZKH_REQUIRE(built.size() <= REC_ALLOWED, "session_build_recursion: %zu programs do not fit the allowed set", built.size());
    std::vector<const uint32_t*> ptrs;
    std::vector<size_t> words;
    std::vector<uint32_t> kinds;
    for (auto& b : built) { ptrs.push_back(b.blob); words.push_back(b.words); kinds.insert(kinds.end(), {b.kind, b.a, b.b}); }
    return zkh_session_set_recursion(s, rdesc, rdesc_words, ptrs.data(), words.data(), kinds.data(), built.size());
}

// one segment on one lane: built-in witness generator, or the caller's traces through prove_begin / accumulate / prove_finish
// records / ram: this segment's preflight output in pinned memory (witness source 1), or NULL: run the preflight here (a retry)
static const char* seal_one(zkh_session* s, Lane& l, const zkh_segment& seg, uint32_t** seal, size_t* words, double* witgen_s,
                            const uint32_t* records = nullptr, const uint32_t* ram = nullptr, double* preflight_cpu_s = nullptr, double* trace_bytes = nullptr) {
    const zkh_circuit* cir = l.circuit;
    const size_t n = (size_t)1 << seg.po2;
    NoiseKey nk;                                   // this segment's blinding key: the caller's, or (all-zero) 256 fresh bits from the OS
    ZKH_TRY(resolve_noise_key(seg.noise_key, &nk));
    const uint32_t* noise = nk.k;
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
    // The code (control) group of a built-in circuit is a function of (circuit, po2) alone: commit it once per lane and size and
    // keep the committed form in HBM (seals are byte-identical to the recomputing prover's).  Upstream's
    // SegmentProver re-commits it per segment; zkh_session_set_resident_code(s, 0) does the same.
    if (s->resident_code && cir->kind <= 2) {
        if (std::find(l.resident_po2.begin(), l.resident_po2.end(), seg.po2) == l.resident_po2.end()) {
            ZKH_TRY(zkh_syn_code(l.ctx, cir, seg.po2, ZKH_ZK_CYCLES, code));
            ZKH_TRY(zkh_prover_cache_code(l.prover, seg.po2, code));
            l.resident_po2.push_back(seg.po2);
        }
        code.out();                                                  // the trace itself is not needed again
    }
    zkh_buf* code_arg = code ? (zkh_buf*)code : nullptr;
    if (s->witness_source == 1 && cir->kind == 1) {
        // trace-driven: the compact per-cycle records (16 bytes per cycle) go up from pinned memory, the row fill expands them
        const size_t A = n - ZKH_ZK_CYCLES;
        std::vector<uint32_t> inline_records, inline_ram;
        if (!records) {                                               // no producer ran ahead for this one (a retry): preflight here
            inline_records.resize(4 * A); inline_ram.resize(zkh_syn_preflight_ram_words());
            double cpu = 0;
            ZKH_TRY(zkh_syn_preflight(seg.seed, seg.po2, ZKH_ZK_CYCLES, inline_records.data(), inline_ram.data(), &cpu));
            if (preflight_cpu_s) *preflight_cpu_s += cpu;
        }
        Tmp recs;
        ZKH_TRY(zkh_alloc(l.ctx, "records", 4 * A, 0, recs.out()));
        if (records) ZKH_TRY(zkh_write_async(l.ctx, recs, records, 0, 4 * A));
        else ZKH_TRY(zkh_write(l.ctx, recs, inline_records.data(), 0, 4 * A));
        if (trace_bytes) *trace_bytes += 16.0 * A + 4.0 * zkh_syn_preflight_ram_words() + 4.0 * out_global.size();
        ZKH_TRY(zkh_syn_witgen_trace(l.ctx, cir, seg.po2, ZKH_ZK_CYCLES, noise, recs, records ? ram : inline_ram.data(), code_arg, data, out_global.data()));
    } else {
        ZKH_TRY(zkh_syn_witgen(l.ctx, cir, seg.po2, ZKH_ZK_CYCLES, seg.seed, noise, seg.n_pub ? seg.pub : nullptr, code_arg, data, out_global.data()));
    }
    *witgen_s = now_s() - t0;
    return zkh_prove_segment(l.prover, seg.po2, ZKH_ZK_CYCLES, noise, code_arg, data, out_global.data(), seal, words);
}

extern "C" const char* zkh_session_prove(zkh_session* s, const zkh_segment* segs, size_t n, int join_tree, size_t join_po2, const uint32_t* join_noise_key,
                                         zkh_prove_info* info) {
    ZKH_REQUIRE(s && segs && n && info, "session_prove: bad argument");
    ZKH_REQUIRE(join_tree != 1 || !s->join_desc.empty(), "session_prove: the session was created without a join circuit");
    ZKH_REQUIRE(join_tree != 2 || !s->rec_kinds.empty(), "session_prove: join_tree 2 needs zkh_session_set_recursion");
    memset(info, 0, sizeof *info);
    info->n_segments = n;
    info->seals = (uint32_t**)calloc(n, sizeof(uint32_t*));
    info->seal_words = (size_t*)calloc(n, sizeof(size_t));
    if (!info->seals || !info->seal_words) {
        free(info->seals); free(info->seal_words);
        memset(info, 0, sizeof *info);
        return make_err("session_prove: out of host memory for %zu receipts", n);
    }
    const bool fold = join_tree == 2;
    if (join_tree == 1 && !s->assum_po2.empty()) {
        zkh_prove_info_free(info);
        return make_err("session_prove: the session has assumption receipts: they are united and resolved by the RECURSION programs (join_tree 2), the P2-JOIN tree has no such node");
    }
    const bool streamed = fold && s->streamed_fold;
    info->streamed = streamed;
    constexpr size_t NONE = (size_t)-1;
    // ---- chained session: the executor's pass.  Every segment's pre-state is fixed BEFORE any segment is proven (one launch over
    // all segments), so the provers stay independent; the pre-state becomes the segment's public input ----
    std::vector<zkh_segment> chained_segs;
    std::vector<uint32_t> pre_states;
    if (s->chained) {
        std::vector<uint64_t> seeds(n);
        std::vector<uint32_t> po2s(n), contrib(n);
        for (size_t i = 0; i < n; i++) {
            if (segs[i].host_code || segs[i].host_data) { zkh_prove_info_free(info); return make_err("session_prove: a chained session takes segments described by their seed"); }
            seeds[i] = segs[i].seed; po2s[i] = segs[i].po2;
        }
        if (const char* e = zkh_syn_chain_contributions(s->lanes[0].ctx, s->lanes[0].circuit, seeds.data(), po2s.data(), n, ZKH_ZK_CYCLES, contrib.data())) {
            zkh_prove_info_free(info);
            return e;
        }
        // SYN-C: one public word per segment (the pre-state).  SYN-S: 19 — pre-state, the exit code pair (SystemSplit (2, 0) for
        // every segment but the last, Halted(0) (0, 0) for the last) and the 16 limbs of SHA-256(journal), zero except in the last
        // segment; the journal of a session is its final state word (canonical residue, 4 bytes little-endian).
        const bool sess = circuit_is_session(s->lanes[0].circuit);
        const size_t pw = sess ? SESSION_OUT_WORDS - 4 : 1;
        pre_states.assign(n * pw, 0);
        uint32_t state = fp_encode(s->initial_state).v;
        for (size_t i = 0; i < n; i++) { pre_states[i * pw] = state; state = add_mod(state, contrib[i]); }
        if (sess) {
            // ... and WHICH receipts the session assumes: the last seal binds Output{journal, assumptions} (zkh_session_set_assumptions first)
            uint32_t limbs[SESSION_JOURNAL_LIMBS], ad[8];
            if (s->has_journal) session_output_limbs(s->journal_ptr(), s->journal.size(), s->assumptions_digest(ad), limbs);
            else session_journal_limbs(state, s->assumptions_digest(ad), limbs);
            for (size_t i = 0; i < n; i++) {
                const bool is_last = i + 1 == n;
                pre_states[i * pw + 1] = fp_encode(is_last ? EXIT_SYS_HALTED : EXIT_SYS_SPLIT).v;
                if (is_last) memcpy(&pre_states[i * pw + 3], limbs, sizeof limbs);
            }
        }
        chained_segs.assign(segs, segs + n);
        for (size_t i = 0; i < n; i++) { chained_segs[i].pub = &pre_states[i * pw]; chained_segs[i].n_pub = pw; }
        segs = chained_segs.data();
    }

    // ---- the fold plan (join_tree 2), fixed before anything runs: the bottom level turns segment receipts into recursion
    // receipts — a pair of segments is ONE proof where the program set has lift2(po2_l, po2_r) for EVERY pair (lift + lift + join
    // fused), else every segment is lifted on its own and the first level above pairs the lifts — and every level above THAT takes
    // three nodes at a time (zeth_amd/recursion.py fold_plan: one join3 where the program set has it for their sizes, else
    // join(join(a, b), c), the same node either way), a remainder of two is a join, of one moves up unchanged.  A node's program
    // follows from its children's sizes, so a missing program is reported here, not after the leaves are sealed. ----
    // (the plan itself and the scheduling state machine: scheduler.h — plain C++, unit-tested without a GPU)

    sched::FoldPlan fplan;
    std::vector<sched::PlanNode>& plan = fplan.nodes;
    struct NodeData { uint32_t* seal = nullptr; size_t words = 0; NodeClaim claim; };     // what a proven node leaves for its parent
    std::vector<NodeData> ndata;
    if (fold) {
        std::vector<uint32_t> seg_po2(n);
        for (size_t i = 0; i < n; i++) seg_po2[i] = segs[i].po2;
        const std::string perr = sched::build_fold_plan(
            seg_po2,
            [&](uint32_t join, uint32_t a, uint32_t b) -> int {
                for (size_t i = 0; i < s->rec_kinds.size(); i++)
                    if (s->rec_kinds[i].join == join && s->rec_kinds[i].a == a && s->rec_kinds[i].b == b) return (int)i;     // lifts: b = circuit family, 0 = the session's
                return -1;
            },
            [&](uint32_t program) { uint32_t inf[8] = {0}; (void)zkh_rec_program_info(s->lanes[0].programs[program], nullptr, inf); return inf[0]; },
            &fplan, s->assum_po2);
        if (!perr.empty()) { zkh_prove_info_free(info); return make_err("session_prove: %s", perr.c_str()); }
        ndata.resize(plan.size());
    }
    const size_t n_bottom = fplan.n_bottom, root_node = fplan.root;
    std::unique_ptr<zkh_circuit, void (*)(zkh_circuit*)> assum_circuit(nullptr, zkh_circuit_destroy);      // host-only: the assumption receipts' claims
    if (fold && !s->assum_po2.empty()) {
        zkh_circuit* ac = nullptr;
        if (const char* e = zkh_circuit_load(nullptr, s->assum_desc.data(), s->assum_desc.size(), &ac)) { zkh_prove_info_free(info); return e; }
        assum_circuit.reset(ac);
    }
    // control root of the leaf circuit per segment size (the lifts' claims are computed against it): before any thread starts
    std::vector<std::pair<uint32_t, std::vector<uint32_t>>> fold_leaf_roots;
    auto leaf_root_of = [&](uint32_t po2) -> const uint32_t* {
        for (auto& r : fold_leaf_roots) if (r.first == po2) return r.second.data();
        return nullptr;
    };
    if (fold)
        for (size_t i = 0; i < n; i++)
            if (!leaf_root_of(segs[i].po2)) {
                std::vector<uint32_t> cr(8);
                if (const char* e = leaf_control_root(s, segs[i], cr.data())) { zkh_prove_info_free(info); return e; }
                fold_leaf_roots.emplace_back(segs[i].po2, cr);
            }
    auto path_of = [&](uint32_t program, std::vector<uint32_t>& out) {      // per level: the direction bit as an element, the sibling
        size_t idx = program;
        for (size_t l = 0; l < REC_DEPTH; l++) {
            out.push_back(fp_encode((uint32_t)(idx & 1)).v);
            out.insert(out.end(), s->allowed[l][idx ^ 1].begin(), s->allowed[l][idx ^ 1].end());
            idx >>= 1;
        }
    };

    // ---- one scheduler for the whole call.  Sealing lanes pull segments from one index (round-robin with work stealing over all
    // lanes of all devices; a segment whose seal fails is handed to ANOTHER lane before the session gives up: ZKH_SEGMENT_RETRIES,
    // default 1).  With join_tree 2 a fold node becomes ready the moment its children exist and is proven by whichever lane is
    // free — the fold-only lanes while segments are still being sealed (the witness schedule of a lift / join is a 300-level
    // latency chain that fills the gaps the VALU-bound seals leave), every lane once the segments are done: upstream's
    // join-as-you-go.  zkh_session_set_streamed_fold(s, 0) holds the fold back until the last segment is sealed (two phases). ----
    ErrorSlot errs;
    std::mutex m;
    std::condition_variable cv;
    const char* renv = getenv("ZKH_SEGMENT_RETRIES");
    sched::Scheduler sc(n, s->lanes.size(), fold ? &fplan : nullptr, streamed, renv ? atoi(renv) : 1);
    // Fault injection (tests of the retry path): ZKH_FAULT_SEGMENT=<i> fails the FIRST attempt at segment i, ZKH_FAULT_SEGMENT_ALWAYS=<i>
    // every attempt.  Armed only by a value that parses STRICTLY as a decimal index (an empty or stray variable arms nothing: atol("")
    // was 0 = segment 0) and only together with ZKH_TEST_HOOKS=1, which no deployment sets.
    auto fault_index = [](const char* name) -> long {
        const char* v = getenv(name);
        const char* armed = getenv("ZKH_TEST_HOOKS");
        if (!v || !*v || !armed || strcmp(armed, "1") != 0) return -1;
        char* end = nullptr;
        const long k = strtol(v, &end, 10);
        return (end && *end == 0 && k >= 0) ? k : -1;
    };
    const long fault_seg = fault_index("ZKH_FAULT_SEGMENT"), fault_always = fault_index("ZKH_FAULT_SEGMENT_ALWAYS");
    const double t0 = now_s();

    auto finished = [&] { return errs.any() || sc.finished(); };                              // call with m held
    auto run_node = [&](Lane* l, size_t id, std::vector<uint32_t>& in) -> const char* {
        const sched::PlanNode& nd = plan[id];
        NodeData& me = ndata[id];
        in.clear();
        const std::vector<uint32_t>& A = s->allowed.back()[0];
        if (nd.kind == 1 || nd.kind == 3) {
            // per child: seal, membership path, then the opening of its claim' (core, pre, post) — all checked in-circuit
            const size_t ch3[3] = {nd.a, nd.b, nd.kind == 3 ? nd.c : NONE};
            for (size_t ch : ch3) {
                if (ch == NONE) continue;
                in.insert(in.end(), ndata[ch].seal, ndata[ch].seal + ndata[ch].words);
                path_of(plan[ch].program, in);
                in.insert(in.end(), ndata[ch].claim.core, ndata[ch].claim.core + 8);
                in.push_back(ndata[ch].claim.pre); in.push_back(ndata[ch].claim.post);
            }
            ZKH_TRY(parent_claim(ndata[nd.a].claim, ndata[nd.b].claim, &me.claim));
            if (nd.kind == 3) { const NodeClaim ab = me.claim; ZKH_TRY(parent_claim(ab, ndata[nd.c].claim, &me.claim)); }     // join3 = join(join(a, b), c)
        } else if (nd.kind == 4) {
            // union: per child its seal and membership path, then the swap bit — (lo, hi) = the two claim' sorted by canonical words
            uint32_t ca[8], cb[8];
            ZKH_TRY(wrap_claim(ndata[nd.a].claim, ca)); ZKH_TRY(wrap_claim(ndata[nd.b].claim, cb));
            bool swap = false;
            for (int i = 0; i < 8; i++) {
                const uint32_t x = fp_decode(Fp::raw(ca[i])), y = fp_decode(Fp::raw(cb[i]));
                if (x != y) { swap = y < x; break; }
            }
            for (size_t ch : {nd.a, nd.b}) {
                in.insert(in.end(), ndata[ch].seal, ndata[ch].seal + ndata[ch].words);
                path_of(plan[ch].program, in);
            }
            in.push_back(fp_encode(swap ? 1u : 0u).v);
            ZKH_TRY(hash_pair_host(swap ? cb : ca, swap ? ca : cb, me.claim.core));
            me.claim.pre = me.claim.post = 0;
        } else if (nd.kind == 5) {
            // resolve: the conditional receipt (the session's root: seal, path, the opening of its claim'), then the assumption receipt
            const NodeData& cond = ndata[nd.a];
            const NodeData& assum = ndata[nd.b];
            in.insert(in.end(), cond.seal, cond.seal + cond.words);
            path_of(plan[nd.a].program, in);
            in.insert(in.end(), cond.claim.core, cond.claim.core + 8);
            in.push_back(cond.claim.pre); in.push_back(cond.claim.post);
            in.insert(in.end(), assum.seal, assum.seal + assum.words);
            path_of(plan[nd.b].program, in);
            uint32_t cc[8], ca[8];
            ZKH_TRY(wrap_claim(cond.claim, cc)); ZKH_TRY(wrap_claim(assum.claim, ca));
            ZKH_TRY(hash_pair_host(cc, ca, me.claim.core));
            me.claim.pre = cond.claim.pre; me.claim.post = cond.claim.post;
        } else if (nd.kind == 0 && nd.family == 1) {
            // the lift of an assumption receipt: its seal, then A; claim = its receipt claim, no state
            const std::vector<uint32_t>& seal = s->assum_seals[nd.a];
            in.assign(seal.begin(), seal.end());
            ZKH_TRY(zkh_receipt_claim(assum_circuit.get(), seal.data(), seal.size(), s->assum_roots[nd.a].data(), nullptr, nullptr, me.claim.core));
            me.claim.pre = me.claim.post = 0;
            in.insert(in.end(), A.begin(), A.end());
        } else {
            const zkh_circuit* lc = s->lanes[0].circuit;
            in.assign(info->seals[nd.a], info->seals[nd.a] + info->seal_words[nd.a]);
            NodeClaim ca;
            ZKH_TRY(leaf_claim(lc, info->seals[nd.a], info->seal_words[nd.a], leaf_root_of(segs[nd.a].po2), &ca));
            if (nd.kind == 2) {
                in.insert(in.end(), info->seals[nd.b], info->seals[nd.b] + info->seal_words[nd.b]);
                NodeClaim cb;
                ZKH_TRY(leaf_claim(lc, info->seals[nd.b], info->seal_words[nd.b], leaf_root_of(segs[nd.b].po2), &cb));
                ZKH_TRY(parent_claim(ca, cb, &me.claim));
            } else {
                me.claim = ca;
            }
            in.insert(in.end(), A.begin(), A.end());
        }
        ZKH_TRY(zkh_rec_prove(l->programs[nd.program], in.data(), in.size(), join_noise_key, nullptr, &me.seal, &me.words));    // NULL: a fresh OS key per proof
        if (nd.kind == 1 || nd.kind >= 3) {                    // children are not kept: the verifier needs the root only
            for (size_t ch : {nd.a, nd.b, nd.kind == 3 ? nd.c : NONE})
                if (ch != NONE) { zkh_free_seal(ndata[ch].seal); ndata[ch].seal = nullptr; }
        }
        return nullptr;
    };
    double wit_sum = 0, seal_sum = 0, fold_busy = 0, pre_cpu_sum = 0, trace_bytes_sum = 0;
    // ---- witness source 1: the sequential host preflight runs AHEAD of the seals.  Every sealing lane has its own producer threads
    // (zkh_session_set_witness_source: default 2 — one preflight of a po2-20 segment is ~70 ms of one core, a lane seals one every
    // ~70 ms) and a pool of pinned record slots (producers + 1): a producer takes the next segment index, replays its cycles into a
    // free slot, and queues (segment, slot) for its lane, which uploads the 16 bytes per cycle and row-fills on the GPU.  The
    // producers are what pull the work index; a lane seals what its producers hand it. ----
    const bool use_pre = s->witness_source == 1 && s->lanes[0].circuit->kind == 1;
    struct Ready { size_t seg; uint32_t* slot; };
    struct LaneQ { std::deque<Ready> ready; std::vector<uint32_t*> free_slots; size_t producers_active = 0; bool accepting = true; };
    std::vector<LaneQ> laneq(s->lanes.size());
    const size_t ram_words = zkh_syn_preflight_ram_words();
    if (use_pre) {
        uint32_t max_po2 = 0;
        for (size_t i = 0; i < n; i++) {
            if (segs[i].host_code || segs[i].host_data || segs[i].n_pub) {       // (info's arrays are allocated: every early return frees them)
                zkh_prove_info_free(info);
                return make_err("session_prove: witness source 1 takes segments described by their seed only");
            }
            max_po2 = std::max(max_po2, segs[i].po2);
        }
        const size_t slot_words = ((size_t)4 << max_po2) + ram_words;          // records, then the RAM image
        for (size_t k = 0; k < s->lanes.size(); k++) {
            Lane& l = s->lanes[k];
            if (l.record_slot_words < slot_words) {                             // (main thread: nobody else touches the contexts yet)
                for (auto p : l.record_slots) zkh_host_free(l.ctx, p);
                l.record_slots.clear(); l.record_slot_words = 0;
            }
            while (l.record_slots.size() < s->producers_per_lane + 1) {
                uint32_t* p = nullptr;
                if (const char* e = zkh_host_alloc(l.ctx, slot_words, &p)) { zkh_prove_info_free(info); return e; }
                l.record_slots.push_back(p);
            }
            l.record_slot_words = slot_words;
            laneq[k].free_slots = l.record_slots;
            laneq[k].producers_active = s->producers_per_lane;
        }
    }
    auto producer = [&](size_t lane_idx) {
        LaneQ& q = laneq[lane_idx];
        const Lane& l = s->lanes[lane_idx];
        { const char* e = zkh_bind_thread_to_device(l.device, 0, 1, nullptr, nullptr); if (e) zkh_free_error(e); }      // (ZKH_AFFINITY=off: a no-op)
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            cv.wait(lk, [&] { return errs.any() || !q.accepting || !sc.indices_left() || !q.free_slots.empty(); });
            if (errs.any() || !q.accepting || !sc.indices_left()) break;
            const size_t i = sc.claim_index();
            uint32_t* slot = q.free_slots.back();
            q.free_slots.pop_back();
            lk.unlock();
            double cpu = 0;
            const char* err = zkh_syn_preflight(segs[i].seed, segs[i].po2, ZKH_ZK_CYCLES, slot, slot + ((size_t)4 << segs[i].po2), &cpu);
            lk.lock();
            pre_cpu_sum += cpu;
            if (err) { errs.set(err, "preflight"); q.free_slots.push_back(slot); break; }
            if (q.accepting) q.ready.push_back(Ready{i, slot});
            else { q.free_slots.push_back(slot); sc.requeue(i); }     // the lane stopped sealing meanwhile: anyone may take it
            cv.notify_all();
        }
        q.producers_active--;
        lk.unlock();
        cv.notify_all();
    };
    auto worker = [&](Lane* l, bool can_seal) {
        const size_t lane_idx = can_seal ? (size_t)(l - s->lanes.data()) : sched::NONE;
        { const char* e = zkh_bind_thread_to_device(l->device, 0, 1, nullptr, nullptr); if (e) zkh_free_error(e); }
        std::vector<uint32_t> in;
        double wit = 0, seal_t = 0, fold_t = 0;
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            if (finished()) break;
            // 1) a segment (scheduler.h take_segment: retries first, then the next index — or, with the host-preflight pipeline, what
            //    this lane's producers have prepared)
            size_t seg = NONE;
            uint32_t* slot = nullptr;
            if (can_seal) {
                seg = sc.take_segment(lane_idx, now_s(), !use_pre).index;
                if (seg == NONE && use_pre) {
                    LaneQ& q = laneq[lane_idx];
                    if (!q.ready.empty()) { seg = q.ready.front().seg; slot = q.ready.front().slot; q.ready.pop_front(); }
                }
            }
            if (seg != NONE) {
                lk.unlock();
                double w = 0, pcpu = 0, tbytes = 0;
                const double ts = now_s();
                const char* err = nullptr;
                if ((long)seg == fault_always || ((long)seg == fault_seg && sc.attempts(seg) == 0)) err = make_err("injected fault (ZKH_FAULT_SEGMENT)");
                else err = seal_one(s, *l, segs[seg], &info->seals[seg], &info->seal_words[seg], &w, slot, slot ? slot + ((size_t)4 << segs[seg].po2) : nullptr, &pcpu, &tbytes);
                const double te = now_s();
                lk.lock();
                pre_cpu_sum += pcpu; trace_bytes_sum += tbytes;
                if (slot) {
                    // a successful seal ended with a host sync, so the upload from this pinned slot is done; a FAILED one may have returned
                    // with the async H2D still in flight: drain the lane's stream before a producer may overwrite the slot
                    if (err) { lk.unlock(); if (const char* se = zkh_sync(l->ctx)) zkh_free_error(se); lk.lock(); }
                    laneq[lane_idx].free_slots.push_back(slot); cv.notify_all();
                }
                if (err) {
                    const sched::Scheduler::Failure f = sc.on_seal_failed(seg, lane_idx, now_s());      // hand it to another lane / device, or give up
                    if (f != sched::Scheduler::Failure::Fatal) {
                        zkh_free_error(err);
                        zkh_free_seal(info->seals[seg]); info->seals[seg] = nullptr; info->seal_words[seg] = 0;
                        if (f == sched::Scheduler::Failure::RetryAndRetireLane) {        // two failures in a row: this lane stops taking segments
                            can_seal = false;
                            if (use_pre) {                            // what its producers already prepared goes to the other lanes
                                LaneQ& q = laneq[lane_idx];
                                q.accepting = false;
                                for (auto& r : q.ready) { q.free_slots.push_back(r.slot); sc.requeue(r.seg); }
                                q.ready.clear();
                            }
                        }
                        cv.notify_all();
                        continue;
                    }
                    char what[64];
                    snprintf(what, sizeof what, "segment %zu (after %d attempt(s))", seg, sc.attempts(seg));
                    errs.set(err, what);
                    cv.notify_all();
                    break;
                }
                wit += w; seal_t += te - ts - w;
                sc.on_seal_done(seg, lane_idx, now_s());
                cv.notify_all();
                continue;
            }
            // 2) a fold node (streamed: any time; two phases: once every segment is sealed)
            const sched::Scheduler::Work nw = sc.take_node();
            if (nw.kind == sched::Scheduler::Kind::Node) {
                const size_t id = nw.index;
                lk.unlock();
                const double ts = now_s();
                const char* err = run_node(l, id, in);
                const double te = now_s();
                lk.lock();
                fold_t += te - ts;
                if (err) {
                    const uint32_t k = plan[id].kind;
                    errs.set(err, k == 1 ? "join" : k == 3 ? "join3" : k == 2 ? "lift2" : k == 4 ? "union (of two assumption receipts)" : k == 5 ? "resolve (the session's root against its assumptions)"
                             : plan[id].family == 1 ? "lift (of an assumption receipt)" : "lift");
                    cv.notify_all();
                    break;
                }
                sc.on_node_done(id, now_s());
                cv.notify_all();
                continue;
            }
            // 3) nothing to do right now (a retry entry held back for another lane wakes us after 20 ms; producers, finished seals and
            //    finished fold nodes notify)
            if (can_seal && sc.retries_waiting()) cv.wait_for(lk, std::chrono::milliseconds(20));
            else cv.wait(lk);
        }
        sc.lane_leaves(can_seal);
        wit_sum += wit; seal_sum += seal_t; fold_busy += fold_t;
        lk.unlock();
        cv.notify_all();
        (void)zkh_sync(l->ctx);
    };
    {
        std::vector<std::thread> th;
        for (auto& lane : s->lanes) th.emplace_back(worker, &lane, true);
        if (fold) for (auto& lane : s->fold_lanes) th.emplace_back(worker, &lane, false);
        if (use_pre) for (size_t k = 0; k < s->lanes.size(); k++) for (size_t p = 0; p < s->producers_per_lane; p++) th.emplace_back(producer, k);
        for (auto& t : th) t.join();
    }
    const double t_end = now_s();
    info->witgen_s_sum = wit_sum; info->seal_s_sum = seal_sum; info->fold_busy_s_sum = fold_busy; info->n_retries = sc.n_retries;
    info->preflight_cpu_s_sum = pre_cpu_sum; info->trace_bytes = trace_bytes_sum;
    info->leaves_s = (sc.t_leaves_done ? sc.t_leaves_done : t_end) - t0;
    if (fold) {
        // bottom level = lifts (or lift2 per pair); with the streamed fold these overlap the leaf phase: lift_s / join_s are what the
        // fold still took AFTER the last segment was sealed (bottom level, then joins); two phases: the two phases' durations
        const double tb = std::max(sc.t_bottom_done ? sc.t_bottom_done : t_end, t0 + info->leaves_s);
        info->n_lifts = n_bottom + fplan.n_assumptions;          // the bottom level + a lift per assumption receipt
        info->n_joins = plan.size() - info->n_lifts;             // joins / join3s, unions, the resolve
        info->lift_s = tb - (t0 + info->leaves_s);
        info->join_s = t_end - tb;
        info->fold_tail_s = t_end - (t0 + info->leaves_s);
        if (!errs.any()) {
            NodeData& r = ndata[root_node];
            info->root_seal = r.seal; info->root_seal_words = r.words; info->root_program = plan[root_node].program;
            memcpy(info->root_core, r.claim.core, 32); info->root_pre = r.claim.pre; info->root_post = r.claim.post;
            r.seal = nullptr;
        }
        for (auto& nd : ndata) zkh_free_seal(nd.seal);
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
                        NoiseKey jk;
                        if (errs.set(resolve_noise_key(join_noise_key, &jk), "join noise")) break;
                        const uint32_t* noise = jk.k;
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
    for (size_t i = 0; i < info->n_segments; i++) {          // the control root of every segment size first (lane 0's GPU for built-in circuits)
        bool have = false;
        for (auto& r : roots) have = have || r.first == segs[i].po2;
        if (!have) {
            std::vector<uint32_t> cr(8);
            ZKH_TRY(leaf_control_root(s, segs[i], cr.data()));
            roots.emplace_back(segs[i].po2, cr);
        }
    }
    {
        // the segment seals are independent and the verifier is pure host arithmetic on read-only data: spread them over host
        // threads (upstream verifies them in a loop; 1024 seals x 7 ms is 7 s on one core)
        ErrorSlot verrs;
        std::atomic<size_t> next{0};
        const size_t n_threads = std::max<size_t>(1, std::min<size_t>({(size_t)std::thread::hardware_concurrency(), (size_t)16, info->n_segments}));
        std::vector<std::thread> th;
        for (size_t t = 0; t < n_threads; t++)
            th.emplace_back([&] {
                for (;;) {
                    const size_t i = next.fetch_add(1);
                    if (i >= info->n_segments || verrs.any()) return;
                    const uint32_t* root = nullptr;
                    for (auto& r : roots) if (r.first == segs[i].po2) root = r.second.data();
                    char what[48];
                    snprintf(what, sizeof what, "segment %zu", i);
                    if (verrs.set(zkh_verify_segment(hc, info->seals[i], info->seal_words[i], root, nullptr, nullptr), what)) return;
                    if (verrs.set(zkh_receipt_claim(hc, info->seals[i], info->seal_words[i], root, nullptr, nullptr, claims[i].data()), what)) return;
                }
            });
        for (auto& t : th) t.join();
        if (verrs.any()) return make_err("session_verify: %s", verrs.first.c_str());
    }
    if (s->chained) {          // continuity (CompositeReceipt::verify_integrity): pre == prev.post, read from the verified seals' `out` words
        uint32_t prev = fp_encode(s->initial_state).v;
        for (size_t i = 0; i < info->n_segments; i++) {
            ZKH_REQUIRE(info->seal_words[i] > 5 && info->seals[i][4] == prev, "session_verify: the session is not continuous: segment %zu does not start from its predecessor's post-state", i);
            prev = info->seals[i][0];
        }
        // SYN-S: the session TERMINATES here — every segment but the last says SystemSplit, the last Halted(0) and carries the digest
        // of the journal (= the final state word): a receipt with trailing segments cut off ends in a SystemSplit and is refused
        if (circuit_is_session(s->lanes[0].circuit)) {
            std::vector<const uint32_t*> seals(info->n_segments);
            for (size_t i = 0; i < info->n_segments; i++) {
                ZKH_REQUIRE(info->seal_words[i] > SESSION_OUT_WORDS, "session_verify: segment %zu: seal too short", i);
                seals[i] = info->seals[i];
            }
            uint32_t ad[8];
            ZKH_TRY(check_session_termination(seals.data(), info->n_segments, s->journal_ptr(), s->journal.size(), s->assumptions_digest(ad)));
        }
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
        // the claim tree over the leaves (receipt claim + state words of every VERIFIED segment seal), as the lift2 / join programs
        // build it in-circuit: parent core = hash_pair(claim'_l, claim'_r), state range [pre_l, post_r], and neighbours must chain
        std::vector<NodeClaim> nodes(info->n_segments);
        for (size_t i = 0; i < info->n_segments; i++) {
            memcpy(nodes[i].core, claims[i].data(), 32);
            const bool chained = circuit_has_state(hc);
            nodes[i].pre = chained ? info->seals[i][4] : 0; nodes[i].post = chained ? info->seals[i][0] : 0;
        }
        // ... in the shape of the fold plan (fold_claim_nodes: pairs, then three at a time)
        NodeClaim top;
        ZKH_TRY(fold_claim_nodes(nodes, &top));
        uint32_t want[8];
        ZKH_TRY(wrap_claim(top, want));
        if (!s->assum_seals.empty()) {
            // the session was RESOLVED against its assumption receipts (verified when they were handed over): the union tree over their
            // lifted claims, then claim' = wrap(hash_pair(claim' of the join tree, claim' of the union tree), pre, post of the session)
            zkh_circuit* ac = nullptr;
            ZKH_TRY(zkh_circuit_load(nullptr, s->assum_desc.data(), s->assum_desc.size(), &ac));
            std::unique_ptr<zkh_circuit, void (*)(zkh_circuit*)> ahold(ac, zkh_circuit_destroy);
            std::vector<std::array<uint32_t, 8>> lifted(s->assum_seals.size());
            for (size_t i = 0; i < lifted.size(); i++) {
                NodeClaim nd;
                ZKH_TRY(zkh_receipt_claim(ac, s->assum_seals[i].data(), s->assum_seals[i].size(), s->assum_roots[i].data(), nullptr, nullptr, nd.core));
                ZKH_TRY(wrap_claim(nd, lifted[i].data()));
            }
            uint32_t assumed[8];
            ZKH_TRY(union_tree(lifted, assumed));
            NodeClaim res;
            ZKH_TRY(hash_pair_host(want, assumed, res.core));
            res.pre = top.pre; res.post = top.post;
            ZKH_TRY(wrap_claim(res, want));
        }
        ZKH_REQUIRE(info->root_seal_words > 16 && memcmp(info->root_seal, want, 32) == 0,
                    s->assum_seals.empty() ? "session_verify: the root receipt does not commit to the claim tree of these segments"
                                           : "session_verify: the root receipt does not commit to the claim tree of these segments resolved against the session's assumption receipts");
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
