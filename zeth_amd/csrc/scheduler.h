// scheduler.h — the session executor's plan and state machine, as plain C++ (no HIP, no threads of its own).
//
// zkh_session_prove (session.hip) is `ProverServer::prove_session` + the lift / join tree of `ProverImpl::{lift, join}` (risc0-zkvm
// 3.0.3, un-vendored: /root/reference/Cargo.lock:5418) behind `default_prover().prove(env, elf)`
// (/root/reference/crates/host/src/lib.rs:137).  Its worker threads do the GPU work; WHAT they do next is decided here:
//   * FoldPlan — which recursion program proves which node, fixed before anything runs (zeth_amd/recursion.py fold_plan: the first
//     level pairs the segments — one lift2 per pair where the program set has it for EVERY pair, else a lift per segment and joins —
//     every level above takes three nodes per proof: join3 where a program exists for their sizes, else join(join(a, b), c); a
//     remainder of two is a join, of one moves up).  A session with ASSUMPTION receipts (keccak batches; ProverServer::{union,
//     resolve}) gets three more kinds of nodes: a lift per assumption receipt (ready from the start: nothing is sealed for it), a
//     union tree over those (neighbours pairwise, an odd one moves up: zeth_amd/recursion.py union_claims), and ONE resolve of the
//     session's root against the union root — the plan's root;
//   * Scheduler — segments handed out through one index (work stealing over all lanes), a failed segment retried on ANOTHER lane,
//     a lane that fails twice in a row retired from sealing, a fold node ready the moment its children exist (streamed) or once the
//     last segment is sealed (two phases), the run finished when every segment is sealed and the root exists.
// Every method is a transition on plain state and is called with the session's mutex held; tests/cpp/scheduler_test.cpp drives the
// object alone (no GPU): plan shapes against the Python fold plan's counts, dependency order under random completion, every
// failure path.
#pragma once
#include <cstddef>
#include <cstdint>
#include <deque>
#include <functional>
#include <string>
#include <vector>

namespace zkh {
namespace sched {

constexpr size_t NONE = (size_t)-1;

struct PlanNode {
    uint32_t kind = 0;                 // 0 lift(a), 2 lift2(a, b): a, b segment indices; 1 join(a, b), 3 join3(a, b, c): node ids;
                                       // 4 union(a, b), 5 resolve(a = the session's root, b = the union root): node ids
    size_t a = 0, b = 0, c = 0, parent = NONE;
    uint32_t program = 0, po2 = 0;
    uint32_t family = 0;               // lifts: 0 = a segment of the session, 1 = an ASSUMPTION receipt (a = its index; pending 0)
    int pending = 0;                   // children not yet available
};

struct FoldPlan {
    std::vector<PlanNode> nodes;
    std::vector<size_t> owner;         // the bottom node that consumes segment i
    size_t n_bottom = 0, root = NONE;
    size_t session_root = NONE;        // the join tree's root (== root without assumptions)
    size_t n_assumptions = 0;
};

// program_of(kind, a, b) -> index or -1: kind 0 lift (a = segment po2, b = circuit family), 1 join (child sizes), 2 lift2 (segment
// sizes), 3 join3 (size of the first two children, size of the third), 4 union / 5 resolve (child sizes); po2_of(program) -> the size
// the program runs at.  assum_po2: the sizes of the session's assumption receipts (lift family 1).  -> "" or what is missing.
inline std::string build_fold_plan(const std::vector<uint32_t>& seg_po2, const std::function<int(uint32_t, uint32_t, uint32_t)>& program_of,
                                   const std::function<uint32_t(uint32_t)>& po2_of, FoldPlan* out, const std::vector<uint32_t>& assum_po2 = {}) {
    FoldPlan& p = *out;
    const size_t n = seg_po2.size();
    p.nodes.clear(); p.owner.assign(n, NONE); p.n_bottom = 0; p.root = NONE; p.session_root = NONE; p.n_assumptions = assum_po2.size();
    if (!n) return "no segments";
    std::string err;
    const size_t n_pairs = n / 2;
    bool all_fused = n > 1;
    for (size_t k = 0; k < n_pairs && all_fused; k++) all_fused = program_of(2, seg_po2[2 * k], seg_po2[2 * k + 1]) >= 0;
    auto add_lift = [&](size_t i) {
        const int pr = program_of(0, seg_po2[i], 0);
        if (pr < 0) { err = "no lift program for po2-" + std::to_string(seg_po2[i]) + " segments"; return; }
        PlanNode nd; nd.kind = 0; nd.a = i; nd.program = (uint32_t)pr; nd.po2 = po2_of((uint32_t)pr); nd.pending = 1;
        p.owner[i] = p.nodes.size(); p.nodes.push_back(nd);
    };
    if (all_fused) {
        for (size_t k = 0; k < n_pairs; k++) {
            PlanNode nd; nd.kind = 2; nd.a = 2 * k; nd.b = 2 * k + 1; nd.pending = 2;
            nd.program = (uint32_t)program_of(2, seg_po2[2 * k], seg_po2[2 * k + 1]); nd.po2 = po2_of(nd.program);
            p.owner[2 * k] = p.owner[2 * k + 1] = p.nodes.size(); p.nodes.push_back(nd);
        }
        if (n % 2) add_lift(n - 1);
    } else {
        for (size_t i = 0; i < n && err.empty(); i++) add_lift(i);
    }
    if (!err.empty()) return err;
    p.n_bottom = p.nodes.size();
    std::vector<size_t> cur(p.n_bottom);
    for (size_t k = 0; k < p.n_bottom; k++) cur[k] = k;
    auto add_join = [&](size_t a, size_t b) -> size_t {
        PlanNode nd; nd.kind = 1; nd.a = a; nd.b = b; nd.pending = 2;
        const int pr = program_of(1, p.nodes[a].po2, p.nodes[b].po2);
        if (pr < 0) { err = "no join program for children of po2 " + std::to_string(p.nodes[a].po2) + " and " + std::to_string(p.nodes[b].po2); return NONE; }
        nd.program = (uint32_t)pr; nd.po2 = po2_of(nd.program);
        p.nodes[a].parent = p.nodes[b].parent = p.nodes.size();
        p.nodes.push_back(nd);
        return p.nodes.size() - 1;
    };
    // the bottom nodes are the first level's pairs when they are lift2s (and a lone lift); when every segment was lifted on its own,
    // the first level above still pairs
    size_t group = all_fused || n < 2 ? 3 : 2;
    while (cur.size() > 1 && err.empty()) {
        std::vector<size_t> nxt;
        size_t k = 0;
        for (; k + group <= cur.size() && err.empty(); k += group) {
            if (group == 2) { nxt.push_back(add_join(cur[k], cur[k + 1])); continue; }
            const size_t a = cur[k], b = cur[k + 1], c = cur[k + 2];
            const int p3 = p.nodes[a].po2 == p.nodes[b].po2 ? program_of(3, p.nodes[a].po2, p.nodes[c].po2) : -1;
            if (p3 >= 0) {
                PlanNode nd; nd.kind = 3; nd.a = a; nd.b = b; nd.c = c; nd.pending = 3;
                nd.program = (uint32_t)p3; nd.po2 = po2_of(nd.program);
                p.nodes[a].parent = p.nodes[b].parent = p.nodes[c].parent = p.nodes.size();
                nxt.push_back(p.nodes.size()); p.nodes.push_back(nd);
            } else {                                                        // the same node as two proofs
                const size_t ab = add_join(a, b);
                nxt.push_back(err.empty() ? add_join(ab, c) : NONE);
            }
        }
        if (err.empty() && cur.size() - k == 2) nxt.push_back(add_join(cur[k], cur[k + 1]));
        else if (err.empty() && cur.size() - k == 1) nxt.push_back(cur[k]);
        cur.swap(nxt);
        group = 3;
    }
    if (!err.empty()) return err;
    p.root = p.session_root = cur[0];
    if (assum_po2.empty()) return "";
    // ---- assumptions: a lift each, the union tree, one resolve ----
    std::vector<size_t> lvl;
    for (size_t i = 0; i < assum_po2.size(); i++) {
        const int pr = program_of(0, assum_po2[i], 1);
        if (pr < 0) return "no lift program for po2-" + std::to_string(assum_po2[i]) + " assumption receipts";
        PlanNode nd; nd.kind = 0; nd.family = 1; nd.a = i; nd.program = (uint32_t)pr; nd.po2 = po2_of((uint32_t)pr); nd.pending = 0;
        lvl.push_back(p.nodes.size()); p.nodes.push_back(nd);
    }
    auto add_pair = [&](uint32_t kind, size_t a, size_t b) -> size_t {
        const int pr = program_of(kind, p.nodes[a].po2, p.nodes[b].po2);
        if (pr < 0) {
            err = std::string("no ") + (kind == 4 ? "union" : "resolve") + " program for children of po2 " + std::to_string(p.nodes[a].po2) + " and " + std::to_string(p.nodes[b].po2);
            return NONE;
        }
        PlanNode nd; nd.kind = kind; nd.a = a; nd.b = b; nd.pending = 2; nd.program = (uint32_t)pr; nd.po2 = po2_of(nd.program);
        p.nodes[a].parent = p.nodes[b].parent = p.nodes.size();
        p.nodes.push_back(nd);
        return p.nodes.size() - 1;
    };
    while (lvl.size() > 1 && err.empty()) {
        std::vector<size_t> nxt;
        for (size_t k = 0; k + 1 < lvl.size() && err.empty(); k += 2) nxt.push_back(add_pair(4, lvl[k], lvl[k + 1]));
        if (lvl.size() % 2) nxt.push_back(lvl.back());
        lvl.swap(nxt);
    }
    if (!err.empty()) return err;
    p.root = add_pair(5, p.session_root, lvl[0]);
    return err;
}

class Scheduler {
  public:
    enum class Kind { None, Seal, Node };
    struct Work { Kind kind = Kind::None; size_t index = NONE; };
    enum class Failure { Retry, RetryAndRetireLane, Fatal };

    // plan == nullptr: no fold (the run ends when every segment is sealed)
    Scheduler(size_t n_segments, size_t n_seal_lanes, FoldPlan* plan, bool streamed, int max_retries)
        : n_(n_segments), plan_(plan), streamed_(streamed), max_retries_(max_retries), attempts_(n_segments, 0),
          consecutive_(n_seal_lanes, 0) {
        seal_lanes_active = n_seal_lanes;
        root_done = plan == nullptr;
        if (plan)
            for (size_t k = 0; k < plan->nodes.size(); k++)
                if (plan->nodes[k].pending == 0) ready_.push_back(k);      // lifts of assumption receipts: nothing to wait for
    }

    // ---- what a lane does next ----
    // Retries first (never on the lane that just failed the segment, unless no other sealing lane is left or nobody else picked it up
    // within 50 ms), then — from_index — the next segment index (false: a producer pipeline hands this lane its segments; the caller
    // looks there when Kind::None comes back with can_seal), then a fold node whose children exist.
    Work take_segment(size_t lane, double now, bool from_index) {
        for (auto it = retry_.begin(); it != retry_.end(); ++it)
            if (it->failed_on != lane || seal_lanes_active <= 1 || now - it->since > 0.05) {
                const Work w{Kind::Seal, it->seg};
                retry_.erase(it);
                return w;
            }
        if (from_index && next_seal < n_) return Work{Kind::Seal, next_seal++};
        return Work{};
    }
    // a producer thread claims the next segment index to prepare (NONE: all handed out)
    size_t claim_index() { return next_seal < n_ ? next_seal++ : NONE; }
    bool indices_left() const { return next_seal < n_; }
    Work take_node() {
        if (!plan_ || ready_.empty() || !(streamed_ || seals_done == n_)) return Work{};
        const Work w{Kind::Node, ready_.front()};
        ready_.pop_front();
        return w;
    }
    bool retries_waiting() const { return !retry_.empty(); }

    // ---- transitions ----
    void on_seal_done(size_t seg, size_t lane, double now) {
        if (lane < consecutive_.size()) consecutive_[lane] = 0;
        if (++seals_done == n_) t_leaves_done = now;
        if (plan_) child_done(plan_->owner[seg]);
    }
    Failure on_seal_failed(size_t seg, size_t lane, double now) {
        if (attempts_[seg]++ >= max_retries_) return Failure::Fatal;
        retry_.push_back(Retry{seg, lane, now});
        n_retries++;
        if (lane < consecutive_.size() && ++consecutive_[lane] >= 2 && seal_lanes_active > 1) {
            seal_lanes_active--;                                   // this lane stops taking segments
            return Failure::RetryAndRetireLane;
        }
        return Failure::Retry;
    }
    // a segment that was prepared for a lane which no longer seals: anyone may take it, at once
    void requeue(size_t seg) { retry_.push_back(Retry{seg, NONE, 0}); }
    void lane_leaves(bool was_sealing) { if (was_sealing) seal_lanes_active = seal_lanes_active ? seal_lanes_active - 1 : 0; }
    void on_node_done(size_t node, double now) {
        if (node < plan_->n_bottom && ++bottom_done == plan_->n_bottom) t_bottom_done = now;
        if (node == plan_->root) root_done = true;
        else if (plan_->nodes[node].parent != NONE) child_done(plan_->nodes[node].parent);
    }
    bool finished() const { return seals_done == n_ && root_done; }
    int attempts(size_t seg) const { return attempts_[seg]; }

    // ---- counters the caller reports ----
    size_t next_seal = 0, seals_done = 0, seal_lanes_active = 0, bottom_done = 0, n_retries = 0;
    bool root_done = false;
    double t_leaves_done = 0, t_bottom_done = 0;

  private:
    struct Retry { size_t seg; size_t failed_on; double since; };
    void child_done(size_t node) { if (--plan_->nodes[node].pending == 0) ready_.push_back(node); }
    size_t n_;
    FoldPlan* plan_;
    bool streamed_;
    int max_retries_;
    std::vector<int> attempts_, consecutive_;
    std::deque<size_t> ready_;
    std::deque<Retry> retry_;
};

}  // namespace sched
}  // namespace zkh
