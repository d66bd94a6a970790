// scheduler_test.cpp — the session executor's plan and state machine (zeth_amd/csrc/scheduler.h) on their own: no GPU, no threads.
//   g++ -std=c++17 -O1 -I zeth_amd/csrc tests/cpp/scheduler_test.cpp -o scheduler_test && ./scheduler_test
// Checks: the fold plan's shape against the counts of zeth_amd/recursion.py fold_plan (pairs, then three at a time; join3 or two joins);
// every node proven exactly once and only after its children under random completion orders, streamed and two-phase; a failed segment
// retried on ANOTHER lane; a lane retired after two failures in a row; a segment that keeps failing ends the run; producers' indices;
// sessions with assumption receipts: their lifts ready from the start, the union tree, one resolve as the root.
#include <cstdio>
#include <cstdlib>
#include <random>
#include <set>

#include "scheduler.h"

using namespace zkh::sched;

static int failures = 0;
#define CHECK(cond)                                                              \
    do {                                                                         \
        if (!(cond)) { printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); failures++; } \
    } while (0)

// a program set like a SYN-A block's: lift2 -> 18, joins close at 18, optionally join3(18, 18, 18) -> 18; a lone lift runs at
// `lift_po2` (17 in the real set: a group of three that contains it has no join3 and is proven as two joins)
struct Programs {
    bool lift2 = true, join3 = true;
    uint32_t lift_po2 = 18;
    bool unions = true;
    int program_of(uint32_t kind, uint32_t a, uint32_t b) const {
        if (kind == 0 && b == 1) return a == 13 ? 4 : -1;                 // the lift of an assumption receipt (family 1): runs at po2 19
        if (kind == 4) return (unions && a >= 18 && a <= 19 && b >= 18 && b <= 19) ? 5 : -1;
        if (kind == 5) return (a >= 17 && a <= 18 && b >= 18 && b <= 19) ? 6 : -1;
        if (kind == 0) return (a == 20 || a == 18) ? 0 : -1;
        if (kind == 2) return lift2 ? 1 : -1;
        if (kind == 1) return (a >= 17 && a <= 18 && b >= 17 && b <= 18) ? 2 : -1;
        if (kind == 3) return (join3 && a == 18 && b == 18) ? 3 : -1;
        return -1;
    }
    uint32_t po2_of(uint32_t program) const { return program == 0 ? lift_po2 : program == 4 ? 19 : 18; }
};

static FoldPlan plan_for(size_t n, const Programs& pr, std::string* err = nullptr, size_t n_assumptions = 0) {
    std::vector<uint32_t> po2(n, 20);
    if (n > 1) po2[n - 1] = 18;
    FoldPlan p;
    const std::string e = build_fold_plan(po2, [&](uint32_t k, uint32_t a, uint32_t b) { return pr.program_of(k, a, b); }, [&](uint32_t g) { return pr.po2_of(g); }, &p,
                                          std::vector<uint32_t>(n_assumptions, 13));
    if (err) *err = e;
    else CHECK(e.empty());
    return p;
}

// zeth_amd/recursion.py fold_plan(n): groups per level -> number of nodes above the bottom when a group of three is ONE proof
static size_t python_plan_proofs(size_t n_leaves, bool join3) {
    size_t cur = (n_leaves + 1) / 2, proofs = 0;          // after the pairing level
    while (cur > 1) {
        const size_t g3 = cur / 3, rem = cur % 3;
        proofs += g3 * (join3 ? 1 : 2) + (rem == 2 ? 1 : 0);
        cur = g3 + (rem ? 1 : 0);
    }
    return proofs;
}

static void test_plan_shapes() {
    for (bool j3 : {true, false}) {
        Programs pr; pr.join3 = j3;
        for (size_t n : {1, 2, 3, 4, 5, 6, 7, 9, 10, 27, 64, 100, 1024}) {
            const FoldPlan p = plan_for(n, pr);
            CHECK(p.n_bottom == (n + 1) / 2);
            CHECK(p.nodes.size() - p.n_bottom == python_plan_proofs(n, j3));
            CHECK(p.root != NONE && p.nodes[p.root].parent == NONE);
            size_t roots = 0;
            for (auto& nd : p.nodes) roots += nd.parent == NONE;
            CHECK(roots == 1);
            for (size_t i = 0; i < n; i++) CHECK(p.owner[i] == i / 2);
            for (size_t k = 0; k < p.nodes.size(); k++) {
                const PlanNode& nd = p.nodes[k];
                CHECK(nd.pending == (nd.kind == 3 ? 3 : nd.kind == 0 ? 1 : 2));
                if (nd.kind == 1 || nd.kind == 3) { CHECK(p.nodes[nd.a].parent == k && p.nodes[nd.b].parent == k); CHECK(nd.a < k && nd.b < k); }
                if (nd.kind == 3) CHECK(p.nodes[nd.c].parent == k);
            }
        }
    }
    Programs pr;
    CHECK(plan_for(1024, pr).nodes.size() == 768);                    // 512 lift2 + 256 join3 / joins (config 5)
    pr.join3 = false;
    CHECK(plan_for(1024, pr).nodes.size() == 1023);
    pr.join3 = true; pr.lift_po2 = 17;                                // 5 segments: lift2, lift2, lift(17): (18, 18, 17) has no join3 -> join(join(a, b), c)
    CHECK(plan_for(5, pr).nodes.size() == 3 + 2);
    pr.lift_po2 = 18;
    CHECK(plan_for(5, pr).nodes.size() == 3 + 1);
    pr.join3 = false; pr.lift2 = false;                               // every segment lifted on its own, the first level above pairs
    const FoldPlan p = plan_for(8, pr);
    CHECK(p.n_bottom == 8 && p.nodes.size() == 8 + 4 + 2 + 1);
    // assumptions: a lift each (ready from the start), the union tree (neighbours pairwise, an odd one moves up), one resolve = the root
    pr = Programs();
    for (size_t na : {1, 2, 3, 4, 5, 8, 9}) {
        const FoldPlan base = plan_for(10, pr), q = plan_for(10, pr, nullptr, na);
        CHECK(q.n_assumptions == na && q.session_root == base.root && q.n_bottom == base.n_bottom);
        CHECK(q.nodes.size() == base.nodes.size() + na + (na - 1) + 1);                 // lifts, unions, the resolve
        const PlanNode& top = q.nodes[q.root];
        CHECK(top.kind == 5 && top.a == q.session_root && top.parent == NONE && q.nodes[q.session_root].parent == q.root);
        size_t lifts = 0, unions = 0, ready = 0;
        for (size_t k = base.nodes.size(); k < q.nodes.size(); k++) {
            const PlanNode& nd = q.nodes[k];
            lifts += nd.kind == 0 && nd.family == 1; unions += nd.kind == 4; ready += nd.pending == 0;
            if (nd.kind == 0) CHECK(nd.family == 1 && nd.pending == 0 && nd.po2 == 19 && nd.a < na);
            if (nd.kind == 4) CHECK(nd.pending == 2 && nd.a < k && nd.b < k && q.nodes[nd.a].parent == k && q.nodes[nd.b].parent == k);
        }
        CHECK(lifts == na && unions == na - 1 && ready == na);
        for (size_t k = 0; k < base.nodes.size(); k++) CHECK(q.nodes[k].kind == base.nodes[k].kind && q.nodes[k].a == base.nodes[k].a && q.nodes[k].family == 0);
        // the union tree's shape: level 0 pairs (0, 1), (2, 3), ..; the union root hangs under the resolve
        CHECK(q.nodes[top.b].parent == q.root && (na == 1 ? q.nodes[top.b].kind == 0 : q.nodes[top.b].kind == 4));
    }
    {
        std::string e2;
        pr.unions = false;
        plan_for(4, pr, &e2, 2);
        CHECK(e2.find("no union program") != std::string::npos);
        pr.unions = true;
        FoldPlan q;
        const std::string e3 = build_fold_plan({20, 20}, [&](uint32_t k, uint32_t a, uint32_t b) { return pr.program_of(k, a, b); }, [&](uint32_t g) { return pr.po2_of(g); }, &q, {12});
        CHECK(e3.find("no lift program for po2-12 assumption") != std::string::npos);
    }
    std::string err;
    std::vector<uint32_t> odd(3, 21);
    FoldPlan q;
    CHECK(!build_fold_plan(odd, [&](uint32_t k, uint32_t a, uint32_t b) { return pr.program_of(k, a, b); }, [&](uint32_t g) { return pr.po2_of(g); }, &q).empty());
}

// drive a scheduler with `lanes` sealing lanes + `fold_lanes` fold-only lanes under a random completion order; returns proofs done
static void simulate(size_t n, size_t lanes, size_t fold_lanes, bool streamed, unsigned seed, size_t n_assumptions = 0) {
    Programs pr;
    FoldPlan plan = plan_for(n, pr, nullptr, n_assumptions);
    Scheduler sc(n, lanes, &plan, streamed, 1);
    std::mt19937 rng(seed);
    struct Busy { bool is_node; size_t index; size_t lane; };
    std::vector<Busy> busy;
    std::set<size_t> sealed, proven;
    std::vector<bool> lane_free(lanes + fold_lanes, true);
    double now = 0;
    size_t guard = 0;
    while (!sc.finished() && guard++ < 100000) {
        for (size_t l = 0; l < lane_free.size(); l++) {
            if (!lane_free[l]) continue;
            Scheduler::Work w;
            if (l < lanes) w = sc.take_segment(l, now, true);
            if (w.kind == Scheduler::Kind::None) w = sc.take_node();
            if (w.kind == Scheduler::Kind::None) continue;
            if (w.kind == Scheduler::Kind::Node) {
                const PlanNode& nd = plan.nodes[w.index];
                if (!streamed) CHECK(sealed.size() == n);                           // two phases: no fold node before the last seal
                if (nd.kind == 0 && nd.family == 1) CHECK(nd.a < n_assumptions);               // an assumption's lift waits for nothing
                else if (nd.kind == 0) CHECK(sealed.count(nd.a));
                else if (nd.kind == 2) CHECK(sealed.count(nd.a) && sealed.count(nd.b));
                else { CHECK(proven.count(nd.a) && proven.count(nd.b)); if (nd.kind == 3) CHECK(proven.count(nd.c)); }
                CHECK(!proven.count(w.index));
            } else {
                CHECK(!sealed.count(w.index));
            }
            busy.push_back(Busy{w.kind == Scheduler::Kind::Node, w.index, l});
            lane_free[l] = false;
        }
        CHECK(!busy.empty());
        if (busy.empty()) break;
        const size_t k = rng() % busy.size();
        const Busy b = busy[k];
        busy.erase(busy.begin() + k);
        now += 0.001;
        lane_free[b.lane] = true;
        if (b.is_node) { proven.insert(b.index); sc.on_node_done(b.index, now); }
        else { sealed.insert(b.index); sc.on_seal_done(b.index, b.lane, now); }
    }
    CHECK(sc.finished() && sealed.size() == n && proven.size() == plan.nodes.size() && sc.root_done);
    CHECK(sc.t_leaves_done > 0 && sc.bottom_done == plan.n_bottom);
}

static void test_failure_paths() {
    // no fold: 6 segments, 2 lanes, one retry allowed
    Scheduler sc(6, 2, nullptr, true, 1);
    CHECK(sc.take_segment(0, 0.0, true).index == 0 && sc.take_segment(1, 0.0, true).index == 1);
    CHECK(sc.on_seal_failed(0, 0, 0.0) == Scheduler::Failure::Retry && sc.n_retries == 1 && sc.retries_waiting());
    // the lane that failed it gets the NEXT index, not the retry (another lane is alive, 50 ms have not passed) ...
    CHECK(sc.take_segment(0, 0.01, true).index == 2);
    // ... the other lane takes the retry first
    CHECK(sc.take_segment(1, 0.01, true).index == 0);
    sc.on_seal_done(0, 1, 0.02);
    sc.on_seal_done(1, 1, 0.02);
    // a second failure in a row on lane 0 retires it
    CHECK(sc.on_seal_failed(2, 0, 0.03) == Scheduler::Failure::RetryAndRetireLane && sc.seal_lanes_active == 1);
    // nobody else picked segment 2 up: after 50 ms (or with one sealing lane left) anyone may, including the last lane standing
    CHECK(sc.take_segment(1, 0.04, true).index == 2);
    // a segment that fails again after its retry is fatal
    CHECK(sc.on_seal_failed(2, 1, 0.05) == Scheduler::Failure::Fatal && sc.attempts(2) == 2);
    // a held-back retry: only the failing lane asks, another lane exists -> held for 50 ms, then released
    Scheduler s2(3, 2, nullptr, true, 2);
    CHECK(s2.take_segment(0, 0.0, true).index == 0);
    CHECK(s2.on_seal_failed(0, 0, 1.0) == Scheduler::Failure::Retry);
    CHECK(s2.take_segment(0, 1.01, false).kind == Scheduler::Kind::None);       // (from_index false: a producer pipeline feeds this lane)
    CHECK(s2.take_segment(0, 1.06, false).index == 0);
    // producers claim indices; a segment prepared for a retired lane is requeued for anyone, at once
    CHECK(s2.claim_index() == 1 && s2.claim_index() == 2 && s2.claim_index() == NONE && !s2.indices_left());
    s2.requeue(2);
    CHECK(s2.take_segment(0, 1.07, false).index == 2);
    // success resets the consecutive-failure count of a lane
    Scheduler s3(4, 2, nullptr, true, 3);
    CHECK(s3.on_seal_failed(0, 0, 0) == Scheduler::Failure::Retry);
    s3.on_seal_done(1, 0, 0);
    CHECK(s3.on_seal_failed(2, 0, 0) == Scheduler::Failure::Retry && s3.seal_lanes_active == 2);
    // a lone sealing lane is never retired
    Scheduler s4(2, 1, nullptr, true, 5);
    CHECK(s4.on_seal_failed(0, 0, 0) == Scheduler::Failure::Retry && s4.on_seal_failed(0, 0, 0) == Scheduler::Failure::Retry && s4.seal_lanes_active == 1);
    CHECK(s4.take_segment(0, 0.0, true).index == 0);                              // and takes its own retries back
}

int main() {
    test_plan_shapes();
    for (unsigned seed = 0; seed < 40; seed++) {
        simulate(1 + seed % 23, 1 + seed % 3, seed % 4, true, seed);
        simulate(1 + seed % 23, 1 + seed % 3, seed % 4, false, seed);
    }
    simulate(1024, 3, 3, true, 7);
    for (unsigned seed = 0; seed < 24; seed++) {                                      // sessions with assumption receipts: union tree + resolve
        simulate(1 + seed % 11, 1 + seed % 3, seed % 3, true, seed, 1 + seed % 6);
        simulate(1 + seed % 11, 1 + seed % 3, seed % 3, false, seed, 1 + seed % 6);
    }
    test_failure_paths();
    if (failures) { printf("%d check(s) failed\n", failures); return 1; }
    printf("scheduler ok\n");
    return 0;
}
