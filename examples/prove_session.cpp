// prove_session — `default_prover().prove(env, elf)` as ONE library call (plain C ABI, compiled with g++).
//
// zeth's whole prove path is that one call (/root/reference/crates/host/src/lib.rs:137); behind it risc0-zkvm 3.0.3's
// ProverServer::prove_session seals the session's segments and (with ProverOpts::succinct) folds them to one receipt, and
// /root/reference/crates/host/src/bin/cli.rs:103 verifies the result.  Here: zkh_session_create -> zkh_session_prove ->
// zkh_session_verify (csrc/session.hip): the segment loop, the lanes, the join tree and the verification all run inside the
// library; this file only builds the segment list and prints what came back.  (examples/seal_segments.cpp is the same session
// written out against the low-level entry points.)
//
//   prove_session (--desc syn_a.desc | --circuit syn_a) [--join-desc p2_join.desc | --recursion-dir DIR | --build-recursion]
//                 [--po2 20] [--tail-po2 18] [--segments 64]
//                 [--devices 1] [--inflight 3] [--join-po2 18] [--noise-seed N] [--two-phase] [--recompute-code] [--no-join3]
//                 [--csv FILE [--block-number N] [--gas-used N]] [--keccak-batches N [--keccak-po2 P]]
//                 [--chained [--initial-state N] [--journal HEX]] [--receipts-dir DIR]
// --chained (SYN-C / SYN-S circuits, e.g. --circuit syn_session): the library's executor pass gives every segment its pre-state (and, SYN-S, its
// exit code; the last seal binds Output{SHA-256(journal), assumptions}) — zkh_session_set_chained; --journal HEX = the bytes the guest
// commits, for zeth the 32-byte block hash (/root/reference/guests/stateless-client/src/lib.rs:33) — zkh_session_set_journal; default: the
// session's final state word.  --receipts-dir DIR writes segment_<i>.zkr (zkh_receipt_encode) and prints the control roots, so that
// `verify_receipts --circuit syn_session --receipts-dir DIR --control-root PO2:HEX .. --initial-state N --journal HEX` is the verifier's side
// of /root/reference/crates/host/src/bin/cli.rs:103-107 with no GPU.
// --keccak-batches N (with --build-recursion): the session ASSUMES N keccak batch receipts (KECCAK-F seals proven first, the shape
// upstream's prove_keccak leaves behind): zkh_session_set_assumptions -> they are lifted, united pairwise (sorted pairs) and the
// session's root is RESOLVED against the union root; the CSV's keccak_calls column counts their permutations.
// --csv appends one row in the reference's stats vocabulary (/root/reference/run-parallel.sh:15 writes the header
// "block_number,execution_time,total_cycles,user_cycles,paging_cycles,keccak_calls,gas_used" and :53-70 scrape the columns out of a
// dev-mode prove): execution_time = this session's wall-clock, total_cycles = sum of 2^po2 over the segments, user_cycles = their
// active rows (2^po2 - ZK_CYCLES), paging_cycles and keccak_calls 0 (these circuits page nothing and call no accelerator).
// --circuit NAME: a circuit description compiled into the library (zkh_shipped_circuit_desc) instead of a file.
// --build-recursion: no files at all — the lift / lift2 / join / join3 (and union / resolve) programs of this block are BUILT here, in-process, by the
// library (zkh_session_build_recursion -> zkh_rec_build_program: this library's STARK verifier restated for the RECURSION circuit) from the
// segment circuit's control roots (computed on the GPU) and the built-in RECURSION description: nothing in the run needs Python.
// --recursion-dir: the directory `python -m zeth_amd.circuits.rec_verify DIR` and `python -m zeth_amd.circuits.recursion
// DIR/recursion.desc` wrote (lift-<po2>.zkr1, lift2-<l>-<r>.zkr1, join-<l>-<r>.zkr1, join3-<a>-<b>-<c>.zkr1): lift the receipts (in pairs: lift2) and join them to one root receipt whose
// every node verified its child seal(s) IN-CIRCUIT (BASELINE.json config 5).
#include <cstdint>
#include <cstdio>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <dirent.h>

#include "zkhal.h"

static bool read_words(const std::string& path, std::vector<uint32_t>& out) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) { perror(path.c_str()); return false; }
    uint32_t w;
    while (fread(&w, 4, 1, f) == 1) out.push_back(w);
    fclose(f);
    return out.size() >= 16;
}

int main(int argc, char** argv) {
    std::string desc_path, join_path, rec_dir, circuit_name, csv_path, gas_used = "N/A";
    unsigned long long block_number = 0;
    bool build_recursion = false;
    size_t po2 = 20, tail_po2 = 18, n = 64, devices = 1, inflight = 3, join_po2 = 18;
    uint64_t noise = 0;
    bool two_phase = false, recompute_code = false, no_join3 = false;
    size_t keccak_batches = 0, keccak_po2 = 13;
    bool chained = false, have_journal = false;
    size_t initial_state = 0;
    std::vector<uint8_t> journal;
    std::string receipts_dir;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto num = [&](size_t& d) { if (i + 1 < argc) d = strtoull(argv[++i], nullptr, 10); };
        if (a == "--desc" && i + 1 < argc) desc_path = argv[++i];
        else if (a == "--circuit" && i + 1 < argc) circuit_name = argv[++i];
        else if (a == "--build-recursion") build_recursion = true;
        else if (a == "--join-desc" && i + 1 < argc) join_path = argv[++i];
        else if (a == "--recursion-dir" && i + 1 < argc) rec_dir = argv[++i];
        else if (a == "--po2") num(po2);
        else if (a == "--tail-po2") num(tail_po2);
        else if (a == "--segments") num(n);
        else if (a == "--devices") num(devices);
        else if (a == "--csv" && i + 1 < argc) csv_path = argv[++i];
        else if (a == "--block-number" && i + 1 < argc) block_number = strtoull(argv[++i], nullptr, 0);
        else if (a == "--gas-used" && i + 1 < argc) gas_used = argv[++i];
        else if (a == "--inflight") num(inflight);
        else if (a == "--join-po2") num(join_po2);
        else if (a == "--noise-seed" && i + 1 < argc) noise = strtoull(argv[++i], nullptr, 0);
        else if (a == "--keccak-batches") num(keccak_batches);
        else if (a == "--keccak-po2") num(keccak_po2);
        else if (a == "--chained") chained = true;
        else if (a == "--initial-state") num(initial_state);
        else if (a == "--receipts-dir" && i + 1 < argc) receipts_dir = argv[++i];
        else if (a == "--journal" && i + 1 < argc) {
            const std::string hex = argv[++i];
            if (hex.size() % 2) { fprintf(stderr, "--journal wants an even number of hex digits\n"); return 2; }
            for (size_t k = 0; k < hex.size(); k += 2) journal.push_back((uint8_t)strtoul(hex.substr(k, 2).c_str(), nullptr, 16));
            have_journal = true;
        }
        else if (a == "--no-join3") no_join3 = true;            // leave the join3 program out: three nodes cost two proofs
        else if (a == "--two-phase") two_phase = true;          // seal everything, then fold (default: one pipeline)
        else if (a == "--recompute-code") recompute_code = true; // re-commit the code group per segment, like upstream's SegmentProver
        else { fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
    }
    std::vector<uint32_t> desc, jdesc;
    if (!circuit_name.empty()) {
        const uint32_t* w = nullptr;
        size_t nw = 0;
        if (const char* e = zkh_shipped_circuit_desc(circuit_name.c_str(), &w, &nw)) { fprintf(stderr, "%s\n", e); zkh_free_error(e); return 2; }
        desc.assign(w, w + nw);
    }
    if ((desc.empty() && (desc_path.empty() || !read_words(desc_path, desc))) || (!join_path.empty() && !read_words(join_path, jdesc))) {
        fprintf(stderr, "usage: %s (--desc FILE | --circuit NAME) [--join-desc FILE | --recursion-dir DIR | --build-recursion] [--po2 N] [--tail-po2 N] [--segments S] [--devices G] [--inflight K] [--join-po2 N] [--noise-seed N]\n", argv[0]);
        return 2;
    }
    std::vector<int> devs(devices);
    for (size_t d = 0; d < devices; d++) devs[d] = (int)d;
    zkh_session* session = nullptr;
    const char* err = zkh_session_create(devs.data(), devs.size(), inflight, desc.data(), desc.size(), jdesc.empty() ? nullptr : jdesc.data(), jdesc.size(), &session);
    if (err) { fprintf(stderr, "zkh_session_create: %s\n", err); zkh_free_error(err); return 1; }
    if (!rec_dir.empty()) {
        // the lift / join programs: every *.zkr1 of the directory, kind from the file name
        std::vector<uint32_t> rdesc;
        if (!read_words(rec_dir + "/recursion.desc", rdesc)) return 2;
        std::vector<std::vector<uint32_t>> blobs;
        std::vector<uint32_t> kinds;
        DIR* d = opendir(rec_dir.c_str());
        if (!d) { perror(rec_dir.c_str()); return 2; }
        std::vector<std::string> names;
        while (dirent* e = readdir(d)) names.push_back(e->d_name);
        closedir(d);
        for (const std::string& name : names) {
            unsigned a = 0, b = 0, c3 = 0;
            uint32_t kind;
            if (name.find(".zkr1") == std::string::npos) continue;
            if (sscanf(name.c_str(), "join3-%u-%u-%u.zkr1", &a, &b, &c3) == 3) {
                if (no_join3) continue;                   // --no-join3: groups of three are proven as join(join(a, b), c)
                if (a != b) { fprintf(stderr, "%s: a join3 whose first two children differ in size\n", name.c_str()); return 2; }
                kind = 3; b = c3;
            }
            else if (sscanf(name.c_str(), "lift2-%u-%u.zkr1", &a, &b) == 2) kind = 2;
            else if (sscanf(name.c_str(), "lift-%u.zkr1", &a) == 1) { kind = 0; b = 0; }
            else if (sscanf(name.c_str(), "join-%u-%u.zkr1", &a, &b) == 2) kind = 1;
            else continue;
            blobs.emplace_back();
            if (!read_words(rec_dir + "/" + name, blobs.back())) return 2;
            kinds.insert(kinds.end(), {kind, a, b});
        }
        std::vector<const uint32_t*> ptrs;
        std::vector<size_t> words;
        for (auto& bl : blobs) { ptrs.push_back(bl.data()); words.push_back(bl.size()); }
        err = zkh_session_set_recursion(session, rdesc.data(), rdesc.size(), ptrs.data(), words.data(), kinds.data(), blobs.size());
        if (err) { fprintf(stderr, "zkh_session_set_recursion: %s\n", err); zkh_free_error(err); return 1; }
    }
    // ---- assumption receipts: N keccak batches sealed by a session of their own, handed to the block's session (verified there) ----
    zkh_prove_info kinfo;
    memset(&kinfo, 0, sizeof kinfo);
    if (keccak_batches) {
        if (!build_recursion) { fprintf(stderr, "--keccak-batches needs --build-recursion (the union / resolve programs are built for the session)\n"); return 2; }
        const uint32_t* kw = nullptr;
        size_t knw = 0;
        if (const char* e = zkh_shipped_circuit_desc("keccak_f", &kw, &knw)) { fprintf(stderr, "%s\n", e); zkh_free_error(e); return 2; }
        zkh_session* ks = nullptr;
        err = zkh_session_create(devs.data(), 1, 1, kw, knw, nullptr, 0, &ks);
        if (err) { fprintf(stderr, "zkh_session_create (keccak): %s\n", err); zkh_free_error(err); return 1; }
        std::vector<zkh_segment> ksegs(keccak_batches);
        for (size_t i = 0; i < keccak_batches; i++) {
            memset(&ksegs[i], 0, sizeof ksegs[i]);
            ksegs[i].po2 = (uint32_t)keccak_po2;
            ksegs[i].seed = 0xCECCull + i;
            ksegs[i].noise_key[0] = (uint32_t)noise; ksegs[i].noise_key[1] = (uint32_t)(noise >> 32);
        }
        err = zkh_session_prove(ks, ksegs.data(), keccak_batches, 0, 0, nullptr, &kinfo);
        if (err) { fprintf(stderr, "zkh_session_prove (keccak): %s\n", err); zkh_free_error(err); return 1; }
        zkh_session_destroy(ks);
        // the keccak circuit's control root at that size (a deployment ships it: upstream's control IDs)
        zkh_ctx* kc = nullptr; zkh_circuit* kcir = nullptr; zkh_prover* kp = nullptr;
        uint32_t kroot[8];
        if ((err = zkh_ctx_create(devs[0], "poseidon2", &kc)) || (err = zkh_circuit_load(kc, kw, knw, &kcir)) || (err = zkh_prover_create(kc, kcir, &kp)) ||
            (err = zkh_syn_control_root(kp, keccak_po2, ZKH_ZK_CYCLES, kroot))) { fprintf(stderr, "keccak control root: %s\n", err); zkh_free_error(err); return 1; }
        zkh_prover_destroy(kp); zkh_circuit_destroy(kcir); zkh_ctx_destroy(kc);
        std::vector<uint32_t> kroots, kpo2s(keccak_batches, (uint32_t)keccak_po2);
        for (size_t i = 0; i < keccak_batches; i++) kroots.insert(kroots.end(), kroot, kroot + 8);
        err = zkh_session_set_assumptions(session, kw, knw, kinfo.seals, kinfo.seal_words, kpo2s.data(), kroots.data(), keccak_batches);
        if (err) { fprintf(stderr, "zkh_session_set_assumptions: %s\n", err); zkh_free_error(err); return 1; }
    }
    double build_s = 0;
    size_t n_built = 0;
    if (build_recursion) {
        // the program set of this block, built and loaded by the library (csrc/rec_builder.hip + the compiled-in RECURSION description)
        const uint32_t tail = (uint32_t)(tail_po2 < po2 ? tail_po2 : po2);
        std::vector<uint32_t> sizes = {(uint32_t)po2};
        if (tail != po2 && n > 1) sizes.push_back(tail);
        const auto t_b = std::chrono::steady_clock::now();
        err = zkh_session_build_recursion(session, sizes.data(), sizes.size(), no_join3 ? 0 : 1);
        if (err) { fprintf(stderr, "zkh_session_build_recursion: %s\n", err); zkh_free_error(err); return 1; }
        build_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_b).count();     // control roots + build + load on every lane
        n_built = 1;
    }
    const bool recursive = !rec_dir.empty() || build_recursion;
    if (two_phase) zkh_session_set_streamed_fold(session, 0);
    if (recompute_code) zkh_session_set_resident_code(session, 0);
    if (chained) {
        err = zkh_session_set_chained(session, 1, (uint32_t)initial_state);
        if (err) { fprintf(stderr, "zkh_session_set_chained: %s\n", err); zkh_free_error(err); return 1; }
    }
    if (have_journal) {
        if (!chained) { fprintf(stderr, "--journal needs --chained (the executor's pass is what writes the output digest into the last segment)\n"); return 2; }
        static const uint8_t none = 0;                     // an empty journal is a journal: the pointer stays non-NULL
        err = zkh_session_set_journal(session, journal.empty() ? &none : journal.data(), journal.size());
        if (err) { fprintf(stderr, "zkh_session_set_journal: %s\n", err); zkh_free_error(err); return 1; }
    }
    // the session's segment list: S distinct segments, the last one the short tail
    std::vector<zkh_segment> segs(n);
    for (size_t i = 0; i < n; i++) {
        memset(&segs[i], 0, sizeof segs[i]);
        segs[i].po2 = (uint32_t)((i + 1 == n && n > 1) ? (tail_po2 < po2 ? tail_po2 : po2) : po2);
        segs[i].seed = 0x5EED0000ull + i;
        segs[i].noise_key[0] = (uint32_t)noise;           // all-zero: a fresh 256-bit OS key per segment, like upstream; --noise-seed N: (N lo, N hi, 0, ..)
        segs[i].noise_key[1] = (uint32_t)(noise >> 32);
    }
    zkh_prove_info info;
    const uint32_t join_key[8] = {(uint32_t)noise, (uint32_t)(noise >> 32), 0, 0, 0, 0, 0, 0};
    err = zkh_session_prove(session, segs.data(), n, recursive ? 2 : !jdesc.empty(), join_po2, noise ? join_key : nullptr, &info);
    if (err) { fprintf(stderr, "zkh_session_prove: %s\n", err); zkh_free_error(err); return 1; }
    err = zkh_session_verify(session, segs.data(), &info, join_po2);
    if (err) { fprintf(stderr, "REJECTED: %s\n", err); zkh_free_error(err); return 1; }
    size_t words = 0;
    for (size_t i = 0; i < info.n_segments; i++) words += info.seal_words[i];
    std::string root_out;                                  // the root receipt's public output (claim ‖ allowed-programs root), hex words
    for (size_t i = 0; i < 16 && i < info.root_seal_words; i++) { char h[12]; snprintf(h, sizeof h, "%08x", info.root_seal[i]); root_out += h; }
    printf("{\"driver\": \"prove_session\", \"library\": \"%s\", \"segments\": %zu, \"po2\": %zu, \"tail_po2\": %u, \"lanes\": %zu, "
           "\"wall_s\": %.4f, \"leaves_s\": %.4f, \"segments_per_s\": %.3f, \"witgen_ms_per_segment\": %.3f, \"lifts\": %zu, \"lift_s\": %.4f, "
           "\"joins\": %zu, \"join_tree_s\": %.4f, \"in_circuit_verification\": %s, \"root_receipt_words\": %zu, \"seal_words_total\": %zu, "
           "\"streamed_fold\": %s, \"fold_tail_s\": %.4f, \"fold_busy_lane_s\": %.3f, \"segment_retries\": %zu, "
           "\"programs_built_by_library\": %s, \"program_build_and_load_s\": %.3f, \"assumption_receipts\": %zu, \"resolved\": %s, \"root_out\": \"%s\", \"verified\": true}\n",
           zkh_version(), n, po2, segs[n - 1].po2, zkh_session_lanes(session), info.wall_s, info.leaves_s, n / info.leaves_s,
           1e3 * info.witgen_s_sum / n, info.n_lifts, info.lift_s, info.n_joins, info.join_s, info.n_lifts ? "true" : "false",
           info.root_seal_words, words, info.streamed ? "true" : "false", info.fold_tail_s, info.fold_busy_s_sum, info.n_retries, n_built ? "true" : "false", build_s, keccak_batches, keccak_batches ? "true" : "false", root_out.c_str());
    if (!receipts_dir.empty()) {
        // the receipts a verifier WITHOUT a GPU checks (examples/verify_receipts.cpp): one container per segment; the control root of each
        // size comes from a prover of the circuit (a deployment ships them: upstream's control IDs)
        zkh_ctx* c = nullptr; zkh_circuit* cir = nullptr; zkh_prover* p = nullptr;
        if ((err = zkh_ctx_create(devs[0], "poseidon2", &c)) || (err = zkh_circuit_load(c, desc.data(), desc.size(), &cir)) || (err = zkh_prover_create(c, cir, &p))) {
            fprintf(stderr, "receipts: %s\n", err); zkh_free_error(err); return 1;
        }
        uint32_t roots[32][8];
        bool have[32] = {false};
        for (size_t i = 0; i < info.n_segments; i++) {
            const uint32_t q = segs[i].po2;
            if (!have[q]) {
                if ((err = zkh_syn_control_root(p, q, ZKH_ZK_CYCLES, roots[q]))) { fprintf(stderr, "control root: %s\n", err); zkh_free_error(err); return 1; }
                have[q] = true;
                fprintf(stderr, "control-root %u:", q);
                for (int k = 0; k < 8; k++) fprintf(stderr, "%08x", roots[q][k]);
                fprintf(stderr, "\n");
            }
            uint32_t* blob = nullptr;
            size_t bw = 0;
            if ((err = zkh_receipt_encode(cir, info.seals[i], info.seal_words[i], (uint32_t)i, roots[q], &blob, &bw))) {
                fprintf(stderr, "zkh_receipt_encode: %s\n", err); zkh_free_error(err); return 1;
            }
            const std::string path = receipts_dir + "/segment_" + std::to_string(i) + ".zkr";
            FILE* f = fopen(path.c_str(), "wb");
            if (!f || fwrite(blob, 4, bw, f) != bw) { perror(path.c_str()); return 1; }
            fclose(f);
            zkh_free_seal(blob);
        }
        zkh_prover_destroy(p); zkh_circuit_destroy(cir); zkh_ctx_destroy(c);
    }
    if (!csv_path.empty()) {
        unsigned long long total = 0, user = 0;
        for (size_t i = 0; i < n; i++) { total += 1ull << segs[i].po2; user += (1ull << segs[i].po2) - ZKH_ZK_CYCLES; }
        FILE* probe = fopen(csv_path.c_str(), "r");
        const bool fresh = probe == nullptr;
        if (probe) fclose(probe);
        if (FILE* f = fopen(csv_path.c_str(), "a")) {
            if (fresh) fprintf(f, "block_number,execution_time,total_cycles,user_cycles,paging_cycles,keccak_calls,gas_used\n");
            // keccak_calls: the permutations the assumed batches prove (25 rows each; upstream counts the guest's accelerator calls)
            const unsigned long long kcalls = keccak_batches * (((1ull << keccak_po2) - ZKH_ZK_CYCLES) / 25);
            fprintf(f, "%llu,%.6f,%llu,%llu,0,%llu,%s\n", block_number, info.wall_s, total, user, kcalls, gas_used.c_str());
            fclose(f);
        } else {
            fprintf(stderr, "cannot write %s\n", csv_path.c_str());
        }
    }
    zkh_prove_info_free(&info);
    zkh_prove_info_free(&kinfo);
    zkh_session_destroy(session);
    return 0;
}
