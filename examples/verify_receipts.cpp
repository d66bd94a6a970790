// examples/verify_receipts.cpp — the VERIFIER side of a block as a plain C++ host: no GPU, no Python.
//
// `receipt.verify(image_id)` (/root/reference/crates/host/src/bin/cli.rs:103) runs wherever the receipt is checked, which is
// not where it was proven: a verifier has no prover hardware.  This tool takes the receipt containers a proving host wrote
// (examples/seal_segments --receipts-dir DIR: segment_<i>.zkr, zkh_receipt_encode) and checks the composite:
//   every container's envelope (zkh_receipt_decode: checksum, circuit hash, po2, claim digest),
//   segment order (index i at position i),
//   every seal against the control root THE VERIFIER expects for its size (zkh_verify_segment, host arithmetic only) — never the
//   root a container carries,
//   with --chained (implied by a SYN-S circuit) the continuity of the session (SYN-C / SYN-S circuits: the first segment starts from --initial-state, every
//   segment's pre-state is its predecessor's post-state: CompositeReceipt::verify_integrity),
//   and that the session is WHOLE: a SYN-S circuit ("syn_session") binds an exit code and the journal's digest in every seal —
//   SystemSplit .. SystemSplit, Halted(0) + SHA-256(journal) — so trailing segments cannot be dropped and the journal cannot be
//   rewritten (zkh_session_check_termination; --journal HEX gives the journal bytes, default: the final state word); a SYN-C circuit
//   binds none of that, so --chained then REQUIRES --segments N, the number of segments the verifier expects.
//
//   verify_receipts (--desc FILE | --circuit NAME) --receipts-dir DIR --control-root PO2:HEX64 [--control-root PO2:HEX64 ...]
//                   [--chained [--initial-state N]] [--segments N] [--journal HEX] [--assumption CLAIMHEX64:ROOTHEX64 ...]
// --assumption (repeatable, in the session's order): claim digest and control root (64 hex digits each, word 0 first) of a receipt the
// session ASSUMED (a keccak batch).  A SYN-S session's last seal binds Output{journal, assumptions}: the list given here must be the one
// the session names, or the session is refused — it cannot be resolved against other receipts.
// PO2:HEX64 = the segment size and the 8 words of the expected control root as 64 hex digits (word 0 first, as
// `python -m zeth_amd.prover` / circuits/control_roots.json print them and zkh_syn_control_root returns them).
// Exit code 0 and {"verified": N} on success; 1 and the reason on the first receipt that does not verify.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "zkhal.h"

static bool read_words(const std::string& path, std::vector<uint32_t>& out) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    uint32_t w;
    while (fread(&w, 4, 1, f) == 1) out.push_back(w);
    fclose(f);
    return !out.empty();
}

int main(int argc, char** argv) {
    std::string desc_path, circuit_name, dir;
    std::map<uint32_t, std::vector<uint32_t>> roots;
    bool chained = false, have_journal = false;
    uint32_t initial_state = 0;
    long expect_segments = -1;
    std::vector<uint8_t> journal;
    std::vector<uint32_t> assum_claims, assum_roots;
    auto hex8 = [](const std::string& h, std::vector<uint32_t>& out) {
        if (h.size() != 64) return false;
        for (int k = 0; k < 8; k++) out.push_back((uint32_t)strtoul(h.substr(8 * k, 8).c_str(), nullptr, 16));
        return true;
    };
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        if (a == "--desc" && i + 1 < argc) desc_path = argv[++i];
        else if (a == "--circuit" && i + 1 < argc) circuit_name = argv[++i];
        else if (a == "--receipts-dir" && i + 1 < argc) dir = argv[++i];
        else if (a == "--chained") chained = true;
        else if (a == "--initial-state" && i + 1 < argc) initial_state = (uint32_t)strtoul(argv[++i], nullptr, 0);
        else if (a == "--segments" && i + 1 < argc) expect_segments = strtol(argv[++i], nullptr, 0);
        else if (a == "--assumption" && i + 1 < argc) {
            const std::string v = argv[++i];
            const size_t colon = v.find(':');
            if (colon == std::string::npos || !hex8(v.substr(0, colon), assum_claims) || !hex8(v.substr(colon + 1), assum_roots)) {
                fprintf(stderr, "--assumption wants CLAIMHEX64:ROOTHEX64\n");
                return 2;
            }
        }
        else if (a == "--journal" && i + 1 < argc) {
            const std::string hex = argv[++i];
            if (hex.size() % 2) { fprintf(stderr, "--journal wants an even number of hex digits\n"); return 2; }
            for (size_t k = 0; k < hex.size(); k += 2) journal.push_back((uint8_t)strtoul(hex.substr(k, 2).c_str(), nullptr, 16));
            have_journal = true;
        }
        else if (a == "--control-root" && i + 1 < argc) {
            unsigned po2 = 0;
            char hex[65] = {0};
            if (sscanf(argv[++i], "%u:%64[0-9a-fA-F]", &po2, hex) != 2 || strlen(hex) != 64) { fprintf(stderr, "--control-root wants PO2:HEX64\n"); return 2; }
            std::vector<uint32_t> r(8);
            for (int k = 0; k < 8; k++) { char w[9] = {0}; memcpy(w, hex + 8 * k, 8); r[k] = (uint32_t)strtoul(w, nullptr, 16); }
            roots[po2] = r;
        } else { fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
    }
    std::vector<uint32_t> desc;
    if (!circuit_name.empty()) {
        const uint32_t* w = nullptr;
        size_t nw = 0;
        if (const char* e = zkh_shipped_circuit_desc(circuit_name.c_str(), &w, &nw)) { fprintf(stderr, "%s\n", e); zkh_free_error(e); return 2; }
        desc.assign(w, w + nw);
    } else if (!desc_path.empty()) read_words(desc_path, desc);
    if (desc.empty() || dir.empty() || roots.empty()) {
        fprintf(stderr, "usage: %s (--desc FILE | --circuit NAME) --receipts-dir DIR --control-root PO2:HEX64 [...] [--chained [--initial-state N]] [--segments N] [--journal HEX]\n", argv[0]);
        return 2;
    }
    zkh_circuit* circuit = nullptr;                        // ctx == NULL: a host-only circuit, enough for everything a verifier does
    const char* err = zkh_circuit_load(nullptr, desc.data(), desc.size(), &circuit);
    if (err) { fprintf(stderr, "zkh_circuit_load: %s\n", err); zkh_free_error(err); return 1; }
    const bool session = desc.size() > 13 && desc[13] == 1 && desc[7] == 23;      // SYN-S: exit code + journal digest bound in every seal
    // A session circuit's seals carry the state words: continuity and the initial state are ALWAYS checked for it (a directory
    // holding segments 0..k of one session followed by the halting segment of another must not verify); --chained is implied.
    if (session) chained = true;
    if (chained && !session && expect_segments < 0) {
        fprintf(stderr, "REJECTED: --chained on a circuit that binds no exit code (SYN-C) needs --segments N: without it a session with its trailing "
                        "segments cut off would verify (use the syn_session circuit to bind termination in the seals)\n");
        return 1;
    }
    std::vector<std::vector<uint32_t>> kept;                                     // SYN-S: the verified seals, for the termination check
    size_t verified = 0;
    uint32_t prev_post = 0;
    bool have_prev = false;
    auto reject = [&](size_t i, const char* what, const char* why) {
        fprintf(stderr, "REJECTED: segment %zu: %s%s%s\n", i, what, why ? ": " : "", why ? why : "");
        return 1;
    };
    for (size_t i = 0;; i++) {
        std::vector<uint32_t> blob;
        if (!read_words(dir + "/segment_" + std::to_string(i) + ".zkr", blob)) break;
        uint32_t info[26];
        size_t off = 0;
        if ((err = zkh_receipt_decode(circuit, blob.data(), blob.size(), info, &off))) { const int rc = reject(i, "container", err); zkh_free_error(err); return rc; }
        if (info[7] != i) return reject(i, "the container holds another segment index (segments out of order)", nullptr);
        auto it = roots.find(info[4]);
        if (it == roots.end()) return reject(i, "no expected control root was given for its size", nullptr);
        // the claim digest a container carries was computed with the root IT names: both must be what this verifier expects
        if (memcmp(info + 10, it->second.data(), 32) != 0) return reject(i, "the container was made under another control root than the expected one", nullptr);
        if ((err = zkh_verify_segment(circuit, blob.data() + off, info[9], it->second.data(), nullptr, nullptr))) { const int rc = reject(i, "seal", err); zkh_free_error(err); return rc; }
        if (chained) {
            // SYN-C: out = (post, 0, 0, 0, pre) as the first words of the seal (Montgomery form); the initial state is a canonical residue
            const uint32_t* seal = blob.data() + off;
            if (info[8] != 5 && info[8] != 23) return reject(i, "--chained: the circuit's segments carry no state words", nullptr);
            uint32_t want = prev_post;
            if (!have_prev) {
                const uint64_t R = ((uint64_t)1 << 32) % 2013265921ull;
                want = (uint32_t)((uint64_t)(initial_state % 2013265921u) * R % 2013265921ull);
            }
            if (seal[4] != want) return reject(i, "the session is not continuous: the segment does not start from its predecessor's post-state", nullptr);
            prev_post = seal[0];
            have_prev = true;
        }
        if (session) kept.emplace_back(blob.begin() + off, blob.begin() + off + info[9]);
        verified++;
    }
    if (!verified) { fprintf(stderr, "no segment_<i>.zkr under %s\n", dir.c_str()); return 1; }
    if (expect_segments >= 0 && (long)verified != expect_segments) {
        fprintf(stderr, "REJECTED: %zu segment receipts found, the session has %ld\n", verified, expect_segments);
        return 1;
    }
    if (session) {          // the session must END here: SystemSplit .. SystemSplit, Halted(0), and the journal's digest in the last seal
        std::vector<const uint32_t*> ptrs;
        std::vector<size_t> words;
        static const uint8_t empty_journal = 0;          // --journal "": an EXPLICITLY empty journal (a NULL pointer would mean "the default journal")
        for (auto& s : kept) { ptrs.push_back(s.data()); words.push_back(s.size()); }
        uint32_t ad[8];
        const bool assumes = !assum_claims.empty();
        if (assumes) zkh_assumptions_digest(assum_claims.data(), assum_roots.data(), assum_claims.size() / 8, ad);
        if ((err = zkh_session_check_output(circuit, ptrs.data(), words.data(), ptrs.size(), have_journal ? (journal.empty() ? &empty_journal : journal.data()) : nullptr, journal.size(),
                                            assumes ? ad : nullptr))) {
            fprintf(stderr, "REJECTED: %s\n", err);
            zkh_free_error(err);
            return 1;
        }
    }
    zkh_circuit_destroy(circuit);
    printf("{\"driver\": \"verify_receipts\", \"library\": \"%s\", \"verified\": %zu, \"chained\": %s, \"gpu\": false}\n", zkh_version(), verified, chained ? "true" : "false");
    return 0;
}
