// seal_segments — a host driver written against include/zkhal.h only (plain C ABI: no HIP, torch or Python in this
// translation unit; it is compiled with g++).  It is the per-session loop that risc0-zkvm 3.0.3's
// ProverImpl::prove_session runs and that zeth reaches from /root/reference/crates/host/src/lib.rs:137
// (`default_prover().prove(env, elf)`): every segment of a session is sealed independently, here spread over G devices
// with K seals in flight per device through one shared work index (segment i -> next free lane, no
// exchange between devices), and every seal is then checked by the host verifier, the analogue of
// `receipt.verify(image_id)` at /root/reference/crates/host/src/bin/cli.rs:103.
//
//   seal_segments --desc syn_a.desc [--po2 20] [--segments 8] [--devices 1] [--inflight 3] [--no-verify] [--noise-seed N]
//                 [--resident-code-group]
//                 [--join-desc p2_join.desc [--join-po2 18]]   (BASELINE config 5: fold the session's receipts through the
//                                           P2-JOIN tree to ONE root receipt — a join constrains parent = Poseidon2
//                                           hash_pair(claim_left, claim_right) in-circuit — then verify the compact
//                                           receipt: root seal + the claim tree recomputed on the host from the leaves)
//                 [--receipts-dir DIR]     (writes segment_<i>.zkr: the receipt container of zkh_receipt_encode)
//                 [--code-objects DIR]     (eval_check kernels of a circuit that is not built in: the .hsaco files +
//                                           manifest.txt written by `python -m zeth_amd.circuits.jit circuit.desc DIR`,
//                                           attached through zkh_circuit_attach_code_object_part — no Python at run time)
//
// The circuit description blob is what zeth_amd/circuits/desc.py serialises (`python -m zeth_amd.circuits.syn_air syn_a syn_a.desc`).
// Witnesses are the declared-synthetic SYN-AIR traces generated on the device (zkh_syn_witgen); with the real rv32im
// circuit the two trace buffers would come from the executor's preflight instead.
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <sys/random.h>

#include "zkhal.h"

namespace {

struct Options {
    std::string desc_path;
    size_t po2 = 20, segments = 8, devices = 1, inflight = 3;
    bool verify = true;
    std::string receipts_dir;        // --receipts-dir: one receipt container per segment
    std::string code_objects_dir;    // --code-objects: generated eval_check kernels to attach after loading the circuit
    bool fixed_noise = false;        // --noise-seed: reproducible seals (tests); default: fresh OS randomness per segment
    bool resident_code = false;      // --resident-code-group: commit the code group once per worker and keep it in HBM
    uint64_t noise_seed = 0;
    std::string join_desc_path;      // --join-desc: the P2-JOIN circuit; enables the join tree
    size_t join_po2 = 18;
};

// The zero-knowledge blinding rows must be unpredictable: upstream fills them from an OS RNG.  Here a NULL key tells the library to
// draw a fresh 256-bit ChaCha12 key from getrandom() for the call (include/zkhal.h, BLINDING ROWS); --noise-seed N fixes the key to
// (N lo, N hi, 0, ..) for reproducible seals (tests).
struct NoiseKey {
    uint32_t k[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    explicit NoiseKey(uint64_t seed) { k[0] = (uint32_t)seed; k[1] = (uint32_t)(seed >> 32); }
};

struct Receipt {
    std::vector<uint32_t> seal;
    double seal_s = 0;
    int device = -1;
};

std::mutex g_err_lock;
std::string g_first_error;

bool failed(const char* err, const char* what) {
    if (!err) return false;
    {
        std::lock_guard<std::mutex> lk(g_err_lock);
        if (g_first_error.empty()) g_first_error = std::string(what) + ": " + err;
    }
    zkh_free_error(err);
    return true;
}

double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// One lane = one context (device + stream) + circuit + prover; lanes of all devices pull from one work index.
// Attach the code objects listed in DIR/manifest.txt ("<part> <n_parts> <kernel> <file>" per line) to a loaded circuit.
bool attach_code_objects(zkh_circuit* circuit, const std::string& dir) {
    FILE* mf = fopen((dir + "/manifest.txt").c_str(), "r");
    if (!mf) { failed(strdup("cannot open manifest.txt"), dir.c_str()); return false; }
    unsigned part = 0, n_parts = 0;
    char kernel[256], file[256];
    bool ok = true;
    while (ok && fscanf(mf, "%u %u %255s %255s", &part, &n_parts, kernel, file) == 4) {
        FILE* f = fopen((dir + "/" + file).c_str(), "rb");
        if (!f) { failed(strdup("cannot open code object"), file); ok = false; break; }
        std::vector<char> image;
        char buf[65536];
        size_t got;
        while ((got = fread(buf, 1, sizeof buf, f)) > 0) image.insert(image.end(), buf, buf + got);
        fclose(f);
        ok = !failed(zkh_circuit_attach_code_object_part(circuit, image.data(), image.size(), kernel, part, n_parts), "zkh_circuit_attach_code_object_part");
    }
    fclose(mf);
    return ok && zkh_circuit_compiled_parts(circuit) > 0;
}

std::mutex g_root_lock;
bool g_have_root = false;
uint32_t g_control_root[8];

void lane(int device, const std::vector<uint32_t>& desc, const Options& opt, std::atomic<size_t>& next,
          std::vector<Receipt>& receipts) {
    zkh_ctx* ctx = nullptr;
    zkh_circuit* circuit = nullptr;
    zkh_prover* prover = nullptr;
    zkh_buf *code = nullptr, *data = nullptr;
    do {
        if (failed(zkh_ctx_create(device, "poseidon2", &ctx), "zkh_ctx_create")) break;
        if (failed(zkh_circuit_load(ctx, desc.data(), desc.size(), &circuit), "zkh_circuit_load")) break;
        if (!opt.code_objects_dir.empty() && !attach_code_objects(circuit, opt.code_objects_dir)) break;
        if (failed(zkh_prover_create(ctx, circuit, &prover), "zkh_prover_create")) break;
        const size_t n = (size_t)1 << opt.po2;
        const size_t w_code = desc[4], w_data = desc[5];          // header: magic, version, 3, W_accum, W_code, W_data
        if (failed(zkh_alloc(ctx, "code", w_code * n, 0, &code), "zkh_alloc(code)")) break;
        if (failed(zkh_alloc(ctx, "data", w_data * n, 0, &data), "zkh_alloc(data)")) break;
        {   // the control root of (circuit, po2): computed once per session, checked by the verifier for every seal
            std::lock_guard<std::mutex> lk(g_root_lock);
            if (!g_have_root) {
                if (failed(zkh_syn_control_root(prover, opt.po2, ZKH_ZK_CYCLES, g_control_root), "zkh_syn_control_root")) break;
                g_have_root = true;
            }
        }
        const size_t out_size = desc[7];
        std::vector<uint32_t> out_global(out_size), pub(out_size > 4 ? out_size - 4 : 0, 0u);
        bool code_resident = false;
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= opt.segments) break;
            const NoiseKey fixed(opt.noise_seed);
            const uint32_t* noise = opt.fixed_noise ? fixed.k : nullptr;
            if (failed(zkh_syn_witgen(ctx, circuit, opt.po2, ZKH_ZK_CYCLES, 0x5EED0000ull + i, noise, pub.empty() ? nullptr : pub.data(),
                                      code, data, out_global.data()),
                       "zkh_syn_witgen"))
                break;
            uint32_t* seal = nullptr;
            size_t words = 0;
            // the code trace of (circuit, po2) is the same for every segment: its committed form can stay resident
            if (opt.resident_code && !code_resident) {
                if (failed(zkh_prover_cache_code(prover, opt.po2, code), "zkh_prover_cache_code")) break;
                code_resident = true;
            }
            const double t0 = now_s();
            if (failed(zkh_prove_segment(prover, opt.po2, ZKH_ZK_CYCLES, noise, code_resident ? nullptr : code, data, out_global.data(),
                                         &seal, &words),
                       "zkh_prove_segment"))
                break;
            receipts[i].seal_s = now_s() - t0;
            receipts[i].seal.assign(seal, seal + words);
            receipts[i].device = device;
            zkh_free_seal(seal);
        }
    } while (false);
    if (code) zkh_release(code);
    if (data) zkh_release(data);
    if (prover) zkh_prover_destroy(prover);
    if (circuit) zkh_circuit_destroy(circuit);
    if (ctx) zkh_ctx_destroy(ctx);
}

// ---- the join tree (host.py join_schedule / JoinExecutor in C++): level l pairs nodes (2k, 2k+1) of level l-1, an unpaired
// last node is carried up; the joins of one level are independent and are pulled from a shared index by the lanes ----
using Claim = std::vector<uint32_t>;          // 8 words

bool hash_pair(const Claim& l, const Claim& r, Claim& out) {
    uint32_t st[24] = {0};
    memcpy(st, l.data(), 32);
    memcpy(st + 8, r.data(), 32);
    if (failed(zkh_poseidon2_mix_host(nullptr, nullptr, st, 1), "zkh_poseidon2_mix_host")) return false;
    out.assign(st, st + 8);
    return true;
}

struct JoinLane {
    zkh_ctx* ctx = nullptr;
    zkh_circuit* circuit = nullptr;
    zkh_prover* prover = nullptr;
    zkh_buf *code = nullptr, *data = nullptr;
    bool open(int device, const std::vector<uint32_t>& jdesc, size_t po2) {
        const size_t n = (size_t)1 << po2;
        return !failed(zkh_ctx_create(device, "poseidon2", &ctx), "zkh_ctx_create(join)") &&
               !failed(zkh_circuit_load(ctx, jdesc.data(), jdesc.size(), &circuit), "zkh_circuit_load(join)") &&
               !failed(zkh_prover_create(ctx, circuit, &prover), "zkh_prover_create(join)") &&
               !failed(zkh_alloc(ctx, "code", jdesc[4] * n, 0, &code), "zkh_alloc(join code)") &&
               !failed(zkh_alloc(ctx, "data", jdesc[5] * n, 0, &data), "zkh_alloc(join data)");
    }
    void close() {
        if (code) zkh_release(code);
        if (data) zkh_release(data);
        if (prover) zkh_prover_destroy(prover);
        if (circuit) zkh_circuit_destroy(circuit);
        if (ctx) zkh_ctx_destroy(ctx);
    }
};

struct JoinResult { size_t joins = 0; double seconds = 0; std::vector<uint32_t> root_seal; Claim root_claim; };

// claims: the leaf claims in segment order.  Returns false on error (g_first_error set).
bool join_tree(const Options& opt, const std::vector<uint32_t>& jdesc, std::vector<Claim> claims, JoinResult& res) {
    std::vector<JoinLane> lanes(opt.devices * opt.inflight);
    bool ok = true;
    for (size_t i = 0; i < lanes.size() && ok; i++) ok = lanes[i].open((int)(i / opt.inflight), jdesc, opt.join_po2);
    const double t0 = now_s();
    while (ok && claims.size() > 1) {
        const size_t pairs = claims.size() / 2;
        std::vector<Claim> next(pairs);
        std::vector<std::vector<uint32_t>> seals(pairs);
        std::atomic<size_t> idx{0};
        std::vector<std::thread> th;
        for (auto& ln : lanes)
            th.emplace_back([&, lane = &ln] {
                std::vector<uint32_t> out(24), pub(16);
                for (;;) {
                    const size_t k = idx.fetch_add(1);
                    if (k >= pairs) return;
                    memcpy(pub.data(), claims[2 * k].data(), 32);
                    memcpy(pub.data() + 8, claims[2 * k + 1].data(), 32);
                    const NoiseKey fixed(opt.noise_seed);
                    const uint32_t* noise = opt.fixed_noise ? fixed.k : nullptr;
                    uint32_t* seal = nullptr;
                    size_t words = 0;
                    if (failed(zkh_syn_witgen(lane->ctx, lane->circuit, opt.join_po2, ZKH_ZK_CYCLES, 0, noise, pub.data(), lane->code, lane->data, out.data()),
                               "zkh_syn_witgen(join)") ||
                        failed(zkh_prove_segment(lane->prover, opt.join_po2, ZKH_ZK_CYCLES, noise, lane->code, lane->data, out.data(), &seal, &words),
                               "zkh_prove_segment(join)"))
                        return;
                    seals[k].assign(seal, seal + words);
                    next[k].assign(seal, seal + 8);                   // the parent claim the join constrains: out[0..8)
                    zkh_free_seal(seal);
                }
            });
        for (auto& t : th) t.join();
        { std::lock_guard<std::mutex> lk(g_err_lock); ok = g_first_error.empty(); }
        if (!ok) break;
        res.joins += pairs;
        if (pairs == 1 && claims.size() == 2) res.root_seal = seals[0];
        if (claims.size() % 2) next.push_back(claims.back());
        claims.swap(next);
    }
    res.seconds = now_s() - t0;
    if (ok && !claims.empty()) res.root_claim = claims[0];
    // the join circuit's control root, for the verifier
    if (ok && !lanes.empty() && !res.root_seal.empty()) {
        uint32_t jroot[8];
        ok = !failed(zkh_syn_control_root(lanes[0].prover, opt.join_po2, ZKH_ZK_CYCLES, jroot), "zkh_syn_control_root(join)");
        zkh_circuit* hc = nullptr;
        ok = ok && !failed(zkh_circuit_load(nullptr, jdesc.data(), jdesc.size(), &hc), "zkh_circuit_load(join host)");
        if (ok) {
            const char* err = zkh_verify_segment(hc, res.root_seal.data(), res.root_seal.size(), jroot, nullptr, nullptr);
            if (err) { failed(err, "root receipt REJECTED"); ok = false; }
        }
        if (hc) zkh_circuit_destroy(hc);
    }
    for (auto& ln : lanes) ln.close();
    return ok;
}

bool parse(int argc, char** argv, Options& o) {
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto val = [&](size_t& dst) { if (i + 1 < argc) dst = strtoull(argv[++i], nullptr, 10); };
        if (a == "--desc" && i + 1 < argc) o.desc_path = argv[++i];
        else if (a == "--po2") val(o.po2);
        else if (a == "--segments") val(o.segments);
        else if (a == "--devices") val(o.devices);
        else if (a == "--inflight") val(o.inflight);
        else if (a == "--no-verify") o.verify = false;
        else if (a == "--receipts-dir" && i + 1 < argc) o.receipts_dir = argv[++i];
        else if (a == "--code-objects" && i + 1 < argc) o.code_objects_dir = argv[++i];
        else if (a == "--resident-code-group") o.resident_code = true;
        else if (a == "--join-desc" && i + 1 < argc) o.join_desc_path = argv[++i];
        else if (a == "--join-po2") val(o.join_po2);
        else if (a == "--noise-seed" && i + 1 < argc) { o.noise_seed = strtoull(argv[++i], nullptr, 0); o.fixed_noise = true; }
        else { fprintf(stderr, "unknown argument %s\n", a.c_str()); return false; }
    }
    return !o.desc_path.empty() && o.devices >= 1 && o.inflight >= 1 && o.segments >= 1;
}

}  // namespace

int main(int argc, char** argv) {
    Options opt;
    if (!parse(argc, argv, opt)) {
        fprintf(stderr, "usage: %s --desc FILE [--po2 N] [--segments S] [--devices G] [--inflight K] [--no-verify] [--noise-seed N] [--receipts-dir DIR] [--code-objects DIR] [--resident-code-group]\n", argv[0]);
        return 2;
    }
    FILE* f = fopen(opt.desc_path.c_str(), "rb");
    if (!f) { perror(opt.desc_path.c_str()); return 2; }
    std::vector<uint32_t> desc;
    uint32_t word;
    while (fread(&word, 4, 1, f) == 1) desc.push_back(word);
    fclose(f);
    if (desc.size() < 16) { fprintf(stderr, "%s: not a circuit description\n", opt.desc_path.c_str()); return 2; }

    std::vector<Receipt> receipts(opt.segments);
    std::atomic<size_t> next{0};
    std::vector<std::thread> lanes;
    const double t0 = now_s();
    for (size_t d = 0; d < opt.devices; d++)
        for (size_t k = 0; k < opt.inflight; k++) lanes.emplace_back(lane, (int)d, std::cref(desc), std::cref(opt), std::ref(next), std::ref(receipts));
    for (auto& t : lanes) t.join();
    const double dt = now_s() - t0;
    if (!g_first_error.empty()) { fprintf(stderr, "error: %s\n", g_first_error.c_str()); return 1; }

    // composite receipt = the seals in segment order; verify each one on the host (no GPU involved)
    size_t verified = 0, total_words = 0;
    double seal_sum = 0;
    JoinResult join_res;
    bool succinct_ok = false;
    if (opt.verify) {
        zkh_circuit* host_circuit = nullptr;
        if (failed(zkh_circuit_load(nullptr, desc.data(), desc.size(), &host_circuit), "zkh_circuit_load(host)")) {
            fprintf(stderr, "error: %s\n", g_first_error.c_str());
            return 1;
        }
        for (size_t i = 0; i < receipts.size(); i++) {
            const char* err = zkh_verify_segment(host_circuit, receipts[i].seal.data(), receipts[i].seal.size(), g_control_root, nullptr, nullptr);
            if (err) { fprintf(stderr, "segment %zu: seal REJECTED: %s\n", i, err); zkh_free_error(err); return 1; }
            verified++;
            if (!opt.receipts_dir.empty()) {
                // the envelope a host would store or ship instead of upstream's bincode SegmentReceipt; read back and re-checked
                uint32_t* blob = nullptr;
                size_t words = 0, off = 0;
                uint32_t info[26];
                if (failed(zkh_receipt_encode(host_circuit, receipts[i].seal.data(), receipts[i].seal.size(), (uint32_t)i, g_control_root, &blob, &words),
                           "zkh_receipt_encode") ||
                    failed(zkh_receipt_decode(host_circuit, blob, words, info, &off), "zkh_receipt_decode")) {
                    fprintf(stderr, "error: %s\n", g_first_error.c_str());
                    return 1;
                }
                const std::string path = opt.receipts_dir + "/segment_" + std::to_string(i) + ".zkr";
                FILE* rf = fopen(path.c_str(), "wb");
                if (!rf || fwrite(blob, 4, words, rf) != words) { perror(path.c_str()); return 1; }
                fclose(rf);
                zkh_free_seal(blob);
            }
        }
        // ---- config 5: the join tree over the verified leaves, then what the holder of the compact receipt checks ----
        if (!opt.join_desc_path.empty() && opt.segments > 1) {
            std::vector<uint32_t> jdesc;
            FILE* jf = fopen(opt.join_desc_path.c_str(), "rb");
            if (!jf) { perror(opt.join_desc_path.c_str()); return 2; }
            while (fread(&word, 4, 1, jf) == 1) jdesc.push_back(word);
            fclose(jf);
            if (jdesc.size() < 16 || jdesc[13] != 3) { fprintf(stderr, "%s: not a P2-JOIN circuit description (kind 3)\n", opt.join_desc_path.c_str()); return 2; }
            std::vector<Claim> claims(receipts.size(), Claim(8));
            for (size_t i = 0; i < receipts.size(); i++)
                if (failed(zkh_receipt_claim(host_circuit, receipts[i].seal.data(), receipts[i].seal.size(), g_control_root, nullptr, nullptr, claims[i].data()),
                           "zkh_receipt_claim")) { fprintf(stderr, "error: %s\n", g_first_error.c_str()); return 1; }
            if (!join_tree(opt, jdesc, claims, join_res)) { fprintf(stderr, "error: %s\n", g_first_error.c_str()); return 1; }
            // the claim tree recomputed on the host from the leaf claims must end in the root receipt's public output
            std::vector<Claim> level = claims;
            while (level.size() > 1) {
                std::vector<Claim> up(level.size() / 2, Claim(8));
                for (size_t k = 0; k < up.size(); k++)
                    if (!hash_pair(level[2 * k], level[2 * k + 1], up[k])) { fprintf(stderr, "error: %s\n", g_first_error.c_str()); return 1; }
                if (level.size() % 2) up.push_back(level.back());
                level.swap(up);
            }
            succinct_ok = level[0] == join_res.root_claim && join_res.root_seal.size() > 24 &&
                          memcmp(join_res.root_seal.data(), level[0].data(), 32) == 0;
            if (!succinct_ok) { fprintf(stderr, "error: the root receipt's output is not the claim tree of the leaves\n"); return 1; }
        }
        zkh_circuit_destroy(host_circuit);
    }
    for (const auto& r : receipts) { total_words += r.seal.size(); seal_sum += r.seal_s; }
    // wall clock here includes context creation, circuit load and witness generation: a session, not the bench metric
    printf("{\"driver\": \"seal_segments\", \"library\": \"%s\", \"po2\": %zu, \"segments\": %zu, \"devices\": %zu, \"inflight\": %zu, "
           "\"session_wall_s\": %.4f, \"segments_per_s_incl_setup\": %.3f, \"mean_seal_call_s\": %.4f, \"seal_words_total\": %zu, "
           "\"verified\": %zu, \"joins\": %zu, \"join_tree_s\": %.4f, \"root_receipt_words\": %zu, \"compact_receipt_verified\": %s}\n",
           zkh_version(), opt.po2, opt.segments, opt.devices, opt.inflight, dt, opt.segments / dt, seal_sum / opt.segments,
           total_words, verified, join_res.joins, join_res.seconds, join_res.root_seal.size(), succinct_ok ? "true" : "false");
    return 0;
}
