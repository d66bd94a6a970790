// rec_builder.hip — the lift / lift2 / join / join3 / union / resolve PROGRAMS of the RECURSION circuit, built on the host in C++ (no GPU, no Python).
//
// Upstream ships its recursion programs as precompiled `.zkr` files (risc0-circuit-recursion 4.0.2, un-vendored:
// /root/reference/Cargo.lock:5305) that its Zirgen toolchain emits offline; ProverServer::{lift, join} (risc0-zkvm 3.0.3
// host/server/prove/prover_impl.rs, /root/reference/Cargo.lock:5418) load them by name.  Here the programs are this library's
// STARK verifier (verifier.hip = risc0-zkp 3.0.2 src/verify/) restated gate by gate for this repository's recursion circuit —
// first written in Python (zeth_amd/circuits/rec_verify.py on top of circuits/recursion.py `Program`), which stays the readable
// statement and the test oracle of this file; this is the same builder for hosts that have no Python: statement for statement
// the same variables, gates, constants and witness ops in the same order, so the blobs are IDENTICAL word for word
// (tests/test_rec_builder.py compares SHA-256 with examples/recursion_programs.manifest.json and with freshly built Python
// blobs).  Host only: nothing here touches a device.
#include <array>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "common.h"

namespace {

using u32 = uint32_t;
using u64 = uint64_t;

constexpr u32 FP = 2013265921u;
constexpr u32 D_MAGIC = 0x5A4B4331u, D_HEADER = 16;
enum : u32 { S_CONST = 0, S_CONST_EXT, S_GET, S_GET_GLOBAL, S_ADD, S_SUB, S_MUL, S_TRUE, S_AND_EQZ, S_AND_COND };
enum : u32 { G_ACCUM = 0, G_CODE = 1, G_DATA = 2 };

u32 mulm(u64 a, u64 b) { return (u32)(a * b % FP); }
u32 powm(u32 b, u64 e) {
    u64 r = 1, x = b % FP;
    for (; e; e >>= 1) { if (e & 1) r = r * x % FP; x = x * x % FP; }
    return (u32)r;
}
u32 invm(u32 a) { return powm(a, FP - 2); }
u32 log2_ceil(u64 x) { u32 l = 0; while (((u64)1 << l) < x) l++; return l; }

// ---- a circuit description (circuits/desc.py Circuit.parse) ----
struct Step { u32 op, a, b, c, d; };
struct Reg { u32 group, offset; std::vector<u32> backs; u32 combo; };
struct Desc {
    u32 group_size[3] = {0, 0, 0}, global_size[2] = {0, 0}, ret = 0, kind = 0;
    std::vector<std::array<u32, 3>> taps;
    std::vector<std::vector<u32>> combos;
    std::vector<Step> steps;
    std::vector<Reg> regs;
};
const char* parse_desc(const u32* d, size_t n, Desc& c) {
    ZKH_REQUIRE(d && n >= D_HEADER && d[0] == D_MAGIC && d[1] == 1, "rec_build: bad circuit description");
    for (int g = 0; g < 3; g++) c.group_size[g] = d[3 + g];
    c.global_size[0] = d[7]; c.global_size[1] = d[8];
    const u32 n_taps = d[9], n_combos = d[10], n_steps = d[11];
    c.ret = d[12]; c.kind = d[13];
    size_t pos = D_HEADER;
    ZKH_REQUIRE(pos + 3 * (size_t)n_taps <= n, "rec_build: circuit description is truncated (taps)");
    for (u32 i = 0; i < n_taps; i++, pos += 3) c.taps.push_back({d[pos], d[pos + 1], d[pos + 2]});
    for (u32 i = 0; i < n_combos; i++) {
        ZKH_REQUIRE(pos < n && pos + 1 + d[pos] <= n, "rec_build: circuit description is truncated (combos)");
        c.combos.emplace_back(d + pos + 1, d + pos + 1 + d[pos]);
        pos += 1 + d[pos];
    }
    ZKH_REQUIRE(pos + 5 * (size_t)n_steps <= n, "rec_build: circuit description is truncated (steps)");
    for (u32 i = 0; i < n_steps; i++, pos += 5) c.steps.push_back({d[pos], d[pos + 1], d[pos + 2], d[pos + 3], d[pos + 4]});
    // operands name earlier values only (what zkh_circuit_load checks before a circuit reaches the GPU): the builder indexes by them
    u32 cf = 0, cm = 0;
    for (const Step& st : c.steps) {
        bool ok = true;
        switch (st.op) {
        case S_CONST: case S_CONST_EXT: break;
        case S_GET: ok = st.a < n_taps; break;
        case S_GET_GLOBAL: ok = st.a <= 1 && st.b < c.global_size[st.a]; break;
        case S_ADD: case S_SUB: case S_MUL: ok = st.a < cf && st.b < cf; break;
        case S_TRUE: break;
        case S_AND_EQZ: ok = st.a < cm && st.b < cf; break;
        case S_AND_COND: ok = st.a < cm && st.b < cf && st.c < cm; break;
        default: ok = false;
        }
        ZKH_REQUIRE(ok, "rec_build: the circuit description has a step with an operand out of range");
        if (st.op >= S_TRUE) cm++; else cf++;
    }
    ZKH_REQUIRE(c.ret < cm, "rec_build: the circuit description's result is not a mix value");
    for (const auto& t : c.taps) ZKH_REQUIRE(t[0] < 3 && t[1] < c.group_size[t[0]], "rec_build: a tap names a column outside its group");
    for (const auto& t : c.taps) {                                   // registers in tap order, each with its back-set's combo
        if (!c.regs.empty() && c.regs.back().group == t[0] && c.regs.back().offset == t[1]) c.regs.back().backs.push_back(t[2]);
        else c.regs.push_back({t[0], t[1], {t[2]}, 0});
    }
    for (auto& r : c.regs) {
        size_t k = 0;
        while (k < c.combos.size() && c.combos[k] != r.backs) k++;
        ZKH_REQUIRE(k < c.combos.size(), "rec_build: a register's back-set is not among the combos");
        r.combo = (u32)k;
    }
    return nullptr;
}

// ---- circuits/recursion.py Program ----
constexpr u32 BLOCK = 12, NW = 6;
enum : u32 { OP_INPUT = 1, OP_GEN, OP_MUX, OP_PACK, OP_UNPACK, OP_INV, OP_BITS, OP_P2, OP_EQ, OP_ISZ };
enum : u32 { GF_MUX = 1, GF_BOOL = 2, GF_EMB = 4, GF_PACK0 = 8, GF_PUB = 128, GF_SWAP = 256 };
constexpr u32 PROG_MAGIC = 0x5a4b5231u, PROG_HEADER = 16, PROG_VERSION = 2;

struct Gate { int pos[6]; u32 q[6]; u32 flags; };
struct Perm { int ins[6]; int out; bool swap; };
using W2 = std::array<int, 2>;                                       // a digest: two packed wires

struct Program {
    int n_vars = 0;
    std::vector<std::array<u32, 8>> ops;
    std::vector<Gate> gates;
    std::vector<Perm> p2s;
    std::vector<u32> consts;
    std::map<std::array<u32, 5>, u32> const_at;
    std::map<std::array<u32, 4>, int> const_var;
    std::vector<int> parent;
    u32 n_inputs = 0;
    bool has_pub = false;

    int var(int count = 1) {
        const int v = n_vars;
        n_vars += count;
        for (int i = 0; i < count; i++) parent.push_back(v + i);
        return v;
    }
    int find(int v) {
        while (parent[v] != v) { parent[v] = parent[parent[v]]; v = parent[v]; }
        return v;
    }
    void op(u32 code, int out, std::initializer_list<int> ins, u32 aux = 0) {
        std::array<u32, 8> o = {code | (aux << 8), (u32)out, 0, 0, 0, 0, 0, 0};
        int k = 2;
        for (int v : ins) o[k++] = (u32)v;
        ops.push_back(o);
    }
    void eq(int x, int y) {
        ops.push_back({OP_EQ, 0, (u32)x, (u32)y, 0, 0, 0, 0});
        const int rx = find(x), ry = find(y);
        if (rx != ry) parent[std::max(rx, ry)] = std::min(rx, ry);
    }
    u32 kidx(u32 q0, u32 q1, u32 q2, u32 q3, u32 q4) {
        const std::array<u32, 5> q = {q0 % FP, q1 % FP, q2 % FP, q3 % FP, q4 % FP};
        auto it = const_at.find(q);
        if (it != const_at.end()) return it->second;
        const u32 at = (u32)consts.size();
        consts.insert(consts.end(), q.begin(), q.end());
        const_at[q] = at;
        return at;
    }
    int input(u32 off, u32 count = 4) {
        const int v = var();
        op(OP_INPUT, v, {(int)off}, count);
        n_inputs = std::max(n_inputs, off + count);
        return v;
    }
    // d = qM a*b + qA a + qB b + qC c + qK
    int gen(int a, int b, int c, u32 qM = 0, u32 qA = 0, u32 qB = 0, u32 qC = 0, u32 qK = 0) {
        const int d = var();
        gates.push_back({{a, b, c, d, -1, -1}, {qM % FP, qA % FP, qB % FP, qC % FP, FP - 1, qK % FP}, 0});
        const u32 k = kidx(qM, qA, qB, qC, qK);
        op(OP_GEN, d, {a, b, c, (int)k});
        return d;
    }
    // qM a*b + qA a + qB b + qC c + qK = 0
    void require(int a, int b, int c, u32 qM = 0, u32 qA = 0, u32 qB = 0, u32 qC = 0, u32 qK = 0) {
        gates.push_back({{a, b, c, -1, -1, -1}, {qM % FP, qA % FP, qB % FP, qC % FP, 0, qK % FP}, 0});
        const int z = var();
        const u32 k = kidx(qM, qA, qB, qC, qK);
        op(OP_GEN, z, {a, b, c, (int)k});
        const int zz = zero();
        ops.push_back({OP_EQ, 0, (u32)z, (u32)zz, 0, 0, 0, 0});
    }
    int constant(u32 k0, u32 k1 = 0, u32 k2 = 0, u32 k3 = 0) {
        const std::array<u32, 4> key = {k0 % FP, k1 % FP, k2 % FP, k3 % FP};
        auto it = const_var.find(key);
        if (it != const_var.end()) return it->second;
        int v;
        if (!key[1] && !key[2] && !key[3]) {
            v = var();
            gates.push_back({{-1, -1, -1, v, -1, -1}, {0, 0, 0, 0, FP - 1, key[0]}, 0});
            const u32 k = kidx(0, 0, 0, 0, key[0]);
            op(OP_GEN, v, {v, v, v, (int)k});
        } else {
            const int c0 = constant(key[0]), c1 = constant(key[1]), c2 = constant(key[2]), c3 = constant(key[3]);
            v = pack(0, c0, c1, c2, c3);
        }
        const_var[key] = v;
        return v;
    }
    int zero() { return constant(0); }
    int add(int a, int b) { return gen(a, b, a, 0, 1, 1); }
    int sub(int a, int b) { return gen(a, b, a, 0, 1, FP - 1); }
    int mul(int a, int b) { return gen(a, b, a, 1); }
    int muladd(int a, int b, int c, u32 k = 1) { return gen(a, b, c, k, 0, 0, 1); }           // c + k a*b
    int scale(int a, u32 k, u32 plus = 0) { return gen(a, a, a, 0, k, 0, 0, plus); }
    int lin(int a, u32 ka, int b, u32 kb, u32 k = 0) { return gen(a, b, a, 0, ka, kb, 0, k); }
    int mux(int bit, int b, int c) {                                                          // bit ? c : b
        const int d = var();
        gates.push_back({{bit, b, c, d, -1, -1}, {0, 0, 0, 0, 0, 0}, GF_MUX});
        op(OP_MUX, d, {bit, b, c});
        return d;
    }
    void boolean(int a) {
        gates.push_back({{a, -1, -1, -1, -1, -1}, {0, 0, 0, 0, 0, 0}, GF_BOOL});
        const int z = var();
        const u32 k = kidx(1, FP - 1, 0, 0, 0);
        op(OP_GEN, z, {a, a, a, (int)k});
        const int zz = zero();
        ops.push_back({OP_EQ, 0, (u32)z, (u32)zz, 0, 0, 0, 0});
    }
    int pack(u32 j, int a, int b, int c, int e, bool embedded = false) {                      // d = (a_j, b_j, c_j, e_j)
        const int d = var();
        gates.push_back({{a, b, c, d, e, -1}, {0, 0, 0, 0, 0, 0}, (GF_PACK0 << j) | (embedded ? GF_EMB : 0u)});
        op(OP_PACK, d, {a, b, c, e}, j);
        return d;
    }
    std::array<int, 4> unpack(int d) {
        const int v = var(4);
        gates.push_back({{v, v + 1, v + 2, d, v + 3, -1}, {0, 0, 0, 0, 0, 0}, GF_PACK0 | GF_EMB});
        op(OP_UNPACK, v, {d});
        return {v, v + 1, v + 2, v + 3};
    }
    int inv(int a) {
        const int b = var();
        op(OP_INV, b, {a});
        require(a, b, a, 1, 0, 0, 0, FP - 1);
        return b;
    }
    int is_zero(int a) {
        const int h = var();
        op(OP_ISZ, h, {a});
        const int z = gen(a, h, a, FP - 1, 0, 0, 0, 1);
        require(a, z, a, 1);
        return z;
    }
    // canonical bits of the base-field wire x: the first `want` (and for want < 31 the partial sum of those bits as last element)
    std::vector<int> bits31(int x, u32 want = 31) {
        const int base = var(31);
        op(OP_BITS, base, {x});
        int acc = -1, partial = -1, low27 = -1;
        for (u32 i = 0; i < 31; i++) {
            const int bit = base + (int)i;
            const int prev = acc >= 0 ? acc : zero();
            acc = var();
            const u32 w = powm(2, i);
            gates.push_back({{bit, -1, prev, acc, -1, -1}, {0, w, 0, 1, FP - 1, 0}, GF_BOOL});
            const u32 k = kidx(0, w, 0, 1, 0);
            op(OP_GEN, acc, {bit, bit, prev, (int)k});
            if (i + 1 == 27) low27 = acc;
            if (i + 1 == want) partial = acc;
        }
        eq(acc, x);
        const int t0 = mul(base + 27, base + 28), t1 = mul(base + 29, base + 30);
        const int t = mul(t0, t1);
        require(t, low27, t, 1);                          // bits 27..30 all set => the low 27 bits are zero (value < P)
        std::vector<int> out;
        for (u32 i = 0; i < want; i++) out.push_back(base + (int)i);
        if (want < 31) out.push_back(partial);
        return out;
    }
    std::array<int, 6> p2(const std::array<int, 6>& ins, bool swap = false) {
        const int out = var(NW);
        Perm p;
        for (int i = 0; i < 6; i++) p.ins[i] = ins[i];
        p.out = out; p.swap = swap;
        p2s.push_back(p);
        op(OP_P2, out, {ins[0], ins[1], ins[2], ins[3], ins[4], ins[5]}, swap ? 1u : 0u);
        return {out, out + 1, out + 2, out + 3, out + 4, out + 5};
    }
    void publish(int a, int b, int c, int d) {
        has_pub = true;
        gates.push_back({{a, b, c, d, -1, -1}, {0, 0, 0, 0, 0, 0}, GF_PUB});
    }
    u32 min_po2(u32 zk) const {
        for (u32 po2 = 1; po2 < 25; po2++) {
            const int64_t A = ((int64_t)1 << po2) - zk;
            if (A <= 0) continue;
            const int64_t K = A / BLOCK;
            if (K >= (int64_t)p2s.size() && A - 2 * K >= (int64_t)gates.size()) return po2;
        }
        return 0;
    }
    // -> header | gate table [rows][13] | position table [rows][6] | consts | ops
    const char* finish(u32 po2, u32 zk, std::vector<u32>& blob) {
        const size_t n = (size_t)1 << po2;
        ZKH_REQUIRE(n > zk + 1, "rec_build: no active rows at po2 %u", po2);
        const size_t A = n - zk, K = A / BLOCK;
        ZKH_REQUIRE(K >= p2s.size(), "rec_build: %zu permutations need more than %zu blocks (po2 %u)", p2s.size(), K, po2);
        ZKH_REQUIRE(A - 2 * K >= gates.size(), "rec_build: %zu gates need more than %zu rows (po2 %u)", gates.size(), A - 2 * K, po2);
        std::vector<int> pos(A * NW, -1);
        std::vector<u32> gate(A * 7, 0);
        size_t g = 0;
        for (size_t r = 0; r < A && g < gates.size(); r++) {
            const size_t k = r % BLOCK;
            if (!(r >= BLOCK * K || (k != 0 && k != BLOCK - 1))) continue;
            for (int w = 0; w < 6; w++) pos[r * NW + w] = gates[g].pos[w];
            for (int i = 0; i < 6; i++) gate[r * 7 + i] = gates[g].q[i];
            gate[r * 7 + 6] = gates[g].flags;
            g++;
        }
        for (size_t p = 0; p < p2s.size(); p++) {
            for (int w = 0; w < 6; w++) { pos[BLOCK * p * NW + w] = p2s[p].ins[w]; pos[(BLOCK * p + BLOCK - 1) * NW + w] = p2s[p].out + w; }
            if (p2s[p].swap) gate[BLOCK * p * 7 + 6] = GF_SWAP;
        }
        // copy classes -> sigma: every used position points at the next position (ascending) of its class, the last at the first
        std::vector<int> root(n_vars);
        for (int v = 0; v < n_vars; v++) root[v] = find(v);
        std::vector<u32> sigma(A * NW), posr(A * NW, 0);
        for (size_t i = 0; i < A * NW; i++) sigma[i] = (u32)i;
        std::vector<int64_t> first(n_vars, -1), last(n_vars, -1);
        for (size_t i = 0; i < A * NW; i++) {
            if (pos[i] < 0) continue;
            const int c = root[pos[i]];
            posr[i] = (u32)c + 1;
            if (first[c] < 0) first[c] = (int64_t)i; else sigma[last[c]] = (u32)i;
            last[c] = (int64_t)i;
        }
        for (int c = 0; c < n_vars; c++) if (last[c] >= 0) sigma[last[c]] = (u32)first[c];
        blob.clear();
        blob.reserve(PROG_HEADER + A * 19 + consts.size() + ops.size() * 8);
        const u32 head[PROG_HEADER] = {PROG_MAGIC, PROG_VERSION, po2, zk, (u32)A, (u32)n_vars, (u32)consts.size(), (u32)ops.size(), n_inputs,
                                       (u32)p2s.size(), (u32)gates.size(), 0, 0, 0, 0, 0};
        blob.insert(blob.end(), head, head + PROG_HEADER);
        for (size_t r = 0; r < A; r++) {
            blob.insert(blob.end(), gate.begin() + r * 7, gate.begin() + r * 7 + 7);
            blob.insert(blob.end(), sigma.begin() + r * NW, sigma.begin() + r * NW + NW);
        }
        blob.insert(blob.end(), posr.begin(), posr.end());
        blob.insert(blob.end(), consts.begin(), consts.end());
        for (const auto& o : ops) blob.insert(blob.end(), o.begin(), o.end());
        return nullptr;
    }
};

// ---- circuits/rec_verify.py ----
constexpr u32 QUERIES = 50, INV_RATE = 4, FRI_FOLD = 16, FRI_MIN_DEGREE = 256, CHECK_SIZE = 16, EXT = 4, ALLOWED_DEPTH = 4;
u32 rou_fwd(u32 k) { return powm(137, (u64)1 << (27 - k)); }
u32 rou_rev(u32 k) { return invm(rou_fwd(k)); }
u32 bitrev4(u32 i) { return ((i & 1) << 3) | ((i & 2) << 1) | ((i & 4) >> 1) | ((i & 8) >> 3); }

struct Sponge {
    Program& pr;
    std::array<int, 6> cells;
    u32 used = 0;
    std::map<int, std::array<int, 4>> unpacked;
    explicit Sponge(Program& p) : pr(p) { const int z = pr.zero(); cells = {z, z, z, z, z, z}; }
    void mix() { cells = pr.p2(cells); used = 0; }
    void commit(const W2& digest) {
        if (used) mix();
        const int c0 = pr.add(cells[0], digest[0]), c1 = pr.add(cells[1], digest[1]);
        cells[0] = c0; cells[1] = c1;
        mix();
    }
    int elem() {
        if (used == 16) mix();
        const int w = cells[used / 4];
        auto it = unpacked.find(w);
        if (it == unpacked.end()) it = unpacked.emplace(w, pr.unpack(w)).first;
        const int e = it->second[used % 4];
        used++;
        return e;
    }
    int ext() {
        if (used == 16) mix();
        if (used % 4 == 0) { const int w = cells[used / 4]; used += 4; return w; }
        const int e0 = elem(), e1 = elem(), e2 = elem(), e3 = elem();
        return pr.pack(0, e0, e1, e2, e3);
    }
};

struct Tree { u32 rows, cols, layers, top_layer; std::vector<W2> top; };
struct FriRound { u32 domain; Tree tree; std::vector<int> mxn; };
struct SealOut { std::vector<int> out, head; W2 code_root; };

struct Verifier {
    Program& pr;
    u32 pos = 0;
    std::unique_ptr<Sponge> io;
    explicit Verifier(Program& p) : pr(p), io(new Sponge(p)) {}

    std::vector<int> read(u32 count) {
        std::vector<int> out;
        for (u32 i = 0; i < (count + 3) / 4; i++) out.push_back(pr.input(pos + 4 * i, std::min<u32>(4, count - 4 * i)));
        pos += count;
        if (count % 4) {                                   // the padding of a partial wire is part of what gets hashed: it must BE zero
            const auto u = pr.unpack(out.back());
            for (u32 k = count % 4; k < 4; k++) { const int z = pr.zero(); pr.eq(u[k], z); }
        }
        return out;
    }
    W2 elems(const std::vector<int>& wires) {
        const int z = pr.zero();
        int cap0 = z, cap1 = z;
        const size_t blocks = std::max<size_t>(1, (wires.size() + 3) / 4);
        std::array<int, 6> o{};
        for (size_t b = 0; b < blocks; b++) {
            std::array<int, 6> in = {z, z, z, z, cap0, cap1};
            for (size_t i = 0; i < 4 && 4 * b + i < wires.size(); i++) in[i] = wires[4 * b + i];
            o = pr.p2(in);
            cap0 = o[4]; cap1 = o[5];
        }
        return {o[0], o[1]};
    }
    W2 pair(const W2& a, const W2& b) {
        const int z = pr.zero();
        const auto o = pr.p2({a[0], a[1], b[0], b[1], z, z});
        return {o[0], o[1]};
    }
    W2 pair_at(int bit, const W2& cur, const W2& sib) {    // one Merkle level as a conditional-swap block
        const int z = pr.zero();
        const auto o = pr.p2({cur[0], cur[1], sib[0], sib[1], bit, z}, true);
        return {o[0], o[1]};
    }
    Tree tree_init(u32 rows, u32 cols) {
        Tree t;
        t.rows = rows; t.cols = cols; t.layers = log2_ceil(rows); t.top_layer = 0;
        for (u32 i = 1; i < t.layers; i++) { if (((u32)1 << i) > QUERIES) break; t.top_layer = i; }
        const u32 top_size = 1u << t.top_layer;
        const auto w = read(8 * top_size);
        t.top.assign(2 * top_size, W2{-1, -1});
        for (u32 i = 0; i < top_size; i++) t.top[top_size + i] = {w[2 * i], w[2 * i + 1]};
        for (u32 i = top_size - 1; i >= 1; i--) t.top[i] = pair(t.top[2 * i], t.top[2 * i + 1]);
        io->commit(t.top[1]);
        return t;
    }
    std::vector<int> tree_open(const Tree& t, const std::vector<int>& idx_bits) {
        const auto col = read(t.cols);
        W2 cur = elems(col);
        const u32 low = t.layers - t.top_layer;
        for (u32 lvl = 0; lvl < low; lvl++) {
            const auto sib = read(8);
            cur = pair_at(idx_bits[lvl], cur, {sib[0], sib[1]});
        }
        std::vector<W2> cand(t.top.begin() + ((size_t)1 << t.top_layer), t.top.end());
        for (size_t bi = low; bi < idx_bits.size(); bi++) {
            std::vector<W2> nxt;
            for (size_t i = 0; i < cand.size() / 2; i++) {
                const int m0 = pr.mux(idx_bits[bi], cand[2 * i][0], cand[2 * i + 1][0]);
                const int m1 = pr.mux(idx_bits[bi], cand[2 * i][1], cand[2 * i + 1][1]);
                nxt.push_back({m0, m1});
            }
            cand.swap(nxt);
        }
        pr.eq(cur[0], cand[0][0]);
        pr.eq(cur[1], cand[0][1]);
        return col;
    }
    int horner(const int* coeffs, size_t n, int x) {
        int acc = coeffs[n - 1];
        for (size_t i = n - 1; i-- > 0;) acc = pr.muladd(acc, x, coeffs[i]);
        return acc;
    }
    int pow_bits(u32 g, const int* bits, size_t n) {       // g^(sum 2^i bit_i) for a constant g
        int acc = pr.constant(1);
        for (size_t i = 0; i < n; i++) {
            const u32 gi = powm(g, (u64)1 << i);
            acc = pr.gen(bits[i], acc, acc, (gi + FP - 1) % FP, 0, 0, 1);
        }
        return acc;
    }
    // PolyExtStepDef::step over ExtElem: -> the constraint polynomial's value
    int poly_ext(const Desc& c, int poly_mix, const std::vector<int>& u, const std::vector<int>& out_words, const std::vector<int>& mix_words) {
        std::vector<int> fv;
        std::vector<std::pair<int, u32>> mv;               // (tot wire or -1 for zero, static exponent of poly_mix)
        std::vector<int> pw = {pr.constant(1), poly_mix};
        auto power = [&](u32 e) { while (pw.size() <= e) pw.push_back(pr.mul(pw.back(), poly_mix)); return pw[e]; };
        for (const Step& s : c.steps) {
            switch (s.op) {
            case S_CONST: fv.push_back(pr.constant(s.a)); break;
            case S_CONST_EXT: fv.push_back(pr.constant(s.a, s.b, s.c, s.d)); break;
            case S_GET: fv.push_back(u[s.a]); break;
            case S_GET_GLOBAL: fv.push_back((s.a == 0 ? out_words : mix_words)[s.b]); break;
            case S_ADD: fv.push_back(pr.add(fv[s.a], fv[s.b])); break;
            case S_SUB: fv.push_back(pr.sub(fv[s.a], fv[s.b])); break;
            case S_MUL: fv.push_back(pr.mul(fv[s.a], fv[s.b])); break;
            case S_TRUE: mv.push_back({-1, 0}); break;
            case S_AND_EQZ: {
                const int tot = mv[s.a].first;
                const u32 e = mv[s.a].second;
                const int pe = power(e);
                mv.push_back({tot >= 0 ? pr.muladd(pe, fv[s.b], tot) : pr.mul(pe, fv[s.b]), e + 1});
                break;
            }
            case S_AND_COND: {
                const int tot = mv[s.a].first, itot = mv[s.c].first;
                const u32 e = mv[s.a].second, ie = mv[s.c].second;
                if (itot < 0) { mv.push_back({tot, e + ie}); break; }
                const int t = pr.mul(fv[s.b], itot);
                const int pe = power(e);
                mv.push_back({tot >= 0 ? pr.muladd(t, pe, tot) : pr.mul(t, pe), e + ie});
                break;
            }
            default: return -1;
            }
        }
        const int tot = mv[c.ret].first;
        return tot >= 0 ? tot : pr.zero();
    }
    // 16 evaluations on a coset -> the folded polynomial's value (verify/fri.rs fold_eval); mxn[i] = mix^i / 16
    int fold_eval(std::vector<int> v, const std::vector<int>& mxn, int inv_wk) {
        for (u32 N = 4; N >= 1; N--) {
            const u32 ln = 1u << N, half = 1u << (N - 1), step = rou_rev(N);
            u32 cur = 1;
            for (u32 i = 0; i < half; i++) {
                for (u32 s = 0; s < 16; s += ln) {
                    const int a = v[s + i], b = v[s + i + half];
                    v[s + i] = pr.add(a, b);
                    v[s + i + half] = pr.lin(a, cur, b, FP - cur);
                }
                cur = mulm(cur, step);
            }
        }
        int tot = -1, mw = -1;
        for (u32 i = 0; i < 16; i++) {
            const int ci = v[bitrev4(i)];
            if (i == 0) { tot = pr.mul(ci, mxn[0]); continue; }
            mw = i == 1 ? inv_wk : pr.mul(mw, inv_wk);
            const int t = pr.mul(ci, mw);
            tot = pr.muladd(t, mxn[i], tot);
        }
        return tot;
    }
    // verifier.hip zkh_verify_segment = risc0-zkp verify/mod.rs, statement by statement (rec_verify.py verify_seal)
    const char* verify_seal(const Desc& c, u32 po2, SealOut& res) {
        const u32 out_size = c.global_size[0], mix_size = c.global_size[1];
        const u32 size = 1u << po2, domain = size * INV_RATE;
        auto head = read(out_size + 1);
        std::vector<int> head_words;
        for (int h : head) { const auto u = pr.unpack(h); head_words.insert(head_words.end(), u.begin(), u.end()); }
        head_words.resize(out_size + 1);
        { const int k = pr.constant(po2); pr.eq(head_words[out_size], k); }
        io->commit(elems(head));
        Tree tg[3];
        tg[G_CODE] = tree_init(domain, c.group_size[G_CODE]);
        tg[G_DATA] = tree_init(domain, c.group_size[G_DATA]);
        std::vector<int> mix_words;
        for (u32 i = 0; i < mix_size; i++) mix_words.push_back(io->elem());
        tg[G_ACCUM] = tree_init(domain, c.group_size[G_ACCUM]);
        const int poly_mix = io->ext();
        const Tree tcheck = tree_init(domain, CHECK_SIZE);
        const int z = io->ext();
        const u32 back_one = rou_rev(po2);
        const size_t n_taps = c.taps.size(), n_u = n_taps + CHECK_SIZE;
        const auto coeff_u = read((u32)(4 * n_u));         // AoS ExtElems: one wire each
        io->commit(elems(coeff_u));
        std::map<u32, int> zb;                              // z * w^-back
        auto z_back = [&](u32 back) {
            auto it = zb.find(back);
            if (it != zb.end()) return it->second;
            const int v = back == 0 ? z : pr.scale(z, powm(back_one, back));
            zb[back] = v;
            return v;
        };
        std::vector<int> eval_u;
        size_t at = 0;
        for (const Reg& r : c.regs) {
            for (u32 bk : r.backs) { const int zbk = z_back(bk); eval_u.push_back(horner(&coeff_u[at], r.backs.size(), zbk)); }
            at += r.backs.size();
        }
        const int result = poly_ext(c, poly_mix, eval_u, head_words, mix_words);
        ZKH_REQUIRE(result >= 0, "rec_build: the circuit description has an unknown step");
        // check(z) from its 16 coefficient planes, times Z(z) = (3 z)^size - 1
        std::vector<int> zi = {pr.constant(1), z};
        zi.push_back(pr.mul(z, z));
        zi.push_back(pr.mul(zi[2], z));
        int basis[4];
        for (u32 k = 0; k < 4; k++) basis[k] = pr.constant(k == 0, k == 1, k == 2, k == 3);
        const u32 remap[4] = {0, 2, 1, 3};
        int check = -1;
        for (u32 i = 0; i < 4; i++)
            for (u32 k = 0; k < 4; k++) {
                const int zb_ik = k == 0 ? zi[i] : pr.mul(zi[i], basis[k]);
                const int cu = coeff_u[n_taps + remap[i] + 4 * k];
                check = check < 0 ? pr.mul(cu, zb_ik) : pr.muladd(cu, zb_ik, check);
            }
        int t3 = pr.scale(z, 3);
        for (u32 i = 0; i < po2; i++) t3 = pr.mul(t3, t3);
        const int zv = pr.scale(t3, 1, FP - 1);
        { const int lhs = pr.mul(check, zv); pr.eq(lhs, result); }
        // DEEP: combine the U coefficients per combo with powers of mix
        const int mix = io->ext();
        std::vector<size_t> combo_begin = {0};
        for (const auto& cb : c.combos) combo_begin.push_back(combo_begin.back() + cb.size());
        const size_t tot_backs = combo_begin.back();
        std::vector<int> combo_u(tot_backs + 1, -1), mix_pows;
        int cur = pr.constant(1);
        at = 0;
        auto acc_into = [&](size_t slot, int a, int b) { combo_u[slot] = combo_u[slot] < 0 ? pr.mul(a, b) : pr.muladd(a, b, combo_u[slot]); };
        for (const Reg& r : c.regs) {
            for (size_t i = 0; i < r.backs.size(); i++) acc_into(combo_begin[r.combo] + i, cur, coeff_u[at + i]);
            mix_pows.push_back(cur);
            cur = pr.mul(cur, mix);
            at += r.backs.size();
        }
        for (u32 i = 0; i < CHECK_SIZE; i++) {
            acc_into(tot_backs, cur, coeff_u[at]);
            at++;
            mix_pows.push_back(cur);
            cur = pr.mul(cur, mix);
        }
        for (auto& x : combo_u) if (x < 0) x = pr.zero();
        // FRI commitments
        std::vector<FriRound> rounds;
        u32 degree = size, dom = domain;
        while (degree > FRI_MIN_DEGREE) {
            FriRound r;
            r.tree = tree_init(dom / FRI_FOLD, FRI_FOLD * EXT);
            const int rmix = io->ext();
            r.mxn.push_back(pr.constant(invm(16)));
            for (int i = 0; i < 15; i++) r.mxn.push_back(pr.mul(r.mxn.back(), rmix));
            r.domain = dom;
            rounds.push_back(std::move(r));
            dom /= FRI_FOLD;
            degree /= FRI_FOLD;
        }
        const auto fin = read(EXT * degree);               // component planes: word j * degree + i = coefficient i, component j
        io->commit(elems(fin));
        ZKH_REQUIRE(degree >= 4, "rec_build: segments below 4 rows are not supported");
        const u32 q4 = degree / 4;
        std::vector<int> final_poly;
        for (u32 i = 0; i < degree; i++) final_poly.push_back(pr.pack(i % 4, fin[i / 4], fin[q4 + i / 4], fin[2 * q4 + i / 4], fin[3 * q4 + i / 4]));
        const u32 gen_final = rou_fwd(log2_ceil(dom)), gen0 = rou_fwd(log2_ceil(domain));
        const int z4 = pr.mul(zi[2], zi[2]);
        const u32 L = log2_ceil(domain);
        for (u32 q = 0; q < QUERIES; q++) {
            int v = io->elem();
            for (int k = 0; k < 3; k++) {
                const int nv = io->elem();
                const int isz = pr.is_zero(v);
                v = pr.mux(isz, v, nv);
            }
            auto bits = pr.bits31(v, L);
            bits.resize(L);
            const int x = pow_bits(gen0, bits.data(), L);
            std::vector<int> rows[3];
            for (int g = 0; g < 3; g++)
                for (int h : tree_open(tg[g], bits)) { const auto u = pr.unpack(h); rows[g].insert(rows[g].end(), u.begin(), u.end()); }
            std::vector<int> check_row;
            for (int h : tree_open(tcheck, bits)) { const auto u = pr.unpack(h); check_row.insert(check_row.end(), u.begin(), u.end()); }
            std::vector<int> tot(c.combos.size() + 1, -1);
            auto acc_tot = [&](size_t slot, int a, int b) { tot[slot] = tot[slot] < 0 ? pr.mul(a, b) : pr.muladd(a, b, tot[slot]); };
            for (size_t r = 0; r < c.regs.size(); r++) acc_tot(c.regs[r].combo, mix_pows[r], rows[c.regs[r].group][c.regs[r].offset]);
            for (u32 i = 0; i < CHECK_SIZE; i++) acc_tot(c.combos.size(), mix_pows[c.regs.size() + i], check_row[i]);
            int goal = -1;
            for (size_t i = 0; i < c.combos.size(); i++) {
                int divisor = -1;
                for (u32 bk : c.combos[i]) {
                    const int zbk = z_back(bk);
                    const int f = pr.sub(x, zbk);
                    divisor = divisor < 0 ? f : pr.mul(divisor, f);
                }
                const int ti = tot[i] >= 0 ? tot[i] : pr.zero();
                const int hv = horner(&combo_u[combo_begin[i]], combo_begin[i + 1] - combo_begin[i], x);
                const int num = pr.sub(ti, hv);
                const int di = pr.inv(divisor);
                goal = goal < 0 ? pr.mul(num, di) : pr.muladd(num, di, goal);
            }
            {
                const int num = pr.sub(tot[c.combos.size()], combo_u[tot_backs]);
                const int f = pr.sub(x, z4);
                const int di = pr.inv(f);
                goal = pr.muladd(num, di, goal);
            }
            std::vector<int> pbits = bits;
            for (const FriRound& r : rounds) {
                const u32 lg = log2_ceil(r.domain / FRI_FOLD);
                std::vector<int> group_bits(pbits.begin(), pbits.begin() + lg), quot_bits(pbits.begin() + lg, pbits.begin() + lg + 4);
                const auto data = tree_open(r.tree, group_bits);
                std::vector<int> vals;
                for (u32 i = 0; i < 16; i++) vals.push_back(pr.pack(i % 4, data[i / 4], data[4 + i / 4], data[8 + i / 4], data[12 + i / 4]));
                std::vector<int> cand = vals;
                for (int b : quot_bits) {
                    std::vector<int> nxt;
                    for (size_t i = 0; i < cand.size() / 2; i++) nxt.push_back(pr.mux(b, cand[2 * i], cand[2 * i + 1]));
                    cand.swap(nxt);
                }
                pr.eq(cand[0], goal);
                const int inv_wk = pow_bits(rou_rev(log2_ceil(r.domain)), group_bits.data(), group_bits.size());
                goal = fold_eval(vals, r.mxn, inv_wk);
                pbits = group_bits;
            }
            const int xf = pow_bits(gen_final, pbits.data(), pbits.size());
            const int hv = horner(final_poly.data(), final_poly.size(), xf);
            pr.eq(hv, goal);
        }
        res.out.assign(head_words.begin(), head_words.begin() + out_size);
        res.head = head;
        res.code_root = tg[G_CODE].top[1];
        return nullptr;
    }
    std::vector<int> repack(const std::vector<int>& words) {
        const int z = pr.zero();
        std::vector<int> out;
        for (size_t i = 0; i < words.size(); i += 4) {
            int w[4] = {z, z, z, z};
            for (size_t k = 0; k < 4 && i + k < words.size(); k++) w[k] = words[i + k];
            out.push_back(pr.pack(0, w[0], w[1], w[2], w[3], true));
        }
        return out;
    }
    std::vector<int> digest_words(const W2& d) {
        std::vector<int> out;
        for (int h : d) { const auto u = pr.unpack(h); out.insert(out.end(), u.begin(), u.end()); }
        return out;
    }
    void allowed_member(const W2& root, const W2& allowed) {
        W2 cur = root;
        for (u32 l = 0; l < ALLOWED_DEPTH; l++) {
            const auto w = read(1);
            const int b = pr.unpack(w[0])[0];
            pr.boolean(b);
            const auto sib = read(8);
            cur = pair_at(b, cur, {sib[0], sib[1]});
        }
        pr.eq(cur[0], allowed[0]);
        pr.eq(cur[1], allowed[1]);
    }
    // claim' = hash_pair(core, (pre, post, 0, 0, 0, 0, 0, 0))
    W2 wrap(const W2& core, int pre, int post) {
        const int z = pr.zero();
        const int st = pr.pack(0, pre, post, z, z, true);
        return pair(core, {st, z});
    }
    // verify one segment seal (fresh transcript), pin its code root -> (claim digest, pre, post)
    const char* segment_claim(const Desc& c, u32 po2, const u32* control_root, W2& claim, int& pre, int& post) {
        io.reset(new Sponge(pr));
        SealOut s;
        ZKH_TRY(verify_seal(c, po2, s));
        const auto root_words = digest_words(s.code_root);
        for (size_t i = 0; i < 8; i++) { const int k = pr.constant(control_root[i]); pr.eq(root_words[i], k); }
        const bool chained = c.kind == 1 && (c.global_size[0] == 5 || c.global_size[0] == 23);   // SYN-C / SYN-S: out = (post, 0, 0, 0, pre, ..)
        if (chained) { pre = s.out[4]; post = s.out[0]; } else { pre = pr.zero(); post = pr.zero(); }
        std::vector<int> words = s.out;
        words.push_back(pr.constant(po2));
        words.insert(words.end(), root_words.begin(), root_words.end());
        claim = elems(repack(words));
        return nullptr;
    }
};

const char* build_lift(Program& pr, const Desc& c, u32 po2, const u32* root) {
    Verifier v(pr);
    W2 claim;
    int pre, post;
    ZKH_TRY(v.segment_claim(c, po2, root, claim, pre, post));
    const W2 wrapped = v.wrap(claim, pre, post);
    const auto allowed = v.read(8);
    pr.publish(wrapped[0], wrapped[1], allowed[0], allowed[1]);
    return nullptr;
}
const char* build_lift2(Program& pr, const Desc& c, u32 po2_l, const u32* root_l, u32 po2_r, const u32* root_r) {
    Verifier v(pr);
    W2 left, right;
    int pre_l, post_l, pre_r, post_r;
    ZKH_TRY(v.segment_claim(c, po2_l, root_l, left, pre_l, post_l));
    ZKH_TRY(v.segment_claim(c, po2_r, root_r, right, pre_r, post_r));
    pr.eq(post_l, pre_r);                                  // continuity: the right segment starts where the left one ended
    const W2 wl = v.wrap(left, pre_l, post_l), wr = v.wrap(right, pre_r, post_r);
    const W2 parent = v.pair(wl, wr);
    const W2 wrapped = v.wrap(parent, pre_l, post_r);
    const auto allowed = v.read(8);
    pr.publish(wrapped[0], wrapped[1], allowed[0], allowed[1]);
    return nullptr;
}
const char* build_join(Program& pr, const Desc& c, const u32* po2s, size_t n_children) {
    Verifier v(pr);
    std::vector<W2> claims;
    std::vector<std::array<int, 2>> states;
    W2 allowed{-1, -1};
    for (size_t k = 0; k < n_children; k++) {
        v.io.reset(new Sponge(pr));
        SealOut s;
        ZKH_TRY(v.verify_seal(c, po2s[k], s));
        ZKH_REQUIRE(s.head.size() >= 4, "rec_build: a join's children must be recursion seals (16 outputs)");
        if (k == 0) allowed = {s.head[2], s.head[3]};
        else { pr.eq(s.head[2], allowed[0]); pr.eq(s.head[3], allowed[1]); }
        v.allowed_member(s.code_root, allowed);
        const auto core = v.read(8);
        const auto stw = v.read(2);
        const auto st = pr.unpack(stw[0]);                 // (pre, post, 0, 0): the padding is constrained by read()
        const W2 opened = v.wrap({core[0], core[1]}, st[0], st[1]);
        pr.eq(opened[0], s.head[0]);
        pr.eq(opened[1], s.head[1]);
        claims.push_back({s.head[0], s.head[1]});
        states.push_back({st[0], st[1]});
    }
    W2 node = claims[0];
    int pre = states[0][0], post = states[0][1];
    for (size_t k = 1; k < n_children; k++) {
        pr.eq(post, states[k][0]);                         // continuity: child k starts where the node so far ended
        node = v.wrap(v.pair(node, claims[k]), pre, states[k][1]);
        post = states[k][1];
    }
    pr.publish(node[0], node[1], allowed[0], allowed[1]);
    return nullptr;
}

// one child RECURSION seal: verified under a fresh transcript, its allowed root tied to the program's, its program a member of the set
const char* rec_child(Verifier& v, Program& pr, const Desc& c, u32 po2, bool first, W2& allowed, W2& claim) {
    v.io.reset(new Sponge(pr));
    SealOut s;
    ZKH_TRY(v.verify_seal(c, po2, s));
    ZKH_REQUIRE(s.head.size() >= 4, "rec_build: the children of a union / resolve must be recursion seals (16 outputs)");
    if (first) allowed = {s.head[2], s.head[3]};
    else { pr.eq(s.head[2], allowed[0]); pr.eq(s.head[3], allowed[1]); }
    v.allowed_member(s.code_root, allowed);
    claim = {s.head[0], s.head[1]};
    return nullptr;
}
// union: two receipts of any claims -> wrap(hash_pair of the pair, swapped when the witness bit is set, 0, 0) (rec_verify.py build_union)
const char* build_union(Program& pr, const Desc& c, const u32* po2s) {
    Verifier v(pr);
    W2 allowed{-1, -1}, left, right;
    ZKH_TRY(rec_child(v, pr, c, po2s[0], true, allowed, left));
    ZKH_TRY(rec_child(v, pr, c, po2s[1], false, allowed, right));
    const auto w = v.read(1);
    const int b = pr.unpack(w[0])[0];
    pr.boolean(b);
    const W2 node = v.wrap(v.pair_at(b, left, right), pr.zero(), pr.zero());
    pr.publish(node[0], node[1], allowed[0], allowed[1]);
    return nullptr;
}
// resolve: the conditional receipt (claim' opened: core, pre, post) bound to its assumption receipt; the state range is the conditional's
const char* build_resolve(Program& pr, const Desc& c, const u32* po2s) {
    Verifier v(pr);
    W2 allowed{-1, -1}, cond, assum;
    ZKH_TRY(rec_child(v, pr, c, po2s[0], true, allowed, cond));
    const auto core = v.read(8);
    const auto stw = v.read(2);
    const auto st = pr.unpack(stw[0]);
    const W2 opened = v.wrap({core[0], core[1]}, st[0], st[1]);
    pr.eq(opened[0], cond[0]);
    pr.eq(opened[1], cond[1]);
    ZKH_TRY(rec_child(v, pr, c, po2s[1], false, allowed, assum));
    const W2 node = v.wrap(v.pair(cond, assum), st[0], st[1]);
    pr.publish(node[0], node[1], allowed[0], allowed[1]);
    return nullptr;
}

}  // namespace

extern "C" const char* zkh_rec_build_program(uint32_t kind, const uint32_t* child_desc, size_t child_desc_words, const uint32_t* po2s,
                                             const uint32_t* control_roots, uint32_t zk_cycles, uint32_t** blob, size_t* words) {
    ZKH_REQUIRE(child_desc && po2s && blob && words, "rec_build_program: null argument");
    ZKH_REQUIRE(kind <= 5, "rec_build_program: kind %u (0 lift, 1 join, 2 lift2, 3 join3, 4 union, 5 resolve)", kind);
    const size_t n_children = kind == 0 ? 1 : kind == 3 ? 3 : 2;
    for (size_t k = 0; k < n_children; k++) ZKH_REQUIRE(po2s[k] >= 4 && po2s[k] <= 24, "rec_build_program: child po2 %u", po2s[k]);
    ZKH_REQUIRE((kind == 0 || kind == 2) == (control_roots != nullptr), "rec_build_program: lifts take the segment circuit's control roots, joins take none");
    // control roots arrive as the library hands them out (zkh_syn_control_root, zkh_prover_cached_code_root: Montgomery words); the
    // program compares the child's code root with CONSTANTS, which are canonical residues
    std::vector<u32> roots;
    if (control_roots) {
        const u32 rinv = invm((u32)(((u64)1 << 32) % FP));
        for (size_t i = 0; i < 8 * n_children; i++) {
            ZKH_REQUIRE(control_roots[i] < FP, "rec_build_program: a control root word is not a reduced field element");
            roots.push_back(mulm(control_roots[i], rinv));
        }
        control_roots = roots.data();
    }
    Desc c;
    ZKH_TRY(parse_desc(child_desc, child_desc_words, c));
    if (kind == 1 || kind >= 3) ZKH_REQUIRE(c.kind == 4 && c.global_size[0] == 16, "rec_build_program: a join / union / resolve verifies seals of the RECURSION circuit");
    Program pr;
    if (kind == 0) ZKH_TRY(build_lift(pr, c, po2s[0], control_roots));
    else if (kind == 2) ZKH_TRY(build_lift2(pr, c, po2s[0], control_roots, po2s[1], control_roots + 8));
    else if (kind == 4) ZKH_TRY(build_union(pr, c, po2s));
    else if (kind == 5) ZKH_TRY(build_resolve(pr, c, po2s));
    else ZKH_TRY(build_join(pr, c, po2s, n_children));
    const u32 po2 = pr.min_po2(zk_cycles);
    ZKH_REQUIRE(po2, "rec_build_program: the program is too large");
    std::vector<u32> out;
    ZKH_TRY(pr.finish(po2, zk_cycles, out));
    *blob = (uint32_t*)malloc(out.size() * 4);
    ZKH_REQUIRE(*blob, "rec_build_program: out of memory");
    memcpy(*blob, out.data(), out.size() * 4);
    *words = out.size();
    return nullptr;
}
