// preflight.hip — the trace-driven witness path: a sequential host "preflight" that emits a COMPACT
// per-cycle trace (16 bytes per cycle), and the GPU row fill that expands it into the data group the HAL commits.
//
// Upstream: `ExecutorEnvBuilder ... write(&input)` (/root/reference/crates/host/src/lib.rs:132-136) feeds the rv32im executor;
// per segment its preflight (risc0-circuit-rv32im 4.0.2 `prove/witgen/preflight.rs`, un-vendored: /root/reference/Cargo.lock:5320)
// replays the cycles on ONE host thread into a per-cycle record list, and witgen kernels (`-sys` crate) fill the trace rows from
// it: one lane per cycle, plus Hal::scatter for the preloaded memory image.  The executor and its ISA are out of scope (and
// unobtainable); what is restated here is the SHAPE of that pipeline, with the same properties: the producer is inherently
// sequential (cycle r needs the machine state cycle r - 1 left), what crosses PCIe is the compact record list (16.7 MB per
// po2-20 segment instead of the 0.94 GB full trace), the expansion runs where the bandwidth is, and the preload is a scatter.
// The machine ("SYN-VM": 8 registers, 64 instructions, 1024 words of RAM, all derived from the segment seed) is stated twice:
// here (product, C++) and in oracle/preflight.c (checker, C).  SYN-AIR (kind 1) circuits without public inputs only.
#include <chrono>

#include "circuit.h"

using namespace zkh;

namespace {

constexpr uint32_t PF_RAM = 1024, PF_PROG = 64, PF_REGS = 8;

inline uint64_t pf_next(uint64_t& st) {
    st += 0x9E3779B97F4A7C15ull;
    uint64_t z = st;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// same hash as circuit.hip's syn_cell (the blinding rows must be the closed-form generator's)
__device__ __forceinline__ uint32_t pf_cell(uint64_t seed, uint32_t group, uint32_t col, uint32_t row) {
    uint64_t z = seed ^ ((uint64_t)(group + 1) * 0x9E3779B97F4A7C15ull);
    z += (uint64_t)col * 0xBF58476D1CE4E5B9ull;
    z += (uint64_t)row * 0x94D049BB133111EBull;
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return mul_mod(R2, (uint32_t)(z >> 32) % P);
}

// One lane per cycle: 16 bytes in (one coalesced uint4 load per lane), wd words out (column-major: lanes of a wave write 64
// consecutive words of every column).  Triple 0 = (w0, w1) and triple 1 = (w3, w2) are the machine's own values, every other
// free cell is hashed from the record; products, the degree-4 product and the running-sum increment as in k_syn_data.
__global__ __launch_bounds__(256) void k_syn_rowfill(uint32_t* __restrict__ data, const uint4* __restrict__ records, uint32_t wd, uint32_t n,
                                                     uint32_t A, NoiseKey nk) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    if (r >= A) {
        for (uint32_t c = 0; c < wd; c++) data[(size_t)c * n + r] = noise_cell(nk, GROUP_DATA, c, r);      // blinding rows: noise.h
        return;
    }
    const uint4 rec = records[r];                      // (w0, w1, w2, w3)
    const uint64_t lo = (uint64_t)rec.y << 32 | rec.x, hi = (uint64_t)rec.z << 32 | rec.w;
    const uint64_t rs = lo ^ (hi << 29 | hi >> 35);
    const uint32_t T = (wd - 2) / 3;
    uint32_t d0 = 0, d1 = 0, d3 = 0, d4 = 0;
    for (uint32_t j = 0; j < T; j++) {
        uint32_t x, y;
        if (j == 0) { x = rec.x; y = rec.y; }
        else if (j == 1) { x = rec.w; y = rec.z; }
        else { x = pf_cell(rs, GROUP_DATA, 3 * j, r); y = pf_cell(rs, GROUP_DATA, 3 * j + 1, r); }
        data[(size_t)(3 * j) * n + r] = x; data[(size_t)(3 * j + 1) * n + r] = y; data[(size_t)(3 * j + 2) * n + r] = mul_mod(x, y);
        if (j == 0) { d0 = x; d1 = y; }
        if (j == 1) { d3 = x; d4 = y; }
    }
    for (uint32_t c = 3 * T; c < wd - 2; c++) data[(size_t)c * n + r] = pf_cell(rs, GROUP_DATA, c, r);
    data[(size_t)(wd - 2) * n + r] = mul_mod(mul_mod(d0, d1), mul_mod(d3, d4));
    data[(size_t)(wd - 1) * n + r] = r == 0 ? d0 : add_mod(d0, mul_mod(mul_mod(R2, r), d1));   // increment; scanned afterwards
}

}  // namespace

extern "C" size_t zkh_syn_preflight_ram_words(void) { return PF_RAM; }

// The sequential machine, on the CALLING host thread.  records: 4 words per active cycle (2^po2 - zk_cycles of them) — pinned
// memory from zkh_host_alloc if it is to be uploaded with zkh_write_async; ram_image (1024 words, may be NULL): the RAM before
// the first cycle; cpu_seconds (may be NULL): thread CPU time spent, the term a pipeline has to hide.
extern "C" const char* zkh_syn_preflight(uint64_t seed, size_t po2, size_t zk_cycles, uint32_t* records, uint32_t* ram_image, double* cpu_seconds) {
    ZKH_REQUIRE(records && po2 >= 1 && po2 <= 24 && ((size_t)1 << po2) > zk_cycles + 1, "syn_preflight: bad argument");
    const auto t0 = std::chrono::steady_clock::now();
    const size_t A = ((size_t)1 << po2) - zk_cycles;
    struct Ins { uint32_t op, rd, rs1, rs2, imm, target; } prog[PF_PROG];
    uint32_t ram[PF_RAM], reg[PF_REGS];
    uint64_t st = seed ^ 0x5EEDF11E5EEDF11Eull;
    for (uint32_t i = 0; i < PF_PROG; i++) {
        const uint64_t w = pf_next(st);
        prog[i] = Ins{(uint32_t)(w % 6), (uint32_t)(w >> 8) & 7, (uint32_t)(w >> 16) & 7, (uint32_t)(w >> 24) & 7, (uint32_t)(w >> 32) % P, 0};
        prog[i].target = (uint32_t)(pf_next(st) % PF_PROG);
    }
    for (uint32_t k = 0; k < PF_RAM; k++) ram[k] = (uint32_t)(pf_next(st) >> 32) % P;
    for (uint32_t k = 0; k < PF_REGS; k++) reg[k] = (uint32_t)(pf_next(st) >> 32) % P;
    if (ram_image) memcpy(ram_image, ram, sizeof ram);
    uint32_t pc = 0;
    uint64_t h = st;
    for (size_t r = 0; r < A; r++) {
        const Ins& in = prog[pc];
        const uint32_t a = reg[in.rs1], b = reg[in.rs2];
        uint32_t v = 0, addr = 0, next = (pc + 1) % PF_PROG;
        switch (in.op) {
        case 0: v = a + b; v = v >= P ? v - P : v; reg[in.rd] = v; break;                                  // ADD (a, b < P < 2^31)
        case 1: v = (uint32_t)(((uint64_t)a * b) % P); reg[in.rd] = v; break;                                // MUL
        case 2: v = a + in.imm; v = v >= P ? v - P : v; reg[in.rd] = v; break;                              // ADDI
        case 3: addr = (a ^ b) & (PF_RAM - 1); v = ram[addr]; reg[in.rd] = v; break;                        // LOAD
        case 4: addr = (a ^ in.imm) & (PF_RAM - 1); ram[addr] = b; v = b; break;                            // STORE
        default: v = a != b; if (v) next = in.target; break;                                                 // BNE
        }
        // the running state digest: (value, address, pc) folded in through 16 DEPENDENT mixing rounds per cycle — the stand-in for the
        // per-cycle bookkeeping of a real preflight (paging, the memory-transaction log); it is what makes a cycle cost tens of
        // nanoseconds of one core, i.e. what the pipeline has to hide
        h ^= (uint64_t)v | (uint64_t)(addr << 8 | pc) << 32;
        for (int k = 0; k < 16; k++) { h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; }
        uint32_t* rec = records + 4 * r;
        rec[0] = v; rec[1] = b; rec[2] = pc | in.op << 8 | in.rd << 12 | addr << 16; rec[3] = (uint32_t)(h >> 33) % P;
        pc = next;
    }
    if (cpu_seconds) *cpu_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return nullptr;
}

// records (DEVICE buffer, 4 x A words, e.g. uploaded with zkh_write_async) + the RAM image (host, or NULL) -> code (NULL: the
// caller holds this size's code group), data, out_global.  One row-fill launch, the running-sum scan, and the preload through
// Hal::scatter: the first unconstrained data column (3 T, where the shape has one) gets the RAM image in rows [0, 1024).
extern "C" const char* zkh_syn_witgen_trace(zkh_ctx* ctx, const zkh_circuit* c, size_t po2, size_t zk_cycles, const uint32_t* noise_key,
                                            const zkh_buf* records, const uint32_t* ram_image, zkh_buf* code, zkh_buf* data, uint32_t* out_global) {
    ZKH_REQUIRE(ctx && c && records && data && out_global, "syn_witgen_trace: null argument");
    ZKH_REQUIRE(c->kind == 1 && c->global_size[GLOBAL_OUT] == 4, "syn_witgen_trace: only SYN-AIR circuits (kind 1) without public inputs are trace-driven");
    const size_t n = (size_t)1 << po2;
    ZKH_REQUIRE(po2 + 2 <= (size_t)MAX_LOG_N && n > zk_cycles + 1, "syn_witgen_trace: po2 out of range");
    const uint32_t wd = c->group_size[GROUP_DATA], wc = c->group_size[GROUP_CODE], A = (uint32_t)(n - zk_cycles), T = (wd - 2) / 3;
    ZKH_REQUIRE(records->len == 4 * (size_t)A, "syn_witgen_trace: %zu record words, expected 4 x %u active cycles", records->len, A);
    ZKH_REQUIRE(data->len == (size_t)wd * n && (!code || code->len == (size_t)wc * n), "syn_witgen_trace: buffer shape mismatch");
    ZKH_REQUIRE(((uintptr_t)records->ptr() & 15) == 0, "syn_witgen_trace: the record buffer must be 16-byte aligned");
    if (code) ZKH_TRY(zkh_syn_code(ctx, c, po2, zk_cycles, code));
    NoiseKey nk;
    ZKH_TRY(resolve_noise_key(noise_key, &nk));
    {
        ProfScope prof(ctx, "syn_rowfill", 16.0 * A + 4.0 * wd * n);
        k_syn_rowfill<<<(unsigned)((n + 255) / 256), 256, 0, ctx->stream>>>(data->ptr(), (const uint4*)records->ptr(), wd, (uint32_t)n, A, nk);
        ZKH_TRY(last_launch_error("syn_rowfill"));
    }
    Tmp last;
    ZKH_TRY(new_buf(ctx, 1, false, last.out()));
    ZKH_TRY(prefix_sum_column(ctx, data->ptr() + (size_t)(wd - 1) * n, A, last->ptr()));
    if (ram_image && wd - 2 > 3 * T) {
        const uint32_t cnt = PF_RAM < A ? PF_RAM : A;
        std::vector<uint32_t> index(cnt), offsets{0, cnt};
        for (uint32_t k = 0; k < cnt; k++) index[k] = (uint32_t)((size_t)(3 * T) * n + k);
        ZKH_TRY(zkh_scatter(ctx, data, index.data(), offsets.data(), ram_image, 1, cnt));
    }
    out_global[1] = out_global[2] = out_global[3] = 0;
    return zkh_read(ctx, last, out_global, 0, 1);
}
