/* preflight.c — CPU restatement of the trace-driven witness path: a sequential per-cycle machine (the
 * "preflight": the stand-in for what `ExecutorEnvBuilder ... write(&input)` + the rv32im executor turn into per-cycle traces
 * upstream, /root/reference/crates/host/src/lib.rs:132-136; risc0-circuit-rv32im 4.0.2 `prove/witgen/preflight.rs`, un-vendored:
 * /root/reference/Cargo.lock:5320) and the row fill that expands its compact records into the SYN-AIR data group.
 *
 * TEST INFRASTRUCTURE (see zkoracle.h): the product's twin is zeth_amd/csrc/preflight.hip (host machine + k_syn_rowfill).
 *
 * The machine ("SYN-VM"): 8 registers, a 64-instruction program and a 1024-word RAM image, all derived from the segment seed;
 * one instruction per cycle (ADD, MUL, ADDI, LOAD, STORE, BNE over BabyBear residues).  It is inherently sequential — cycle r
 * needs the registers and memory cycle r - 1 left — which is the property of the real preflight that matters for the pipeline.
 * Per cycle it emits ONE 16-byte record: w0 = the value produced, w1 = operand b, w2 = pc | op << 8 | rd << 12 | addr << 16,
 * w3 = the machine's running state digest (16 dependent mixing rounds per cycle over value, address and pc) reduced mod P.
 * All four words are < P, i.e. valid raw Elem words. */
#include <string.h>

#include "circuit.h"
#include "field.h"
#include "zkoracle.h"

#define PF_RAM 1024u
#define PF_PROG 64u
#define PF_REGS 8u

static uint64_t pf_next(uint64_t* st) {
    *st += 0x9E3779B97F4A7C15ull;
    uint64_t z = *st;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

size_t zko_syn_preflight_ram_words(void) { return PF_RAM; }

void zko_syn_preflight(uint64_t seed, unsigned po2, unsigned zk, uint32_t* records, uint32_t* ram_image) {
    const size_t A = ((size_t)1 << po2) - zk;
    struct { uint32_t op, rd, rs1, rs2, imm, target; } prog[PF_PROG];
    uint32_t ram[PF_RAM], reg[PF_REGS];
    uint64_t st = seed ^ 0x5EEDF11E5EEDF11Eull;
    for (uint32_t i = 0; i < PF_PROG; i++) {
        const uint64_t w = pf_next(&st);
        prog[i].op = (uint32_t)(w % 6); prog[i].rd = (uint32_t)(w >> 8) & 7; prog[i].rs1 = (uint32_t)(w >> 16) & 7;
        prog[i].rs2 = (uint32_t)(w >> 24) & 7; prog[i].imm = (uint32_t)(w >> 32) % FP_P;
        prog[i].target = (uint32_t)(pf_next(&st) % PF_PROG);
    }
    for (uint32_t k = 0; k < PF_RAM; k++) ram[k] = (uint32_t)(pf_next(&st) >> 32) % FP_P;
    for (uint32_t k = 0; k < PF_REGS; k++) reg[k] = (uint32_t)(pf_next(&st) >> 32) % FP_P;
    if (ram_image) memcpy(ram_image, ram, sizeof ram);          /* the preload: RAM as it is BEFORE the first cycle */
    uint32_t pc = 0;
    uint64_t h = st;
    for (size_t r = 0; r < A; r++) {
        const uint32_t op = prog[pc].op, rd = prog[pc].rd, a = reg[prog[pc].rs1], b = reg[prog[pc].rs2], imm = prog[pc].imm;
        uint32_t v = 0, addr = 0, next = (pc + 1) % PF_PROG;
        switch (op) {
        case 0: v = (uint32_t)(((uint64_t)a + b) % FP_P); reg[rd] = v; break;                 /* ADD  */
        case 1: v = (uint32_t)(((uint64_t)a * b) % FP_P); reg[rd] = v; break;                 /* MUL  */
        case 2: v = (uint32_t)(((uint64_t)a + imm) % FP_P); reg[rd] = v; break;               /* ADDI */
        case 3: addr = (a ^ b) & (PF_RAM - 1); v = ram[addr]; reg[rd] = v; break;              /* LOAD */
        case 4: addr = (a ^ imm) & (PF_RAM - 1); ram[addr] = b; v = b; break;                  /* STORE */
        default: v = a != b; if (v) next = prog[pc].target; break;                             /* BNE  */
        }
        /* the machine's running state digest: every cycle folds (value, address, pc) in through 16 dependent mixing rounds — the
         * stand-in for the per-cycle bookkeeping of a real preflight (paging, memory-transaction log), and what makes a cycle cost
         * tens of nanoseconds rather than one */
        h ^= (uint64_t)v | (uint64_t)(addr << 8 | pc) << 32;
        for (int k = 0; k < 16; k++) { h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; }
        (void)a;
        records[4 * r] = v; records[4 * r + 1] = b; records[4 * r + 2] = pc | op << 8 | rd << 12 | addr << 16; records[4 * r + 3] = (uint32_t)(h >> 33) % FP_P;
        pc = next;
    }
}

/* seed of the hashed cells of row r: the record, folded */
static uint64_t pf_rowseed(const uint32_t* rec) {
    const uint64_t lo = (uint64_t)rec[1] << 32 | rec[0], hi = (uint64_t)rec[2] << 32 | rec[3];
    return lo ^ (hi << 29 | hi >> 35);
}

/* records (4 words per active row) + the RAM image -> code, data, out_global of a SYN-AIR segment (kind 1, no public inputs):
 * triple 0 = (w0, w1), triple 1 = (w3, w2) — the machine's own values, so the running sum s and with it out[0] is a checksum of
 * the execution — every other free cell is hashed from the record; products, the degree-4 product and s as in zko_syn_witgen;
 * rows >= A are blinding noise; then the preload: the first unconstrained column (3 T, where the shape has one) gets the RAM
 * image in rows [0, min(1024, A)) — a scatter upstream's witgen does with Hal::scatter. */
void zko_syn_witgen_trace(const zko_circuit* c, unsigned po2, unsigned zk, const uint32_t* noise_key, const uint32_t* records,
                          const uint32_t* ram_image, uint32_t* code, uint32_t* data, uint32_t* out_global) {
    const size_t n = (size_t)1 << po2, A = n - zk, wd = c->group_size[ZKC_GROUP_DATA], T = (wd - 2) / 3;
    if (code) zko_syn_code(c, po2, zk, code);
    fp s = 0;
    for (size_t r = 0; r < n; r++) {
        if (r >= A) {
            for (size_t col = 0; col < wd; col++) data[col * n + r] = zko_noise_cell(noise_key, ZKC_GROUP_DATA, (uint32_t)col, (uint32_t)r);
            continue;
        }
        const uint32_t* rec = records + 4 * r;
        const uint64_t rs = pf_rowseed(rec);
        for (size_t col = 0; col < wd - 2; col++) data[col * n + r] = zko_syn_cell(rs, ZKC_GROUP_DATA, (uint32_t)col, (uint32_t)r);
        data[0 * n + r] = rec[0]; data[1 * n + r] = rec[1]; data[3 * n + r] = rec[3]; data[4 * n + r] = rec[2];
        for (size_t j = 0; j < T; j++) data[(3 * j + 2) * n + r] = fp_mul(data[(3 * j) * n + r], data[(3 * j + 1) * n + r]);
        const fp d0 = data[r], d1 = data[n + r], d3 = data[3 * n + r], d4 = data[4 * n + r];
        data[(wd - 2) * n + r] = fp_mul(fp_mul(d0, d1), fp_mul(d3, d4));
        s = r == 0 ? d0 : fp_add(s, fp_add(d0, fp_mul(fp_from_u32((uint32_t)r), d1)));
        data[(wd - 1) * n + r] = s;
    }
    if (ram_image && wd - 2 > 3 * T)
        for (size_t k = 0; k < PF_RAM && k < A; k++) data[(3 * T) * n + k] = ram_image[k];
    out_global[0] = s; out_global[1] = out_global[2] = out_global[3] = 0;
}
