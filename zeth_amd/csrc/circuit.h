// circuit.h — host-side circuit description (TapSet + PolyExtStep list as data) shared by circuit.hip, the
// generated eval_check kernels and the segment prover.  Mirrors risc0-zkp 3.0.2 src/taps.rs + src/adapter.rs
// (un-vendored; /root/reference/Cargo.lock:5393); blob layout: zeth_amd/circuits/desc.py.
#pragma once
#include <cstdint>
#include <vector>

#include "common.h"

namespace zkh {

constexpr uint32_t DESC_MAGIC = 0x5a4b4331u;
constexpr uint32_t DESC_HEADER = 16;
enum : uint32_t { GROUP_ACCUM = 0, GROUP_CODE = 1, GROUP_DATA = 2 };
enum : uint32_t { GLOBAL_OUT = 0, GLOBAL_MIX = 1 };
enum : uint32_t { OP_CONST = 0, OP_CONST_EXT, OP_GET, OP_GET_GLOBAL, OP_ADD, OP_SUB, OP_MUL, OP_TRUE, OP_AND_EQZ, OP_AND_COND };

struct Tap { uint32_t group, offset, back; };
struct Step { uint32_t op, a, b, c, d; };
struct Reg { uint32_t group, offset, tap_begin, size, combo_id; };

// Arguments every eval_check kernel (generated or interpreted) receives.
struct EvalCheckArgs {
    uint32_t* check;                 // 4 planes x dom
    const uint32_t* groups[3];       // evaluated accum, code, data (W x dom)
    const uint32_t* globals[2];      // out, mix (device)
    const uint32_t* mix_pows;        // poly_mix^e, e < n_mix_pows (ExtElem each, device)
    uint32_t zinv[4];                // 1 / (3^n * i^(idx mod 4) - 1), Montgomery
    uint32_t dom;                    // 4n
    uint32_t accumulate;             // 0: check = part; 1: check += part (parts 1.. of a split constraint system)
};
// Tap read of a generated eval_check kernel: wave-uniform column base (SGPR pair) + 32-bit per-lane byte offset, which is
// the addressing form global_load has natively (no 64-bit VGPR address per load).
__device__ __forceinline__ uint32_t tap_load(const uint32_t* __restrict__ group, size_t column_words, uint32_t lane_byte_offset) {
    return *(const uint32_t*)((const char*)(group + column_words) + lane_byte_offset);
}
// tot += p * x for an Fp4-valued constraint x and the mix power p, into the four unreduced 64-bit component sums of the
// generated kernels: only the three overflow coefficients (x^4..x^6, which meet -11) are reduced; the sixteen products
// land in the sums as they are (at most four per component: the caller accounts four units of room).
__device__ __forceinline__ void ext_accumulate(uint64_t& s0, uint64_t& s1, uint64_t& s2, uint64_t& s3, const uint4 p, const zkh::Fp4& x) {
    using namespace zkh;
    const uint64_t x0 = x.c[0].v, x1 = x.c[1].v, x2 = x.c[2].v, x3 = x.c[3].v;
    const uint64_t h0 = mont_reduce_wide(p.y * x3 + p.z * x2 + p.w * x1);
    const uint64_t h1 = mont_reduce_wide(p.z * x3 + p.w * x2);
    const uint64_t h2 = mont_reduce(p.w * x3);
    s0 += p.x * x0 + NBETA_M * h0;
    s1 += p.x * x1 + p.y * x0 + NBETA_M * h1;
    s2 += p.x * x2 + p.y * x1 + p.z * x0 + NBETA_M * h2;
    s3 += p.x * x3 + p.y * x2 + p.z * x1 + p.w * x0;
}
// x * v with lazy components (each in [0, 2P)): for an Fp4 whose every consumer multiplies it by a canonical base value
__device__ __forceinline__ zkh::Fp4 ext_mul_base_lazy(const zkh::Fp4& x, uint32_t v) {
    using namespace zkh;
    return Fp4(Fp::raw(mul_lazy(x.c[0].v, v)), Fp::raw(mul_lazy(x.c[1].v, v)), Fp::raw(mul_lazy(x.c[2].v, v)), Fp::raw(mul_lazy(x.c[3].v, v)));
}
// x + v for an Fp4 x and a base-field v (only the constant coefficient moves)
__device__ __forceinline__ zkh::Fp4 ext_add_base(const zkh::Fp4& x, uint32_t v) {
    return zkh::Fp4(x.c[0] + zkh::Fp::raw(v), x.c[1], x.c[2], x.c[3]);
}
__device__ __forceinline__ zkh::Fp4 ext_sub_base(const zkh::Fp4& x, uint32_t v) {
    return zkh::Fp4(x.c[0] - zkh::Fp::raw(v), x.c[1], x.c[2], x.c[3]);
}
typedef void (*eval_check_launch_fn)(const EvalCheckArgs&, hipStream_t);
// A circuit's generated eval_check: n_parts kernels over disjoint constraint ranges, launched back to back on one stream;
// part 0 writes `check`, the others add their share (circuits/codegen.py).
// gather_exps (optional): per part, {count, exponent of slot 0, 1, ...} — the part reads its mix powers from a table of its own,
// gathered into the order its code touches them (codegen.py GATHER); NULL: every part indexes the plain table mix^0, mix^1, ...
// gather_consts (optional, with gather_exps): per part, {count, then (slot, c0, c1, c2, c3) each} — slots whose power is multiplied by an
// Fp4 constant (Montgomery words) when the table is built (codegen.py LINFORM: an Fp4 constraint linear over constants becomes base leaves).
struct CompiledEvalCheck { uint64_t desc_hash; const char* name; const eval_check_launch_fn* parts; uint32_t n_parts; uint32_t n_mix_pows;
                           const uint32_t* const* gather_exps; const uint32_t* const* gather_consts; };
// registry filled by the generated translation unit (eval_check_gen.hip)
const CompiledEvalCheck* find_compiled_eval_check(uint64_t desc_hash);

uint64_t desc_hash64(const uint32_t* words, size_t n);
// inclusive prefix sum (mod P) of the first A words of a device column, in place; *last_out (device) = the grand total (circuit.hip)
const char* prefix_sum_column(zkh_ctx* ctx, uint32_t* col, uint32_t A, uint32_t* last_out);

// device program for the generic interpreter (slots allocated on the host by liveness).  op = opcode | kind(a) << 8 |
// kind(b) << 11 | (dst is Fp4) << 14; operand kinds: taps, constants and globals are operands, not slots, so a tap that
// thousands of constraints read does not pin a slot for the whole program.
struct InterpInsn { uint32_t op, dst, a, b, c, w; };   // w = index into mix_pows (ConstExt: 4th word)
enum : uint32_t { OPK_FP = 0, OPK_EXT = 1, OPK_TAP = 2, OPK_CONST = 3, OPK_GLOBAL = 4 };

}  // namespace zkh

struct zkh_circuit {
    zkh_ctx* ctx;
    std::vector<uint32_t> desc;
    uint64_t hash;
    uint32_t group_size[3];
    uint32_t global_size[2];
    uint32_t ret, kind;
    std::vector<zkh::Tap> taps;
    std::vector<std::vector<uint32_t>> combos;
    std::vector<zkh::Step> steps;
    std::vector<zkh::Reg> regs;
    size_t tot_combo_backs;
    const zkh::CompiledEvalCheck* compiled;
    // code objects attached at run time (zkh_circuit_attach_code_object[_part]): eval_check kernels generated for THIS desc
    std::vector<hipModule_t> jit_modules;
    std::vector<hipFunction_t> jit_kernels;      // one per part; all non-null once every part is attached
    // gathered power tables ([0] built-in kernels, [1] attached code objects): device exponent list of all parts back to back,
    // and where each part's slots start (n_parts + 1 entries); empty = the parts index the plain table
    uint32_t* d_gather[2];
    std::vector<uint32_t> gather_off[2];
    uint32_t* d_gconst[2];                         // (slot, c0..c3) records of all parts, slots rebased to the concatenated table
    uint32_t n_gconst[2];
    bool gather_centered[2];                       // some slot is read centred (bit 31 of its exponent word): the table build ends with k_ext_center_at
    std::vector<std::vector<uint32_t>> jit_exps;   // per attached part: its exponent list ({} = none exported)
    std::vector<std::vector<uint32_t>> jit_pwc;    // per attached part: its slot-constant list ({} = none exported)
    bool jit_mixed;       // the attached parts disagree about the table (a set half replaced): not launched until repaired
    bool interp_ok;       // the step interpreter's live values fit its LDS
    // interpreter program
    std::vector<zkh::InterpInsn> prog;
    uint32_t n_fp_slots, n_mix_slots, n_mix_pows, ret_slot;   // n_mix_slots: 16-byte slots (mix totals and Fp4-valued values)
    uint32_t* d_prog;     // device copy of prog
    uint32_t* d_taps;     // device copy of taps (group, offset, back)
};

// Circuits whose segments chain: SYN-C (kind 1, out = (post, 0, 0, 0, pre): 5 words) and SYN-S (zeth_amd/circuits/syn_air.py
// syn_session: the same plus the exit code pair and the 16 limbs of the output digest: 23 words).  Every word past the first four
// is a public input bound by a `first`-gated constraint.
constexpr uint32_t SESSION_OUT_WORDS = 23, SESSION_EXIT_SYS = 5, SESSION_EXIT_USER = 6, SESSION_JOURNAL = 7, SESSION_JOURNAL_LIMBS = 16;
constexpr uint32_t EXIT_SYS_HALTED = 0, EXIT_SYS_PAUSED = 1, EXIT_SYS_SPLIT = 2;        // ExitCode::into_pair (risc0-binfmt, recalled)
inline bool circuit_is_session(const zkh_circuit* c) { return c->kind == 1 && c->global_size[0] == SESSION_OUT_WORDS; }
inline bool circuit_has_state(const zkh_circuit* c) { return c->kind == 1 && (c->global_size[0] == 5 || c->global_size[0] == SESSION_OUT_WORDS); }
namespace zkh {
// the OUTPUT digest a session's last seal binds = tagged_struct("risc0.Output", [SHA-256(journal), Assumptions digest]) as sixteen
// 16-bit limbs (assumptions == NULL: the session assumed nothing, the zero digest) — verifier.hip
void session_output_limbs(const uint8_t* journal, size_t journal_len, const uint32_t assumptions[8], uint32_t limbs[16]);
void session_journal_limbs(uint32_t final_state_mont, const uint32_t assumptions[8], uint32_t limbs[16]);         // journal = the final state word
void assumptions_digest(const uint32_t* claims, const uint32_t* control_roots, size_t n, uint32_t out[8]);
const char* check_session_termination(const uint32_t* const* seals, size_t n, const uint8_t* journal, size_t journal_len, const uint32_t assumptions[8]);
}
