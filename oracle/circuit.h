/* oracle/circuit.h — internal circuit-description structs for the CPU oracle (TEST INFRASTRUCTURE ONLY).
 * Blob layout is documented in zeth_amd/circuits/desc.py; semantics follow risc0-zkp 3.0.2 src/taps.rs and
 * src/adapter.rs (un-vendored, /root/reference/Cargo.lock:5393). */
#ifndef ZKORACLE_CIRCUIT_H
#define ZKORACLE_CIRCUIT_H
#include "zkoracle.h"

#define ZKC_MAGIC 0x5a4b4331u
#define ZKC_HEADER_WORDS 16
enum { ZKC_GROUP_ACCUM = 0, ZKC_GROUP_CODE = 1, ZKC_GROUP_DATA = 2 };
enum { ZKC_GLOBAL_OUT = 0, ZKC_GLOBAL_MIX = 1 };
enum { ZKC_CONST = 0, ZKC_CONST_EXT = 1, ZKC_GET = 2, ZKC_GET_GLOBAL = 3, ZKC_ADD = 4, ZKC_SUB = 5, ZKC_MUL = 6,
       ZKC_TRUE = 7, ZKC_AND_EQZ = 8, ZKC_AND_COND = 9 };

typedef struct { uint32_t group, offset, back; } zkc_tap;
typedef struct { uint32_t op, a[4]; } zkc_step;
typedef struct { uint32_t group, offset, tap_begin, size, combo_id; } zkc_reg;

struct zko_circuit {
    uint32_t group_size[3];
    uint32_t global_size[2];
    size_t n_taps, n_combos, n_steps, n_regs, tot_combo_backs;
    uint32_t ret, kind;
    zkc_tap* taps;
    uint32_t* combo_begin;   /* n_combos + 1 */
    uint32_t* combo_backs;
    zkc_step* steps;
    zkc_reg* regs;
};
#endif
