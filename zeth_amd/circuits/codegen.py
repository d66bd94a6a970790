"""eval_check code generator: circuit desc (PolyExtStep list) -> straight-line HIP for gfx950.

Upstream ships machine-generated CUDA for `CircuitHal::eval_check` (Zirgen output inside
risc0-circuit-rv32im-sys, un-vendored: /root/reference/Cargo.lock:5320).  Here the generator is part of the
build: `zeth_amd.build` runs it for the shipped circuit shapes and compiles the result into
libzkhal_mi355x.so; at run time `zkh_circuit_load` picks the kernel whose desc hash matches, and any other
desc falls back to the on-device interpreter (circuit.hip).

MI355X-first choices (all result-preserving — field arithmetic is exact):
  * value registers are Fp (the prover evaluates on the base-field coset), only mix totals are Fp4;
  * every MixState's `mul` is a *static* power of poly_mix (True = mix^0, AndEqz adds 1, AndCond adds the
    inner exponent), so the per-step Fp4 x Fp4 product of the literal algorithm disappears: powers come from a
    precomputed table through scalar loads, and AndEqz is 4 multiply-adds;
  * one lane per domain point, every tap read is a coalesced column read;
  * 1/((3x)^n - 1) takes only 4 values on the coset (3^n * i^(idx mod 4)): a 4-entry kernel argument.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np

from . import syn_air
from .desc import (OP_ADD, OP_AND_COND, OP_AND_EQZ, OP_CONST, OP_CONST_EXT, OP_GET, OP_GET_GLOBAL, OP_MUL, OP_SUB,
                   OP_TRUE, Circuit, P)

R2 = pow(2, 64, P)


def desc_hash64(desc: np.ndarray) -> int:
    """FNV-1a over the little-endian bytes of the desc words (must match circuit.hip desc_hash64)."""
    h = 0xCBF29CE484222325
    for b in np.asarray(desc, dtype="<u4").tobytes():
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def mont(x: int) -> int:
    return (x % P) * pow(2, 32, P) % P


def analyse(c: Circuit):
    """-> (fp_of_step, mix_of_step, mix_exp, used_fp, used_mix, n_pows)"""
    mix_exp: List[int] = []
    nf = nm = 0
    kinds = []
    for op, a, b, cc, d in c.steps:
        if op >= OP_TRUE:
            if op == OP_TRUE:
                mix_exp.append(0)
            elif op == OP_AND_EQZ:
                mix_exp.append(mix_exp[a] + 1)
            else:
                mix_exp.append(mix_exp[a] + mix_exp[cc])
            kinds.append(("m", nm))
            nm += 1
        else:
            kinds.append(("f", nf))
            nf += 1
    # liveness from ret backwards
    used_f = [False] * nf
    used_m = [False] * nm
    used_m[c.ret] = True
    for i in range(len(c.steps) - 1, -1, -1):
        op, a, b, cc, d = c.steps[i]
        k, vid = kinds[i]
        live = used_m[vid] if k == "m" else used_f[vid]
        if not live:
            continue
        if op in (OP_ADD, OP_SUB, OP_MUL):
            used_f[a] = used_f[b] = True
        elif op == OP_AND_EQZ:
            used_m[a] = True
            used_f[b] = True
        elif op == OP_AND_COND:
            used_m[a] = used_m[cc] = True
            used_f[b] = True
    max_pow = 0
    for i, (op, a, b, cc, d) in enumerate(c.steps):
        k, vid = kinds[i]
        if k == "m" and used_m[vid] and op in (OP_AND_EQZ, OP_AND_COND):
            max_pow = max(max_pow, mix_exp[a])
    return kinds, mix_exp, used_f, used_m, max_pow + 1


def emit_kernel(name: str, desc: np.ndarray, standalone: bool = False) -> Tuple[str, int, int]:
    """HIP source of `k_eval_check_<name>` for this desc.  standalone=True: an `extern "C"` kernel with no host-side
    launcher, for a code object that is attached at run time (circuits/jit.py)."""
    c = Circuit.parse(desc)
    kinds, mix_exp, used_f, used_m, n_pows = analyse(c)
    L: List[str] = []
    w = L.append
    w(f"// {name}: groups (accum, code, data) = {c.group_sizes}, {len(c.taps)} taps, {len(c.steps)} steps")
    linkage = 'extern "C" ' if standalone else ""
    w(f"{linkage}__global__ __launch_bounds__(256) void k_eval_check_{name}(EvalCheckArgs a) {{")
    w("    const uint32_t idx = blockIdx.x * 256u + threadIdx.x;")
    w("    if (idx >= a.dom) return;")
    w("    const uint32_t mask = a.dom - 1;")
    w("    const size_t dom = a.dom;")
    w("    const uint4* __restrict__ pw = (const uint4*)a.mix_pows;")
    # how often each live mix var is consumed (a var consumed once by an AndEqz can stay a lazy 64-bit sum)
    m_uses: Dict[int, int] = {}
    for i, (op, x, y, z, d) in enumerate(c.steps):
        k, vid = kinds[i]
        if k == "m" and used_m[vid]:
            if op == OP_AND_EQZ:
                m_uses[x] = m_uses.get(x, 0) + 1
            elif op == OP_AND_COND:
                m_uses[x] = m_uses.get(x, 0) + 1
                m_uses[z] = m_uses.get(z, 0) + 1
    m_uses[c.ret] = m_uses.get(c.ret, 0) + 1
    lazy: Dict[int, Tuple] = {}          # vid -> (base vid or None, [(step index of the power load, fp var)])
    materialised = set()
    zero_vars = set()

    def materialise(vid: int) -> None:
        if vid in materialised:
            return
        base, pend = lazy.pop(vid)
        materialised.add(vid)
        for kk, comp in enumerate("xyzw"):
            prods = " + ".join(f"(uint64_t)p{si}.{comp} * f{fv}" for si, fv in pend)
            if base is not None and len(pend) <= 2:
                w(f"    const uint32_t m{vid}_{kk} = mont_reduce_wide(((uint64_t)m{base}_{kk} << 32) + {prods});")
            elif base is not None:
                w(f"    const uint32_t m{vid}_{kk} = add_mod(m{base}_{kk}, mont_reduce_wide({prods}));")
            else:
                w(f"    const uint32_t m{vid}_{kk} = mont_reduce_wide({prods});")

    for i, (op, x, y, z, d) in enumerate(c.steps):
        k, vid = kinds[i]
        if k == "f":
            if not used_f[vid]:
                continue
            v = f"f{vid}"
            if op == OP_CONST:
                w(f"    const uint32_t {v} = {mont(x)}u;")
            elif op == OP_CONST_EXT:
                raise ValueError("ConstExt is not supported on the device path")
            elif op == OP_GET:
                g, off, back = c.taps[x]
                pos = "idx" if back == 0 else f"((idx - {4 * back}u) & mask)"
                w(f"    const uint32_t {v} = a.groups[{g}][(size_t){off} * dom + {pos}];")
            elif op == OP_GET_GLOBAL:
                w(f"    const uint32_t {v} = a.globals[{x}][{y}];")
            elif op == OP_ADD:
                w(f"    const uint32_t {v} = add_mod(f{x}, f{y});")
            elif op == OP_SUB:
                w(f"    const uint32_t {v} = sub_mod(f{x}, f{y});")
            elif op == OP_MUL:
                w(f"    const uint32_t {v} = mul_mod(f{x}, f{y});")
        else:
            if not used_m[vid]:
                continue
            m = f"m{vid}"
            if op == OP_TRUE:
                zero_vars.add(vid)
                materialised.add(vid)
                w(f"    const uint32_t {m}_0 = 0, {m}_1 = 0, {m}_2 = 0, {m}_3 = 0;")
            elif op == OP_AND_EQZ:
                # tot = x.tot + mix^e(x) * v.  Products are accumulated as 64-bit sums (v_mad_u64_u32 chains) and
                # reduced once per <= 4 terms (4 P^2 < 2 P 2^32, the bound of mont_reduce_wide) instead of once each.
                e = mix_exp[x]
                w(f"    const uint4 p{i} = pw[{e}];")
                if x in lazy and m_uses[x] == 1:
                    base, pend = lazy.pop(x)
                else:
                    materialise(x)
                    base, pend = (None if x in zero_vars else x), []
                pend = pend + [(i, y)]
                lazy[vid] = (base, pend)
                if len(pend) == 4 or m_uses[vid] != 1 or vid == c.ret:
                    materialise(vid)
            elif op == OP_AND_COND:
                e = mix_exp[x]
                materialise(x)
                materialise(z)
                materialised.add(vid)
                w(f"    const uint4 p{i} = pw[{e}];")
                w(f"    const Fp4 t{i} = (Fp4(Fp::raw(m{z}_0), Fp::raw(m{z}_1), Fp::raw(m{z}_2), Fp::raw(m{z}_3)) * Fp::raw(f{y})) *"
                  f" Fp4(Fp::raw(p{i}.x), Fp::raw(p{i}.y), Fp::raw(p{i}.z), Fp::raw(p{i}.w));")
                for kk in range(4):
                    w(f"    const uint32_t {m}_{kk} = add_mod(m{x}_{kk}, t{i}.c[{kk}].v);")
    materialise(c.ret)
    r = f"m{c.ret}"
    w("    const uint32_t zi = a.zinv[idx & 3];")
    for kk in range(4):
        off = "" if kk == 0 else f"{kk} * dom + "
        w(f"    a.check[{off}idx] = mul_mod({r}_{kk}, zi);")
    w("}")
    if not standalone:
        w(f"static void launch_{name}(const EvalCheckArgs& a, hipStream_t s) {{")
        w(f"    k_eval_check_{name}<<<(a.dom + 255u) / 256u, 256, 0, s>>>(a);")
        w("}")
    return "\n".join(L), desc_hash64(desc), n_pows


SHIPPED: Dict[str, np.ndarray] = {}


def shipped() -> Dict[str, np.ndarray]:
    if not SHIPPED:
        SHIPPED["syn_a"] = syn_air.syn_a()
        SHIPPED["syn_small"] = syn_air.syn_small()
        SHIPPED["syn_tiny"] = syn_air.syn_tiny()
    return SHIPPED


def generate_source() -> str:
    parts = ["// GENERATED by zeth_amd/circuits/codegen.py — do not edit.  Straight-line eval_check kernels (gfx950) for the",
             "// shipped circuit descriptions; selected at run time by desc hash (circuit.hip).",
             '#include "circuit.h"', "", "using namespace zkh;", "", "namespace {", ""]
    table = []
    for name, desc in shipped().items():
        src, h, n_pows = emit_kernel(name, desc)
        parts.append(src)
        parts.append("")
        table.append(f'    {{0x{h:016x}ull, "{name}", launch_{name}, {n_pows}u}},')
    parts += ["const CompiledEvalCheck k_table[] = {", *table, "};", "", "}  // namespace", "",
              "namespace zkh {", "const CompiledEvalCheck* find_compiled_eval_check(uint64_t h) {",
              "    for (const auto& e : k_table) if (e.desc_hash == h) return &e;", "    return nullptr;", "}",
              "}  // namespace zkh", ""]
    return "\n".join(parts)


def write_generated(path: str) -> None:
    src = generate_source()
    try:
        with open(path) as fh:
            if fh.read() == src:
                return
    except FileNotFoundError:
        pass
    with open(path, "w") as fh:
        fh.write(src)
