"""Row f3: the Rust side of the boundary as FILES, checked mechanically without a Rust toolchain.

rust/risc0-sys-hip/src/lib.rs is the complete `extern "C"` block (generated from include/zkhal.h); rust/hal_hip.rs is the
hand-written `impl Hal for HipHal` / `Buffer` / `CircuitHal`.  Here both sides are parsed INDEPENDENTLY of the generator and
diffed: names, arity, pointer depth / constness and integer widths of every function; field order and types of the `repr(C)`
structs; every `sys::zkh_*` call in hal_hip.rs against the extern block; every method of the recalled traits has a body."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "zkhal.h")
LIB_RS = os.path.join(ROOT, "rust", "risc0-sys-hip", "src", "lib.rs")
HAL_RS = os.path.join(ROOT, "rust", "hal_hip.rs")

C_BASE = {"size_t": "usize", "int": "c_int", "uint32_t": "u32", "uint64_t": "u64", "double": "f64", "char": "c_char", "void": "c_void", "uint8_t": "u8"}


def _camel(c_name: str) -> str:
    return "".join(p.capitalize() for p in c_name.split("_"))


def _norm_c(decl: str):
    """a C parameter / return type, name stripped -> (rust base name, tuple of pointer levels 'const' / 'mut' outermost first)"""
    d = decl.strip()
    arr = d.endswith("]")
    d = re.sub(r"\[[^\]]*\]$", "", d).strip()
    toks = re.findall(r"[A-Za-z_][A-Za-z0-9_]*|\*", d)
    if toks and toks[-1] != "*" and toks[-1] != "const" and len([t for t in toks if t not in ("const", "*")]) == 2:
        toks = toks[:-1]                              # drop the parameter name
    base = next(t for t in toks if t != "const")
    i = toks.index(base)
    lead_const = "const" in toks[:i]
    levels, pending = [], lead_const
    for t in toks[i + 1:]:
        if t == "const":
            pending = True
        else:
            levels.append("const" if pending else "mut")
            pending = False
    if arr:
        levels.append("const" if pending or (lead_const and not levels) else "mut")
    rb = C_BASE.get(base) or (_camel(base) if base.startswith("zkh_") else None)
    assert rb, f"unknown C base type in '{decl}'"
    return rb, tuple(reversed(levels))


def _norm_rust(ty: str):
    ty = ty.strip()
    levels = []
    while ty.startswith("*"):
        m = re.match(r"\*(const|mut)\s+", ty)
        levels.append(m.group(1))
        ty = ty[m.end():]
    return ty, tuple(levels)


def c_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"^\s*#.*$", " ", text, flags=re.M)
    text = re.sub(r"typedef\s+struct\s*\{.*?\}\s*\w+\s*;", " ", text, flags=re.S)
    text = re.sub(r"typedef[^;]*\(\*\w+\)\s*\([^;]*\)\s*;", " ", text, flags=re.S)
    text = re.sub(r"typedef[^;]*;", " ", text)
    out = {}
    for m in re.finditer(r"([\w\s\*]+?)\b(zkh_\w+)\s*\(([^;{}]*)\)\s*;", text, flags=re.S):
        ret, name, args = " ".join(m.group(1).split()), m.group(2), " ".join(m.group(3).split())
        params = [] if args in ("", "void") else [_norm_c(a) for a in args.split(",")]
        out[name] = (None if ret == "void" else _norm_c(ret), params)
    return out


def rust_functions():
    text = open(LIB_RS).read()
    block = text[text.index('extern "C" {'):]
    block = block[: block.index("\n}\n")]
    out = {}
    for m in re.finditer(r"pub fn (zkh_\w+)\((.*?)\)( -> ([^;]+))?;", block, flags=re.S):
        params = [] if not m.group(2).strip() else [_norm_rust(p.split(":", 1)[1]) for p in re.split(r",\s*(?=\w+:)", m.group(2).strip())]
        out[m.group(1)] = (_norm_rust(m.group(4)) if m.group(4) else None, params)
    return out


def test_extern_block_matches_the_header_name_by_name_and_type_by_type():
    c, r = c_functions(), rust_functions()
    assert len(c) >= 100 and set(c) == set(r), (sorted(set(c) - set(r)), sorted(set(r) - set(c)))
    for name in sorted(c):
        assert c[name][0] == r[name][0], (name, "return", c[name][0], r[name][0])
        assert len(c[name][1]) == len(r[name][1]), (name, "arity", len(c[name][1]), len(r[name][1]))
        for k, (a, b) in enumerate(zip(c[name][1], r[name][1])):
            if a[0] == "ZkhAccumulateFn":
                assert b == ("ZkhAccumulateFn", ()), name
                continue
            assert a == b, (name, k, a, b)
    # ... and the library exports every one of them (the .so is what the crate links)
    exported = subprocess.run(["nm", "-D", "--defined-only", os.path.join(ROOT, "zeth_amd", "libzkhal_mi355x.so")], capture_output=True, text=True).stdout
    have = set(re.findall(r" T (zkh_\w+)", exported))
    assert set(r) <= have, sorted(set(r) - have)


def test_generated_file_is_what_the_generator_emits():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_rust_ffi.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def _c_struct_fields(name: str):
    text = re.sub(r"/\*.*?\*/", " ", open(HEADER).read(), flags=re.S)
    body = re.search(r"typedef\s+struct\s*\{([^}]*)\}\s*" + name + r"\s*;", text, flags=re.S).group(1)
    fields = []
    for decl in (d.strip() for d in body.split(";")):
        if not decl:
            continue
        # `double a, b, c` declares several fields of one type; pointer stars belong to each declarator
        first = re.match(r"^((?:const\s+)?\w+)\s*(.*)$", decl)
        base, rest = first.group(1), first.group(2)
        for d in (x.strip() for x in rest.split(",")):
            stars = d.count("*")
            fname = re.sub(r"[\*\s]|\[.*\]", "", d)
            arr = re.search(r"\[(\d+)\]", d)
            fields.append((fname, base.replace("const ", ""), "const" in base, stars, int(arr.group(1)) if arr else None))
    return fields


def _rust_struct_fields(name: str):
    text = open(LIB_RS).read()
    body = re.search(r"pub struct " + name + r" \{(.*?)\n\}", text, flags=re.S).group(1)
    return [(m.group(1), m.group(2).strip()) for m in re.finditer(r"pub (\w+): ([^,\n]+),", body)]


def test_repr_c_structs_have_the_headers_fields_in_order():
    for cname, rname in (("zkh_segment", "ZkhSegment"), ("zkh_prove_info", "ZkhProveInfo"), ("zkh_prof_rec", "ZkhProfRec")):
        cf, rf = _c_struct_fields(cname), _rust_struct_fields(rname)
        assert len(cf) == len(rf), (cname, [f[0] for f in cf], [f[0] for f in rf])
        for (fname, base, const, stars, arr), (rn, rt) in zip(cf, rf):
            assert rn.rstrip("_") == fname, (cname, fname, rn)
            want = C_BASE[base]
            if arr is not None:
                assert rt == f"[{want}; {arr}]", (cname, fname, rt)
            else:
                ptr = "".join("*const " if const and k == stars - 1 else "*mut " for k in range(stars))
                # outermost-first spelling: the constness of the C declaration applies to the innermost pointee
                got_base, got_levels = _norm_rust(rt)
                assert got_base == want and len(got_levels) == stars, (cname, fname, rt)
                if stars:
                    assert (got_levels[-1] == "const") == const, (cname, fname, rt, ptr)


# the traits as recalled from risc0-zkp 3.0.2 src/hal/mod.rs (un-vendored: /root/reference/Cargo.lock:5393): method -> the
# library entry point its body must reach
HAL_METHODS = {
    "has_unified_memory": None, "get_hash_suite": None,
    "alloc_digest": "zkh_alloc", "alloc_elem": "zkh_alloc", "alloc_elem_init": "zkh_copy_from", "alloc_extelem": "zkh_alloc",
    "alloc_extelem_zeroed": "zkh_alloc", "alloc_u32": "zkh_alloc",
    "copy_from_digest": "zkh_copy_from", "copy_from_elem": "zkh_copy_from", "copy_from_extelem": "zkh_copy_from", "copy_from_u32": "zkh_copy_from",
    "batch_expand_into_evaluate_ntt": "zkh_batch_expand_into_evaluate_ntt", "batch_interpolate_ntt": "zkh_batch_interpolate_ntt",
    "batch_bit_reverse": "zkh_batch_bit_reverse", "batch_evaluate_any": "zkh_batch_evaluate_any", "zk_shift": "zkh_zk_shift",
    "mix_poly_coeffs": "zkh_mix_poly_coeffs", "eltwise_add_elem": "zkh_eltwise_add_elem", "eltwise_sum_extelem": "zkh_eltwise_sum_extelem",
    "eltwise_copy_elem": "zkh_eltwise_copy_elem", "eltwise_zeroize_elem": "zkh_eltwise_zeroize_elem", "fri_fold": "zkh_fri_fold",
    "hash_rows": "zkh_hash_rows", "hash_fold": "zkh_hash_fold", "gather_sample": "zkh_gather_sample", "scatter": "zkh_scatter",
    "prefix_products": "zkh_prefix_products", "combos_prepare": "zkh_combos_prepare_regs", "combos_divide": "zkh_combos_divide",
}
BUFFER_METHODS = {"name": None, "size": "zkh_size", "slice": "zkh_slice", "get_at": "read_words", "view": "read_words", "view_mut": "zkh_write", "to_vec": "read_words"}


def _impl_block(text: str, header_regex: str) -> str:
    m = re.search(header_regex, text)
    assert m, header_regex
    depth, i = 0, text.index("{", m.end() - 1)
    for j in range(i, len(text)):
        depth += text[j] == "{"
        depth -= text[j] == "}"
        if depth == 0:
            return text[i:j + 1]
    raise AssertionError("unbalanced braces")


def _method_bodies(block: str):
    out = {}
    for m in re.finditer(r"\n    fn (\w+)\s*(?:<[^>]*>)?\s*\(", block):
        i = block.index("{", m.end())
        depth = 0
        for j in range(i, len(block)):
            depth += block[j] == "{"
            depth -= block[j] == "}"
            if depth == 0:
                out[m.group(1)] = block[i:j + 1]
                break
    return out


def test_every_trait_method_is_spelled_out_and_reaches_its_entry_point():
    text = open(HAL_RS).read()
    hal = _method_bodies(_impl_block(text, r"impl Hal for HipHal\s*\{"))
    assert set(HAL_METHODS) <= set(hal), sorted(set(HAL_METHODS) - set(hal))
    for name, entry in HAL_METHODS.items():
        body = hal[name]
        assert "todo!" not in body and "unimplemented!" not in body and "/* " not in body and len(body.strip("{} \n")) > 0, name
        if entry:
            helper = {"zkh_alloc": "self.alloc(", "zkh_copy_from": "self.copy_from("}.get(entry)
            assert f"sys::{entry}(" in body or (helper and helper in body), (name, entry)
    buf = _method_bodies(_impl_block(text, r"impl<T: Pod \+ Clone> Buffer<T> for HipBuffer<T>\s*\{"))
    assert set(BUFFER_METHODS) <= set(buf), sorted(set(BUFFER_METHODS) - set(buf))
    for name, entry in BUFFER_METHODS.items():
        assert entry is None or entry in buf[name], (name, entry)
    circ = _method_bodies(_impl_block(text, r"impl CircuitHal<HipHal> for HipCircuitHal\s*\{"))
    assert "sys::zkh_eval_check(" in circ["eval_check"]
    for helper, entry in (("fn alloc<", "sys::zkh_alloc("), ("fn copy_from<", "sys::zkh_copy_from("), ("fn read_words", "sys::zkh_read(")):
        i = text.index(helper)
        assert entry in text[i:i + 900], helper


def test_every_call_into_the_library_exists_with_that_arity():
    text = open(HAL_RS).read()
    r = rust_functions()
    calls = 0
    for m in re.finditer(r"sys::(zkh_\w+)\(", text):
        name = m.group(1)
        assert name in r, f"hal_hip.rs calls {name}, which include/zkhal.h does not declare"
        depth, args, cur = 0, [], ""
        for ch in text[m.end():]:
            if ch in "([{":
                depth += 1
            elif ch in ")]}":
                if depth == 0:
                    break
                depth -= 1
            if ch == "," and depth == 0:
                args.append(cur)
                cur = ""
            else:
                cur += ch
        if cur.strip():
            args.append(cur)
        assert len(args) == len(r[name][1]), (name, len(args), len(r[name][1]))
        calls += 1
    assert calls >= 35
