"""Exact integer bound verifier + CPU executor for GENERATED eval_check kernels — the library side of tools/check_bounds.py.

It reads only the EMITTED SOURCE TEXT of a kernel (what hipcc is about to compile) and the contracts of the primitives that text
calls (csrc/fp.h, csrc/circuit.h, restated here with the preconditions their comments give); it shares no code with the generator
(circuits/codegen.py) whose lazy-arithmetic accounting it checks, and it never imports it.  `check_source` runs the straight-line
code over exact Python integer INTERVALS and reports every place where a 32/64-bit word could wrap or a reduction could see an operand
outside its domain; `execute_source` runs the same text on single values (one domain point, Montgomery words, the gathered power
table rebuilt the way the library builds it) and returns the check words.  Used three ways:

  * zeth_amd/build.py and circuits/jit.py call `check_source` on every generated translation unit BEFORE it is compiled: the library
    is never built with — and a circuit that arrives as data never gets — a kernel whose sums can wrap;
  * tools/check_bounds.py is the command line (shipped circuits, random circuits, files on disk);
  * tests/test_bounds_checker.py: the round-5 overflow bug must FAIL, every shipped / random / SYN-HUGE kernel must pass and, executed
    on the CPU, equal the oracle's literal step interpreter.
The rules are listed in tools/check_bounds.py's docstring."""
from __future__ import annotations

import re
from typing import Dict, List, Tuple

P = 2013265921
R1 = (1 << 32) % P
NBETA_M = P - (11 << 32) % P
W32 = 1 << 32


class Violation(Exception):
    pass


class S:
    """an integer scalar: true-value interval [lo, hi], the C type's width and signedness"""
    __slots__ = ("lo", "hi", "w", "sg")

    def __init__(self, lo: int, hi: int, w: int = 32, sg: bool = False):
        self.lo, self.hi, self.w, self.sg = lo, hi, w, sg

    def __repr__(self):
        return f"S[{self.lo}, {self.hi}]{'i' if self.sg else 'u'}{self.w}"


class FP:
    """Fp::raw(x)"""
    __slots__ = ("s",)

    def __init__(self, s: S):
        self.s = s


class E:
    """an Fp4: four component words"""
    __slots__ = ("c",)

    def __init__(self, c: List[S]):
        self.c = c


class U4:
    """a uint4 (one mix power)"""
    __slots__ = ("c",)

    def __init__(self, c: List[S]):
        self.c = c


CANON = lambda: S(0, P - 1, 32)          # noqa: E731
NEG_PINV = (-pow(P, -1, W32)) % W32
M32 = W32 - 1


def exact(*vals) -> bool:
    """all operands are single values (concrete execution: tools/check_bounds.py execute_source), not intervals"""
    return all(v.lo == v.hi for v in vals)


def pt(v: int, w: int = 32) -> S:
    return S(v, v, w)


def _mont(t: int, correct: bool) -> int:
    m = ((t & M32) * NEG_PINV) & M32
    r = (t + m * P) >> 32
    return r - P if correct and r >= P else r


def fits(s: S, what: str) -> S:
    """the value is stored / cast / multiplied / passed on: it must be what the machine word holds"""
    if s.sg:
        if s.lo < -(1 << (s.w - 1)) or s.hi >= (1 << (s.w - 1)):
            raise Violation(f"{what}: [{s.lo}, {s.hi}] does not fit i{s.w} (|.| up to 2^{max(abs(s.lo), abs(s.hi)).bit_length() - 1}.., limit 2^{s.w - 1})")
        return s
    if s.lo < 0:
        raise Violation(f"{what}: may be negative (lowest true value {s.lo}): the u{s.w} word wraps")
    if s.hi >= (1 << s.w):
        raise Violation(f"{what}: worst case {s.hi} = 2^{s.hi.bit_length() - 1}.. does not fit u{s.w} (limit 2^{s.w})")
    return s


# ---- the primitives' contracts (csrc/fp.h, csrc/circuit.h) ----
def reduce_once(s: S, what: str) -> S:
    if s.hi >= 2 * P:
        raise Violation(f"{what}: reduce_once needs s < 2P, worst case {s.hi}")
    if exact(s):
        return pt(s.hi - P if s.hi >= P else s.hi)
    return CANON()


def mont_reduce(t: S, what="mont_reduce") -> S:
    fits(t, what)
    if t.hi >= P << 32:
        raise Violation(f"{what}: needs t < P 2^32 = {P << 32}, worst case {t.hi} ({t.hi / (P << 32):.3f} x)")
    if exact(t):
        return pt(_mont(t.hi, True))
    return CANON()


def mont_reduce_lazy(t: S, what="mont_reduce_lazy") -> S:
    fits(t, what)
    if t.hi >= P << 32:
        raise Violation(f"{what}: needs t < P 2^32 = {P << 32}, worst case {t.hi} ({t.hi / (P << 32):.3f} x)")
    if exact(t):
        return pt(_mont(t.hi, False))
    return S(0, (t.hi + (W32 - 1) * P) >> 32, 32)


def mont_reduce_wide(t: S, what="mont_reduce_wide", lazy=False) -> S:
    fits(t, what)
    if t.hi >= (2 * P) << 32:
        raise Violation(f"{what}: needs t < 2 P 2^32 = {(2 * P) << 32}, worst case {t.hi} ({t.hi / ((2 * P) << 32):.3f} x)")
    if exact(t):
        hi = t.hi >> 32
        hi = hi - P if hi >= P else hi
        return pt(_mont((hi << 32) | (t.hi & M32), not lazy))
    if not lazy:
        return CANON()
    return S(0, ((min(t.hi, (P << 32) - 1)) + (W32 - 1) * P) >> 32, 32)


def fold_acc(s: S, what="fold_acc") -> S:
    fits(s, what)
    if exact(s):
        return pt((s.hi >> 32) * R1 + (s.hi & M32), 64)
    return S(0, s.hi if s.hi < W32 else (s.hi >> 32) * R1 + W32 - 1, 64)


def as_i32(x: S, what: str) -> S:
    """an operand of a signed multiply-add: the 32-bit word read as int32 must BE the value (no wrap)"""
    if x.lo < -(1 << 31) or x.hi >= (1 << 31):
        raise Violation(f"{what}: operand [{x.lo}, {x.hi}] does not fit int32")
    return x


def mad_i64(a: S, b: S, acc: S, what="mad_i64") -> S:
    """v_mad_i64_i32: acc + a * b, exact in signed 64 bits"""
    as_i32(a, what); as_i32(b, what)
    prods = [a.lo * b.lo, a.lo * b.hi, a.hi * b.lo, a.hi * b.hi]
    return fits(S(acc.lo + min(prods), acc.hi + max(prods), 64, True), what)


def fold_acc_s(s: S, what="fold_acc_s") -> S:
    """hi R + lo with hi = s >> 32 signed, lo = the low word unsigned"""
    fits(S(s.lo, s.hi, 64, True), what)
    if exact(s):
        return S((s.hi >> 32) * R1 + (s.hi & M32), (s.hi >> 32) * R1 + (s.hi & M32), 64, True)
    return S((s.lo >> 32) * R1, (s.hi >> 32) * R1 + W32 - 1, 64, True)


def smont_canon(t: S, what="smont_canon") -> S:
    """signed Montgomery step (|t| < P 2^31 -> (-P, P)) and one conditional + P -> [0, P)"""
    fits(S(t.lo, t.hi, 64, True), what)
    if t.lo <= -(P << 31) or t.hi >= (P << 31):
        raise Violation(f"{what}: needs |t| < P 2^31 = {P << 31}, worst case [{t.lo}, {t.hi}] ({max(abs(t.lo), abs(t.hi)) / (P << 31):.3f} x)")
    if exact(t):
        m = ((t.hi & M32) * NEG_PINV) & M32
        m = m - W32 if m >= (1 << 31) else m
        r = (t.hi + m * P) >> 32
        return pt(r + P if r < 0 else r)
    return CANON()


def add_mod(a: S, b: S, what="add_mod") -> S:
    fits(a, what); fits(b, what)
    return reduce_once(S(a.lo + b.lo, a.hi + b.hi, 32), what)


def sub_mod(a: S, b: S, what="sub_mod") -> S:
    fits(a, what); fits(b, what)
    if a.hi > P - 1 or b.hi > P - 1:
        raise Violation(f"{what}: needs canonical operands (< P), worst cases {a.hi}, {b.hi}")
    if exact(a, b):
        return pt(a.hi - b.hi if a.hi >= b.hi else a.hi - b.hi + P)
    return CANON()


def mul64(a: S, b: S, what: str) -> S:
    fits(a, what); fits(b, what)
    return S(a.lo * b.lo, a.hi * b.hi, 64)


def mul_mod(a: S, b: S, what="mul_mod") -> S:
    return mont_reduce(mul64(a, b, what), what)


def mul_lazy(a: S, b: S, what="mul_lazy") -> S:
    return mont_reduce_lazy(mul64(a, b, what), what)


def sum64(terms: List[S], what: str) -> S:
    return fits(S(sum(t.lo for t in terms), sum(t.hi for t in terms), 64), what)


def fp4_mul(a: E, b: E, what="Fp4 * Fp4") -> E:
    A, B = a.c, b.c
    m = lambda i, j: mul64(A[i], B[j], what)                     # noqa: E731
    nb = S(NBETA_M, NBETA_M, 32)
    h0 = mont_reduce_wide(sum64([m(1, 3), m(2, 2), m(3, 1)], what), what + " (x^4)")
    h1 = mont_reduce_wide(sum64([m(2, 3), m(3, 2)], what), what + " (x^5)")
    h2 = mont_reduce(m(3, 3), what + " (x^6)")
    r0 = mont_reduce_wide(sum64([m(0, 0), mul64(nb, h0, what)], what), what + " (c0)")
    r1 = mont_reduce_wide(sum64([m(0, 1), m(1, 0), mul64(nb, h1, what)], what), what + " (c1)")
    r2 = mont_reduce_wide(sum64([m(0, 2), m(1, 1), m(2, 0), mul64(nb, h2, what)], what), what + " (c2)")
    r3 = mont_reduce_wide(sum64([m(0, 3), m(1, 2), m(2, 1), m(3, 0)], what), what + " (c3)")
    return E([r0, r1, r2, r3])


def ext_accumulate(s: List[S], p: U4, x: E, what="ext_accumulate") -> List[S]:
    pc, xc = p.c, x.c
    m = lambda i, j: mul64(pc[i], xc[j], what)                   # noqa: E731
    nb = S(NBETA_M, NBETA_M, 32)
    h0 = mont_reduce_wide(sum64([m(1, 3), m(2, 2), m(3, 1)], what), what + " (x^4)")
    h1 = mont_reduce_wide(sum64([m(2, 3), m(3, 2)], what), what + " (x^5)")
    h2 = mont_reduce(m(3, 3), what + " (x^6)")
    return [sum64([s[0], m(0, 0), mul64(nb, h0, what)], what + " s0"),
            sum64([s[1], m(0, 1), m(1, 0), mul64(nb, h1, what)], what + " s1"),
            sum64([s[2], m(0, 2), m(1, 1), m(2, 0), mul64(nb, h2, what)], what + " s2"),
            sum64([s[3], m(0, 3), m(1, 2), m(2, 1), m(3, 0)], what + " s3")]


# ---- a small expression parser for the emitted subset of C++ ----
TOKEN = re.compile(r"\s*(?:(\d+)(ull|u)?|([A-Za-z_][A-Za-z_0-9]*(?:::[A-Za-z_][A-Za-z_0-9]*)?)|(\+=|[-+*()\[\],.=&]))")


def tokenize(text: str) -> List[Tuple[str, str]]:
    out, pos = [], 0
    text = text.strip()
    while pos < len(text):
        m = TOKEN.match(text, pos)
        if not m:
            raise Violation(f"cannot tokenise `{text[pos:pos + 40]}`")
        pos = m.end()
        if m.group(1) is not None:
            out.append(("num", m.group(1) + (m.group(2) or "")))
        elif m.group(3) is not None:
            out.append(("id", m.group(3)))
        else:
            out.append(("op", m.group(4)))
    return out


class Parser:
    def __init__(self, toks, env: Dict[str, object], ctx: str):
        self.t, self.i, self.env, self.ctx = toks, 0, env, ctx
        self.conc = env.get("__concrete__")             # concrete execution: the inputs of ONE domain point (execute_source)

    def peek(self, k=0):
        return self.t[self.i + k] if self.i + k < len(self.t) else ("end", "")

    def eat(self, val=None):
        tok = self.peek()
        if val is not None and tok[1] != val:
            raise Violation(f"parse: expected `{val}`, found `{tok[1]}` in `{self.ctx}`")
        self.i += 1
        return tok

    def done(self) -> bool:
        return self.i >= len(self.t)

    # additive
    def expr(self):
        left = self.term()
        while self.peek()[1] in ("+", "-") and self.peek()[0] == "op":
            op = self.eat()[1]
            right = self.term()
            left = self.additive(op, left, right)
        return left

    def additive(self, op, a, b):
        if isinstance(a, E) or isinstance(b, E):
            if not (isinstance(a, E) and isinstance(b, E)):
                raise Violation(f"Fp4 {op} non-Fp4 in `{self.ctx}`")
            f = add_mod if op == "+" else sub_mod
            return E([f(x, y, f"Fp4 {op}") for x, y in zip(a.c, b.c)])
        if not (isinstance(a, S) and isinstance(b, S)):
            raise Violation(f"unsupported operands of {op} in `{self.ctx}`")
        w = max(a.w, b.w)
        return S(a.lo + b.lo, a.hi + b.hi, w) if op == "+" else S(a.lo - b.hi, a.hi - b.lo, w)

    def term(self):
        left = self.unary()
        while self.peek() == ("op", "*"):
            self.eat()
            right = self.unary()
            left = self.mul(left, right)
        return left

    def mul(self, a, b):
        if isinstance(a, E) and isinstance(b, E):
            return fp4_mul(a, b)
        if isinstance(a, E) and isinstance(b, FP):
            return E([mul_mod(x, b.s, "Fp4 * Fp") for x in a.c])
        if isinstance(a, S) and isinstance(b, S):
            w = max(a.w, b.w)
            fits(a, "multiplicand"); fits(b, "multiplicand")
            if a.sg or b.sg:
                if not (a.sg and b.sg and a.w == b.w):
                    raise Violation(f"a product of a signed and an unsigned (or differently wide) operand in `{self.ctx}`")
                prods = [a.lo * b.lo, a.lo * b.hi, a.hi * b.lo, a.hi * b.hi]
                return S(min(prods), max(prods), w, True)
            return S(a.lo * b.lo, a.hi * b.hi, w)
        raise Violation(f"unsupported operands of * in `{self.ctx}`")

    def unary(self):
        # a cast: `(uint64_t)` / `(size_t)` followed by a unary expression
        if self.peek() == ("op", "(") and self.peek(1) == ("id", "int32_t") and self.peek(2) == ("op", ")"):
            # (int32_t)x and (int32_t)(x - Pu): the 32-bit word REINTERPRETED as signed — the u32 expression may wrap (that is the point:
            # x - P for x in [0, 2P)), the true value must fit int32
            self.eat(); self.eat(); self.eat()
            if self.peek() == ("op", "("):
                self.eat()
                v = self.expr()
                self.eat(")")
            else:
                v = self.unary()
            if not isinstance(v, S):
                raise Violation(f"cast of a non-scalar in `{self.ctx}`")
            if v.w != 32:
                raise Violation(f"(int32_t) of a {v.w}-bit value in `{self.ctx}`")
            return as_i32(S(v.lo, v.hi, 32, True), "(int32_t)")
        if self.peek() == ("op", "(") and self.peek(1) == ("id", "int64_t") and self.peek(2) == ("op", ")"):
            self.eat(); self.eat(); self.eat()
            v = self.unary()
            if not isinstance(v, S) or not v.sg:
                raise Violation(f"(int64_t) of something that is not a signed scalar in `{self.ctx}`")
            fits(v, "(int64_t) operand")
            return S(v.lo, v.hi, 64, True)
        if self.peek() == ("op", "(") and self.peek(1) == ("id", "uint64_t") and self.peek(2) == ("op", ")"):
            self.eat(); self.eat(); self.eat()
            v = self.unary()
            if not isinstance(v, S):
                raise Violation(f"cast of a non-scalar in `{self.ctx}`")
            fits(v, "cast operand")
            return S(v.lo, v.hi, 64)
        return self.postfix()

    def postfix(self):
        v = self.primary()
        while True:
            tok = self.peek()
            if tok == ("op", "."):
                self.eat()
                name = self.eat()[1]
                if isinstance(v, U4) and name in "xyzw" and len(name) == 1:
                    v = v.c["xyzw".index(name)]
                elif isinstance(v, E) and name == "c":
                    self.eat("[")
                    k = int(self.eat()[1].rstrip("ul"))
                    self.eat("]")
                    v = FP(v.c[k])
                elif isinstance(v, FP) and name == "v":
                    v = v.s
                else:
                    raise Violation(f"unsupported member .{name} in `{self.ctx}`")
            elif tok == ("op", "[") and isinstance(v, str) and v == "pwp":
                self.eat()
                k = self.expr()
                self.eat("]")
                centred = bool(self.env.get("__centred__", {}).get(k.hi)) if exact(k) else False
                if self.conc is not None:
                    v = U4([pt(int(x)) for x in self.conc["pwp"][k.hi]])          # (centred slots: negative values)
                elif centred:                                                     # |p| <= (P - 1) / 2, read as int32 by its consumer
                    v = U4([S(-((P - 1) // 2), (P - 1) // 2, 32) for _ in range(4)])
                else:
                    v = U4([CANON() for _ in range(4)])
            else:
                return v

    def args(self) -> list:
        self.eat("(")
        out = []
        if self.peek() != ("op", ")"):
            out.append(self.expr())
            while self.peek() == ("op", ","):
                self.eat()
                out.append(self.expr())
        self.eat(")")
        return out

    def primary(self):
        kind, val = self.peek()
        if kind == "num":
            self.eat()
            n = int(val.rstrip("ul"))
            return S(n, n, 64 if val.endswith("ull") else 32)
        if kind == "op" and val == "(":
            self.eat()
            v = self.expr()
            self.eat(")")
            if isinstance(v, S):
                fits(v, "parenthesised value")          # every parenthesised scalar in the emitted text is an operand of a product
            return v
        if kind != "id":
            raise Violation(f"parse: unexpected `{val}` in `{self.ctx}`")
        self.eat()
        if val == "pwp":
            return "pwp"
        if val == "a":                                   # a.globals[x][y], a.zinv[idx & 3], a.check[...]: canonical input words
            text = "".join(v for _, v in self.t[self.i:])
            depth = 0
            while not self.done():
                k, v = self.peek()
                if v in ("[", "("):
                    depth += 1
                elif v in ("]", ")"):
                    if depth == 0:
                        break
                    depth -= 1
                elif depth == 0 and v in (",", "+", "-", "*", "=") and k == "op":
                    break
                self.eat()
            if self.conc is not None:
                m = re.match(r"^\.globals\[(\d+)\]\[(\d+)\]", text)
                if m:
                    return pt(int(self.conc["globals"][int(m.group(1))][int(m.group(2))]))
                if text.startswith(".zinv["):
                    return pt(int(self.conc["zinv"]))
                raise Violation(f"concrete execution: unsupported input `a{text[:30]}`")
            return CANON()
        if val == "Fp::raw":
            (x,) = self.args()
            if not isinstance(x, S):
                raise Violation(f"Fp::raw of a non-scalar in `{self.ctx}`")
            return FP(fits(x, "Fp::raw"))
        if val == "Fp4":
            a = self.args()
            if len(a) == 1 and isinstance(a[0], FP):
                return E([a[0].s, S(0, 0), S(0, 0), S(0, 0)])
            if len(a) == 4 and all(isinstance(x, FP) for x in a):
                return E([x.s for x in a])
            raise Violation(f"unsupported Fp4 constructor in `{self.ctx}`")
        if self.peek() == ("op", "("):
            if val == "tap_load":                        # address arithmetic inside: a canonical trace word comes back
                depth, inner = 0, []
                while True:
                    v = self.eat()[1]
                    inner.append(v)
                    depth += v == "("
                    depth -= v == ")"
                    if depth == 0:
                        break
                if self.conc is not None:                 # tap_load(g<group>, (size_t)<offset> * dw<k>, o<back>_<k>)
                    m = re.match(r"^\(g(\d)\,\(size_t\)(\d+)\*dw\d+\,o(\d+)_\d+\)$", "".join(inner))
                    if not m:
                        raise Violation(f"concrete execution: tap_load form not understood in `{self.ctx}`")
                    return pt(int(self.conc["tap"](int(m.group(1)), int(m.group(2)), int(m.group(3)))))
                return CANON()
            a = self.args()
            if val in ("mad_i64", "mad_i64_k"):
                if len(a) != 3 or not all(isinstance(x, S) for x in a):
                    raise Violation(f"{val} wants three scalars in `{self.ctx}`")
                return mad_i64(a[0], a[1], a[2], val)
            if val == "fold_acc_s":
                return fold_acc_s(a[0])
            if val == "smont_canon":
                return smont_canon(a[0])
            fn = {"mont_reduce": mont_reduce, "mont_reduce_lazy": mont_reduce_lazy, "mont_reduce_wide": mont_reduce_wide,
                  "mont_reduce_wide_lazy": lambda t, what="mont_reduce_wide_lazy": mont_reduce_wide(t, what, lazy=True),
                  "fold_acc": fold_acc, "add_mod": add_mod, "sub_mod": sub_mod, "mul_mod": mul_mod, "mul_lazy": mul_lazy}.get(val)
            if fn is not None:
                if not all(isinstance(x, S) for x in a):
                    raise Violation(f"{val} of a non-scalar in `{self.ctx}`")
                return fn(*a)
            if val == "ext_mul_base_lazy":
                return E([mul_lazy(x, a[1], val) for x in a[0].c])
            if val in ("ext_add_base", "ext_sub_base"):
                f = add_mod if val == "ext_add_base" else sub_mod
                return E([f(a[0].c[0], a[1], val)] + [fits(x, val) for x in a[0].c[1:]])
            raise Violation(f"unknown function {val} in `{self.ctx}`")
        if val not in self.env:
            raise Violation(f"unknown name {val} in `{self.ctx}`")
        return self.env[val]


SKIP_PREFIXES = ("const uint32_t idx", "if (idx >= a.dom)", "const uint32_t mask", "const size_t dom", "const uint4* __restrict__",
                 "const uint32_t* __restrict__", "if (a.accumulate)", "} else {")


def split_statements(line: str) -> List[str]:
    """statements of one emitted line (`;`-separated, braces of `{ ... }` one-liners dropped)"""
    s = line.strip()
    if s.startswith("{"):
        s = s[1:]
    if s.endswith("}"):
        s = s[:-1]
    return [x.strip() for x in s.split(";") if x.strip()]


class KernelCheck:
    def __init__(self, name: str):
        self.name = name
        self.env: Dict[str, object] = {}
        self.violations: List[str] = []
        self.n_statements = self.n_reductions = 0
        self.max_acc = 0

    def store(self, name: str, v, what: str):
        if isinstance(v, S):
            fits(v, f"{what} {name}")
            if v.w == 64:
                self.max_acc = max(self.max_acc, 2 * max(abs(v.lo), abs(v.hi)) if v.sg else v.hi)      # (a signed sum uses one bit for the sign)
        elif isinstance(v, E):
            for x in v.c:
                fits(x, f"{what} {name}")
        self.env[name] = v

    def statement(self, st: str):
        self.n_statements += 1
        self.n_reductions += st.count("mont_reduce") + st.count("mul_mod") + st.count("mul_lazy")
        env = self.env
        if st.startswith("ext_accumulate("):
            inner = st[len("ext_accumulate("):-1]
            parts = [x.strip() for x in inner.split(",", 4)]
            names, rest = parts[:4], parts[4]
            p = Parser(tokenize("pair(" + rest + ")"), env, st)
            p.eat(); a = p.args()
            new = ext_accumulate([env[n] for n in names], a[0], a[1] if isinstance(a[1], E) else _promote(a[1]))
            for n, v in zip(names, new):
                self.store(n, v, "accumulator")
            return
        m = re.match(r"^(?:const\s+)?(uint32_t|uint64_t|int32_t|int64_t|Fp4|uint4)\s+(.*)$", st)
        if m:
            typ, rest = m.group(1), m.group(2)
            if typ == "Fp4" and re.match(r"^[A-Za-z_0-9]+\(", rest):           # const Fp4 x7(Fp::raw(..), ...)
                name = rest[:rest.index("(")]
                p = Parser(tokenize("Fp4" + rest[len(name):]), env, st)
                self.store(name, p.expr(), "value")
                return
            # `uint32_t t0_0 = 0, t0_1 = 0, ...` (several declarators) or one `NAME = EXPR`
            for decl in _split_top(rest, ","):
                decl = re.sub(r"^(uint32_t|uint64_t|int32_t|int64_t)\s+", "", decl.strip())
                name, expr = [x.strip() for x in decl.split("=", 1)]
                p = Parser(tokenize(expr), env, st)
                v = p.expr()
                if not p.done():
                    raise Violation(f"parse: trailing tokens in `{st}`")
                if isinstance(v, S):
                    v = S(v.lo, v.hi, 64 if typ in ("uint64_t", "int64_t") else 32, typ in ("int32_t", "int64_t"))
                self.store(name, v, "value")
            return
        m = re.match(r"^([A-Za-z_0-9]+)\s*\+=\s*(.*)$", st)
        if m:
            name, expr = m.group(1), m.group(2)
            p = Parser(tokenize(expr), env, st)
            v = p.expr()
            cur = env[name]
            self.store(name, S(cur.lo + v.lo, cur.hi + v.hi, cur.w, cur.sg), "accumulator")
            return
        m = re.match(r"^a\.check\[[^\]]*\]\s*=\s*(.*)$", st)
        if m:
            v = Parser(tokenize(m.group(1)), env, st).expr()
            if v.hi > P - 1:
                raise Violation(f"check word not canonical in `{st}`")
            return
        if "=" in st:                                     # NAME = NAME2 = ... = EXPR
            parts = [x.strip() for x in st.split("=")]
            p = Parser(tokenize(parts[-1]), env, st)
            v = p.expr()
            if not p.done():
                raise Violation(f"parse: trailing tokens in `{st}`")
            for name in parts[:-1]:
                if name not in env:
                    raise Violation(f"assignment to an undeclared name {name} in `{st}`")
                cur = env[name]
                self.store(name, S(v.lo, v.hi, cur.w, cur.sg) if isinstance(v, S) else v, "value")
            return
        raise Violation(f"statement form not understood: `{st}`")

    def run(self, lines: List[Tuple[int, str]]):
        for no, raw in lines:
            line = raw.strip()
            if not line or line in ("{", "}"):
                continue
            if line.startswith("//"):
                continue
            if line.startswith(SKIP_PREFIXES) or "asm(" in line:
                continue
            if re.match(r"^const uint32_t b\d+ = ", line):        # lane byte offsets
                continue
            try:
                for st in split_statements(line):
                    self.statement(st)
            except Violation as e:
                self.violations.append(f"{self.name}: line {no}: {e}\n      {line[:200]}")
                if len(self.violations) >= 8:
                    return
            except (KeyError, ValueError, IndexError, AttributeError, TypeError) as e:
                self.violations.append(f"{self.name}: line {no}: checker could not follow the statement ({type(e).__name__}: {e})\n      {line[:200]}")
                return


def _promote(v) -> E:
    if isinstance(v, E):
        return v
    raise Violation("ext_accumulate of a non-Fp4 value")


def _split_top(text: str, sep: str) -> List[str]:
    out, depth, cur = [], 0, []
    for ch in text:
        if ch in "([":
            depth += 1
        elif ch in ")]":
            depth -= 1
        if ch == sep and depth == 0:
            out.append("".join(cur)); cur = []
        else:
            cur.append(ch)
    out.append("".join(cur))
    return out


def check_source(src: str, label: str = "") -> Tuple[List[str], dict]:
    """Every `__global__` kernel of a generated translation unit -> (violations, statistics)."""
    lines = src.split("\n")
    violations: List[str] = []
    stats = {"kernels": 0, "statements": 0, "reductions": 0, "claims": 0, "max_acc_bits": 0.0}
    i = 0
    while i < len(lines):
        m = re.search(r"__global__ .* void (k_eval_check_\w+)\(EvalCheckArgs a\) \{", lines[i])
        if not m:
            i += 1
            continue
        name = m.group(1)
        j = i + 1
        body: List[Tuple[int, str]] = []
        while j < len(lines) and lines[j] != "}":
            body.append((j + 1, lines[j]))
            j += 1
        kc = KernelCheck((label + ":" if label else "") + name)
        # which slots of its power table this kernel reads CENTRED (bit 31 of the exported exponent word): |p| <= (P - 1) / 2 there
        mm = re.search(r"const uint32_t (?:exps_%s|%s_exps)\[\] = \{([^}]*)\}" % (name, name), src)
        if mm:
            words = [int(x) for x in mm.group(1).split(",")]
            kc.env["__centred__"] = {k: True for k, w in enumerate(words[1:]) if w >> 31}
        # the generator's claims are checked where they stand: a `// BOUND x <= N` line follows the statement that completes x
        claims_inline: List[Tuple[int, str, int]] = []
        kc_lines: List[Tuple[int, str]] = []
        for no, raw in body:
            mm = re.match(r"^\s*// BOUND (\S+) <= (\d+)", raw)
            if mm:
                claims_inline.append((len(kc_lines), mm.group(1), int(mm.group(2))))
            else:
                kc_lines.append((no, raw))
        pos = 0
        for at, cname, claimed in claims_inline + [(len(kc_lines), None, 0)]:
            kc.run(kc_lines[pos:at])
            pos = at
            if kc.violations or cname is None:
                if kc.violations:
                    break
                continue
            stats["claims"] += 1
            v = kc.env.get(cname)
            got = max(x.hi for x in v.c) if isinstance(v, E) else (max(abs(v.lo), abs(v.hi)) if isinstance(v, S) else None)      # (a signed sum: its magnitude)
            if got is None:
                kc.violations.append(f"{kc.name}: the bounds trace names {cname}, which the code has not defined at that point")
                break
            if got > claimed:
                kc.violations.append(f"{kc.name}: the generator claims {cname} <= {claimed}, the code allows {got}")
                break
        violations.extend(kc.violations)
        stats["kernels"] += 1
        stats["statements"] += kc.n_statements
        stats["reductions"] += kc.n_reductions
        if kc.max_acc:
            stats["max_acc_bits"] = max(stats["max_acc_bits"], round(kc.max_acc.bit_length() - 1 + (kc.max_acc / (1 << (kc.max_acc.bit_length() - 1)) - 1), 3))
        i = j + 1
    return violations, stats


# ---- concrete execution: the emitted kernels run on the CPU for ONE domain point (the same parser, single values instead of intervals) ----
def fp4_mul_canon(a, b):
    r = [0] * 7
    for i in range(4):
        for j in range(4):
            r[i + j] += a[i] * b[j]
    return tuple((r[k] - 11 * (r[k + 4] if k + 4 < 7 else 0)) % P for k in range(4))


def fp4_pow_canon(a, e: int):
    r = (1, 0, 0, 0)
    while e:
        if e & 1:
            r = fp4_mul_canon(r, a)
        a = fp4_mul_canon(a, a)
        e >>= 1
    return r


def kernel_tables(src: str) -> Tuple[List[int], List[Tuple[int, Tuple[int, int, int, int]]]]:
    """the gathered-table description a kernel exports: exponent per slot, and (slot, Fp4 constant as Montgomery words) records"""
    m = re.search(r"const uint32_t (?:exps_\w+|\w+_exps)\[\] = \{([^}]*)\}", src)
    exps = [int(x) for x in m.group(1).split(",")] if m else None
    m = re.search(r"const uint32_t (?:pwc_\w+|\w+_pwc)\[\] = \{([^}]*)\}", src)
    recs = [int(x) for x in m.group(1).split(",")] if m else [0]
    consts = [(recs[1 + 5 * i], tuple(recs[2 + 5 * i: 6 + 5 * i])) for i in range(recs[0])]
    return (exps[1:] if exps else None), consts


def execute_source(src: str, groups, globals_, poly_mix, po2: int, idx: int) -> List[int]:
    """Run every `__global__` kernel of a generated translation unit for domain point `idx` in exact integer arithmetic, the way the
    device does (Montgomery words, the library's gathered power table incl. slot constants, zinv), and return the four words this
    unit contributes to check[k * dom + idx] (parts of a split circuit: add them mod P).  groups: three W x dom arrays of raw words;
    globals_: (out, mix) raw words; poly_mix: four raw words.  Every precondition the bound checker knows is ALSO checked on the
    concrete values, so a wrapped accumulator raises instead of silently computing garbage."""
    n, dom = 1 << po2, 4 << po2
    RINV = pow(W32, -1, P)
    canon = lambda w: (int(w) * RINV) % P            # noqa: E731
    montw = lambda x: (x * W32) % P                  # noqa: E731
    mixc = tuple(canon(w) for w in poly_mix)
    exps, consts = kernel_tables(src)
    if exps is None:
        raise Violation("concrete execution needs the kernel's exported exponent list (GATHER)")
    cmap = {slot: tuple(canon(w) for w in C) for slot, C in consts}
    cache: Dict[int, tuple] = {}
    pwp = []
    for slot, word in enumerate(exps):
        e, centred = word & 0x7FFFFFFF, word >> 31
        if e not in cache:
            cache[e] = fp4_pow_canon(mixc, e)
        v = cache[e]
        if slot in cmap:
            v = fp4_mul_canon(v, cmap[slot])
        ws = tuple(montw(x) for x in v)
        if centred:                                  # the library's last table step: x or x - P in [-(P-1)/2, (P-1)/2] (k_ext_center_at)
            ws = tuple(x - P if x > (P - 1) // 2 else x for x in ws)
        pwp.append(ws)
    w = pow(137, 1 << (27 - (po2 + 2)), P)
    y = pow(3 * pow(w, idx, P) % P, n, P)
    zinv = montw(pow((y - 1) % P, P - 2, P))
    conc = {"pwp": pwp, "globals": globals_, "zinv": zinv,
            "tap": lambda g, off, back: groups[g][off * dom + ((idx - 4 * back) & (dom - 1))]}
    lines = src.split("\n")
    out = [0, 0, 0, 0]
    i = 0
    while i < len(lines):
        m = re.search(r"__global__ .* void (k_eval_check_\w+)\(EvalCheckArgs a\) \{", lines[i])
        if not m:
            i += 1
            continue
        j = i + 1
        body = []
        while j < len(lines) and lines[j] != "}":
            if not re.match(r"^\s*(if \(a\.accumulate\)|\} else \{|a\.check\[|\}$)", lines[j]) and "a.check[" not in lines[j]:
                body.append((j + 1, lines[j]))
            j += 1
        kc = KernelCheck(m.group(1))
        kc.env["__concrete__"] = conc
        kc.run(body)
        if kc.violations:
            raise Violation(kc.violations[0])
        zi = kc.env["zi"]
        for k in range(4):
            r = mul_mod(kc.env[f"t0_{k}"], zi)
            out[k] = (out[k] + r.hi) % P
        i = j + 1
    return out
