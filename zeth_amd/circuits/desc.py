"""Circuit description blob: TapSet + PolyExtStep list as flat u32 data.

Upstream the circuit is a Zirgen-generated Rust/C++ artefact (risc0-circuit-rv32im 4.0.2
``src/zirgen/{taps.rs,poly_ext.rs}``, un-vendored: /root/reference/Cargo.lock:5320) consumed through
``risc0_zkp::adapter::{TapsProvider, PolyExtStepDef}`` (risc0-zkp 3.0.2 ``src/adapter.rs``, ``src/taps.rs``).
Here the same information is plain data so any circuit (SYN-AIR today; rv32im / recursion / keccak
when their generated tables are supplied) drops in without code changes.

Blob layout (u32 words)::

    [0] magic 'ZKC1' = 0x5a4b4331      [1] version = 1        [2] n_groups = 3
    [3..6)  group sizes: accum, code, data   (REGISTER_GROUP_ACCUM/CODE/DATA = 0/1/2, taps.rs)
    [6] n_global_groups = 2   [7] out size   [8] mix size
    [9] n_taps  [10] n_combos  [11] n_steps  [12] ret (mix-var index of the result)
    [13] kind (1 = SYN-AIR)   [14..16) reserved
    taps:   n_taps x (group, offset, back), sorted by (group, offset, back)
    combos: n_combos x (count, back_0 .. back_{count-1})
    steps:  n_steps x (op, a, b, c, d)

Step ops follow ``PolyExtStep``: value steps append to ``fp_vars``, mix steps append to ``mix_vars``;
operands index those two lists.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Sequence, Tuple

import numpy as np

MAGIC = 0x5A4B4331
HEADER_WORDS = 16
GROUP_ACCUM, GROUP_CODE, GROUP_DATA = 0, 1, 2
GLOBAL_OUT, GLOBAL_MIX = 0, 1
OP_CONST, OP_CONST_EXT, OP_GET, OP_GET_GLOBAL, OP_ADD, OP_SUB, OP_MUL, OP_TRUE, OP_AND_EQZ, OP_AND_COND = range(10)
P = 2013265921


@dataclass(frozen=True)
class Fp:
    """Handle to an fp_vars entry."""
    idx: int


@dataclass(frozen=True)
class Mix:
    """Handle to a mix_vars entry."""
    idx: int


@dataclass
class CircuitBuilder:
    """Tiny DSL that records PolyExtSteps and the taps they touch."""
    group_sizes: Tuple[int, int, int]
    global_sizes: Tuple[int, int]
    kind: int = 1
    steps: List[Tuple[int, int, int, int, int]] = field(default_factory=list)
    n_fp: int = 0
    n_mix: int = 0
    _taps: dict = field(default_factory=dict)           # (group, offset, back) -> [step indices]
    _cache: dict = field(default_factory=dict)

    def _fp(self, op, a=0, b=0, c=0, d=0) -> Fp:
        key = (op, a, b, c, d)
        if op in (OP_CONST, OP_GET_GLOBAL) and key in self._cache:
            return self._cache[key]
        self.steps.append((op, a, b, c, d))
        r = Fp(self.n_fp)
        self.n_fp += 1
        if op in (OP_CONST, OP_GET_GLOBAL):
            self._cache[key] = r
        return r

    def _mix(self, op, a=0, b=0, c=0) -> Mix:
        self.steps.append((op, a, b, c, 0))
        r = Mix(self.n_mix)
        self.n_mix += 1
        return r

    # value steps
    def const(self, v: int) -> Fp:
        return self._fp(OP_CONST, v % P)

    def get(self, group: int, offset: int, back: int = 0) -> Fp:
        key = ("tap", group, offset, back)
        if key in self._cache:
            return self._cache[key]
        self.steps.append((OP_GET, -1, 0, 0, 0))          # tap index patched in finish()
        self._taps.setdefault((group, offset, back), []).append(len(self.steps) - 1)
        r = Fp(self.n_fp)
        self.n_fp += 1
        self._cache[key] = r
        return r

    def get_global(self, base: int, off: int) -> Fp:
        return self._fp(OP_GET_GLOBAL, base, off)

    def const_ext(self, c0: int, c1: int, c2: int, c3: int) -> Fp:
        """PolyExtStep::ConstExt: an extension-field constant (the value it flows into becomes Fp4-valued)."""
        return self._fp(OP_CONST_EXT, c0 % P, c1 % P, c2 % P, c3 % P)

    def add(self, a: Fp, b: Fp) -> Fp:
        return self._fp(OP_ADD, a.idx, b.idx)

    def sub(self, a: Fp, b: Fp) -> Fp:
        return self._fp(OP_SUB, a.idx, b.idx)

    def mul(self, a: Fp, b: Fp) -> Fp:
        return self._fp(OP_MUL, a.idx, b.idx)

    # mix steps
    def true(self) -> Mix:
        return self._mix(OP_TRUE)

    def and_eqz(self, x: Mix, v: Fp) -> Mix:
        return self._mix(OP_AND_EQZ, x.idx, v.idx)

    def and_cond(self, x: Mix, cond: Fp, inner: Mix) -> Mix:
        return self._mix(OP_AND_COND, x.idx, cond.idx, inner.idx)

    def finish(self, ret: Mix) -> np.ndarray:
        """Every column of every group gets at least a back-0 tap (upstream: every register is tapped)."""
        for g, size in enumerate(self.group_sizes):
            for off in range(size):
                if not any(k[0] == g and k[1] == off for k in self._taps):
                    self._taps.setdefault((g, off, 0), [])
        taps = sorted(self._taps)
        for ti, key in enumerate(taps):
            for si in self._taps[key]:
                self.steps[si] = (OP_GET, ti, 0, 0, 0)
        # combos = distinct back-sets of the registers, sorted
        regs = {}
        for g, off, back in taps:
            regs.setdefault((g, off), []).append(back)
        combos = sorted({tuple(v) for v in regs.values()})
        words = [MAGIC, 1, 3, *self.group_sizes, 2, *self.global_sizes, len(taps), len(combos), len(self.steps),
                 ret.idx, self.kind, 0, 0]
        assert len(words) == HEADER_WORDS
        for t in taps:
            words.extend(t)
        for c in combos:
            words.append(len(c))
            words.extend(c)
        for s in self.steps:
            words.extend(s)
        return np.asarray(words, dtype=np.uint32)


@dataclass
class Circuit:
    """Parsed view of a desc blob (host-side mirror of TapSet)."""
    desc: np.ndarray
    group_sizes: Tuple[int, int, int]
    global_sizes: Tuple[int, int]
    taps: List[Tuple[int, int, int]]
    combos: List[Tuple[int, ...]]
    steps: List[Tuple[int, int, int, int, int]]
    ret: int
    kind: int

    @staticmethod
    def parse(desc: Sequence[int]) -> "Circuit":
        d = np.asarray(desc, dtype=np.uint32)
        assert d[0] == MAGIC and d[1] == 1, "bad circuit desc"
        gs = tuple(int(x) for x in d[3:6])
        gl = (int(d[7]), int(d[8]))
        n_taps, n_combos, n_steps, ret, kind = (int(x) for x in d[9:14])
        pos = HEADER_WORDS
        taps = [tuple(int(x) for x in d[pos + 3 * i: pos + 3 * i + 3]) for i in range(n_taps)]
        pos += 3 * n_taps
        combos = []
        for _ in range(n_combos):
            cnt = int(d[pos])
            combos.append(tuple(int(x) for x in d[pos + 1: pos + 1 + cnt]))
            pos += 1 + cnt
        steps = [tuple(int(x) for x in d[pos + 5 * i: pos + 5 * i + 5]) for i in range(n_steps)]
        return Circuit(d, gs, gl, taps, combos, steps, ret, kind)

    @property
    def regs(self):
        """[(group, offset, [backs], combo_id)] in tap order."""
        out = []
        for g, off, back in self.taps:
            if out and out[-1][0] == g and out[-1][1] == off:
                out[-1][2].append(back)
            else:
                out.append([g, off, [back], None])
        for r in out:
            r[3] = self.combos.index(tuple(r[2]))
        return [tuple(r) for r in out]
