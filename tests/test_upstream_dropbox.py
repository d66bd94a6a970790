"""tests/upstream/ — the auto-pin drop box (round-5 verdict, item 7).  Every upstream artefact found there is run through its
importer / checker as a test and the outcome lands in tests/upstream/parity_pins.json (bench.py copies it into config.parity_pins);
an absent artefact is a SKIP that says what to drop where.  See tests/upstream/README.md."""
import glob
import json
import os
import subprocess
import sys
import time

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BOX = os.path.join(ROOT, "tests", "upstream")
PINS = os.path.join(BOX, "parity_pins.json")


def _pin(key: str, ok: bool, detail: str) -> None:
    try:
        pins = json.load(open(PINS))
    except (OSError, ValueError):
        pins = {}
    pins[key] = {"ok": bool(ok), "detail": detail[-600:], "checked_at": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime())}
    with open(PINS, "w") as fh:
        json.dump(pins, fh, indent=1, sort_keys=True)


def _tool(*argv, timeout=1800):
    return subprocess.run([sys.executable, *argv], capture_output=True, text=True, timeout=timeout, cwd=ROOT)


def test_the_drop_box_is_documented_and_its_contents_stay_out_of_history():
    assert os.path.exists(os.path.join(BOX, "README.md"))
    ignore = open(os.path.join(ROOT, ".gitignore")).read()
    assert "tests/upstream/*" in ignore and "!tests/upstream/README.md" in ignore


def test_upstream_poseidon2_constants():
    path = os.path.join(BOX, "consts.rs")
    if not os.path.exists(path):
        pytest.skip("drop risc0-zkp/src/core/hash/poseidon2/consts.rs into tests/upstream/ to pin the Poseidon2 tables to upstream's")
    r = _tool(os.path.join(ROOT, "tools", "import_upstream_consts.py"), path)
    _pin("poseidon2_consts", r.returncode == 0, r.stdout + r.stderr)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_upstream_circuit_tables_load_as_data():
    taps, poly = os.path.join(BOX, "taps.rs"), os.path.join(BOX, "poly_ext.rs")
    if not (os.path.exists(taps) and os.path.exists(poly)):
        pytest.skip("drop a Zirgen circuit's taps.rs + poly_ext.rs (+ info.rs) into tests/upstream/ to load the real circuit as data")
    out = os.path.join(BOX, "circuit.desc.npy")
    args = [os.path.join(ROOT, "tools", "import_upstream_circuit.py"), taps, poly]
    if os.path.exists(os.path.join(BOX, "info.rs")):
        args.append(os.path.join(BOX, "info.rs"))
    if os.path.exists(os.path.join(BOX, "circuit.kind")):          # which built-in witness generator the blob names (0 = none: a foreign circuit)
        args += ["--kind", open(os.path.join(BOX, "circuit.kind")).read().strip()]
    r = _tool(*args, "-o", out)
    ok = r.returncode == 0 and os.path.exists(out)
    detail = r.stdout + r.stderr
    if ok:
        # the kernels the generator emits for it: within bounds for every input, and equal to the oracle's literal interpreter
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import check_bounds
        import zko
        from zeth_amd.circuits import codegen
        desc = np.load(out)
        viol, st = check_bounds.check_desc("upstream", desc)
        ok = ok and not viol
        detail += f"\nbounds: {len(viol)} violations over {st['kernels']} kernels, {st['statements']} statements"
        if not viol:
            import ctypes as C
            oracle = zko.load()
            oc = zko.OracleCircuit(oracle, desc)
            po2, P = 5, 2013265921
            dom = 4 << po2
            rng = np.random.default_rng(1)
            gs = [rng.integers(0, P, size=int(w) * dom, dtype=np.uint64).astype(np.uint32) for w in desc[3:6]]
            outg = rng.integers(0, P, size=max(1, int(desc[7])), dtype=np.uint64).astype(np.uint32)
            mix = rng.integers(0, P, size=max(1, int(desc[8])), dtype=np.uint64).astype(np.uint32)
            pm = rng.integers(0, P, size=4, dtype=np.uint64).astype(np.uint32)
            want = np.zeros(4 * dom, np.uint32)
            oracle.zko_eval_check(oc.h, want, (C.c_void_p * 3)(*[a.ctypes.data for a in gs]), (C.c_void_p * 2)(outg.ctypes.data, mix.ctypes.data), pm, po2)
            parts, _, _ = codegen.emit_parts("upstream", desc)
            tot = [0] * 4
            for _, src in parts:
                tot = [(a + b) % P for a, b in zip(tot, check_bounds.execute_source(src, gs, (outg, mix), pm, po2, 3))]
            same = tot == [int(want[k * dom + 3]) for k in range(4)]
            ok = ok and same
            detail += f"\ngenerated kernels executed on the CPU {'equal' if same else 'DIFFER FROM'} the literal interpreter"
    _pin("circuit_tables", ok, detail)
    assert ok, detail[-3000:]


def test_upstream_seals():
    seals = sorted(glob.glob(os.path.join(BOX, "*.seal.bin")))
    if not seals:
        pytest.skip("drop an upstream SegmentReceipt.seal as <name>.seal.bin (+ circuit.desc.npy or <name>.circuit) into tests/upstream/ "
                    "to pin the seal layout and the Fiat-Shamir order")
    bad = []
    for seal in seals:
        stem = seal[:-len(".seal.bin")]
        circuit = os.path.join(BOX, "circuit.desc.npy")
        if os.path.exists(stem + ".circuit"):
            circuit = open(stem + ".circuit").read().strip()
        args = [os.path.join(ROOT, "tools", "check_upstream_seal.py"), seal, circuit]
        if os.path.exists(stem + ".control_root"):
            args += ["--control-root", *open(stem + ".control_root").read().split()]
        r = _tool(*args)
        _pin("seal:" + os.path.basename(seal), r.returncode == 0, r.stdout + r.stderr)
        if r.returncode != 0:
            bad.append(os.path.basename(seal) + ": " + (r.stdout + r.stderr)[-1500:])
    assert not bad, "\n".join(bad)


def test_upstream_receipts():
    receipts = sorted(glob.glob(os.path.join(BOX, "*.receipt.bin")))
    if not receipts:
        pytest.skip("drop bincode::serialize(&receipt) as <name>.receipt.bin into tests/upstream/ to pin the Receipt family's field order")
    bad = []
    for path in receipts:
        r = _tool(os.path.join(ROOT, "tools", "check_upstream_receipt.py"), path)
        _pin("receipt:" + os.path.basename(path), r.returncode == 0, r.stdout + r.stderr)
        if r.returncode != 0:
            bad.append(os.path.basename(path) + ": " + (r.stdout + r.stderr)[-1500:])
    assert not bad, "\n".join(bad)


def test_the_drop_box_works_end_to_end_on_this_repositorys_own_artefacts(tmp_path, monkeypatch, oracle):
    """The same tests against a box filled with what THIS repository can produce (its Poseidon2 tables in Rust syntax, a shipped
    circuit exported as taps.rs / poly_ext.rs, an oracle seal): they must all pass and write their pins — so that the day real files
    arrive, a failure is about the files."""
    import importlib
    import zko
    from zeth_amd.circuits import syn_air
    box = tmp_path / "upstream"
    box.mkdir()
    me = importlib.import_module("test_upstream_dropbox")
    monkeypatch.setattr(me, "BOX", str(box))
    monkeypatch.setattr(me, "PINS", str(box / "parity_pins.json"))
    r = _tool(os.path.join(ROOT, "tools", "export_rust_syntax.py"), "consts", str(box))
    assert r.returncode == 0, r.stdout + r.stderr
    r = _tool(os.path.join(ROOT, "tools", "export_rust_syntax.py"), "circuit", "syn_small", str(box))
    assert r.returncode == 0, r.stdout + r.stderr
    (box / "circuit.kind").write_text("1")                           # SYN-AIR: the oracle's witness generator makes the seal below
    me.test_upstream_poseidon2_constants()
    me.test_upstream_circuit_tables_load_as_data()
    # a seal of the imported circuit, made by the oracle at the protocol's blinding size; the control root the checker is told = the seal's own
    desc = np.load(box / "circuit.desc.npy")
    assert np.array_equal(desc, syn_air.syn_small())
    oc = zko.OracleCircuit(oracle, desc)
    seal = oc.prove(12, 1994, 5, 6)
    np.asarray(seal, dtype="<u4").tofile(box / "small.seal.bin")
    (box / "small.control_root").write_text("self")
    me.test_upstream_seals()
    pins = json.load(open(box / "parity_pins.json"))
    assert pins["poseidon2_consts"]["ok"] is True and pins["circuit_tables"]["ok"] is True and pins["seal:small.seal.bin"]["ok"] is True
    assert "equal the literal interpreter" in pins["circuit_tables"]["detail"]
