"""tools/import_upstream_{consts,circuit}.py and tools/check_upstream_seal.py: the one-command checks that turn the
un-vendored upstream artefacts (risc0-zkp consts.rs; Zirgen taps.rs / poly_ext.rs / info.rs; a seal) into this repository's
data the moment they are supplied (/root/reference/Cargo.lock:5393, :5320).  Exercised here on Rust-syntax fixtures emitted
from the repository's own data by tools/export_rust_syntax.py, in every encoding the importers claim to understand, plus
the failure modes."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_upstream_seal as cus  # noqa: E402
import export_rust_syntax as ers  # noqa: E402
import import_upstream_circuit as iuc  # noqa: E402
import import_upstream_consts as iuk  # noqa: E402

from zeth_amd.circuits import syn_air, syn_heavy, syn_random  # noqa: E402

P = 2013265921


# ---------------------------------------------------------------- consts.rs
@pytest.mark.parametrize("montgomery", [False, True])
@pytest.mark.parametrize("compact", [False, True])
def test_consts_importer_recognises_every_encoding_of_the_shipped_tables(tmp_path, montgomery, compact):
    f = tmp_path / "consts.rs"
    f.write_text(ers.consts_rust(montgomery, compact))
    log = []
    assert iuk.compare(str(f), out=log.append) == 0, log
    assert ("Montgomery words" if montgomery else "canonical residues") in log[1]
    assert ("compact" if compact else "[round][cell]") in log[1]
    assert "identical" in log[-1]


def test_consts_importer_reports_a_differing_word_and_can_adopt_upstreams_tables(tmp_path):
    txt = ers.consts_rust(False, False)
    # change one DIAGONAL entry: the tables still... do NOT reproduce the known-answer vector -> unusable, exit 2
    rc, diag, _ = iuk.shipped_header()
    bad = txt.replace(f"Elem::new({diag[5]})", f"Elem::new({(diag[5] + 1) % P})")
    assert bad != txt
    f = tmp_path / "consts.rs"
    f.write_text(bad)
    log = []
    assert iuk.compare(str(f), out=log.append) == 2 and "known-answer" in log[-1]
    # a header that differs from (correct) upstream tables in a word the KAT cannot see is impossible (every word is used),
    # so "exit 1" is reached the other way round: a SHIPPED header with a wrong word against correct upstream tables
    f.write_text(txt)
    wrong = tmp_path / "zkh_poseidon2_consts.h"
    hdr = open(os.path.join(ROOT, "include", "zkh_poseidon2_consts.h")).read()
    wrong.write_text(hdr.replace("0x0795bb97u", "0x0795bb98u"))
    real = iuk.shipped_header
    iuk.shipped_header = lambda: (lambda r, d, p: (r[:1] + [r[1] + 1] + r[2:], d, p))(*real())
    try:
        log = []
        assert iuk.compare(str(f), write=True, header_path=str(wrong), out=log.append) == 1
        assert any("ROUND_CONSTANTS[round 0][cell 1]" in ln for ln in log)
    finally:
        iuk.shipped_header = real
    # --write-header adopted upstream's tables: the rewritten header now parses back to exactly the shipped tables
    body = wrong.read_text()
    assert "ZKH_P2_CONSTS_ARE_DERIVED 0" in body and "0x0795bb97u" in body


def test_consts_importer_rejects_what_it_cannot_parse(tmp_path):
    f = tmp_path / "consts.rs"
    f.write_text("pub const SOMETHING_ELSE: [u32; 2] = [1, 2];")
    assert iuk.compare(str(f), out=lambda *_: None) == 2
    f.write_text("pub const ROUND_CONSTANTS: [Elem; 3] = [Elem::new(1), Elem::new(2), Elem::new(3)];\npub const M_INT_DIAG_HZN: [Elem; 24] = [" +
                 ", ".join("Elem::new(1)" for _ in range(24)) + "];")
    assert iuk.compare(str(f), out=lambda *_: None) == 2


def test_the_tools_literal_permutation_reproduces_the_published_vector():
    rc, diag, _ = iuk.shipped_header()
    kin, kout = iuk.load_kat()
    assert iuk.permute(kin, rc, diag) == kout


# ---------------------------------------------------------------- taps.rs / poly_ext.rs / info.rs
CIRCUITS = {"syn_a": syn_air.syn_a, "syn_join": syn_air.syn_join, "syn_heavy_small": syn_heavy.syn_heavy_small,
            "syn_heavy": syn_heavy.syn_heavy, "random_3": lambda: syn_random.random_circuit(3)}


@pytest.mark.parametrize("name", sorted(CIRCUITS))
def test_circuit_importer_round_trips_rust_syntax_to_the_identical_blob(tmp_path, name):
    desc = CIRCUITS[name]()
    files = ers.circuit_rust(desc)
    for fn, txt in files.items():
        (tmp_path / fn).write_text(txt)
    back = iuc.import_circuit(str(tmp_path / "taps.rs"), str(tmp_path / "poly_ext.rs"), str(tmp_path / "info.rs"), kind=int(desc[13]))
    assert np.array_equal(back, desc)
    # sizes on the command line instead of info.rs; comments, odd spacing and typed literals do not matter
    noisy = files["taps.rs"].replace("TapData {", "/* reg */ TapData{").replace("offset: ", "offset:  0_").replace(", group", "u16 , group")
    (tmp_path / "taps.rs").write_text(noisy)
    c = iuc.D.Circuit.parse(desc)
    back = iuc.import_circuit(str(tmp_path / "taps.rs"), str(tmp_path / "poly_ext.rs"), None, c.global_sizes[0], c.global_sizes[1], int(desc[13]))
    assert np.array_equal(back, desc)


def test_circuit_importer_cross_checks_the_redundant_tables(tmp_path):
    files = ers.circuit_rust(syn_air.syn_small())

    def run(taps=None, poly=None):
        (tmp_path / "taps.rs").write_text(taps or files["taps.rs"])
        (tmp_path / "poly_ext.rs").write_text(poly or files["poly_ext.rs"])
        (tmp_path / "info.rs").write_text(files["info.rs"])
        return iuc.import_circuit(str(tmp_path / "taps.rs"), str(tmp_path / "poly_ext.rs"), str(tmp_path / "info.rs"))

    run()
    with pytest.raises(iuc.ImportError_, match="skip"):
        run(taps=files["taps.rs"].replace("combo: 1, skip: 2", "combo: 1, skip: 3", 1))
    with pytest.raises(iuc.ImportError_, match="says combo"):
        run(taps=files["taps.rs"].replace("combo: 1, skip: 2", "combo: 0, skip: 2", 1))
    with pytest.raises(iuc.ImportError_, match="reg_count"):
        run(taps=files["taps.rs"].replace("reg_count: ", "reg_count: 1"))
    with pytest.raises(iuc.ImportError_, match="unknown step"):
        run(poly=files["poly_ext.rs"].replace("PolyExtStep::True", "PolyExtStep::Maybe", 1))
    with pytest.raises(iuc.ImportError_, match="only .* defined so far"):
        run(poly=files["poly_ext.rs"].replace("PolyExtStep::Mul(", "PolyExtStep::Mul(4000000", 1))


# ---------------------------------------------------------------- a seal against the layout
def _seal(po2=12):
    import zko
    desc = syn_air.syn_small()
    oc = zko.OracleCircuit(zko.load(), desc)
    return desc, oc.prove(po2, 1994, 0x5EED0000, 0x2E80), oc.control_root(po2, 1994)


def test_seal_checker_accepts_an_oracle_seal_and_names_the_section_of_a_disagreement(tmp_path):
    desc, seal, root = _seal()
    log = []
    assert cus.check(seal, desc, root, out=log.append) == 0, log[-3:]
    assert cus.check(seal, desc, None, out=lambda *_: None) == 0                  # root taken from the seal itself
    from zeth_amd.circuits.desc import Circuit
    lay = cus.layout(Circuit.parse(desc), 12)
    assert lay[-1][3] == seal.size
    # the file formats: little-endian bytes and .npy
    (tmp_path / "seal.bin").write_bytes(seal.astype("<u4").tobytes())
    assert np.array_equal(cus.load_words(str(tmp_path / "seal.bin")), seal)

    def section_of(word, mutate=lambda w: (w + 1) % P, root_=root):
        bad = seal.copy()
        bad[word] = mutate(int(bad[word]))
        log = []
        assert cus.check(bad, desc, root_, out=log.append) == 1
        return "\n".join(log)
    sec = {name: (a, b) for name, kind, a, b in lay}
    kinds = {name: kind for name, kind, a, b in lay}
    txt = section_of(sec["code tree top layer"][0] + 3)
    assert "code root does not match the control root" in txt and "`code tree top layer`" in txt and "check_code" in txt
    txt = section_of(sec["coeff_u"][0] + 5)
    assert "constraint check failed" in txt and "`coeff_u`" in txt and "poly_interpolate" in txt
    txt = section_of(sec["coeff_u"][0] + 5, mutate=lambda w: 0xFFFFFFFF)
    assert "UNREDUCED" in txt and "`coeff_u`" in txt
    first_q = next(name for name in sec if name.startswith("query 0: accum"))
    txt = section_of(sec[first_q][0])
    assert "authentication path" in txt and "query 0: accum opening" in txt and "random_bits" in txt
    fin = next(name for name in sec if name.startswith("final coefficients"))
    txt = section_of(sec[fin][0] + 1)
    assert "FIRST DISAGREEMENT" in txt
    # a seal of another length: the layout table says by how much
    log = []
    assert cus.check(seal[:-8], desc, root, out=log.append) == 1
    assert any("LENGTH" in ln and "-8" in ln for ln in log)
    # po2 that is not a size
    bad = seal.copy()
    bad[4] = 5
    log = []
    assert cus.check(bad, desc, root, out=log.append) == 1 and "po2" in log[0]
