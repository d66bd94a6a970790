"""Regenerates tests/golden/recursion_digests.json: SHA-256 of the RECURSION circuit description, of a small program's blob, of
the CPU oracle's witness and seal for it, and of the lift program of a SYN-tiny po2-8 segment with its oracle witness.  A change
of the circuit, the program format, the assembler or the verifier-as-program shows up here first.
    python tests/golden/make_golden_recursion.py"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(HERE, ".."))


def digests():
    import zko
    from test_recursion import enc, small_program
    from zeth_amd.circuits import rec_verify as V, recursion as R, syn_air
    from zeth_amd.circuits.desc import P
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a, dtype="<u4").tobytes()).hexdigest()
    lib = zko.load()
    rec = zko.OracleCircuit(lib, R.recursion_circuit())
    out = {"recursion_desc": sha(R.recursion_circuit())}
    pr, _ = small_program()
    blob = pr.finish(7, 50)
    code, data, o = rec.rec_witgen(blob, enc([5, 6, 7, 8, 11, 12, 13, 14]))
    out.update(small_blob=sha(blob), small_code=sha(code), small_data=sha(data), small_out=sha(o),
               small_seal=sha(rec.prove_traces(7, code, data, o, 50)))
    desc = syn_air.syn_tiny()
    child = zko.OracleCircuit(lib, desc)
    rinv = pow((1 << 32) % P, -1, P)
    seal = child.prove(8, 50)
    lift = V.build_lift(desc, 8, [int(w) * rinv % P for w in child.control_root(8, 50)])
    lblob = lift.finish(lift.min_po2())
    _, ldata, lout = rec.rec_witgen(lblob, np.concatenate([seal, np.arange(1, 9, dtype=np.uint32)]))
    out.update(child_seal=sha(seal), lift_blob=sha(lblob), lift_data=sha(ldata), lift_out=sha(lout),
               lift_shape=[lift.min_po2(), len(lift.p2s), len(lift.gates), lift.n_inputs])
    return out


if __name__ == "__main__":
    with open(os.path.join(HERE, "recursion_digests.json"), "w") as fh:
        json.dump(digests(), fh, indent=1)
    print(open(os.path.join(HERE, "recursion_digests.json")).read())
