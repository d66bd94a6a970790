#!/usr/bin/env python3
"""GPU experiment (round-3 verdict item 7): SYN-HEAVY eval_check with every part's mix powers GATHERED into emission order
(codegen.py GATHER), in circuit order and under tap-set locality ordering, against the shipped generator.

    python tools/exp_gather.py default .variants/libzkhal_oldgen.so
Each library is a build of the same ABI (the round-3 generator: ZKH_BUILD_VARIANT=oldgen ZKH_CODEGEN_GATHER=0 ZKH_CODEGEN_LOCALITY=0
ZKH_CODEGEN_PART=3200 ZKH_CODEGEN_PREFETCH=4 python -m zeth_amd.build)
run in its own process on RANDOM evaluated groups; the result of the generated kernels is compared word for word with the
step-list interpreter's in the same process, and its digest across libraries."""
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(po2: int, reps: int) -> None:
    import numpy as np
    from zeth_amd.circuits import codegen
    from zeth_amd.circuits.desc import Circuit, P
    from zeth_amd.hal import HipHal
    hal = HipHal(0)
    rng = np.random.default_rng(7)
    dom = 4 << po2
    out = {"lib": os.environ.get("ZKH_LIBRARY", "default")}
    for name in ("syn_heavy", "syn_a", "keccak_f", "recursion", "p2_join"):
        p = {"keccak_f": min(po2, 14), "recursion": min(po2, 18), "p2_join": min(po2, 16)}.get(name, po2)
        d = 4 << p
        desc = codegen.shipped()[name]
        c = Circuit.parse(desc)
        circ = hal.load_circuit(desc)
        groups = [hal.copy_from(f"g{i}", rng.integers(0, P, w * d, dtype=np.uint32)) for i, w in enumerate(c.group_sizes)]
        gl = [hal.copy_from(f"gl{i}", rng.integers(0, P, max(1, s), dtype=np.uint32)) for i, s in enumerate(c.global_sizes)]
        check = hal.alloc_elem("check", 4 * d)
        mix = rng.integers(1, P, 4, dtype=np.uint32)
        circ.eval_check(check, groups, gl, mix, p)
        hal.sync()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(reps):
                circ.eval_check(check, groups, gl, mix, p)
            hal.sync()
            best = min(best, (time.perf_counter() - t0) / reps * 1e3)
        got = check.to_vec()
        rec = {"po2": p, "parts": circ.compiled_parts(), "ms": round(best, 3), "digest": hashlib.sha256(got.tobytes()).hexdigest()[:16]}
        circ.eval_check(check, groups, gl, mix, p, use_interpreter=True)
        rec["equals_interpreter"] = bool(np.array_equal(check.to_vec(), got))
        out[name] = rec
        del groups, gl, check
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(int(sys.argv[2]), int(sys.argv[3]))
    else:
        for rnd in range(2):
            for lib in sys.argv[1:]:
                env = dict(os.environ)
                env.pop("ZKH_LIBRARY", None)
                if lib != "default":
                    env["ZKH_LIBRARY"] = os.path.join(ROOT, lib)
                subprocess.run([sys.executable, __file__, "--child", os.environ.get("EXP_PO2", "20"), "5"], env=env, check=False)
