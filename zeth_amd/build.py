"""Build libzkhal_mi355x.so (gfx950) in-tree with hipcc, and the CPU oracle with gcc.

`python -m zeth_amd.build` or `__graft_entry__.build()`.  hipcc cross-compiles without a GPU.  The shared
library lands next to this file (git-ignored, but shipped to the GPU box by gpurun).
"""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
# experiment builds of the same ABI (tools/gpu_*.sh A/B runs; loaded through ZKH_LIBRARY): ZKH_BUILD_VARIANT=<name>
# ZKH_BUILD_FLAGS="-D..." -> .variants/libzkhal_<name>.so with its own object directory
VARIANT = os.environ.get("ZKH_BUILD_VARIANT", "")
LIB = os.path.join(ROOT, ".variants", f"libzkhal_{VARIANT}.so") if VARIANT else os.path.join(HERE, "libzkhal_mi355x.so")
OBJ_DIR = os.path.join(ROOT, ".variants", f"_obj_{VARIANT}") if VARIANT else os.path.join(HERE, "csrc", "_obj")
SOURCES = ["hal.hip", "ntt.hip", "hash.hip", "poly.hip", "circuit.hip", "prover.hip", "verifier.hip", "session.hip", "recursion.hip", "topology.hip", "preflight.hip"]   # + generated eval_check units
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-comment", "-Wno-unused-result"] + os.environ.get("ZKH_BUILD_FLAGS", "").split()
# hash.hip: the unrolled Poseidon2 source order already interleaves 24 independent cells; LLVM's machine scheduler
# re-interleaves it up to the register budget (126 VGPRs instead of 86) for no measurable gain on MI355X (hash_rows
# M3 9.55 vs 9.59 ms): the source order is kept for the smaller footprint.
EXTRA_FLAGS = {"hash.hip": ["-mllvm", "-enable-misched=0"]}


def _extra(src: str) -> list:
    """Per-file flags; the generated eval_check units share theirs with the load-time compiler (circuits/jit.py)."""
    if src.startswith("eval_check_gen"):
        from .circuits import codegen
        return EXTRA_FLAGS.get(src, []) + codegen.KERNEL_FLAGS
    return EXTRA_FLAGS.get(src, [])


def _deps_digest(src: str) -> str:
    h = hashlib.sha256()
    files = [os.path.join(CSRC, src)]
    files += sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h"))
    files += [os.path.join(ROOT, "include", f) for f in sorted(os.listdir(os.path.join(ROOT, "include")))]
    for f in files:
        with open(f, "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(FLAGS + _extra(src)).encode())
    return h.hexdigest()


def _compile(src: str, force: bool) -> str:
    obj = os.path.join(OBJ_DIR, src.replace(".hip", ".o"))
    stamp = obj + ".sha"
    digest = _deps_digest(src)
    if not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == digest:
        return obj
    cmd = [HIPCC, *FLAGS, *_extra(src), "-c", os.path.join(CSRC, src), "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr[-4000:]}")
    with open(stamp, "w") as fh:
        fh.write(digest)
    return obj


def generate_eval_check() -> list:
    """Emit csrc/eval_check_gen*.hip: straight-line eval_check kernels for the shipped circuits (one translation unit per
    part of a split constraint system, so that the parts compile in parallel).  Returns the generated file names."""
    from .circuits import codegen
    return codegen.write_generated(CSRC)


def build_examples() -> str:
    """examples/seal_segments (the session written out against the low-level entry points) and examples/prove_session (the
    same session as one zkh_session_prove call): plain g++ consumers of include/zkhal.h (no HIP headers, no Python).
    Returns the path of seal_segments."""
    outs = []
    for name in ("seal_segments", "prove_session"):
        src = os.path.join(ROOT, "examples", name + ".cpp")
        out = os.path.join(ROOT, "examples", name)
        outs.append(out)
        deps = [src, os.path.join(ROOT, "include", "zkhal.h"), LIB]
        if os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in deps):
            continue
        cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-I", os.path.join(ROOT, "include"), src, "-o", out, "-L", HERE,
               "-lzkhal_mi355x", "-Wl,-rpath,$ORIGIN/../zeth_amd", "-Wl,-rpath-link,/opt/rocm/lib", "-lpthread"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"g++ failed for examples/{name}.cpp:\n{r.stderr[-4000:]}")
    return outs[0]


def build_oracle() -> None:
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ_DIR, exist_ok=True)
    sources = SOURCES + generate_eval_check()
    wanted = {s.replace(".hip", ".o") for s in sources}
    for f in os.listdir(OBJ_DIR):                       # objects of generated units that no longer exist
        if f.startswith("eval_check_gen") and f.endswith(".o") and f not in wanted:
            os.remove(os.path.join(OBJ_DIR, f))
    with ThreadPoolExecutor(max_workers=min(os.cpu_count() or 4, len(sources))) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), sources))
    newest = max(os.path.getmtime(o) for o in objs)
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < newest:
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
    if not VARIANT:
        build_examples()
    if verbose:
        print("built", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    build_oracle()
