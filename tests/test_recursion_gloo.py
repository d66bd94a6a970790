"""The N > 1 shape of the recursive fold on CPU (gloo, world size 2): every rank folds its own aligned range of leaves (here one
leaf each: a lift - the fold inside a rank is tests/test_recursion.py's join test), rank 0 gathers the local roots over the control
plane and joins them - with the CPU oracle standing in for the GPU (test-only), real seals all the way: the top join's witness
exists, satisfies every constraint, and carries the claim tree of all leaves."""
import os
import socket
import sys

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPO2, CZK, N_LEAVES = 8, 50, 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="8")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import zko
    from zeth_amd import recursion as host_rec
    from zeth_amd.circuits import rec_verify as V, recursion as R, syn_air
    from zeth_amd.circuits.desc import P
    rinv = pow((1 << 32) % P, -1, P)
    lib = zko.load()
    desc = syn_air.syn_tiny()
    child, rec = zko.OracleCircuit(lib, desc), zko.OracleCircuit(lib, R.recursion_circuit())
    croot = child.control_root(CPO2, CZK)
    # the program set (every rank builds the same one, as every lane loads the same blobs)
    lift = V.build_lift(desc, CPO2, [int(w) * rinv % P for w in croot])
    lpo2 = lift.min_po2()
    lblob = lift.finish(lpo2)
    j1 = V.build_join(R.recursion_circuit(), lpo2, lpo2)
    jpo2 = j1.min_po2()
    j1blob = j1.finish(jpo2)
    roots = []
    for blob, po2 in ((lblob, lpo2), (j1blob, jpo2)):
        code = np.zeros(R.WC << po2, np.uint32)
        assert lib.zko_rec_code(blob, blob.size, code) is None
        roots.append(rec.root_of_code(po2, code))
    levels = host_rec.allowed_tree(roots)
    A = levels[-1][0]

    def run(blob, po2, inputs, seal=True):
        code, data, out = rec.rec_witgen(blob, inputs)
        return (rec.prove_traces(po2, code, data, out) if seal else None), (code, data, out)
    mine = host_rec.aligned_range(N_LEAVES, world, rank)
    leaves = {i: child.prove(CPO2, CZK, seed=300 + i) for i in mine}
    claims = {}
    for i, s in leaves.items():
        cin = np.concatenate([s[:5], croot])
        c = np.zeros(8, np.uint32)
        lib.zko_hash_elem_slice(np.ascontiguousarray(cin), cin.size, 1, c)
        claims[i] = c
    assert len(mine) == 1
    local_root, _ = run(lblob, lpo2, np.concatenate([leaves[mine[0]], A]))          # this rank's fold: one leaf, lifted and sealed
    gathered = [None] * world if rank == 0 else None
    dist.gather_object((local_root, claims), gathered, dst=0)
    if rank == 0:
        path0 = host_rec.membership_words(levels, 0)
        allc = {k: v for _, part in gathered for k, v in part.items()}
        opening = lambda i: np.concatenate([allc[i], np.zeros(2, np.uint32)])        # a lift's claim' opens to (receipt claim, pre 0, post 0)
        _, (code, data, out) = run(j1blob, jpo2, np.concatenate([gathered[0][0], path0, opening(0), gathered[1][0], path0, opening(1)]), seal=False)
        mix = np.array([(i * 7919 + 13) % P for i in range(20)], dtype=np.uint32)
        bad = rec.check_rows(jpo2, rec.rec_accum(jpo2, code, data, mix), code, data, out, mix)
        allc = {k: v for _, part in gathered for k, v in part.items()}
        want = host_rec.fold_leaf_claims([allc[i] for i in range(N_LEAVES)], ranks=world)
        q.put({"bad_row": bad, "claim_ok": bool(np.array_equal(out[:8], want)), "allowed_ok": bool(np.array_equal(out[8:], A)),
               "ranges": [list(host_rec.aligned_range(N_LEAVES, world, r)) for r in range(world)]})
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_fold_their_ranges_and_rank0_joins_the_local_roots():
    from zeth_amd import recursion as host_rec
    import pytest
    with pytest.raises(ValueError):
        host_rec.aligned_range(6, 4, 0)
    assert list(host_rec.aligned_range(1024, 8, 3)) == list(range(384, 512))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=900)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert res == {"bad_row": -1, "claim_ok": True, "allowed_ok": True, "ranges": [[0], [1]]}
