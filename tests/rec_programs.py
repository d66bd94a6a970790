"""Seeded random RECURSION programs over every gate kind (shared by the CPU and the GPU tests)."""
import numpy as np

from zeth_amd.circuits import recursion as R
from zeth_amd.circuits.desc import P


def random_program(seed: int):
    """-> (program, input word count, the four public wires)"""
    rng = np.random.default_rng(1000 + seed)
    pr = R.Program()
    n_in = int(rng.integers(2, 6))
    wires = [pr.input(4 * i, int(rng.integers(1, 5))) for i in range(n_in)]
    emb = [pr.const(int(rng.integers(0, P))) for _ in range(3)]
    for _ in range(int(rng.integers(20, 60))):
        k = int(rng.integers(0, 9))
        a, b, c = (wires[int(rng.integers(0, len(wires)))] for _ in range(3))
        if k == 0:
            wires.append(pr.gen(a, b, c, *(int(rng.integers(0, P)) for _ in range(5))))
        elif k == 1:
            wires.append(pr.muladd(a, b, c, int(rng.integers(1, P))))
        elif k == 2:
            bits = pr.bits31(emb[int(rng.integers(0, len(emb)))], int(rng.integers(1, 31)))
            wires.append(pr.mux(bits[int(rng.integers(0, len(bits) - 1))], a, b))
        elif k == 3:
            wires.append(pr.pack(int(rng.integers(0, 4)), a, b, c, wires[int(rng.integers(0, len(wires)))]))
        elif k == 4:
            u = pr.unpack(a)
            emb.append(u[int(rng.integers(0, 4))])
            wires.append(u[0])
        elif k == 5:
            o = pr.p2([wires[int(rng.integers(0, len(wires)))] for _ in range(6)])
            wires.extend(o[:2])
        elif k == 6:
            wires.append(pr.is_zero(emb[int(rng.integers(0, len(emb)))]))
        elif k == 7:
            wires.append(pr.sub(a, b))
        else:
            wires.append(pr.add(pr.mul(a, a), pr.const(1, 0, 0, 1)))
    pub = [wires[-1 - i] for i in range(4)]
    pr.public(*pub)
    words = [int(x) for x in rng.integers(0, P, 4 * n_in)][:pr.n_inputs]      # exactly what the program reads
    return pr, words, pub
