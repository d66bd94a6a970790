"""zeth_amd — MI355X-native STARK seal path (risc0_zkp::hal::Hal drop-in) for Zeth's block-validity proofs.

Only the hot path lives here: csrc/ (HIP kernels + C-ABI + segment-prover driver) and the thin host-side
mirrors of the reference interfaces (hal.py, prover.py, host.py).  See DESIGN.md.
"""
__version__ = "0.1.0"
