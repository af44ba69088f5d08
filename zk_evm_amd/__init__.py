"""zk_evm_amd -- MI355X-native STARK commitment/proving backend for evm_arithmetization.

Only the hot path of the reference (0xPolygonZero/zk_evm) lives here: Goldilocks NTT/LDE,
Poseidon/Keccak Merkle-cap commitment and (in later stages) quotient/FRI, implemented as
hand-written HIP kernels for gfx950 in ``csrc/`` behind the C ABI of ``include/zkstark.h``.
The Python layer mirrors the reference-side interface of that path
(``PolynomialBatch::from_values`` etc.) over ctypes; it contains no arithmetic of its own and
has no CPU fallback: importing works anywhere, but every compute call requires the HIP library
and a GPU and raises ``ZkStarkError`` otherwise.
"""
from .config import FriConfig, StarkConfig  # noqa: F401
from ._lib import ZkStarkError, lib_path, load_library  # noqa: F401
from .context import Context, default_context  # noqa: F401
from .polynomial_batch import MerkleCap, MerkleProof, PolynomialBatch  # noqa: F401
from .challenger import Challenger  # noqa: F401
from .fri import (FriBatchInfo, FriInstanceInfo, fri_openings, prove_openings,  # noqa: F401
                  stark_fri_instance)

__all__ = [
    "FriConfig", "StarkConfig", "ZkStarkError", "Context", "default_context",
    "PolynomialBatch", "MerkleCap", "MerkleProof", "lib_path", "load_library", "Challenger",
    "FriBatchInfo", "FriInstanceInfo", "fri_openings", "prove_openings", "stark_fri_instance",
]
