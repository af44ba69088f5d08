"""Host mirror of the FRI opening interface: ``FriInstanceInfo`` / ``FriBatchInfo`` /
``PolynomialBatch::prove_openings`` ([EXT] plonky2 fri/structure.rs, fri/oracle.rs) and starky's
``Stark::fri_instance`` shape.  Reached in the reference via ``prove_with_commitment``
(evm_arithmetization/src/prover.rs:322).  Marshalling only; all arithmetic is in the HIP library."""
import ctypes as C
from dataclasses import dataclass
from typing import List, Sequence, Tuple

import numpy as np

from ._lib import ZkStarkError
from .challenger import Challenger
from .config import StarkConfig, ZkCfg
from .polynomial_batch import PolynomialBatch


class ZkFriBatch(C.Structure):
    _fields_ = [("point", C.c_uint64 * 2), ("n_polys", C.c_size_t), ("oracle_idx", C.c_void_p),
                ("poly_idx", C.c_void_p)]


@dataclass
class FriBatchInfo:
    point: Tuple[int, int]                      # element of F_{p^2}
    polynomials: List[Tuple[int, int]]          # (oracle_index, polynomial_index)


@dataclass
class FriInstanceInfo:
    batches: List[FriBatchInfo]

    def to_c(self):
        arr = (ZkFriBatch * len(self.batches))()
        keep = []
        for i, b in enumerate(self.batches):
            oi = np.array([p[0] for p in b.polynomials], dtype=np.uint32)
            pi = np.array([p[1] for p in b.polynomials], dtype=np.uint32)
            keep += [oi, pi]
            arr[i].point[0], arr[i].point[1] = int(b.point[0]), int(b.point[1])
            arr[i].n_polys = len(b.polynomials)
            arr[i].oracle_idx = oi.ctypes.data
            arr[i].poly_idx = pi.ctypes.data
        return arr, keep

    @property
    def n_openings(self) -> int:
        return sum(len(b.polynomials) for b in self.batches)


def stark_fri_instance(zeta, g_zeta, num_trace: int, num_aux: int, num_quotient: int,
                       ctl_zs_range=None) -> FriInstanceInfo:
    """starky ``Stark::fri_instance``: oracles = [trace, auxiliary (if any), quotient]; batches at
    zeta (all polys), g*zeta (trace + aux) and 1 (the CTL Z columns) when the table has CTLs."""
    trace = [(0, i) for i in range(num_trace)]
    aux = [(1, i) for i in range(num_aux)]
    qo = 2 if num_aux else 1
    quot = [(qo, i) for i in range(num_quotient)]
    batches = [FriBatchInfo(zeta, trace + aux + quot), FriBatchInfo(g_zeta, trace + aux)]
    if ctl_zs_range is not None:
        batches.append(FriBatchInfo((1, 0), [(1, i) for i in range(*ctl_zs_range)]))
    return FriInstanceInfo(batches)


def _oracle_array(oracles: Sequence[PolynomialBatch]):
    return (C.c_void_p * len(oracles))(*[o.handle for o in oracles])


def fri_openings(instance: FriInstanceInfo, oracles: Sequence[PolynomialBatch]) -> np.ndarray:
    """Evaluate every (oracle, poly) of every batch at the batch point -> (n_openings, 2) uint64."""
    ctx = oracles[0].ctx
    arr, keep = instance.to_c()
    out = np.zeros((instance.n_openings, 2), dtype=np.uint64)
    ctx.check(ctx.lib.zk_fri_openings(ctx.handle, _oracle_array(oracles), len(oracles), arr,
                                      len(instance.batches), out.ctypes.data))
    return out


def prove_openings(instance: FriInstanceInfo, oracles: Sequence[PolynomialBatch],
                   challenger: Challenger, config: StarkConfig, openings: np.ndarray) -> np.ndarray:
    """``PolynomialBatch::prove_openings``.  Returns the flat FriProof (layout: include/zkstark.h)."""
    ctx = oracles[0].ctx
    o0 = oracles[0]
    cfg = config.to_c(rate_bits=o0.rate_bits, cap_height=o0.cap_height)
    cfg.hasher = o0.hasher
    cols = np.array([o.num_polys for o in oracles], dtype=np.uint64)
    nw = ctx.lib.zk_fri_proof_words(C.byref(cfg), o0.degree_log, cols.ctypes.data, len(oracles))
    if nw == 0:
        raise ZkStarkError(-1, "unsupported FRI configuration")
    proof = np.zeros(nw, dtype=np.uint64)
    arr, keep = instance.to_c()
    opn = np.ascontiguousarray(openings, dtype=np.uint64)
    ctx.check(ctx.lib.zk_fri_prove_openings(ctx.handle, C.byref(cfg), _oracle_array(oracles),
                                            len(oracles), arr, len(instance.batches), opn.ctypes.data,
                                            challenger.handle, proof.ctypes.data))
    return proof
