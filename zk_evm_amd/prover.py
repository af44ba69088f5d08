"""Host mirror of starky's per-table prover (``starky::prover::prove_with_commitment``, [EXT]) as the
reference drives it from ``prove_single_table`` (evm_arithmetization/src/prover.rs:301-341):
lookup helper columns -> auxiliary commitment -> alphas -> quotient -> zeta -> openings -> FRI.
Every arithmetic step is a C-ABI call into the HIP library; this file is sequencing only (what the
Rust host keeps doing)."""
import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

from ._lib import ZkStarkError
from .challenger import Challenger
from .config import StarkConfig
from .fri import fri_openings, prove_openings, stark_fri_instance
from .polynomial_batch import PolynomialBatch
from .stark import Column, Filter, Lookup, ctl_partial_sums, encode_program, lookup_helper_columns

AIR_NONE, AIR_MEM_CONTINUATION, AIR_LOGIC, AIR_MEMORY, AIR_BYTE_PACKING, AIR_ARITHMETIC = 0, 1, 2, 3, 4, 5
AIR_KECCAK = 6
AIR_KECCAK_SPONGE = 7
AIR_CPU = 8
P = 0xFFFFFFFF00000001


@dataclass
class CtlZData:
    """starky ``CtlZData``: one (challenge, looking entries of this table) pair with its device
    columns: ``aux`` = helper columns (possibly none) followed by Z."""
    beta: int
    gamma: int
    columns_filters: List[Tuple[Sequence[Column], Filter]]
    aux: "object"  # CUDA tensor (n_helpers + 1, n)

    @property
    def n_helpers(self) -> int:
        return self.aux.shape[0] - 1


@dataclass
class StarkProof:
    """Flat mirror of starky ``StarkProof`` (+ ``init_challenger_state`` of StarkProofWithMetadata)."""
    trace_cap: np.ndarray
    auxiliary_polys_cap: Optional[np.ndarray]
    quotient_polys_cap: np.ndarray
    openings: np.ndarray          # (n_openings, 2): batches zeta, g*zeta, (1) in starky order
    opening_proof: np.ndarray     # flat FriProof (include/zkstark.h)
    init_challenger_state: Optional[np.ndarray] = None
    num_ctl_zs: int = 0
    degree_bits: Optional[int] = None


def encode_lookup_set(lookups: Sequence[Lookup]) -> Optional[np.ndarray]:
    if not lookups:
        return None
    subs = [encode_program([([c], f) for c, f in zip(l.columns, l.filter_columns)], l.table_column,
                           l.frequencies_column) for l in lookups]
    head = 1 + len(subs)
    offs, pos = [], head
    for s in subs:
        offs.append(pos)
        pos += s.size
    return np.concatenate([np.array([len(subs)] + offs, dtype=np.uint64)] + subs)


def encode_ctl_set(zdatas: Sequence[CtlZData]) -> Optional[np.ndarray]:
    if not zdatas:
        return None
    subs = []
    for z in zdatas:
        prog = encode_program(z.columns_filters)
        subs.append(np.concatenate([np.array([z.beta % P, z.gamma % P, z.n_helpers], dtype=np.uint64), prog]))
    head = 1 + len(subs)
    offs, pos = [], head
    for s in subs:
        offs.append(pos)
        pos += s.size
    return np.concatenate([np.array([len(subs)] + offs, dtype=np.uint64)] + subs)


def quotient_polys(air_id: int, config: StarkConfig, trace: PolynomialBatch, aux: Optional[PolynomialBatch],
                   alphas: Sequence[int], lookups: Sequence[Lookup], lookup_challenges: Sequence[int],
                   zdatas: Sequence[CtlZData], constraint_degree: int,
                   air_consts: Sequence[int] = ()) -> PolynomialBatch:
    ctx = trace.ctx
    cfg = config.to_c(rate_bits=trace.rate_bits, cap_height=trace.cap_height)
    cfg.hasher = trace.hasher
    lp = encode_lookup_set(lookups)
    cp = encode_ctl_set(zdatas)
    al = np.array([a % (1 << 64) for a in alphas], dtype=np.uint64)
    lc = np.array([c % (1 << 64) for c in lookup_challenges], dtype=np.uint64) if lookups else np.zeros(0, np.uint64)
    ac = np.array(list(air_consts), dtype=np.uint64)
    h = C.c_void_p()
    rc = ctx.lib.zk_quotient_polys(
        ctx.handle, C.byref(cfg), air_id, ac.ctypes.data if ac.size else None, ac.size, trace.handle,
        aux.handle if aux is not None else None, al.ctypes.data,
        lp.ctypes.data if lp is not None else None, lp.size if lp is not None else 0,
        lc.ctypes.data if lc.size else None, lc.size,
        cp.ctypes.data if cp is not None else None, cp.size if cp is not None else 0,
        constraint_degree, C.byref(h))
    ctx.check(rc)
    return PolynomialBatch(ctx, h, trace.rate_bits, trace.cap_height, trace.hasher)


def prove_with_commitment(air_id: int, config: StarkConfig, trace_values, trace_commitment: PolynomialBatch,
                          lookups: Sequence[Lookup], ctl_zdatas: Sequence[CtlZData],
                          ctl_challenges: Optional[Sequence[Tuple[int, int]]], challenger: Challenger,
                          constraint_degree: int = 3, requires_ctls: bool = True,
                          air_consts: Sequence[int] = ()) -> StarkProof:
    """trace_values: CUDA tensor (n_cols, n) (the same values trace_commitment was built from).
    ctl_challenges: [(beta, gamma)] * num_challenges (lookup challenges = the betas, as starky does
    when `ctl_challenges` is Some; otherwise drawn from the challenger)."""
    import torch
    nchal = config.num_challenges
    # ---- lookup helper columns --------------------------------------------------------------
    aux_parts = []
    lookup_challenges: List[int] = []
    if lookups:
        lookup_challenges = [b for b, _ in ctl_challenges] if ctl_challenges is not None \
            else challenger.get_n_challenges(nchal)
        for l in lookups:
            for ch in lookup_challenges:
                aux_parts.append(lookup_helper_columns(l, trace_values, ch, constraint_degree, ctx=trace_commitment.ctx))
    # get_ctl_auxiliary_polys: all helper polys of all z-data, then all Z polys
    if ctl_zdatas:
        for z in ctl_zdatas:
            if z.n_helpers:
                aux_parts.append(z.aux[:-1])
        for z in ctl_zdatas:
            aux_parts.append(z.aux[-1:])
    aux = None
    aux_cap = None
    if aux_parts:
        aux_values = torch.cat(aux_parts, dim=0).contiguous()
        aux = PolynomialBatch.from_values(aux_values, trace_commitment.rate_bits, False, trace_commitment.cap_height,
                                          hasher=trace_commitment.hasher, ctx=trace_commitment.ctx)
        aux_cap = aux.merkle_tree.cap.elements
        challenger.observe_cap(aux_cap)
    alphas = challenger.get_n_challenges(nchal)
    quotient = quotient_polys(air_id, config, trace_commitment, aux, alphas, lookups, lookup_challenges,
                              ctl_zdatas, constraint_degree, air_consts)
    q_cap = quotient.merkle_tree.cap.elements
    challenger.observe_cap(q_cap)
    zeta = challenger.get_extension_challenge()
    # g = primitive_root_of_unity(degree_bits)
    g = pow(7277203076849721926, 1 << (32 - trace_commitment.degree_log), P)
    # (zeta^n == 1 would leak witness data; starky bails out.)
    n_trace = trace_commitment.num_polys
    n_aux = aux.num_polys if aux is not None else 0
    n_ctl_zs = len(ctl_zdatas)
    ctl_range = (n_aux - n_ctl_zs, n_aux) if (requires_ctls and n_ctl_zs) else None
    g_zeta = (zeta[0] * g % P, zeta[1] * g % P)
    inst = stark_fri_instance(zeta, g_zeta, n_trace, n_aux, quotient.num_polys, ctl_zs_range=ctl_range)
    oracles = [trace_commitment] + ([aux] if aux is not None else []) + [quotient]
    openings = fri_openings(inst, oracles)
    challenger.observe_extension_elements(openings)
    proof = prove_openings(inst, oracles, challenger, config, openings)
    quotient.free()                    # release HBM now (the trace commitment belongs to the caller)
    if aux is not None:
        aux.free()
    return StarkProof(trace_cap=trace_commitment.merkle_tree.cap.elements, auxiliary_polys_cap=aux_cap,
                      quotient_polys_cap=q_cap, openings=openings, opening_proof=proof, num_ctl_zs=n_ctl_zs)
