"""Binding of the per-table prover: the reference's ``prove_single_table`` -> starky
``prove_with_commitment`` (evm_arithmetization/src/prover.rs:301-341) is the C-ABI call ``zk_prove_table``
(sequencing compiled in csrc/segment_host.inc); this file marshals the `Lookup` / `CtlZData` descriptions into
the program encoding and copies the proof out.  ``quotient_polys`` exposes the K8/K9 step on its own."""
import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

from ._lib import ZkStarkError
from .challenger import Challenger
from .config import StarkConfig
from .polynomial_batch import PolynomialBatch
from .stark import Column, Filter, Lookup, encode_program

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

    # ---- flat words (what moves between ranks: a fixed header and u64 payloads, no pickling) ------------------------
    def to_words(self) -> np.ndarray:
        """header: degree_bits, num_ctl_zs, cap digests, has_aux_cap, n_openings, proof words, has_init_state; then
        trace cap, [aux cap], quotient cap, openings, opening proof, [12 init-state words]."""
        nd = int(np.asarray(self.trace_cap).reshape(-1, 4).shape[0])
        has_aux = self.auxiliary_polys_cap is not None
        has_st = self.init_challenger_state is not None
        op = np.asarray(self.openings, dtype=np.uint64).reshape(-1)
        fr = np.asarray(self.opening_proof, dtype=np.uint64).reshape(-1)
        head = np.array([self.degree_bits if self.degree_bits is not None else (1 << 63), self.num_ctl_zs, nd, int(has_aux),
                         op.size // 2, fr.size, int(has_st)], dtype=np.uint64)
        parts = [head, np.asarray(self.trace_cap, dtype=np.uint64).reshape(-1)]
        if has_aux:
            parts.append(np.asarray(self.auxiliary_polys_cap, dtype=np.uint64).reshape(-1))
        parts += [np.asarray(self.quotient_polys_cap, dtype=np.uint64).reshape(-1), op, fr]
        if has_st:
            parts.append(np.asarray(self.init_challenger_state, dtype=np.uint64).reshape(-1))
        return np.concatenate(parts)

    @staticmethod
    def from_words(w: np.ndarray) -> Tuple["StarkProof", int]:
        """-> (proof, words consumed)"""
        w = np.asarray(w, dtype=np.uint64)
        db, nz, nd, has_aux, n_op, n_fri, has_st = (int(x) for x in w[:7])
        pos = 7

        def take(n, shape=None):
            nonlocal pos
            a = w[pos: pos + n].copy()
            if a.size != n:
                raise ZkStarkError(-1, "truncated proof words")
            pos += n
            return a.reshape(shape) if shape else a
        tc = take(4 * nd, (nd, 4))
        ac = take(4 * nd, (nd, 4)) if has_aux else None
        qc = take(4 * nd, (nd, 4))
        op = take(2 * n_op, (n_op, 2))
        fr = take(n_fri)
        st = take(12) if has_st else None
        return StarkProof(tc, ac, qc, op, fr, st, nz, None if db == (1 << 63) else db), pos


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


class ZkTableProofView(C.Structure):
    """include/zkstark.h zk_table_proof_view"""
    _fields_ = [("degree_bits", C.c_uint), ("n_trace_cols", C.c_size_t), ("n_aux_cols", C.c_size_t),
                ("n_quotient_cols", C.c_size_t), ("n_ctl_zs", C.c_size_t), ("cap_digests", C.c_size_t),
                ("trace_cap", C.c_void_p), ("aux_cap", C.c_void_p), ("quotient_cap", C.c_void_p),
                ("openings", C.c_void_p), ("n_openings", C.c_size_t), ("opening_proof", C.c_void_p),
                ("proof_words", C.c_size_t), ("init_challenger_state", C.c_uint64 * 12)]


def _copy_words(ptr, n_words, shape=None) -> np.ndarray:
    a = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint64)), shape=(n_words,)).copy()
    return a.reshape(shape) if shape else a


def table_proof_from_handle(lib, handle) -> StarkProof:
    """Copy a library-owned zk_table_proof into a StarkProof (the caller still frees the handle)."""
    v = ZkTableProofView()
    rc = lib.zk_table_proof_get(handle, C.byref(v))
    if rc != 0:
        raise ZkStarkError(rc, "zk_table_proof_get failed")
    nd = v.cap_digests
    return StarkProof(
        trace_cap=_copy_words(v.trace_cap, 4 * nd, (nd, 4)),
        auxiliary_polys_cap=_copy_words(v.aux_cap, 4 * nd, (nd, 4)) if v.aux_cap else None,
        quotient_polys_cap=_copy_words(v.quotient_cap, 4 * nd, (nd, 4)),
        openings=_copy_words(v.openings, 2 * v.n_openings, (v.n_openings, 2)),
        opening_proof=_copy_words(v.opening_proof, v.proof_words),
        init_challenger_state=np.array(list(v.init_challenger_state), dtype=np.uint64),
        num_ctl_zs=int(v.n_ctl_zs), degree_bits=int(v.degree_bits))


def _table_args(trace_values, lookups, ctl_zdatas, ctl_challenges):
    """the marshalled pieces shared by zk_prove_table_with_aux and zk_table_aux_commit (kept alive by the caller)"""
    import torch
    from .stark import _trace_args
    n_cols, n, log_n, stride = _trace_args(trace_values)
    lp = encode_lookup_set(lookups)
    cp = encode_ctl_set(ctl_zdatas)
    ctl_cols = torch.cat([z.aux for z in ctl_zdatas], dim=0).contiguous() if ctl_zdatas else None
    cc = None
    if ctl_challenges is not None:
        cc = np.array([x % (1 << 64) for bg in ctl_challenges for x in bg], dtype=np.uint64)
    return n_cols, n, log_n, stride, lp, cp, ctl_cols, cc


def table_aux_commit(config: StarkConfig, trace_values, lookups: Sequence[Lookup], ctl_zdatas: Sequence[CtlZData],
                     ctl_challenges: Sequence[Tuple[int, int]], constraint_degree: int = 3, hasher: Optional[int] = None,
                     ctx=None) -> Optional[PolynomialBatch]:
    """Phase 2 of a table (zk_table_aux_commit): its auxiliary polynomials -- logUp helper columns under the CTL betas,
    the CTL helper / Z columns of `ctl_zdatas` -- committed.  Depends on the CTL challenges only (prover.rs:134-144,328),
    so every table's owner runs it at once before the serial chain (sharding.prove_segment_table_parallel).  None when
    the table has no auxiliary polynomials."""
    from .context import default_context
    if not lookups and not ctl_zdatas:
        return None
    ctx = ctx or default_context(trace_values.device.index or 0)
    ctx.use_torch_current_stream()
    n_cols, n, log_n, stride, lp, cp, ctl_cols, cc = _table_args(trace_values, lookups, ctl_zdatas, ctl_challenges)
    cfg = config.to_c()
    if hasher is not None:
        cfg.hasher = hasher
    h = C.c_void_p()
    rc = ctx.lib.zk_table_aux_commit(
        ctx.handle, C.byref(cfg), C.c_void_p(trace_values.data_ptr()), stride, n_cols, log_n,
        lp.ctypes.data if lp is not None else None, lp.size if lp is not None else 0,
        cp.ctypes.data if cp is not None else None, cp.size if cp is not None else 0,
        C.c_void_p(ctl_cols.data_ptr()) if ctl_cols is not None else None, n, cc.ctypes.data, constraint_degree, C.byref(h))
    ctx.check(rc)
    return PolynomialBatch(ctx, h, cfg.rate_bits, cfg.cap_height, cfg.hasher)


def prove_single_table(air_id: int, config: StarkConfig, trace_values, trace_commitment: PolynomialBatch,
                          lookups: Sequence[Lookup], ctl_zdatas: Sequence[CtlZData],
                          ctl_challenges: Optional[Sequence[Tuple[int, int]]], challenger: Challenger,
                          constraint_degree: int = 3, requires_ctls: bool = True,
                          air_consts: Sequence[int] = (), aux_commitment: Optional[PolynomialBatch] = None) -> StarkProof:
    """`prove_single_table` -> starky `prove_with_commitment` as ONE C-ABI call (zk_prove_table_with_aux; the sequencing
    lives in csrc/segment_host.inc).  The challenger is compacted first, exactly as prover.rs:318-320 does, and
    its state is returned in `init_challenger_state`.
    trace_values: CUDA tensor (n_cols, n) (the same values trace_commitment was built from).
    ctl_challenges: [(beta, gamma)] * num_challenges (lookup challenges = the betas, as starky does when
    `ctl_challenges` is Some; otherwise they are drawn from the challenger).
    aux_commitment: the result of `table_aux_commit` for this table, when phase 2 ran ahead of the chain."""
    ctx = trace_commitment.ctx
    ctx.use_torch_current_stream()
    n_cols, n, log_n, stride, lp, cp, ctl_cols, cc = _table_args(trace_values, lookups, ctl_zdatas, ctl_challenges)
    cfg = config.to_c(rate_bits=trace_commitment.rate_bits, cap_height=trace_commitment.cap_height)
    cfg.hasher = trace_commitment.hasher
    ac = np.array(list(air_consts), dtype=np.uint64)
    h = C.c_void_p()
    rc = ctx.lib.zk_prove_table_with_aux(
        ctx.handle, C.byref(cfg), air_id, ac.ctypes.data if ac.size else None, ac.size,
        C.c_void_p(trace_values.data_ptr()), stride, trace_commitment.handle,
        lp.ctypes.data if lp is not None else None, lp.size if lp is not None else 0,
        cp.ctypes.data if cp is not None else None, cp.size if cp is not None else 0,
        C.c_void_p(ctl_cols.data_ptr()) if ctl_cols is not None else None, n,
        cc.ctypes.data if cc is not None else None, constraint_degree, 1 if requires_ctls else 0,
        aux_commitment.handle if aux_commitment is not None else None, challenger.handle, C.byref(h))
    ctx.check(rc)
    try:
        return table_proof_from_handle(ctx.lib, h)
    finally:
        ctx.lib.zk_table_proof_free(h)
