"""`prove_single_table` with ONE table spread over the ranks of a job -- SURVEY 8(e) level 3, the `north_star`'s "RCCL all-gather
over xGMI for FRI folding and Merkle-cap reduction".  Reference seam: the per-table commit loop
`evm_arithmetization/src/prover.rs:90-111` and `prove_single_table` `prover.rs:301-341` (starky `prove_with_commitment`).

Since r05 the prover itself is in the library behind the C ABI (`zk_commit_rows_sharded`, `zk_prove_table_sharded`:
csrc/shard_prove_host.inc on a `zk_comm`, csrc/comm_host.inc -- RCCL's C API or the host-staged transport); what it shards and
how is described there and in include/zkstark.h.  This file marshals the table description into the program encoding, makes
the communicator of a torch.distributed group (comm.py) and copies the proof out: a thin call, as the Rust shim's would be."""
import ctypes as C
from typing import List, Optional, Sequence, Tuple

import numpy as np

from .comm import Comm, comm_for
from .sharding import _bitrev, split_columns  # noqa: F401  (re-exported: tests, tools)

P = 0xFFFFFFFF00000001
FRI_MODES = {"replicated": 0, "sharded": 1}


class ShardedOracle:
    """A `zk_sharded_batch`: what a rank keeps of one sharded commitment (its coefficient column shard, its leaf-ordered row
    shard with the local subtrees, the whole cap)."""

    def __init__(self, ctx, handle, n_cols: int, log_n: int, cap_height: int):
        self.ctx, self.handle, self.n_cols, self.log_n = ctx, handle, n_cols, log_n
        cap = np.zeros((1 << cap_height, 4), dtype=np.uint64)
        ctx.check(ctx.lib.zk_sharded_batch_cap(handle, cap.ctypes.data))
        self.cap = cap

    @property
    def row_batch(self):
        return C.c_void_p(self.ctx.lib.zk_sharded_batch_rows(self.handle))

    @property
    def col_batch(self):
        h = self.ctx.lib.zk_sharded_batch_columns(self.handle)
        return C.c_void_p(h) if h else None

    def free(self):
        if self.handle and getattr(self.ctx, "handle", None):
            self.ctx.lib.zk_sharded_batch_free(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _comm(ctx, group, comm) -> Comm:
    return comm if comm is not None else comm_for(ctx, group)


def _block_args(block) -> Tuple[int, int, int]:
    if block.dim() != 2 or block.stride(1) != 1:
        raise ValueError("a row block is a 2-D CUDA tensor (columns, rows) with contiguous rows")
    return int(block.shape[0]), int(block.shape[1]), int(block.stride(0)) if block.shape[0] > 1 else int(block.shape[1])


def commit_rows_sharded(block, config, ctx, group=None, timing: Optional[dict] = None, comm: Optional[Comm] = None) -> ShardedOracle:
    """`PolynomialBatch::from_values` of the matrix whose row block `rank` is `block` (K, n / W) -- zk_commit_rows_sharded.
    Returns this rank's ShardedOracle; its `cap` equals the single-GPU commitment's."""
    cm = _comm(ctx, group, comm)
    ctx.use_torch_current_stream()
    K, nb, stride = _block_args(block)
    n = nb * cm.world
    log_n = n.bit_length() - 1
    if n != 1 << log_n:
        raise ValueError("row blocks must be a power-of-two fraction of the table")
    cfg = config.to_c()
    h = C.c_void_p()
    cm.timing_ms(reset=True)
    ctx.check(ctx.lib.zk_commit_rows_sharded(ctx.handle, cm.handle, C.byref(cfg), C.c_void_p(block.data_ptr()), stride, K, log_n, C.byref(h)))
    if timing is not None:
        for k, v in cm.timing_ms().items():
            if v:
                timing[k] = timing.get(k, 0.0) + v / 1e3
    return ShardedOracle(ctx, h, K, log_n, config.fri_config.cap_height)


def table_ctl_specs(all_stark, table: int, ctl_challenges) -> List[Tuple[int, int, list]]:
    """The z-data of `table` in starky's order (`cross_table_lookup_data`): per CTL, per challenge, the run of this table's
    looking entries, then the looked entry -> [(beta, gamma, [(columns, filter)])]"""
    from itertools import groupby
    out = []
    for ctl in all_stark.cross_table_lookups:
        for beta, gamma in ctl_challenges:
            for t, grp in groupby(ctl.looking_tables, key=lambda x: x.table):
                if t == table:
                    out.append((beta, gamma, [(x.columns, x.filter) for x in grp]))
            if ctl.looked_table.table == table:
                out.append((beta, gamma, [(ctl.looked_table.columns, ctl.looked_table.filter)]))
    return out


def encode_ctl_specs(ctl_specs: Sequence[Tuple[int, int, list]], constraint_degree: int) -> Optional[np.ndarray]:
    """the `ctl_zdata` words of zk_prove_table_sharded: n, off[n], per z-data: beta, gamma, n_helpers, partial-sums program"""
    from .stark import encode_program
    if not ctl_specs:
        return None
    subs = []
    for beta, gamma, entries in ctl_specs:
        nh = -(-len(entries) // (constraint_degree - 1)) if len(entries) > 1 else 0
        subs.append(np.concatenate([np.array([beta % P, gamma % P, nh], dtype=np.uint64), encode_program(entries)]))
    offs, pos = [], 1 + len(subs)
    for s in subs:
        offs.append(pos)
        pos += s.size
    return np.concatenate([np.array([len(subs)] + offs, dtype=np.uint64)] + subs)


def prove_table_row_sharded(air_id: int, config, block, ctl_specs: Sequence[Tuple[int, int, list]], ctl_challenges, challenger,
                            constraint_degree: int = 3, air_consts: Sequence[int] = (), lookups=(), requires_ctls: bool = True,
                            group=None, ctx=None, timing: Optional[dict] = None, trace_oracle: "Optional[ShardedOracle]" = None,
                            fri: str = "replicated", comm: Optional[Comm] = None):
    """`prove_single_table` (prover.rs:301-341) of ONE table over the ranks of `group` / `comm` -- zk_prove_table_sharded.
    block: CUDA int64 (C, n / W), this rank's contiguous row block of the trace; `challenger`: the transcript, replicated, in
    the state the single-GPU call would receive it in (it is advanced identically on every rank).  `trace_oracle`: the trace
    commitment when the caller made it earlier (`commit_rows_sharded`); it is consumed here.  `fri`: "replicated" or "sharded"
    (include/zkstark.h).  Returns the `StarkProof` -- the single-GPU proof, word for word -- on EVERY rank."""
    from .context import default_context
    from .prover import encode_lookup_set, table_proof_from_handle
    if fri not in FRI_MODES:
        raise ValueError("fri must be 'replicated' or 'sharded'")
    ctx = ctx or default_context(block.device.index or 0)
    cm = _comm(ctx, group, comm)
    ctx.use_torch_current_stream()
    K, nb, stride = _block_args(block)
    n = nb * cm.world
    log_n = n.bit_length() - 1
    if n != 1 << log_n:
        raise ValueError("row blocks must be a power-of-two fraction of the table")
    lookups = list(lookups or [])
    cfg = config.to_c()
    lp = encode_lookup_set(lookups) if lookups else None
    cp = encode_ctl_specs(ctl_specs, constraint_degree)
    cc = np.array([x % (1 << 64) for bg in ctl_challenges for x in bg], dtype=np.uint64) if ctl_challenges is not None else None
    ac = np.array(list(air_consts), dtype=np.uint64)
    h = C.c_void_p()
    cm.timing_ms(reset=True)
    try:
        ctx.check(ctx.lib.zk_prove_table_sharded(
            ctx.handle, cm.handle, C.byref(cfg), air_id, ac.ctypes.data if ac.size else None, ac.size,
            C.c_void_p(block.data_ptr()), stride, K, log_n, trace_oracle.handle if trace_oracle is not None else None,
            lp.ctypes.data if lp is not None else None, lp.size if lp is not None else 0,
            cp.ctypes.data if cp is not None else None, cp.size if cp is not None else 0,
            cc.ctypes.data if cc is not None else None, constraint_degree, 1 if requires_ctls else 0, FRI_MODES[fri],
            challenger.handle, C.byref(h)))
    finally:
        if trace_oracle is not None:
            trace_oracle.free()
    if timing is not None:
        t = cm.timing_ms()
        timing.update({k: v / 1e3 for k, v in t.items()})
        timing.update({"ranks": cm.world, "rows per rank": (n << config.fri_config.rate_bits) // cm.world, "transport": cm.transport})
    try:
        return table_proof_from_handle(ctx.lib, h)
    finally:
        ctx.lib.zk_table_proof_free(h)
