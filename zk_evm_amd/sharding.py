"""Multi-GPU work distribution for the STARK commit path.

Two levels shard naturally in the reference:
  * trace segments are fully independent (fresh Challenger per segment, prover.rs:118; the
    reference maps them onto workers at zero/src/prover.rs:221-224)  -> `assign_segments`;
  * inside one segment the per-table trace commitments are transcript-independent
    (prover.rs:90-111) until their caps are observed in fixed table order (prover.rs:118-127)
    -> `assign_tables` (largest-first onto the least-loaded rank) + `gather_caps`: an all-gather of
    2^cap_height x 32 bytes per table; after it only the 31-word challenger state travels, owner -> all, once per table.
No bulk data ever moves between GPUs, and nothing that moves is a pickled object (collectives.py).  One process per GPU; `torch.distributed` backend "nccl"
(RCCL) on the GPU box, "gloo" in the CPU tests.
"""
from typing import Dict, List, Sequence

import numpy as np


def assign_segments(n_segments: int, world_size: int) -> List[List[int]]:
    """Round-robin: rank r proves segments r, r + W, r + 2W, ..."""
    return [list(range(r, n_segments, world_size)) for r in range(world_size)]


def table_cost(n_cols: int, log_n: int) -> float:
    """Relative commit cost of a table: dominated by Poseidon leaf hashing, i.e. by
    ceil(cols/8) permutations per LDE row, plus the n log n NTT term."""
    n = float(1 << log_n)
    perms = 2.0 * n * ((n_cols + 7) // 8 + 1)
    ntt = 3.0 * n * n_cols * max(log_n, 1) / 64.0
    return perms + ntt


def assign_tables(shapes: Sequence[tuple], world_size: int) -> List[List[int]]:
    """shapes[t] = (n_cols, log_n).  Longest-processing-time-first bin packing.  Deterministic,
    so every rank computes the same assignment without communication."""
    order = sorted(range(len(shapes)), key=lambda t: (-table_cost(*shapes[t]), t))
    load = [0.0] * world_size
    out: List[List[int]] = [[] for _ in range(world_size)]
    for t in order:
        r = min(range(world_size), key=lambda k: (load[k], k))
        out[r].append(t)
        load[r] += table_cost(*shapes[t])
    for lst in out:
        lst.sort()
    return out


def gather_caps(local_caps: Dict[int, np.ndarray], n_tables: int, cap_len: int = 16,
                group=None) -> List[np.ndarray]:
    """All-gather the Merkle caps computed by each rank into table order.
    local_caps: {table_index: (cap_len, 4) uint64}.  Returns the list of all n_tables caps on
    every rank (what `prove_with_traces` feeds to the Challenger, prover.rs:113-127).
    One fixed-shape all-gather ([owner flag, cap words] per table; RCCL moves it from device memory under nccl)."""
    from .collectives import all_gather_words
    row = 1 + cap_len * 4
    buf = np.zeros((n_tables, row), dtype=np.uint64)
    for t, cap in local_caps.items():
        buf[t, 0] = 1
        buf[t, 1:] = np.ascontiguousarray(cap, dtype=np.uint64).reshape(-1)
    out: List[np.ndarray] = [None] * n_tables  # type: ignore
    for p in all_gather_words(buf.reshape(-1), n_tables * row, group):
        p = p.reshape(n_tables, row)
        for t in range(n_tables):
            if int(p[t, 0]) == 1:
                assert out[t] is None, f"table {t} committed by two ranks"
                out[t] = p[t, 1:].reshape(cap_len, 4).copy()
    missing = [t for t in range(n_tables) if out[t] is None]
    assert not missing, f"tables {missing} were committed by no rank"
    return out


# ---- table-parallel proof of ONE segment (latency mode, SURVEY 8(e) level 2) ------------------------------------------
def table_owners(shapes: Sequence[tuple], world_size: int, row_sharded: Sequence[int] = (), lib=None) -> List[int]:
    """`zk_assign_tables`: the rank that commits and proves each table (row-sharded tables: every rank; listed as 0)."""
    import ctypes as C
    if lib is None:
        from ._lib import load_library
        lib = load_library()
    n = len(shapes)
    cols = (C.c_size_t * n)(*[int(c) for c, _ in shapes])
    logs = (C.c_uint * n)(*[int(l) for _, l in shapes])
    wide = (C.c_uint8 * n)(*[1 if t in set(row_sharded) else 0 for t in range(n)])
    own = (C.c_uint32 * n)()
    rc = lib.zk_assign_tables(cols, logs, n, world_size, wide, own)
    if rc != 0:
        raise ValueError("zk_assign_tables failed (%d)" % rc)
    return [int(x) for x in own]


def prove_segment_table_parallel(all_stark, config, trace_poly_values, table_in_use, public_values, group=None, ctx=None,
                                 timing=None, row_sharded=None, comm=None, fri: str = "replicated"):
    """`prove_with_traces` (prover.rs:72-194) with the tables of ONE segment spread over the ranks of `group` / `comm` --
    `zk_prove_segment_table_parallel` (csrc/shard_prove_host.inc; what shards and what does not is described in
    include/zkstark.h): ONE library call per rank on a `zk_comm`.

    trace_poly_values[t] is only read on the owner of t (`table_owners`; others may pass None).
    row_sharded: {table: this rank's contiguous ROW BLOCK of that table's trace, CUDA (C, n / W)} -- tables whose commitment and
    proof are spread over ALL ranks (level 3); every rank passes its block, trace_poly_values[t] is ignored for them.
    Returns the `AllProof` on EVERY rank, bit-identical to the single-GPU `prove_with_traces`.  Latency, not throughput:
    independent segments on independent GPUs (`scheduler.run_distributed`) remain the throughput path."""
    import ctypes as C

    from . import segment as sg
    from .comm import comm_for
    from .context import default_context
    from .shard_prover import FRI_MODES
    from .stark import _trace_args
    n_tab = all_stark.num_tables
    row_sharded = dict(row_sharded or {})
    dev0 = next((tr.device for tr in list(trace_poly_values) + list(row_sharded.values()) if tr is not None), None)
    if ctx is None:
        import torch
        ctx = default_context((dev0.index or 0) if dev0 is not None else torch.cuda.current_device())
    cm = comm if comm is not None else comm_for(ctx, group)
    ctx.use_torch_current_stream()
    world, rank = cm.world, cm.rank
    # every rank needs every table's height: all-gather of what each rank was given (0 = absent)
    mine_logs = np.zeros(n_tab, dtype=np.uint64)
    for t in range(n_tab):
        if t in row_sharded:
            mine_logs[t] = 1 + (int(row_sharded[t].shape[1]) * world).bit_length() - 1
        elif trace_poly_values[t] is not None:
            mine_logs[t] = 1 + _trace_args(trace_poly_values[t])[2]
    logs = cm.all_gather_words(mine_logs).max(axis=0)
    if (logs == 0).any():
        raise ValueError("every table's trace must be present on at least one rank")
    shapes = [(all_stark.table_columns[t], int(logs[t]) - 1) for t in range(n_tab)]
    owner = table_owners(shapes, world, sorted(row_sharded), ctx.lib)
    mine = [t for t in range(n_tab) if t not in row_sharded and owner[t] == rank]
    held = [tr if (t in mine) else None for t, tr in enumerate(trace_poly_values)]
    tables, wiring, keep = sg.segment_tables(all_stark, held, table_in_use, shapes=[l for _, l in shapes], row_blocks=row_sharded)
    wide = (C.c_uint8 * n_tab)(*[1 if t in row_sharded else 0 for t in range(n_tab)])
    cfg = config.to_c()
    pv = np.array(sg.public_values_elements(public_values), dtype=np.uint64)
    h = C.c_void_p()
    cm.timing_ms(reset=True)
    ctx.check(ctx.lib.zk_prove_segment_table_parallel(
        ctx.handle, cm.handle, C.byref(cfg), C.cast(tables, C.c_void_p), n_tab, wide if row_sharded else None, wiring.ctypes.data, wiring.size,
        pv.ctypes.data, pv.size, all_stark.constraint_degree, sg.Table.MemBefore, sg.Table.MemAfter, FRI_MODES[fri], C.byref(h)))
    try:
        ms = (C.c_double * (2 + n_tab))()
        ctx.lib.zk_segment_proof_stage_ms(h, ms, 2 + n_tab)
        proof = sg.segment_proof_from_handle(ctx.lib, h, all_stark, config, table_in_use, public_values)
    finally:
        ctx.lib.zk_segment_proof_free(h)
    if timing is not None:
        timing.update({"compute owned trace commitments + cap all-gather": ms[0] / 1e3,
                       "CTL data + auxiliary commitments (owned tables, parallel over ranks)": ms[1] / 1e3,
                       "per-table proofs (serial chain over owners)": sum(ms[2:]) / 1e3, "tables owned": mine,
                       "transport": cm.transport})
        if row_sharded:
            timing["row-sharded stages (s)"] = {k: v / 1e3 for k, v in cm.timing_ms().items()}
    return proof


# ---- sharding INSIDE one table's commitment (SURVEY 8(e) level 3; prototype of the commit phase) -------------------------
def split_columns(n_cols: int, world: int) -> List[range]:
    """Contiguous column ranges, sizes differing by at most one; deterministic."""
    base, extra = divmod(n_cols, world)
    out, pos = [], 0
    for r in range(world):
        k = base + (1 if r < extra else 0)
        out.append(range(pos, pos + k))
        pos += k
    return out


def _bitrev(i: int, bits: int) -> int:
    r = 0
    for _ in range(bits):
        r = (r << 1) | (i & 1)
        i >>= 1
    return r
