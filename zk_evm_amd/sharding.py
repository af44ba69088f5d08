"""Multi-GPU work distribution for the STARK commit path.

Two levels shard naturally in the reference:
  * trace segments are fully independent (fresh Challenger per segment, prover.rs:118; the
    reference maps them onto workers at zero/src/prover.rs:221-224)  -> `assign_segments`;
  * inside one segment the per-table trace commitments are transcript-independent
    (prover.rs:90-111) until their caps are observed in fixed table order (prover.rs:118-127)
    -> `assign_tables` (largest-first onto the least-loaded rank) + `gather_caps`, the only
    exchange step: an all-gather of 2^cap_height x 32 bytes per table.
No bulk data ever moves between GPUs.  One process per GPU; `torch.distributed` backend "nccl"
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
    every rank (what `prove_with_traces` feeds to the Challenger, prover.rs:113-127)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        assert len(local_caps) == n_tables
        return [np.ascontiguousarray(local_caps[t], dtype=np.uint64) for t in range(n_tables)]
    world = dist.get_world_size(group)
    backend = dist.get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    # fixed-size payload: [owner_flag, cap words...] per table so a plain all_gather suffices
    buf = torch.zeros((n_tables, 1 + cap_len * 4), dtype=torch.int64)
    for t, cap in local_caps.items():
        buf[t, 0] = 1
        buf[t, 1:] = torch.from_numpy(np.ascontiguousarray(cap, dtype=np.uint64).view(np.int64).reshape(-1))
    buf = buf.to(dev)
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf, group=group)
    out: List[np.ndarray] = [None] * n_tables  # type: ignore
    for p in parts:
        p = p.cpu()
        for t in range(n_tables):
            if int(p[t, 0]) == 1:
                assert out[t] is None, f"table {t} committed by two ranks"
                out[t] = p[t, 1:].numpy().view(np.uint64).reshape(cap_len, 4).copy()
    missing = [t for t in range(n_tables) if out[t] is None]
    assert not missing, f"tables {missing} were committed by no rank"
    return out


# ---- table-parallel proof of ONE segment (latency mode, SURVEY 8(e) level 2) ------------------------------------------
def prove_segment_table_parallel(all_stark, config, trace_poly_values, table_in_use, public_values, group=None, ctx=None,
                                 timing=None):
    """`prove_with_traces` (prover.rs:72-194) with the tables of ONE segment spread over the ranks of `group`.

    What shards (prover.rs:90-111): every table's trace commitment is independent of the transcript, and a table's CTL /
    logUp columns and its whole `prove_single_table` only read that table's own trace and LDEs -- so table t lives on
    exactly one rank (`assign_tables`, largest first) and no bulk data ever moves.  What does not: Fiat-Shamir.  The trace
    caps are observed in table order before anything else (prover.rs:118-127) -> ONE all-gather of 2^cap_height x 32 B per
    table (`gather_caps`); from then on every rank replays the same transcript (public values, CTL challenges), and the
    per-table proofs run in table order on their owners with the 31-word challenger state broadcast from owner to all
    after each table (prover.rs:251-259: the chain is serial by construction, the recursive verifier enforces it).

    trace_poly_values[t] is only read on the owner of t (others may pass None).  Returns the `AllProof` on rank 0 of the
    group (None elsewhere); bit-identical to the single-GPU `prove_with_traces`.  Latency, not throughput: the chain is
    serial, so the gain is bounded by (largest trace commitment + sum of the per-table proofs) / (single-GPU time) --
    independent segments on independent GPUs (`scheduler.run_distributed`) remain the throughput path."""
    import time

    import torch
    import torch.distributed as dist

    from . import segment as sg
    from .challenger import Challenger
    from .context import default_context
    from .polynomial_batch import PolynomialBatch
    from .prover import CtlZData
    from .stark import _trace_args, ctl_partial_sums
    multi = dist.is_available() and dist.is_initialized()
    world, rank = (dist.get_world_size(group), dist.get_rank(group)) if multi else (1, 0)
    n_tab = all_stark.num_tables
    hasher = config.hasher
    fri = config.fri_config
    # every rank needs the shapes to compute the same assignment: exchange (cols, log_n) of the tables it was given
    shapes = [None] * n_tab
    for t, tr in enumerate(trace_poly_values):
        if tr is not None:
            c, n, ln, _ = _trace_args(tr)
            shapes[t] = (c, ln)
    if multi:
        allsh = [None] * world
        dist.all_gather_object(allsh, shapes, group=group)
        for sh in allsh:
            for t, s in enumerate(sh):
                if s is not None:
                    shapes[t] = s
    if any(s is None for s in shapes):
        raise ValueError("every table's trace must be present on at least one rank")
    owner = [0] * n_tab
    for r, ts in enumerate(assign_tables(shapes, world)):
        for t in ts:
            owner[t] = r
    mine = [t for t in range(n_tab) if owner[t] == rank]
    missing = [t for t in mine if trace_poly_values[t] is None]
    if missing:
        raise ValueError("rank %d owns tables %s but was not given their traces" % (rank, missing))
    dev0 = trace_poly_values[mine[0]].device if mine else None
    ctx = ctx or default_context((dev0.index or 0) if dev0 is not None else torch.cuda.current_device())
    t0 = time.perf_counter()
    # ---- phase 1: trace commitments of the owned tables, one all-gather of caps -----------------------------------
    batches = {t: PolynomialBatch.from_values(trace_poly_values[t], fri.rate_bits, False, fri.cap_height, hasher=hasher, ctx=ctx)
               for t in mine}
    caps = gather_caps({t: batches[t].merkle_tree.cap.elements for t in mine}, n_tab, 1 << fri.cap_height, group)
    t1 = time.perf_counter()
    # ---- transcript seed, replicated (prover.rs:114-144) ---------------------------------------------------------------
    ch = Challenger(hasher)
    for t in range(n_tab):
        if t in all_stark.optional_table_indices and not table_in_use[t]:
            ch.observe_elements([0] * (4 << fri.cap_height))
        else:
            ch.observe_cap(caps[t])
    sg.observe_public_values(ch, public_values)
    ctl_challenges = [(ch.get_challenge(), ch.get_challenge()) for _ in range(config.num_challenges)]
    # ---- CTL data of the owned tables (starky cross_table_lookup_data, restricted to this rank's tables) -------------
    from itertools import groupby
    zdata = {t: [] for t in mine}
    deg = all_stark.constraint_degree
    for ctl in all_stark.cross_table_lookups:
        looked = ctl.looked_table
        for beta, gamma in ctl_challenges:
            for table, grp in groupby(ctl.looking_tables, key=lambda x: x.table):
                entries = [(x.columns, x.filter) for x in grp]
                if table in zdata and table_in_use[table]:
                    aux = ctl_partial_sums(trace_poly_values[table], entries, beta, gamma, deg, ctx=ctx)
                    allc = [(x.columns, x.filter) for x in ctl.looking_tables if x.table == table]
                    zdata[table].append(CtlZData(beta, gamma, allc, aux))
            if looked.table in zdata and table_in_use[looked.table]:
                z = ctl_partial_sums(trace_poly_values[looked.table], [(looked.columns, looked.filter)], beta, gamma, deg, ctx=ctx)
                zdata[looked.table].append(CtlZData(beta, gamma, [(looked.columns, looked.filter)], z))
    t2 = time.perf_counter()
    # ---- the chain: tables in order on their owners, challenger state handed on (prover.rs:251-259) -------------------
    proofs = {}
    for t in range(n_tab):
        if not table_in_use[t]:
            continue
        if owner[t] == rank:
            p = sg.prove_single_table(all_stark, t, config, trace_poly_values[t], batches[t], zdata[t], ctl_challenges, ch)
            proofs[t] = p
            batches[t].free() if t not in (sg.Table.MemBefore, sg.Table.MemAfter) else None
        if multi:
            state = [ch.export_state() if owner[t] == rank else None]
            dist.broadcast_object_list(state, src=dist.get_global_rank(group, owner[t]) if group is not None else owner[t], group=group)
            if owner[t] != rank:
                ch.import_state(state[0])
    t3 = time.perf_counter()
    for t in list(batches):
        if batches[t].handle:
            batches[t].free()
    if timing is not None:
        timing.update({"compute owned trace commitments + cap all-gather": t1 - t0, "compute CTL data (owned tables)": t2 - t1,
                       "per-table proofs (serial chain over owners)": t3 - t2, "tables owned": mine})
    parts = [proofs]
    if multi:
        parts = [None] * world if rank == 0 else None
        dist.gather_object(proofs, parts, dst=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    if rank != 0:
        return None
    merged = {}
    for part in parts:
        merged.update(part)
    mb, ma = caps[sg.Table.MemBefore], caps[sg.Table.MemAfter].copy()
    if not table_in_use[sg.Table.MemAfter]:
        ma[:] = 0
    public_values.mem_before = sg.MemCap.from_merkle_cap(mb, hasher)
    public_values.mem_after = sg.MemCap.from_merkle_cap(ma, hasher)
    return sg.AllProof(sg.MultiProof([merged.get(t) for t in range(n_tab)], ctl_challenges), public_values, list(table_in_use))
