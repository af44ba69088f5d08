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
def prove_segment_table_parallel(all_stark, config, trace_poly_values, table_in_use, public_values, group=None, ctx=None,
                                 timing=None, row_sharded=None):
    """`prove_with_traces` (prover.rs:72-194) with the tables of ONE segment spread over the ranks of `group`.

    What shards (prover.rs:90-111): every table's trace commitment is independent of the transcript, and a table's CTL /
    logUp columns and its whole `prove_single_table` only read that table's own trace and LDEs -- so table t lives on
    exactly one rank (`assign_tables`, largest first) and no bulk data ever moves.  What does not: Fiat-Shamir.
      phase 1  trace commitments of the owned tables, in parallel; the caps are observed in table order before anything
               else (prover.rs:118-127) -> ONE all-gather of 2^cap_height x 32 B per table (`gather_caps`); every rank
               then replays the same transcript (public values, CTL challenges);
      phase 2  CTL running sums, logUp helper columns and the AUXILIARY COMMITMENT of every owned table, in parallel on all
               ranks: they depend on the CTL challenges only (prover.rs:134-144; lookup challenges = the CTL betas, :328);
      phase 3  the per-table proofs in table order on their owners (prover.rs:251-259: serial by construction, the
               recursive verifier enforces the order), the 31-word challenger state broadcast owner -> all after each.
    Everything that crosses ranks is a fixed-shape int64 tensor (collectives.py): table shapes, a status word after every
    local step (a failing rank never strands the others in a collective: all of them raise), caps, challenger states,
    and at the end the flat proof words gathered on rank 0.

    row_sharded: {table: this rank's contiguous ROW BLOCK of that table's trace, CUDA (C, n / W)} -- tables whose commitment
    and proof are spread over ALL ranks (level 3, shard_prover.py: the 2431-column Keccak table is what bounds this mode
    otherwise); every rank passes its block, trace_poly_values[t] is ignored for them.  Their steps of the chain run on every
    rank at once on the replicated transcript, so no state is broadcast after them.

    trace_poly_values[t] is only read on the owner of t (others may pass None).  Returns the `AllProof` on rank 0 of the
    group (None elsewhere); bit-identical to the single-GPU `prove_with_traces`.  Latency, not throughput: the chain is
    serial, so the gain is bounded by (largest trace commitment + largest phase 2 + sum of the chain steps) / (single-GPU
    time) -- independent segments on independent GPUs (`scheduler.run_distributed`) remain the throughput path."""
    import time
    from itertools import groupby

    import torch
    import torch.distributed as dist

    from . import collectives as co
    from . import segment as sg
    from .challenger import Challenger
    from .context import default_context
    from .polynomial_batch import PolynomialBatch
    from .prover import CtlZData, StarkProof, table_aux_commit
    from .stark import _trace_args, ctl_partial_sums
    multi = dist.is_available() and dist.is_initialized()
    world, rank = (dist.get_world_size(group), dist.get_rank(group)) if multi else (1, 0)
    n_tab = all_stark.num_tables
    hasher = config.hasher
    fri = config.fri_config
    deg = all_stark.constraint_degree

    def step(what, fn):
        """run a local step; afterwards every rank knows whether all ranks succeeded"""
        err, val = None, None
        try:
            val = fn()
        except Exception as e:                      # noqa: BLE001 -- re-raised by agree() on this rank
            err = e
        co.agree(err, what, group)
        return val
    row_sharded = dict(row_sharded or {})
    # every rank needs the shapes to compute the same assignment: (cols, log_n) of the tables it was given, (0, 0) = absent
    mine_shapes = np.zeros((n_tab, 2), dtype=np.uint64)
    for t, tr in enumerate(trace_poly_values):
        if t in row_sharded:
            blk = row_sharded[t]
            mine_shapes[t] = (int(blk.shape[0]), (int(blk.shape[1]) * world).bit_length() - 1)
        elif tr is not None:
            c, n, ln, _ = _trace_args(tr)
            mine_shapes[t] = (c, ln)
    shapes = [None] * n_tab
    for part in co.all_gather_words(mine_shapes.reshape(-1), 2 * n_tab, group):
        for t, (c, ln) in enumerate(part.reshape(n_tab, 2)):
            if int(c):
                shapes[t] = (int(c), int(ln))

    def plan():
        if any(s is None for s in shapes):
            raise ValueError("every table's trace must be present on at least one rank")
        owner = [0] * n_tab
        solo = [t for t in range(n_tab) if t not in row_sharded]
        for r, ts in enumerate(assign_tables([shapes[t] for t in solo], world)):
            for k in ts:
                owner[solo[k]] = r
        mine = [t for t in solo if owner[t] == rank]
        missing = [t for t in mine if trace_poly_values[t] is None]
        if missing:
            raise ValueError("rank %d owns tables %s but was not given their traces" % (rank, missing))
        return owner, mine
    owner, mine = step("the table assignment", plan)
    dev0 = trace_poly_values[mine[0]].device if mine else (next(iter(row_sharded.values())).device if row_sharded else None)
    ctx = ctx or default_context((dev0.index or 0) if dev0 is not None else torch.cuda.current_device())
    t0 = time.perf_counter()
    # ---- phase 1: trace commitments of the owned tables, one all-gather of caps -----------------------------------
    batches = step("a trace commitment", lambda: {
        t: PolynomialBatch.from_values(trace_poly_values[t], fri.rate_bits, False, fri.cap_height, hasher=hasher, ctx=ctx)
        for t in mine})
    aux = {}
    wide = {}
    try:
        # the row-sharded tables: every rank takes part in each commitment (column-sharded NTT, all-to-all, sub-root all-gather)
        from .shard_prover import commit_rows_sharded, prove_table_row_sharded, table_ctl_specs
        for t in sorted(row_sharded):
            wide[t] = step("a row-sharded trace commitment", lambda t=t: commit_rows_sharded(row_sharded[t], config, ctx, group))
        local = {t: batches[t].merkle_tree.cap.elements for t in mine}
        if rank == 0:                                  # (every rank holds a row-sharded table's cap; one of them reports it)
            local.update({t: wide[t].cap for t in wide})
        caps = gather_caps(local, n_tab, 1 << fri.cap_height, group)
        t1 = time.perf_counter()
        # ---- transcript seed, replicated (prover.rs:114-144) ---------------------------------------------------------------
        ch = Challenger(hasher)
        for t in range(n_tab):
            if t in all_stark.optional_table_indices and not table_in_use[t]:
                ch.observe_elements([0] * (4 << fri.cap_height))
            else:
                ch.observe_cap(caps[t])
        step("the public values", lambda: sg.observe_public_values(ch, public_values))
        ctl_challenges = [(ch.get_challenge(), ch.get_challenge()) for _ in range(config.num_challenges)]
        # ---- phase 2, parallel over the ranks: CTL data, logUp columns and the auxiliary commitment of the owned tables ----
        zdata = {t: [] for t in mine}

        def phase2():
            for ctl in all_stark.cross_table_lookups:        # starky cross_table_lookup_data, restricted to this rank's tables
                looked = ctl.looked_table
                for beta, gamma in ctl_challenges:
                    for table, grp in groupby(ctl.looking_tables, key=lambda x: x.table):
                        entries = [(x.columns, x.filter) for x in grp]
                        if table in zdata and table_in_use[table]:
                            cols = ctl_partial_sums(trace_poly_values[table], entries, beta, gamma, deg, ctx=ctx)
                            allc = [(x.columns, x.filter) for x in ctl.looking_tables if x.table == table]
                            zdata[table].append(CtlZData(beta, gamma, allc, cols))
                    if looked.table in zdata and table_in_use[looked.table]:
                        z = ctl_partial_sums(trace_poly_values[looked.table], [(looked.columns, looked.filter)], beta, gamma, deg, ctx=ctx)
                        zdata[looked.table].append(CtlZData(beta, gamma, [(looked.columns, looked.filter)], z))
            for t in mine:
                if table_in_use[t]:
                    aux[t] = table_aux_commit(config, trace_poly_values[t], all_stark.lookups[t], zdata[t], ctl_challenges, deg,
                                              hasher=hasher, ctx=ctx)
        step("the auxiliary commitments (phase 2)", phase2)
        t2 = time.perf_counter()
        # ---- phase 3, the chain: tables in order on their owners, challenger state handed on (prover.rs:251-259) ----------
        proofs = {}
        for t in range(n_tab):
            if not table_in_use[t]:
                continue
            if t in wide:                              # all ranks together, on the replicated transcript
                pr = step("the row-sharded proof of table %d" % t, lambda t=t: prove_table_row_sharded(
                    all_stark.table_air[t], config, row_sharded[t], table_ctl_specs(all_stark, t, ctl_challenges), ctl_challenges, ch,
                    constraint_degree=deg, air_consts=all_stark.air_consts[t], lookups=all_stark.lookups[t], group=group, ctx=ctx,
                    trace_oracle=wide.pop(t)))
                if rank == 0:                          # (the sharded prover returns the proof on every rank)
                    proofs[t] = sg.StarkProofWithMetadata(pr, pr.init_challenger_state)
                continue
            state, err = np.zeros(32, dtype=np.uint64), None
            if owner[t] == rank:
                try:
                    proofs[t] = sg.prove_single_table(all_stark, t, config, trace_poly_values[t], batches[t], zdata[t],
                                                      ctl_challenges, ch, aux_commitment=aux.get(t))
                    state[1:] = ch.export_state()
                except Exception as e:              # noqa: BLE001 -- announced to the other ranks below, then re-raised
                    err = e
                    state[0] = 1
            if multi:
                state = co.broadcast_words(state, 32, owner[t], group)
            if err is not None:
                raise err
            if int(state[0]):
                raise co.RemoteRankError("the proof of table %d failed on rank %d" % (t, owner[t]))
            if owner[t] != rank:
                ch.import_state(state[1:])
            else:
                if aux.get(t) is not None:
                    aux.pop(t).free()
                if t not in (sg.Table.MemBefore, sg.Table.MemAfter):
                    batches[t].free()
        t3 = time.perf_counter()
    finally:
        for b in list(batches.values()) + [a for a in aux.values() if a is not None]:
            if b.handle:
                b.free()
        for o in wide.values():
            o.free()
    if timing is not None:
        timing.update({"compute owned trace commitments + cap all-gather": t1 - t0,
                       "CTL data + auxiliary commitments (owned tables, parallel over ranks)": t2 - t1,
                       "per-table proofs (serial chain over owners)": t3 - t2, "tables owned": mine})
    # ---- the proofs to rank 0: flat words, one padded gather ------------------------------------------------------------
    recs = []
    for t in sorted(proofs):
        recs.append(np.concatenate([np.array([t], dtype=np.uint64), proofs[t].proof.to_words()]))
    parts = co.gather_varlen_words(co.pack_records(recs), dst=0, group=group)
    if rank != 0:
        return None
    merged = {}
    for part in parts:
        for rec in co.unpack_records(part):
            pr, _ = StarkProof.from_words(rec[1:])
            merged[int(rec[0])] = sg.StarkProofWithMetadata(pr, pr.init_challenger_state)
    mb, ma = caps[sg.Table.MemBefore], caps[sg.Table.MemAfter].copy()
    if not table_in_use[sg.Table.MemAfter]:
        ma[:] = 0
    public_values.mem_before = sg.MemCap.from_merkle_cap(mb, hasher)
    public_values.mem_after = sg.MemCap.from_merkle_cap(ma, hasher)
    return sg.AllProof(sg.MultiProof([merged.get(t) for t in range(n_tab)], ctl_challenges), public_values, list(table_in_use))


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
