"""`prove_single_table` with ONE table spread over the ranks of a process group -- SURVEY 8(e) level 3, the `north_star`'s
"RCCL all-gather over xGMI for FRI folding and Merkle-cap reduction".  Reference seam: the per-table commit loop
`evm_arithmetization/src/prover.rs:90-111` and `prove_single_table` `prover.rs:301-341` (starky `prove_with_commitment`).

What is sharded, and how (W = 2^k ranks, one per GPU; N = 2n LDE points):

  input        rank q holds the contiguous ROW BLOCK q of the trace, all C columns (witness generation is row-parallel:
               `zk_keccak_generate_trace` & co. write row blocks);
  aux columns  CTL helper / Z columns are row-wise sums over the block (`zk_ctl_partial_sums` on the block); a Z column is a
               reverse running sum, so block q adds the totals of the blocks after it: ONE all-gather of a word per column;
  commitment   all-to-all #1 row blocks -> COLUMN shards (NTTs are per column): iNTT + coset LDE; the coefficients stay on
               the column owner (openings).  all-to-all #2 column shards -> ROW shards: `zk_shard_pack_leaf_rows` writes the
               send buffers in LEAF order in one pass, so the receive buffers ARE the row shard: rank q = the leaves
               [q N/W, (q+1) N/W) = the natural rows j with j mod W = bitrev_W(q).  Leaf hashing + the local subtrees;
               all-gather of the 2^cap_height / W sub-roots per rank = the cap ("Merkle-cap reduction");
  quotient     needs rows j and j + 2: the next rows of a whole shard live on ONE other rank (residue + 2 mod W) -- a
               point-to-point exchange of the trace + auxiliary shards when W > 2, nothing when W <= 2.  Values on the local
               rows (`zk_quotient_values_sharded`), all-gather (16 B per point), then the 4-column chunk batch is committed
               on every rank alike (`zk_quotient_commit_values`): replicated, it is 4 columns wide;
  openings     each column owner evaluates its coefficient columns at zeta, g zeta (and 1 for the CTL Z columns); all-gather;
  FRI          batch combination on the local rows (`zk_fri_combine_sharded`: the pass that reads every LDE column),
               then either (fri="replicated") ONE all-gather of the combined polynomial (2 columns: 16 B per point) and the
               commit-phase trees, folds, final polynomial and proof of work on every rank alike (`zk_fri_prove_from_values`)
               or (fri="sharded") every layer on the rank that owns its leaves: local subtrees and one sub-root all-gather per
               round, folds on the local values (`_fri_sharded`).  The initial-tree opening of query x comes from the rank
               that owns leaf x (its row + the path inside its subtree); in the sharded form so do the round openings.

The proof equals the single-GPU `zk_prove_table` proof word for word (tests/test_gpu_multirank.py), for every table of the
AllStark: logUp lookups (forward running sums with carries from the blocks before), CTL looking runs with helper columns, and
columns that read the next row across a block boundary (Memory's range-check lookup, the Cpu table's CTL entries: the block's
last row is redone on a "seam" trace holding the next block's first row).  The Python side is orchestration only -- every
kernel is the library's."""
import ctypes as C
from typing import List, Optional, Sequence, Tuple

import numpy as np

from ._lib import ZkStarkError
from .sharding import _bitrev, split_columns

P = 0xFFFFFFFF00000001
NCCL_PIECE_BYTES = 256 << 20      # the largest piece handed to one RCCL send / recv (see all_to_all)


# ---- collectives on lists of device tensors (RCCL under nccl; host round trips under gloo) ------------------------------
def _dist(group):
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist, dist.get_world_size(group), dist.get_rank(group)
    return None, 1, 0


def all_to_all(send: List, recv: List, group=None) -> None:
    """recv[p] <- what rank p holds in its send[this rank] (the pieces may differ in size: column counts that W does not
    divide).  `dist.all_to_all` on device tensors under nccl (RCCL: every pair of GPUs on its own xGMI link); pairwise
    sends of host copies under gloo (the CPU-side tests)."""
    import torch
    dist, world, rank = _dist(group)
    if dist is None:
        recv[0].copy_(send[0])
        return
    if dist.get_backend(group) == "nccl":
        # In pieces of at most NCCL_PIECE_BYTES: RCCL (2.26, this image) returned CORRUPTED data, silently, for a send / recv of
        # more than 2^30 bytes (measured with one rank: 1.07 GB intact, 1.27 GB not -- tools/l3_one_rank_overhead.py found it
        # as a wrong cap), and the pieces of a wide table over few ranks are larger than that.  Every piece is cut along dim 0
        # (columns) into the SAME number of nearly equal parts -- agreed by one all-reduce -- so that no round has an empty
        # part unless a piece has fewer columns than there are rounds; sender and receiver of a piece see the same shape.
        rounds = 1
        for x in list(send) + list(recv):
            if x.numel():
                rounds = max(rounds, -(-(x.numel() * x.element_size()) // NCCL_PIECE_BYTES))
        if world > 1:
            r_t = torch.tensor([rounds], dtype=torch.int64, device=send[0].device)
            dist.all_reduce(r_t, op=dist.ReduceOp.MAX, group=group)
            rounds = int(r_t.item())
        if rounds == 1:
            dist.all_to_all(recv, send, group=group)
            return

        def cut(x, i):                         # part i of `rounds` nearly equal parts along dim 0: the same cut at both ends of a piece
            n = int(x.shape[0])
            return x[i * n // rounds: (i + 1) * n // rounds]
        for i in range(rounds):
            dist.all_to_all([cut(x, i) for x in recv], [cut(x, i) for x in send], group=group)
        return
    g = (lambda r: dist.get_global_rank(group, r)) if group is not None else (lambda r: r)
    recv[rank].copy_(send[rank])
    host_out = {p: torch.empty(recv[p].shape, dtype=recv[p].dtype) for p in range(world) if p != rank and recv[p].numel()}
    ops = [dist.P2POp(dist.isend, send[p].cpu().contiguous(), g(p), group) for p in range(world) if p != rank and send[p].numel()]
    ops += [dist.P2POp(dist.irecv, host_out[p], g(p), group) for p in host_out]
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    for p, t in host_out.items():
        recv[p].copy_(t)


def all_gather_tensor(t, group=None) -> List:
    """every rank's `t` (same shape everywhere), in rank order"""
    import torch
    dist, world, rank = _dist(group)
    if dist is None:
        return [t]
    if dist.get_backend(group) == "nccl":
        parts = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(parts, t.contiguous(), group=group)
        return parts
    host = t.cpu().contiguous()
    parts = [torch.empty_like(host) for _ in range(world)]
    dist.all_gather(parts, host, group=group)
    return [p.to(t.device) for p in parts]


def _p2p_in_pieces(dist, buf, out, dst_global: int, src_global: int, group, piece_bytes: int) -> None:
    """buf -> rank dst, out <- rank src, at most `piece_bytes` per send / recv (cut along dim 0; both ends hold one shape)"""
    rows = int(buf.shape[0]) if buf.dim() else 1
    per_row = max(1, buf.numel() // max(1, rows)) * buf.element_size()
    step = rows if not (buf.dim() and buf.numel()) else max(1, piece_bytes // per_row)
    for lo in range(0, max(1, rows), max(1, step)):
        sl = slice(lo, lo + step) if buf.dim() else Ellipsis
        ops = [dist.P2POp(dist.isend, buf[sl], dst_global, group), dist.P2POp(dist.irecv, out[sl], src_global, group)]
        for w in dist.batch_isend_irecv(ops):
            w.wait()


def exchange(send, dst: int, src: int, group=None):
    """point to point: this rank's `send` goes to group rank `dst`, the result comes from group rank `src`"""
    import torch
    dist, world, rank = _dist(group)
    if dist is None or (dst == rank and src == rank):
        return send
    nccl = dist.get_backend(group) == "nccl"
    out = torch.empty_like(send) if nccl else torch.empty(send.shape, dtype=send.dtype)
    buf = send.contiguous() if nccl else send.cpu().contiguous()
    g = (lambda r: dist.get_global_rank(group, r)) if group is not None else (lambda r: r)
    # (under nccl in pieces of at most NCCL_PIECE_BYTES: see all_to_all)
    _p2p_in_pieces(dist, buf, out, g(dst), g(src), group, NCCL_PIECE_BYTES if nccl else 1 << 62)
    return out if nccl else out.to(send.device)


# ---- one sharded oracle -------------------------------------------------------------------------------------------------
class ShardedOracle:
    """What a rank keeps of one committed matrix: the coefficient columns of its COLUMN shard, the leaf-ordered rows of
    its ROW shard with the local subtrees, and the whole cap.  `col_batch` / `row_batch` are library views over this
    object's tensors (zk_batch_from_parts)."""

    def __init__(self, ctx, cfg, n_cols, log_n, cols, coeffs, rows, digests, cap, lw, rank):
        import torch  # noqa: F401
        self.ctx, self.n_cols, self.log_n, self.cols = ctx, n_cols, log_n, cols
        self.coeffs, self.rows, self.digests, self.cap = coeffs, rows, digests, cap
        lib = ctx.lib
        capw = np.ascontiguousarray(cap, dtype=np.uint64).reshape(-1)
        self.col_batch = None
        if coeffs is not None and coeffs.shape[0]:
            h = C.c_void_p()
            ctx.check(lib.zk_batch_from_parts(ctx.handle, C.byref(cfg), coeffs.shape[0], log_n, C.c_void_p(coeffs.data_ptr()),
                                              None, None, capw.ctypes.data, 0, 0, C.byref(h)))
            self.col_batch = h
        h = C.c_void_p()
        ctx.check(lib.zk_batch_from_parts(ctx.handle, C.byref(cfg), n_cols, log_n, None, C.c_void_p(rows.data_ptr()),
                                          C.c_void_p(digests.data_ptr()) if digests is not None else None, capw.ctypes.data,
                                          lw, rank, C.byref(h)))
        self.row_batch = h

    def free(self):
        for h in (self.col_batch, self.row_batch):
            if h and getattr(self.ctx, "handle", None):
                self.ctx.lib.zk_batch_free(h)
        self.col_batch = self.row_batch = None


def commit_rows_sharded(block, config, ctx, group=None, timing=None) -> ShardedOracle:
    """`PolynomialBatch::from_values` of the matrix whose row block `rank` is `block` (K, n / W): steps "commitment" of the
    module docstring.  Returns this rank's ShardedOracle; its `cap` equals the single-GPU commitment's."""
    import time

    import torch

    from .collectives import all_gather_words
    dist, world, rank = _dist(group)
    ctx.use_torch_current_stream()         # the torch ops around the library calls (all_to_all, stack) must be ordered with them
    fri = config.fri_config
    lw = world.bit_length() - 1
    if world != 1 << lw or lw > fri.cap_height:
        raise ValueError("the number of ranks must be a power of two and at most 2^cap_height")
    K, nb = int(block.shape[0]), int(block.shape[1])
    n = nb * world
    log_n = n.bit_length() - 1
    if n != 1 << log_n or block.stride(1) != 1:
        raise ValueError("row blocks must be contiguous and a power-of-two fraction of the table")
    log_N = log_n + fri.rate_bits
    N, Nl = 1 << log_N, (1 << log_N) >> lw
    if Nl < 1 << (fri.cap_height - lw) or Nl < 2:
        raise ValueError("the table is too small for %d ranks" % world)
    dev = block.device
    lib = ctx.lib
    cfg = config.to_c()
    cols = split_columns(K, world)
    k_me = len(cols[rank])
    t0 = time.perf_counter()
    # all-to-all #1: row blocks -> column shards (the send pieces are contiguous slices of the column-major block)
    block = block.contiguous()
    send = [block[cols[p].start: cols[p].stop] for p in range(world)]
    recv = [torch.empty((k_me, nb), dtype=torch.int64, device=dev) for _ in range(world)]
    all_to_all(send, recv, group)
    values = torch.stack(recv, dim=1).reshape(k_me, n).contiguous()          # column c = its W row blocks in order
    del recv
    coeffs = torch.empty((k_me, n), dtype=torch.int64, device=dev)            # bit-reversed coefficient order (zk_batch layout)
    packed = torch.empty((world, k_me, Nl), dtype=torch.int64, device=dev)
    if k_me:
        lde = torch.empty((k_me, N), dtype=torch.int64, device=dev)
        ctx.check(lib.zk_shard_values_to_lde(ctx.handle, C.c_void_p(values.data_ptr()), k_me, log_n, fri.rate_bits,
                                             C.c_void_p(coeffs.data_ptr()), C.c_void_p(lde.data_ptr())))
        # all-to-all #2, send side: every destination's rows in leaf order, one pass over the LDE
        ctx.check(lib.zk_shard_pack_leaf_rows(ctx.handle, C.c_void_p(lde.data_ptr()), N, k_me, log_N, lw, C.c_void_p(packed.data_ptr())))
        torch.cuda.synchronize(dev)
        del lde
    del values
    t1 = time.perf_counter()
    rows = torch.empty((K, Nl), dtype=torch.int64, device=dev)
    all_to_all([packed[q] for q in range(world)], [rows[cols[p].start: cols[p].stop] for p in range(world)], group)
    del packed
    t2 = time.perf_counter()
    # leaf hashing + local subtrees; the sub-roots of all ranks, in rank order, are the cap
    cap_local_h = fri.cap_height - lw
    log_leaves = log_N - lw
    n_dig = int(lib.zk_merkle_num_digests(log_leaves, cap_local_h))
    dig = torch.zeros((n_dig, 4), dtype=torch.int64, device=dev)
    ctx.check(lib.zk_hash_rows(ctx.handle, config.hasher, C.c_void_p(rows.data_ptr()), Nl, K, Nl, C.c_void_p(dig.data_ptr())))
    ctx.check(lib.zk_merkle_build(ctx.handle, config.hasher, C.c_void_p(dig.data_ptr()), log_leaves, cap_local_h))
    sub = dig[n_dig - (1 << cap_local_h):].cpu().numpy().view(np.uint64).reshape(-1)
    cap = np.concatenate(all_gather_words(sub, sub.size, group)).reshape(1 << fri.cap_height, 4)
    if timing is not None:
        timing["column shards: all-to-all #1 + iNTT + LDE + pack"] = timing.get("column shards: all-to-all #1 + iNTT + LDE + pack", 0.0) + t1 - t0
        timing["all-to-all #2 to row shards"] = timing.get("all-to-all #2 to row shards", 0.0) + t2 - t1
        timing["row shards: leaf hashing + subtrees + cap all-gather"] = timing.get("row shards: leaf hashing + subtrees + cap all-gather", 0.0) + time.perf_counter() - t2
    return ShardedOracle(ctx, cfg, K, log_n, cols, coeffs, rows, dig, cap, lw, rank)


# ---- helpers -------------------------------------------------------------------------------------------------------------
def _entries_use_next_row(columns_filters) -> bool:
    def col_next(c):
        return bool(getattr(c, "next_row_linear_combination", None))
    for cols, filt in columns_filters:
        if any(col_next(c) for c in cols):
            return True
        if filt is not None:
            for a, b in getattr(filt, "products", []):
                if col_next(a) or col_next(b):
                    return True
            if any(col_next(c) for c in getattr(filt, "constants", [])):
                return True
    return False


def _ext_mul(a, b):
    return ((a[0] * b[0] + 7 * a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


_BITREV_IDX = {}


def _leaf_to_natural(t, log_N: int):
    """columns given in leaf (bit-reversed) order -> natural order"""
    import torch
    key = (log_N, str(t.device))
    idx = _BITREV_IDX.get(key)
    if idx is None:
        # on the device, one pass per bit (a host-built table took seconds at 2^21 points -- measured on one rank, where
        # nothing else hides it: tools/l3_one_rank_overhead.py)
        j = torch.arange(1 << log_N, dtype=torch.int64, device=t.device)
        idx = torch.zeros_like(j)
        for b in range(log_N):
            idx |= ((j >> b) & 1) << (log_N - 1 - b)
        if len(_BITREV_IDX) > 64:
            _BITREV_IDX.clear()
        _BITREV_IDX[key] = idx
    return t.index_select(1, idx).contiguous()


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


def _fri_sharded(ctx, cfg, config, comb, log_n: int, lw: int, rank: int, world: int, group, challenger, oracles, ocols, nw: int):
    """`fri_proof` ([EXT] fri/prover.rs: committed trees, final polynomial, proof of work, query rounds) with every layer kept on
    the rank that owns its leaves.  comb: (2, N / W) -- the batch combination at this rank's leaves (zk_fri_combine_sharded).
    Per round: the leaves are 2^arity_bits consecutive local values, the local subtrees are hashed here, ONE all-gather of their
    roots gives the round's cap ("Merkle-cap reduction"), beta comes from the replicated transcript, and the fold runs on the
    local VALUES (zk_fri_fold_values_sharded) -- leaf s of a layer is point bitrev(s) of the next, so the next layer is again
    this rank's contiguous run of leaves.  The last layer (2^5 .. 2^8 values) is all-gathered and interpolated for the final
    polynomial.  A query is answered entirely by the rank that owns its leaf: the same top bits select the rank in every layer.
    Returns the flat FriProof (layout: include/zkstark.h) on rank 0, None elsewhere; the transcript advances on every rank."""
    import torch

    from .collectives import all_gather_words, gather_varlen_words
    lib = ctx.lib
    fc = config.fri_config
    ab = int(cfg.arity_bits)
    arity = 1 << ab
    log_N = log_n + fc.rate_bits
    N = 1 << log_N
    ar = (C.c_uint32 * 32)()
    R = int(lib.zk_fri_reduction_arity_bits(C.byref(cfg), log_n, ar, 32))
    if R < 0 or R > 32:
        raise ZkStarkError(-1, "unsupported FRI configuration")
    dev = comb.device
    cur, lg, shift = comb.contiguous(), log_N, 14293326489335486720            # coset_shift()
    caps, rounds = [], []
    for r in range(R):
        if lg - lw < ab or lg - ab < fc.cap_height:
            raise ValueError("FRI layer %d (2^%d values) is too small to stay sharded over %d ranks: use fri='replicated'" % (r, lg, world))
        leaf_log = lg - ab - lw
        n_dig = int(lib.zk_merkle_num_digests(leaf_log, fc.cap_height - lw))
        dig = torch.zeros((n_dig, 4), dtype=torch.int64, device=dev)
        ctx.check(lib.zk_fri_commit_round_sharded(ctx.handle, C.byref(cfg), C.c_void_p(cur.data_ptr()), lg, lw, C.c_void_p(dig.data_ptr())))
        sub = dig[n_dig - (1 << (fc.cap_height - lw)):].cpu().numpy().view(np.uint64).reshape(-1)
        cap = np.concatenate(all_gather_words(sub, sub.size, group)).reshape(1 << fc.cap_height, 4)     # sub-roots in rank order
        challenger.observe_cap(cap)
        beta = np.array(challenger.get_extension_challenge(), dtype=np.uint64)
        nxt = torch.empty((2, cur.shape[1] >> ab), dtype=torch.int64, device=dev)
        ctx.check(lib.zk_fri_fold_values_sharded(ctx.handle, C.byref(cfg), C.c_void_p(cur.data_ptr()), lg, lw, rank, C.c_uint64(shift),
                                                 beta.ctypes.data, C.c_void_p(nxt.data_ptr())))
        caps.append(cap)
        rounds.append((cur.cpu().numpy().view(np.uint64), dig.cpu().numpy().view(np.uint64), lg))
        cur, lg = nxt, lg - ab
        shift = pow(shift, arity, P)
    # final polynomial: the last layer in full, natural order, coset iNTT -> natural coefficients; the first len >> rate_bits
    torch.cuda.synchronize(dev)
    last = _leaf_to_natural(torch.cat(all_gather_tensor(cur, group), dim=1), lg).contiguous()
    ctx.check(lib.zk_coset_ifft(ctx.handle, C.c_void_p(last.data_ptr()), 1 << lg, 2, lg, C.c_uint64(shift)))
    co = last.cpu().numpy().view(np.uint64) % np.uint64(P)
    flen = (1 << lg) >> fc.rate_bits
    final = np.stack([co[0, :flen], co[1, :flen]], axis=1).reshape(-1)
    challenger.observe_elements(final)
    wit = np.zeros(1, dtype=np.uint64)
    ctx.check(lib.zk_fri_proof_of_work(ctx.handle, C.byref(cfg), challenger.handle, wit.ctypes.data))
    Q = fc.num_query_rounds
    xs = np.array([challenger.get_challenge() % N for _ in range(Q)], dtype=np.uint64)
    # ---- query rounds: the owner of leaf x answers the whole round --------------------------------------------------------------
    per_init = sum(int(c) + 4 * (log_N - fc.cap_height) for c in ocols)
    init = np.zeros(Q * per_init, dtype=np.uint64)
    ctx.check(lib.zk_fri_initial_openings(ctx.handle, C.byref(cfg), (C.c_void_p * len(oracles))(*oracles), len(oracles),
                                          xs.ctypes.data, Q, init.ctypes.data))
    mine = []
    for q in range(Q):
        x = int(xs[q])
        if x >> (log_N - lw) != rank:
            continue
        rec = [np.array([q], dtype=np.uint64), init[q * per_init: (q + 1) * per_init]]
        for vals, dig, lgr in rounds:
            x >>= ab                                                   # the leaf of this round
            leaf_log = lgr - ab - lw
            slot = x & ((1 << leaf_log) - 1)
            ev = np.stack([vals[0, slot * arity: (slot + 1) * arity], vals[1, slot * arity: (slot + 1) * arity]], axis=1).reshape(-1)
            rec.append(ev)
            off, idx = 0, slot
            for lvl in range(leaf_log, fc.cap_height - lw, -1):        # siblings inside this rank's subtrees
                rec.append(dig[off + (idx ^ 1)])
                off += 1 << lvl
                idx >>= 1
        mine.append(np.concatenate([np.asarray(a, dtype=np.uint64).reshape(-1) for a in rec]))
    payload = np.concatenate(mine) if mine else np.zeros(0, dtype=np.uint64)
    parts = gather_varlen_words(payload, dst=0, group=group)
    if rank != 0:
        return None
    proof = np.zeros(nw, dtype=np.uint64)
    K = len(ocols)
    proof[:6] = [R, 1 << fc.cap_height, Q, K, flen, log_N]
    proof[6: 6 + R] = [ab] * R
    proof[6 + R: 6 + R + K] = ocols
    pos = 6 + R + K
    for cap in caps:
        proof[pos: pos + cap.size] = cap.reshape(-1)
        pos += cap.size
    proof[pos: pos + final.size] = final
    pos += final.size
    proof[pos] = wit[0]
    pos += 1
    query_words = (nw - pos) // Q if Q else 0
    for part in parts:
        for k in range(0, part.size, 1 + query_words):
            q = int(part[k])
            proof[pos + q * query_words: pos + (q + 1) * query_words] = part[k + 1: k + 1 + query_words]
    return proof


def prove_table_row_sharded(air_id: int, config, block, ctl_specs: Sequence[Tuple[int, int, list]], ctl_challenges, challenger,
                            constraint_degree: int = 3, air_consts: Sequence[int] = (), lookups=(), requires_ctls: bool = True,
                            group=None, ctx=None, timing: Optional[dict] = None, trace_oracle: "Optional[ShardedOracle]" = None,
                            fri: str = "replicated"):
    """`prove_single_table` (prover.rs:301-341) of ONE table over the ranks of `group` (module docstring).
    block: CUDA int64 (C, n / W), this rank's contiguous row block of the trace; `challenger`: the transcript, replicated,
    in the state the single-GPU call would receive it in (it is advanced identically on every rank).  `trace_oracle`: the
    trace commitment when the caller made it earlier (`commit_rows_sharded(block, ...)`: a segment commits every trace before
    the transcript starts, prover.rs:90-127); it is consumed here.  `fri`: "replicated" -- the combined polynomial is
    all-gathered once and every rank runs the (two-column) commit phase -- or "sharded" -- every FRI layer stays on the rank that
    owns its leaves: local trees, one sub-root all-gather per round, folds on values (`_fri_sharded`); the same proof either way.
    Returns the `StarkProof` on group rank 0, None elsewhere."""
    import time

    import torch

    from .collectives import all_gather_words, gather_varlen_words
    from .context import default_context
    from .fri import FriBatchInfo, FriInstanceInfo, stark_fri_instance
    from .polynomial_batch import PolynomialBatch
    from .prover import StarkProof, encode_ctl_set, encode_lookup_set, CtlZData
    from .stark import ctl_partial_sums, lookup_helper_columns
    lookups = list(lookups or [])
    dist, world, rank = _dist(group)
    dev = block.device
    ctx = ctx or default_context(dev.index or 0)
    ctx.use_torch_current_stream()
    lib = ctx.lib
    fri_cfg = config.fri_config
    nchal = config.num_challenges
    lw = world.bit_length() - 1
    C_tr, nb = int(block.shape[0]), int(block.shape[1])
    n = nb * world
    log_n = n.bit_length() - 1
    log_N = log_n + fri_cfg.rate_bits
    N, Nl = 1 << log_N, (1 << log_N) >> lw
    cfg = config.to_c()
    t_start = time.perf_counter()
    # ---- trace commitment ----------------------------------------------------------------------------------------------------
    trace = trace_oracle if trace_oracle is not None else commit_rows_sharded(block, config, ctx, group, timing)
    init_state = challenger.compact()                                  # "Clear buffered outputs." (prover.rs:320)
    # ---- auxiliary polynomials: CTL helper / Z columns on the row block, carries across blocks ------------------------------
    aux = None
    n_helpers_of, zdatas = [], []
    # logUp lookups (`lookup_helper_columns`; the lookup challenges are the CTL betas, prover.rs:328): the helper columns are
    # row-wise; a lookup's Z is a FORWARD running sum from 0 (z[i + 1] = z[i] + sum_h h[i] - freq[i] / (table[i] + alpha)), so a
    # block adds the totals of the blocks BEFORE it -- its own total needs the increment of its last row, evaluated here from
    # that row's values.  Order of the auxiliary polynomials: per lookup, per challenge, the helpers then Z; then the CTL columns.
    lookup_cols = []
    lookup_challenges = [b for b, _ in ctl_challenges] if lookups else []
    # Columns that read the NEXT row (Memory's range-check lookup, the Cpu table's CTL entries): the last row of a block needs
    # the first row of the block after it (the last block: row 0).  The builders run on the block and wrap around inside it, so
    # their values AT the block's last row are redone -- by the same kernels, on a 16-row "seam" trace whose row 0 is the
    # block's last row and whose other rows are the next block's first row (an all-gather of one row per rank).
    MINI = 16
    firsts = all_gather_words(block[:, 0].cpu().numpy().view(np.uint64), int(block.shape[0]), group)
    seam = torch.from_numpy(np.ascontiguousarray(firsts[(rank + 1) % world]).view(np.int64)).to(dev).reshape(-1, 1).repeat(1, MINI).contiguous()
    seam[:, 0] = block[:, nb - 1]
    # ... except in the LAST block: starky's `Column::eval_table` takes the next row's values to be 0 at the last row of the
    # trace ("If the lookups are correctly written, the filter should be 0 in that case anyway") -- which is what the builders
    # do at the last row of whatever they are given, so the last block is right as it stands
    use_seam = rank + 1 < world

    def word(x):
        return int(x.cpu().numpy().reshape(-1).view(np.uint64)[0]) % P

    if lookups:
        lz, lz_tot = [], []
        for lk in lookups:
            for alpha in lookup_challenges:
                cols = lookup_helper_columns(lk, block, alpha, constraint_degree, ctx=ctx)        # helpers ..., Z (from 0 in this block)
                mini = lookup_helper_columns(lk, seam, alpha, constraint_degree, ctx=ctx)
                if use_seam:
                    cols[:-1, nb - 1] = mini[:-1, 0]                   # the last row's helpers with the true next row
                # Z[i + 1] = Z[i] + (sum of the helpers - freq / (table + alpha))[i]: the seam's Z[1] is the last row's increment
                # (the last block's total is not used by anybody)
                lz.append(cols)
                lz_tot.append((word(cols[-1, nb - 1]) + word(mini[-1, 1])) % P)
        tots = all_gather_words(np.array(lz_tot, dtype=np.uint64), len(lz_tot), group)
        for k, cols in enumerate(lz):
            carry = sum(int(tots[q][k]) for q in range(rank)) % P          # the blocks before this one
            if carry:
                z = cols[-1:]
                add = np.array([carry], dtype=np.uint64)          # (bound to a name: `.ctypes.data` of a temporary dangles)
                ctx.check(lib.zk_gl_add_scalar_columns(ctx.handle, C.c_void_p(z.data_ptr()), nb, 1, nb, add.ctypes.data))
            lookup_cols.append(cols)
    if ctl_specs or lookup_cols:
        helpers, zs = [], []
        for beta, gamma, entries in ctl_specs:
            cols = ctl_partial_sums(block, entries, beta, gamma, constraint_degree, ctx=ctx)      # helpers (if any), then Z
            if use_seam and _entries_use_next_row(entries):
                # Z[i] = Z[i + 1] + term(i), Z[last] = term(last): the block's Z all contain its last row's term, computed
                # with the wrong next row -- replace it by the seam's (Z[0] - Z[1] there is term(row 0))
                mini = ctl_partial_sums(seam, entries, beta, gamma, constraint_degree, ctx=ctx)
                cols[:-1, nb - 1] = mini[:-1, 0]
                delta = (word(mini[-1, 0]) - word(mini[-1, 1]) - word(cols[-1, nb - 1])) % P
                if delta:
                    z = cols[-1:]
                    add = np.array([delta], dtype=np.uint64)
                    ctx.check(lib.zk_gl_add_scalar_columns(ctx.handle, C.c_void_p(z.data_ptr()), nb, 1, nb, add.ctypes.data))
            n_helpers_of.append(int(cols.shape[0]) - 1)
            helpers.append(cols[:-1])
            zs.append(cols[-1:])
            zdatas.append(CtlZData(beta, gamma, entries, cols))
        pieces = list(lookup_cols)
        if zs:
            zmat = torch.cat(zs, dim=0).contiguous()                   # (n_z, nb): reverse running sums WITHIN the block
            tot = zmat[:, 0].cpu().numpy().view(np.uint64)             # block totals
            parts = all_gather_words(tot, tot.size, group)
            carry = np.zeros(tot.size, dtype=np.uint64)
            for z in range(tot.size):                                  # the blocks after this one
                carry[z] = sum(int(parts[q][z]) % P for q in range(rank + 1, world)) % P
            if carry.any():
                ctx.check(lib.zk_gl_add_scalar_columns(ctx.handle, C.c_void_p(zmat.data_ptr()), nb, zmat.shape[0], nb, carry.ctypes.data))
            pieces += [h for h in helpers if h.shape[0]] + [zmat]
        # starky's order of the auxiliary polynomials: lookup columns, all CTL helper columns, all CTL Z columns
        aux_block = torch.cat(pieces, dim=0).contiguous()
        aux = commit_rows_sharded(aux_block, config, ctx, group, timing)
        challenger.observe_cap(aux.cap)
    n_aux = aux.n_cols if aux is not None else 0
    n_z = len(ctl_specs)
    t_commit = time.perf_counter()
    # ---- quotient -------------------------------------------------------------------------------------------------------------
    alphas = np.array(challenger.get_n_challenges(nchal), dtype=np.uint64)
    qd_bits = 1 if constraint_degree - 1 >= 2 else 0
    if fri_cfg.rate_bits != qd_bits:
        raise NotImplementedError("row-sharded quotient needs rate_bits == quotient_degree_bits")
    res = _bitrev(rank, lw)                                            # this rank's rows: natural j = i W + res
    nxt = _bitrev((res + (1 << qd_bits)) % world, lw) if world > 1 else 0     # the rank that holds rows j + 2^qd_bits
    prv = _bitrev((res - (1 << qd_bits)) % world, lw) if world > 1 else 0     # ... and the rank whose next rows are ours
    trace_next = exchange(trace.rows, prv, nxt, group) if nxt != rank else trace.rows
    aux_next = (exchange(aux.rows, prv, nxt, group) if nxt != rank else aux.rows) if aux is not None else None
    qloc = torch.empty((nchal, Nl), dtype=torch.int64, device=dev)
    ac = np.array(list(air_consts), dtype=np.uint64)
    # the ctl program carries the per-z-data helper counts; the column tensors themselves are not read (aux rows are)
    cp = encode_ctl_set(zdatas) if zdatas else None
    lp = encode_lookup_set(lookups) if lookups else None
    lch = np.array([a % (1 << 64) for a in lookup_challenges], dtype=np.uint64)
    ctx.check(lib.zk_quotient_values_sharded(
        ctx.handle, C.byref(cfg), air_id, ac.ctypes.data if ac.size else None, ac.size,
        C.c_void_p(trace.rows.data_ptr()), C.c_void_p(trace_next.data_ptr()), C_tr,
        C.c_void_p(aux.rows.data_ptr()) if aux is not None else None, C.c_void_p(aux_next.data_ptr()) if aux is not None else None,
        n_aux, log_n, lw, rank, alphas.ctypes.data, lp.ctypes.data if lp is not None else None, lp.size if lp is not None else 0,
        lch.ctypes.data if lch.size else None, lch.size,
        cp.ctypes.data if cp is not None else None, cp.size if cp is not None else 0, constraint_degree, C.c_void_p(qloc.data_ptr())))
    torch.cuda.synchronize(dev)
    del trace_next, aux_next
    q_leaf = torch.cat(all_gather_tensor(qloc, group), dim=1)          # (nchal, N), leaf order (rank q = leaves [q Nl, (q+1) Nl))
    q_nat = _leaf_to_natural(q_leaf, log_N)
    h = C.c_void_p()
    ctx.check(lib.zk_quotient_commit_values(ctx.handle, C.byref(cfg), C.c_void_p(q_nat.data_ptr()), log_n, constraint_degree, C.byref(h)))
    quotient = PolynomialBatch(ctx, h, fri_cfg.rate_bits, fri_cfg.cap_height, config.hasher)
    n_quot = quotient.num_polys
    quotient_cap = quotient.merkle_tree.cap.elements.copy()
    challenger.observe_cap(quotient_cap)
    t_quot = time.perf_counter()
    # ---- openings -------------------------------------------------------------------------------------------------------------
    zeta = challenger.get_extension_challenge()
    zp = (zeta[0] % P, zeta[1] % P)
    for _ in range(log_n):
        zp = _ext_mul(zp, zp)
    if zp == (1, 0):
        raise ZkStarkError(-1, "Opening point is in the subgroup.")
    g = pow(7277203076849721926, 1 << (32 - log_n), P)                 # primitive_root_of_unity(degree_bits)
    g_zeta = (zeta[0] % P * g % P, zeta[1] % P * g % P)
    ctl_batch = requires_ctls and n_z > 0
    instance = stark_fri_instance(zeta, g_zeta, C_tr, n_aux, n_quot, (n_aux - n_z, n_aux) if ctl_batch else None)

    def local_openings(oracle, points):
        """this rank's columns of `oracle` at `points` -> (len(points), K_me, 2) on every rank, gathered to (len(points), K, 2)"""
        k_me = len(oracle.cols[rank])
        kmax = max(len(c) for c in oracle.cols)
        mine = np.zeros((len(points), kmax, 2), dtype=np.uint64)
        if k_me:
            inst = FriInstanceInfo([FriBatchInfo(pt, [(0, i) for i in range(k_me)]) for pt in points])
            arr, keep = inst.to_c()
            out = np.zeros((len(points) * k_me, 2), dtype=np.uint64)
            ctx.check(lib.zk_fri_openings(ctx.handle, (C.c_void_p * 1)(oracle.col_batch), 1, arr, len(points), out.ctypes.data))
            mine[:, :k_me] = out.reshape(len(points), k_me, 2)
        parts = all_gather_words(mine.reshape(-1), mine.size, group)
        full = np.zeros((len(points), oracle.n_cols, 2), dtype=np.uint64)
        for p, part in enumerate(parts):
            cr = oracle.cols[p]
            full[:, cr.start: cr.stop] = part.reshape(len(points), kmax, 2)[:, : len(cr)]
        return full
    tr_op = local_openings(trace, [zeta, g_zeta])
    ax_op = local_openings(aux, [zeta, g_zeta, (1, 0)]) if aux is not None else None
    qinst = FriInstanceInfo([FriBatchInfo(zeta, [(0, i) for i in range(n_quot)])])
    arr, keep = qinst.to_c()
    q_op = np.zeros((n_quot, 2), dtype=np.uint64)
    ctx.check(lib.zk_fri_openings(ctx.handle, (C.c_void_p * 1)(quotient.handle), 1, arr, 1, q_op.ctypes.data))
    pieces = [tr_op[0]] + ([ax_op[0]] if aux is not None else []) + [q_op, tr_op[1]] + ([ax_op[1]] if aux is not None else [])
    if ctl_batch:
        pieces.append(ax_op[2][n_aux - n_z:])
    openings = np.concatenate(pieces, axis=0)
    assert openings.shape[0] == instance.n_openings
    challenger.observe_elements(openings.reshape(-1))                  # observe_openings
    t_open = time.perf_counter()
    # ---- FRI --------------------------------------------------------------------------------------------------------------------
    alpha = np.array(challenger.get_extension_challenge(), dtype=np.uint64)
    # the quotient's rows of this shard (leaf order) out of the replicated batch
    qpack = torch.empty((world, n_quot, Nl), dtype=torch.int64, device=dev)
    ctx.check(lib.zk_shard_pack_leaf_rows(ctx.handle, C.c_void_p(quotient.lde_device_ptr()), N, n_quot, log_N, lw, C.c_void_p(qpack.data_ptr())))
    q_rows = qpack[rank].contiguous()
    hq = C.c_void_p()
    ctx.check(lib.zk_batch_from_parts(ctx.handle, C.byref(cfg), n_quot, log_n, None, C.c_void_p(q_rows.data_ptr()), None,
                                      np.ascontiguousarray(quotient_cap, dtype=np.uint64).ctypes.data, lw, rank, C.byref(hq)))
    arr, keep = instance.to_c()
    shard_oracles = [trace.row_batch] + ([aux.row_batch] if aux is not None else []) + [hq]
    comb = torch.empty((2, Nl), dtype=torch.int64, device=dev)
    opn = np.ascontiguousarray(openings, dtype=np.uint64)
    ctx.check(lib.zk_fri_combine_sharded(ctx.handle, C.byref(cfg), (C.c_void_p * len(shard_oracles))(*shard_oracles), len(shard_oracles),
                                         arr, len(instance.batches), opn.ctypes.data, alpha.ctypes.data, C.c_void_p(comb.data_ptr())))
    lib.zk_batch_free(hq)
    layout_oracles = [trace.row_batch] + ([aux.row_batch] if aux is not None else []) + [quotient.handle]
    ocols = np.array([C_tr] + ([n_aux] if aux is not None else []) + [n_quot], dtype=np.uint64)
    nw = int(lib.zk_fri_proof_words(C.byref(cfg), log_n, ocols.ctypes.data, len(ocols)))
    if nw == 0:
        raise ZkStarkError(-1, "unsupported FRI configuration")
    if fri == "sharded":
        proof = _fri_sharded(ctx, cfg, config, comb, log_n, lw, rank, world, group, challenger, layout_oracles, ocols, nw)
    elif fri != "replicated":
        raise ValueError("fri must be 'replicated' or 'sharded'")
    else:
        vals = _leaf_to_natural(torch.cat(all_gather_tensor(comb, group), dim=1), log_N)     # (2, N), natural order
        proof = np.zeros(nw, dtype=np.uint64)
        xs = np.zeros(fri_cfg.num_query_rounds, dtype=np.uint64)
        ctx.check(lib.zk_fri_prove_from_values(ctx.handle, C.byref(cfg), (C.c_void_p * len(layout_oracles))(*layout_oracles), len(layout_oracles),
                                               arr, len(instance.batches), C.c_void_p(vals.data_ptr()), challenger.handle,
                                               proof.ctypes.data, xs.ctypes.data))
        # ---- the initial-tree openings of every query from the rank that owns its leaf -------------------------------------------
        R, cap_len, Q, K, F = (int(x) for x in proof[:5])
        off_queries = 6 + R + K + R * cap_len * 4 + 2 * F + 1
        query_words = (nw - off_queries) // Q if Q else 0
        n_shard = len(ocols) - 1                                       # trace (+ aux): the quotient is whole on every rank
        per_oracle = [int(c) + 4 * (log_N - fri_cfg.cap_height) for c in ocols]
        mine = []
        for q in range(Q):
            if int(xs[q]) >> (log_N - lw) == rank:
                base = off_queries + q * query_words
                mine.append(np.concatenate([np.array([q], dtype=np.uint64), proof[base: base + sum(per_oracle[:n_shard])]]))
        payload = np.concatenate(mine) if mine else np.zeros(0, dtype=np.uint64)
        parts = gather_varlen_words(payload, dst=0, group=group)
        if rank == 0:
            rec = 1 + sum(per_oracle[:n_shard])
            for part in parts:
                for k in range(0, part.size, rec):
                    q = int(part[k])
                    base = off_queries + q * query_words
                    proof[base: base + rec - 1] = part[k + 1: k + rec]
    trace_cap, aux_cap = trace.cap, (aux.cap if aux is not None else None)
    trace.free()
    if aux is not None:
        aux.free()
    quotient.free()
    if timing is not None:
        t_end = time.perf_counter()
        timing.update({"commitments (trace + auxiliary)": t_commit - t_start, "quotient": t_quot - t_commit,
                       "openings": t_open - t_quot, "FRI": t_end - t_open, "ranks": world, "rows per rank": Nl})
    if rank != 0:
        return None
    return StarkProof(trace_cap=np.asarray(trace_cap, dtype=np.uint64), auxiliary_polys_cap=None if aux_cap is None else np.asarray(aux_cap, dtype=np.uint64),
                      quotient_polys_cap=np.asarray(quotient_cap, dtype=np.uint64), openings=openings, opening_proof=proof,
                      init_challenger_state=init_state, num_ctl_zs=n_z, degree_bits=log_n)
