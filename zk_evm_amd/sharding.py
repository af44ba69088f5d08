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
