"""Fixed-shape tensor collectives for the multi-GPU paths (scheduler.run_distributed, sharding): everything that crosses
ranks is an int64 tensor -- a status word, the 31-word challenger state, 16 x 4-word caps, flat proof words -- moved with
all_gather / all_reduce / broadcast / gather of `torch.distributed` (backend "nccl" = RCCL over xGMI on the GPU node, "gloo"
in the CPU tests).  No `*_object` collective: those pickle Python objects through device staging buffers.

Failure protocol: a rank never leaves its peers inside a collective.  Local work runs under `try`, the outcome travels as
a status word (`agree`) or inside the payload itself, every rank reaches every collective, and only then is the error
raised -- on every rank, naming the rank that failed."""
from typing import List, Optional, Sequence

import numpy as np


class RemoteRankError(RuntimeError):
    """Another rank of the group failed at this step."""


def _dist():
    import torch.distributed as dist
    return dist if (dist.is_available() and dist.is_initialized()) else None


def device_for(group=None):
    """Where collective payloads live: the current CUDA device under nccl (RCCL moves device memory), host otherwise."""
    import torch
    dist = _dist()
    if dist is not None and dist.get_backend(group) == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def _to_tensor(words: np.ndarray, n: Optional[int], dev):
    import torch
    a = np.ascontiguousarray(words, dtype=np.uint64).reshape(-1)
    t = torch.zeros(a.size if n is None else n, dtype=torch.int64)
    t[: a.size] = torch.from_numpy(a.view(np.int64))
    return t.to(dev)


def _to_words(t) -> np.ndarray:
    return t.detach().cpu().contiguous().numpy().view(np.uint64)


def agree(error: Optional[BaseException], what: str, group=None) -> None:
    """Status all-reduce (MAX over `1 + rank` of the failing ranks).  Every rank calls it after a local step; if any
    rank failed, every rank raises: the failing one its own exception, the others `RemoteRankError`."""
    dist = _dist()
    if dist is None:
        if error is not None:
            raise error
        return
    import torch
    rank = dist.get_rank(group)
    t = torch.tensor([0 if error is None else 1 + rank], dtype=torch.int64, device=device_for(group))
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    bad = int(t.item())
    if error is not None:
        raise error
    if bad:
        raise RemoteRankError("%s failed on rank %d" % (what, bad - 1))


def all_gather_words(words: np.ndarray, n: int, group=None) -> List[np.ndarray]:
    """All-gather of exactly `n` u64 words per rank."""
    dist = _dist()
    if dist is None:
        return [np.ascontiguousarray(words, dtype=np.uint64).reshape(-1)[:n].copy()]
    import torch
    t = _to_tensor(words, n, device_for(group))
    parts = [torch.empty_like(t) for _ in range(dist.get_world_size(group))]
    dist.all_gather(parts, t, group=group)
    return [_to_words(p) for p in parts]


def broadcast_words(words: Optional[np.ndarray], n: int, src: int, group=None) -> np.ndarray:
    """`n` u64 words from group rank `src` to every rank."""
    dist = _dist()
    if dist is None:
        return np.ascontiguousarray(words, dtype=np.uint64).reshape(-1)[:n].copy()
    t = _to_tensor(words if words is not None else np.zeros(0, np.uint64), n, device_for(group))
    dist.broadcast(t, src=dist.get_global_rank(group, src) if group is not None else src, group=group)
    return _to_words(t)


def gather_varlen_words(words: np.ndarray, dst: int = 0, group=None) -> Optional[List[np.ndarray]]:
    """Variable-length u64 payloads to group rank `dst`: one all-gather of the lengths, then ONE `gather` of tensors
    padded to the longest.  Returns the per-rank payloads on `dst`, None elsewhere."""
    dist = _dist()
    a = np.ascontiguousarray(words, dtype=np.uint64).reshape(-1)
    if dist is None:
        return [a.copy()]
    import torch
    lens = [int(x[0]) for x in all_gather_words(np.array([a.size], dtype=np.uint64), 1, group)]
    n = max(max(lens), 1)
    dev = device_for(group)
    t = _to_tensor(a, n, dev)
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    outs = [torch.empty(n, dtype=torch.int64, device=dev) for _ in range(world)] if rank == dst else None
    dist.gather(t, outs, dst=dist.get_global_rank(group, dst) if group is not None else dst, group=group)
    if rank != dst:
        return None
    return [_to_words(o)[:l].copy() for o, l in zip(outs, lens)]


def pack_records(records: Sequence[np.ndarray]) -> np.ndarray:
    """[n, len_0, .., len_{n-1}, payload_0, ...]"""
    recs = [np.ascontiguousarray(r, dtype=np.uint64).reshape(-1) for r in records]
    return np.concatenate([np.array([len(recs)] + [r.size for r in recs], dtype=np.uint64)] + recs)


def unpack_records(w: np.ndarray) -> List[np.ndarray]:
    w = np.asarray(w, dtype=np.uint64)
    n = int(w[0])
    lens = [int(x) for x in w[1: 1 + n]]
    out, pos = [], 1 + n
    for l in lens:
        out.append(w[pos: pos + l].copy())
        pos += l
    return out


def text_words(s: str, limit: int = 480) -> np.ndarray:
    b = s.encode("utf-8", "replace")[:limit]
    b += bytes(-len(b) % 8)
    return np.concatenate([np.array([len(s.encode("utf-8", "replace")[:limit])], dtype=np.uint64), np.frombuffer(b, dtype=np.uint64)])


def words_text(w: np.ndarray) -> str:
    n = int(w[0])
    return np.asarray(w[1:], dtype=np.uint64).tobytes()[:n].decode("utf-8", "replace")
