"""``zk_comm`` wrapper: the ranks of one multi-GPU job as the library sees them (include/zkstark.h, csrc/comm_host.inc).

The provers that cross GPUs -- ``zk_prove_table_sharded`` (SURVEY 8(e) level 3) and ``zk_prove_segment_table_parallel`` (level
2) -- live in the library and talk RCCL's C API (or the host-staged shared-memory transport) through a communicator handle.
This file only creates that handle for a Python caller: from explicit (rank, world) numbers, or from a ``torch.distributed``
process group, which is used for ONE thing -- handing rank 0's rendezvous token (the 128-byte RCCL unique id, or the name of the
shared region) to the other ranks.  No tensor ever travels through torch.distributed on these paths."""
import ctypes as C
import os
import threading
from typing import Optional

import numpy as np

from ._lib import ZkStarkError

ID_BYTES = 128
_counter = [0]
_lock = threading.Lock()


class Comm:
    """Owns a ``zk_comm``.  ``transport``: "rccl" (one rank per GPU) or "host" (shared memory; ranks may share a GPU)."""

    def __init__(self, ctx, handle, keep=None):
        self.ctx, self.handle, self._keep = ctx, handle, keep
        self.lib = ctx.lib

    # ---- construction ------------------------------------------------------------------------------------------------------
    @classmethod
    def host(cls, ctx, name: str, rank: int, world: int, slot_bytes: int = 0) -> "Comm":
        h = C.c_void_p()
        rc = ctx.lib.zk_comm_create_host(ctx.handle, name.encode(), rank, world, slot_bytes, C.byref(h))
        ctx.check(rc)
        return cls(ctx, h)

    @classmethod
    def rccl(cls, ctx, unique_id: bytes, rank: int, world: int) -> "Comm":
        assert len(unique_id) == ID_BYTES
        buf = (C.c_uint8 * ID_BYTES).from_buffer_copy(unique_id)
        h = C.c_void_p()
        rc = ctx.lib.zk_comm_create(ctx.handle, buf, rank, world, C.byref(h))
        ctx.check(rc)
        return cls(ctx, h)

    @staticmethod
    def unique_id(lib) -> bytes:
        buf = (C.c_uint8 * ID_BYTES)()
        rc = lib.zk_comm_unique_id(buf)
        if rc != 0:
            raise ZkStarkError(rc, "zk_comm_unique_id failed (RCCL not available)")
        return bytes(buf)

    @classmethod
    def single(cls, ctx) -> "Comm":
        """a communicator of ONE rank (no process group): the sharded provers run on it as they stand"""
        with _lock:
            _counter[0] += 1
            name = "zk_%d_%d_single" % (os.getpid(), _counter[0])
        return cls.host(ctx, name, 0, 1)

    @classmethod
    def from_group(cls, ctx, group=None, transport: Optional[str] = None) -> "Comm":
        """The communicator of a torch.distributed process group: RCCL when the group's backend is nccl (transport="host"
        forces the shared-memory transport), the host transport otherwise.  Collective over the group."""
        import torch
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return cls.single(ctx)
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        backend = dist.get_backend(group)
        if transport is None:
            transport = "rccl" if backend == "nccl" else "host"
        dev = torch.device("cuda", ctx.device) if backend == "nccl" else torch.device("cpu")
        src = dist.get_global_rank(group, 0) if group is not None else 0
        token = torch.zeros(ID_BYTES, dtype=torch.uint8)
        if rank == 0:
            if transport == "rccl":
                raw = cls.unique_id(ctx.lib)
            else:
                with _lock:
                    _counter[0] += 1
                    raw = ("zk_%d_%d_%s" % (os.getpid(), _counter[0], os.urandom(4).hex())).encode().ljust(ID_BYTES, b"\0")
            token = torch.from_numpy(np.frombuffer(raw, dtype=np.uint8).copy())
        token = token.to(dev)
        dist.broadcast(token, src=src, group=group)
        raw = bytes(token.cpu().numpy().tobytes())
        if transport == "rccl":
            return cls.rccl(ctx, raw, rank, world)
        return cls.host(ctx, raw.rstrip(b"\0").decode(), rank, world)

    # ---- queries -------------------------------------------------------------------------------------------------------------
    @property
    def rank(self) -> int:
        return int(self.lib.zk_comm_rank(self.handle))

    @property
    def world(self) -> int:
        return int(self.lib.zk_comm_world(self.handle))

    @property
    def transport(self) -> str:
        return self.lib.zk_comm_transport(self.handle).decode()

    def stats(self) -> dict:
        out = (C.c_uint64 * 3)()
        self.lib.zk_comm_stats(self.handle, out)
        return {"bytes_sent": int(out[0]), "bytes_received": int(out[1]), "collectives": int(out[2])}

    STAGES = ("column shards: all-to-all #1 + iNTT + LDE + pack", "all-to-all #2 to row shards",
              "row shards: leaf hashing + subtrees + cap all-gather", "auxiliary columns + carries", "quotient", "openings", "FRI")

    def timing_ms(self, reset: bool = False) -> dict:
        arr = (C.c_double * 8)()
        n = int(self.lib.zk_comm_last_timing(self.handle, arr, 8, 1 if reset else 0))
        return {k: float(arr[i]) for i, k in enumerate(self.STAGES[:n])}

    def barrier(self):
        self.ctx.check(self.lib.zk_comm_barrier(self.handle))

    # ---- the exported collectives (host payloads), mostly for tests ------------------------------------------------------------
    def all_gather_words(self, words: np.ndarray) -> np.ndarray:
        a = np.ascontiguousarray(words, dtype=np.uint64).reshape(-1)
        out = np.zeros((self.world, a.size), dtype=np.uint64)
        self.ctx.check(self.lib.zk_comm_all_gather_host(self.handle, a.ctypes.data, a.size * 8, out.ctypes.data))
        return out

    def broadcast_words(self, words: np.ndarray, root: int = 0) -> np.ndarray:
        a = np.ascontiguousarray(words, dtype=np.uint64).reshape(-1).copy()
        self.ctx.check(self.lib.zk_comm_broadcast_host(self.handle, a.ctypes.data, a.size * 8, root))
        return a

    def close(self):
        if getattr(self, "handle", None):
            self.lib.zk_comm_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_group_comms = {}


def comm_for(ctx, group=None, transport: Optional[str] = None) -> Comm:
    """the (cached) communicator of `group` on `ctx`: created collectively on first use"""
    key = (id(ctx), id(group) if group is not None else None, transport)
    c = _group_comms.get(key)
    if c is None or c.handle is None or c.ctx is not ctx or getattr(ctx, "handle", None) is None:
        c = _group_comms[key] = Comm.from_group(ctx, group, transport)
    return c


def drop_comms():
    """free every cached communicator (before the process group or the contexts go away)"""
    for c in list(_group_comms.values()):
        c.close()
    _group_comms.clear()
