"""Host mirror of plonky2 ``Challenger<F, H>`` ([EXT] iop/challenger.rs) over the C ABI
(``zk_challenger_*``).  Reference use: evm_arithmetization/src/prover.rs:118-127 (observe trace
caps), get_challenges.rs:11-227 (observe public values), prover.rs:320 (``compact``)."""
import ctypes as C
from typing import List, Sequence

import numpy as np

from ._lib import ZkStarkError, load_library
from .config import HASH_POSEIDON


class Challenger:
    def __init__(self, hasher: int = HASH_POSEIDON, _handle=None):
        self.lib = load_library()
        self.hasher = hasher
        if _handle is not None:
            self.handle = _handle
            return
        h = C.c_void_p()
        rc = self.lib.zk_challenger_create(hasher, C.byref(h))
        if rc != 0:
            raise ZkStarkError(rc, "zk_challenger_create failed")
        self.handle = h

    def clone(self) -> "Challenger":
        h = C.c_void_p()
        rc = self.lib.zk_challenger_clone(self.handle, C.byref(h))
        if rc != 0:
            raise ZkStarkError(rc, "zk_challenger_clone failed")
        return Challenger(self.hasher, _handle=h)

    def observe_element(self, e: int):
        self.observe_elements([e])

    def observe_elements(self, elements: Sequence[int]):
        a = np.ascontiguousarray(elements, dtype=np.uint64).reshape(-1)
        if a.size:
            self.lib.zk_challenger_observe_elements(self.handle, a.ctypes.data, a.size)

    def observe_extension_elements(self, elements):
        self.observe_elements(np.ascontiguousarray(elements, dtype=np.uint64).reshape(-1))

    def observe_cap(self, cap):
        """cap: MerkleCap or (n, 4) uint64 array of 32-byte digest slots."""
        a = np.ascontiguousarray(getattr(cap, "elements", cap), dtype=np.uint64).reshape(-1, 4)
        self.lib.zk_challenger_observe_cap(self.handle, a.ctypes.data, a.shape[0])

    def get_challenge(self) -> int:
        return int(self.lib.zk_challenger_get_challenge(self.handle))

    def get_n_challenges(self, n: int) -> List[int]:
        return [self.get_challenge() for _ in range(n)]

    def get_extension_challenge(self):
        out = np.zeros(2, dtype=np.uint64)
        self.lib.zk_challenger_get_extension_challenge(self.handle, out.ctypes.data)
        return int(out[0]), int(out[1])

    def compact(self) -> np.ndarray:
        out = np.zeros(12, dtype=np.uint64)
        self.lib.zk_challenger_compact(self.handle, out.ctypes.data)
        return out

    def export_state(self) -> np.ndarray:
        """the whole transcript state (31 words, include/zkstark.h zk_challenger_export)"""
        out = np.zeros(31, dtype=np.uint64)
        rc = self.lib.zk_challenger_export(self.handle, out.ctypes.data)
        if rc != 0:
            raise ZkStarkError(rc, "zk_challenger_export failed")
        return out

    def import_state(self, words) -> None:
        a = np.ascontiguousarray(words, dtype=np.uint64).reshape(-1)
        if a.size != 31 or self.lib.zk_challenger_import(self.handle, a.ctypes.data) != 0:
            raise ZkStarkError(-1, "bad challenger state")
        self.hasher = int(a[30])

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.zk_challenger_free(self.handle)
                self.handle = None
        except Exception:
            pass
