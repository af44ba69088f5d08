"""Host mirror of plonky2 ``PolynomialBatch`` / ``MerkleTree`` for the commitment path.

Reference interface being mirrored ([EXT] plonky2 1.0.0 ``fri/oracle.rs``, ``hash/merkle_tree.rs``)
as the reference calls it: ``PolynomialBatch::from_values(values, rate_bits, false, cap_height,
timing, None)`` at evm_arithmetization/src/prover.rs:100-107 and verifier.rs:68-77, then
``.merkle_tree.cap`` (prover.rs:113-116).  Same argument meaning and error behaviour
(mismatched column lengths / non-power-of-two / cap_height too large are rejected).
All arithmetic happens in the HIP library; this file only marshals pointers.
"""
import ctypes as C
import weakref
from dataclasses import dataclass
from typing import List, Sequence

import numpy as np

from ._lib import ZkStarkError
from .config import HASH_POSEIDON, StarkConfig, ZkCfg
from .context import Context, default_context


@dataclass
class MerkleCap:
    """``MerkleCap<F, H>``: 2^cap_height digests.  ``elements`` is (2^h, 4) uint64 (32-byte
    slots; Keccak-25 digests use the first 25 bytes)."""
    elements: np.ndarray

    def __len__(self):
        return self.elements.shape[0]

    def height(self) -> int:
        return int(self.elements.shape[0]).bit_length() - 1

    def flatten(self) -> np.ndarray:
        return self.elements.reshape(-1)


@dataclass
class MerkleProof:
    """``MerkleProof<F, H>``: sibling digests from the leaf level up to (excluding) the cap."""
    siblings: np.ndarray  # (log_leaves - cap_height, 4) uint64


def _is_torch(x) -> bool:
    return type(x).__module__.split(".")[0] == "torch"


class _MerkleTreeView:
    """Read-only view with the ``MerkleTree`` accessors the prover uses."""

    def __init__(self, batch: "PolynomialBatch"):
        # weak: batch.merkle_tree -> view -> batch would be a cycle, and a cycle defers the release of
        # tens of GB of HBM to the cyclic collector
        self._ref = weakref.ref(batch)

    @property
    def _b(self) -> "PolynomialBatch":
        b = self._ref()
        if b is None or not b.handle:
            raise ZkStarkError(-1, "the PolynomialBatch behind this MerkleTree was freed")
        return b

    @property
    def cap(self) -> MerkleCap:
        b = self._b
        out = np.zeros((1 << b.cap_height, 4), dtype=np.uint64)
        b.ctx.check(b.ctx.lib.zk_batch_cap(b.handle, out.ctypes.data))
        return MerkleCap(out)

    def get(self, leaf_index: int) -> np.ndarray:
        b = self._b
        out = np.zeros(b.num_polys, dtype=np.uint64)
        b.ctx.check(b.ctx.lib.zk_batch_leaf(b.handle, leaf_index, out.ctypes.data))
        return out

    def prove(self, leaf_index: int) -> MerkleProof:
        b = self._b
        k = b.degree_log + b.rate_bits - b.cap_height
        out = np.zeros((max(k, 0), 4), dtype=np.uint64)
        buf = out if out.size else np.zeros((1, 4), dtype=np.uint64)
        b.ctx.check(b.ctx.lib.zk_batch_merkle_path(b.handle, leaf_index, buf.ctypes.data))
        return MerkleProof(out)

    def __len__(self):
        return 1 << (self._b.degree_log + self._b.rate_bits)


class PolynomialBatch:
    """Device-resident batch of committed polynomials (coefficients, LDE, Merkle tree)."""

    def __init__(self, ctx: Context, handle, rate_bits: int, cap_height: int, hasher: int):
        self.ctx = ctx
        self.handle = handle
        self.rate_bits = rate_bits
        self.cap_height = cap_height
        self.hasher = hasher
        self.blinding = False
        lib = ctx.lib
        self.degree_log = int(lib.zk_batch_log_n(handle))
        self.num_polys = int(lib.zk_batch_num_cols(handle))
        self.merkle_tree = _MerkleTreeView(self)

    # ---- constructors ------------------------------------------------------------------
    @staticmethod
    def _cfg(rate_bits, cap_height, hasher, config) -> ZkCfg:
        config = config or StarkConfig(hasher=hasher)
        c = config.to_c(rate_bits=rate_bits, cap_height=cap_height)
        c.hasher = hasher
        return c

    @classmethod
    def from_values(cls, values, rate_bits: int, blinding: bool, cap_height: int, *,
                    hasher: int = HASH_POSEIDON, ctx: Context = None,
                    config: StarkConfig = None) -> "PolynomialBatch":
        """values: a sequence of equal-length 1-D uint64 columns (numpy, host), a 2-D numpy array
        (n_cols, n), or a 2-D CUDA torch tensor (n_cols, n) of dtype int64/uint64 (bit pattern)."""
        return cls._build(values, rate_bits, blinding, cap_height, hasher, ctx, config, True)

    @classmethod
    def from_coeffs(cls, coeffs, rate_bits: int, blinding: bool, cap_height: int, *,
                    hasher: int = HASH_POSEIDON, ctx: Context = None,
                    config: StarkConfig = None) -> "PolynomialBatch":
        return cls._build(coeffs, rate_bits, blinding, cap_height, hasher, ctx, config, False)

    @classmethod
    def _build(cls, data, rate_bits, blinding, cap_height, hasher, ctx, config, is_values):
        if blinding:
            # the reference never blinds on this path (prover.rs:103 passes `false`)
            raise ZkStarkError(-5, "blinding=true is not supported on this path")
        h = C.c_void_p()
        cfg = cls._cfg(rate_bits, cap_height, hasher, config)
        if _is_torch(data):
            import torch
            if data.dim() != 2 or not data.is_cuda:
                raise ZkStarkError(-1, "device input must be a 2-D CUDA tensor (n_cols, n)")
            if data.dtype not in (torch.int64, torch.uint64):
                raise ZkStarkError(-1, "device input must be int64/uint64")
            if data.stride(1) != 1:
                raise ZkStarkError(-1, "columns must be contiguous")
            n_cols, n = data.shape
            log_n = cls._log2(n)
            ctx = ctx or default_context(data.device.index or 0)
            ctx.use_torch_current_stream()
            fn = ctx.lib.zk_commit_columns_device if is_values else ctx.lib.zk_commit_coeffs_device
            rc = fn(ctx.handle, C.byref(cfg), C.c_void_p(data.data_ptr()), data.stride(0) if n_cols > 1 else n,
                    n_cols, log_n, C.byref(h))
            ctx.check(rc)
            return cls(ctx, h, rate_bits, cap_height, hasher)
        cols = cls._host_columns(data)
        n_cols = len(cols)
        if n_cols == 0:
            raise ZkStarkError(-1, "empty batch")
        n = cols[0].shape[0]
        if any(c.shape[0] != n for c in cols):
            raise ZkStarkError(-1, "all polynomials must have the same length")
        log_n = cls._log2(n)
        ctx = ctx or default_context(0)
        if is_values:
            ptrs = (C.c_void_p * n_cols)(*[c.ctypes.data for c in cols])
            rc = ctx.lib.zk_commit_columns(ctx.handle, C.byref(cfg), ptrs, n_cols, log_n, C.byref(h))
            ctx.check(rc)
            return cls(ctx, h, rate_bits, cap_height, hasher)
        import torch
        host = np.stack(cols)
        dev = torch.from_numpy(host.view(np.int64)).to(f"cuda:{ctx.device}")
        return cls._build(dev, rate_bits, blinding, cap_height, hasher, ctx, config, False)

    @staticmethod
    def _log2(n: int) -> int:
        if n <= 0 or n & (n - 1):
            raise ZkStarkError(-1, f"polynomial length {n} is not a power of two")
        return n.bit_length() - 1

    @staticmethod
    def _host_columns(data) -> List[np.ndarray]:
        if isinstance(data, np.ndarray) and data.ndim == 2:
            data = [data[i] for i in range(data.shape[0])]
        cols = []
        for c in data:
            a = np.ascontiguousarray(c, dtype=np.uint64)
            if a.ndim != 1:
                raise ZkStarkError(-1, "each polynomial must be 1-D")
            cols.append(a)
        return cols

    # ---- accessors ---------------------------------------------------------------------
    def polynomial_coeffs(self, col: int) -> np.ndarray:
        """``self.polynomials[col].coeffs`` (natural order)."""
        out = np.zeros(1 << self.degree_log, dtype=np.uint64)
        self.ctx.check(self.ctx.lib.zk_batch_coeffs(self.handle, col, out.ctypes.data))
        return out

    def get_lde_values(self, index: int, step: int = 1) -> np.ndarray:
        """``PolynomialBatch::get_lde_values(index, step)`` = leaves[bitrev(index * step)]."""
        out = np.zeros(self.num_polys, dtype=np.uint64)
        self.ctx.check(self.ctx.lib.zk_batch_lde_values(self.handle, index, step, out.ctypes.data))
        return out

    def lde_device_ptr(self) -> int:
        return int(self.ctx.lib.zk_batch_lde_device(self.handle) or 0)

    def digests_device_ptr(self) -> int:
        return int(self.ctx.lib.zk_batch_digests_device(self.handle) or 0)

    def free(self):
        # zk_ctx_destroy frees any batch still alive on that ctx, so never touch a handle whose
        # context is already closed (finaliser order during garbage collection is arbitrary)
        if self.handle and getattr(self.ctx, "handle", None):
            self.ctx.lib.zk_batch_free(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
