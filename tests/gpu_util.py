"""Helpers for the -m gpu parity tests: device buffers via torch, calls through the C ABI."""
import ctypes as C

import numpy as np
import torch

P = 0xFFFFFFFF00000001


def to_dev(a: np.ndarray) -> torch.Tensor:
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return torch.from_numpy(a.view(np.int64)).cuda()


def to_host(t: torch.Tensor) -> np.ndarray:
    return t.cpu().numpy().view(np.uint64)


def ptr(t: torch.Tensor):
    return C.c_void_p(t.data_ptr())


def rand_u64(rng, shape, noncanonical_frac=0.05):
    """Random u64 with a sprinkling of non-canonical representatives and edge values."""
    a = rng.integers(0, 1 << 64, size=shape, dtype=np.uint64)
    flat = a.reshape(-1)
    k = max(1, int(flat.size * noncanonical_frac))
    idx = rng.integers(0, flat.size, size=k)
    edge = np.array([0, 1, P - 1, P, P + 1, (1 << 64) - 1, (1 << 32) - 1, 1 << 32,
                     0xFFFFFFFF00000000], dtype=np.uint64)
    flat[idx] = edge[rng.integers(0, edge.size, size=k)]
    return a
