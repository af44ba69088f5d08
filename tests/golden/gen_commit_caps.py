#!/usr/bin/env python3
"""Generate tests/golden/commit_caps.json: *self-golden* Merkle caps of seeded matrices, produced
by the CPU oracle (which is itself pinned by the reference-tree KATs, tests/test_oracle_kat.py).
The reference (Rust + un-vendored plonky2) cannot run in this environment, so these are labelled
self-golden, not reference-golden."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.oracle_lib import load_oracle, splitmix64  # noqa: E402

CASES = [
    # name, n_cols, log_n, rate_bits, cap_height, hasher, seed
    ("membefore_2^7_poseidon", 12, 7, 1, 4, 0, 1),
    ("memory_2^10_poseidon", 30, 10, 1, 4, 0, 2),
    ("arithmetic_2^12_poseidon", 116, 12, 1, 4, 0, 3),
    ("arithmetic_2^10_keccak", 116, 10, 1, 4, 1, 4),
    ("cpu_2^10_poseidon", 85, 10, 1, 4, 0, 5),
    ("tiny_noop_leaves", 3, 5, 1, 2, 0, 6),
    ("rate3_2^8_poseidon", 9, 8, 3, 4, 0, 7),
]


def main():
    o = load_oracle()
    out = []
    for name, n_cols, log_n, rb, ch, hasher, seed in CASES:
        vals = np.stack([splitmix64(seed + c, 1 << log_n) for c in range(n_cols)])
        r = o.commit_values(vals, rate_bits=rb, cap_height=ch, hasher=hasher, want_leaves=False)
        out.append(dict(name=name, n_cols=n_cols, log_n=log_n, rate_bits=rb, cap_height=ch,
                        hasher=hasher, seed=seed,
                        cap=[[int(x) for x in row] for row in r["cap"]]))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "commit_caps.json")
    json.dump(dict(kind="self-golden (oracle pinned by reference KATs)", cases=out),
              open(path, "w"), indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
