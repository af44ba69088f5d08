#!/usr/bin/env python3
"""Generate tests/golden/segment_proof.json: a *self-golden* digest of one tiny whole-segment proof per case
(all nine tables, real all_stark.rs CTL wiring), produced by the CPU oracle's restatement of prove_with_traces
(oracle/segment.py), which is pinned by the reference-tree KATs only at the hash / field level -- see DESIGN.md 2.
The CPU suite checks that the oracle still reproduces it, the GPU suite that zk_prove_segment does."""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import tests.oracle_lib as ol  # noqa: E402
from tests.test_gpu_segment import make_pv, make_traces  # noqa: E402

CASES = [("poseidon_all_tables", 0, [True] * 9, 2024),
         ("keccak_optional_tables_unused", 1, [True, False, True, False, False, False, True, True, False], 2025)]
KW = dict(pow_bits=3, queries=2)


def digest(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.uint64).tobytes()).hexdigest()


def case_inputs(hasher, in_use, seed):
    rng = np.random.default_rng(seed)
    traces = make_traces(rng)
    for t, used in enumerate(in_use):
        if not used:
            traces[t] = np.zeros((traces[t].shape[0], 16), dtype=np.uint64)
    return traces, make_pv(rng)


def summarize(ctl_challenges, tables, mem_before, mem_after):
    """tables: list of None | dict(init, aux_cap, quotient_cap, openings, fri)"""
    out = dict(ctl_challenges=[[int(b), int(g)] for b, g in ctl_challenges], tables=[],
               mem_before=digest(mem_before), mem_after=digest(mem_after))
    for t in tables:
        if t is None:
            out["tables"].append(None)
            continue
        out["tables"].append(dict(init=[int(x) for x in t["init"]],
                                  aux_cap=None if t["aux_cap"] is None else digest(t["aux_cap"]),
                                  quotient_cap=digest(t["quotient_cap"]), openings=digest(np.asarray(t["openings"]).reshape(-1)),
                                  n_openings=int(np.asarray(t["openings"]).size // 2), fri=digest(t["fri"]),
                                  fri_words=int(np.asarray(t["fri"]).size)))
    return out


def oracle_case(o, hasher, in_use, seed):
    from oracle import airs as oairs
    from oracle import segment as oseg
    traces, pvd = case_inputs(hasher, in_use, seed)
    cfg = ol.make_cfg(hasher=hasher, **KW)
    exp = oseg.prove_with_traces(o, ol, cfg, traces, in_use, pvd, oairs.CPU_TEST_CONSTS)
    tables = [None if p is None else dict(init=exp["init_states"][t], aux_cap=p["aux_cap"], quotient_cap=p["quotient_cap"],
                                          openings=p["openings"], fri=p["fri"]) for t, p in enumerate(exp["proofs"])]
    return summarize(exp["ctl_challenges"], tables, exp["mem_before"], exp["mem_after"])


def main():
    o = ol.load_oracle()
    ol.setup_fri_api(o)
    out = dict(kind="self-golden (oracle restatement of prove_with_traces; see DESIGN.md section 2)", cases=[])
    for name, hasher, in_use, seed in CASES:
        out["cases"].append(dict(name=name, hasher=hasher, in_use=in_use, seed=seed, proof=oracle_case(o, hasher, in_use, seed)))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "segment_proof.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
