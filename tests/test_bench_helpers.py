"""CPU: the pieces of the bench / multi-GPU tooling that need no GPU -- the block-shaped job list of BASELINE configs[3] and
the z-data order the row-sharded prover hands to the library."""
import numpy as np

from tools.benchlib import B19807080_RANGES, block_segment_shapes


def test_block_segment_shapes_follow_the_reference_ranges():
    shapes = block_segment_shapes(40, seed=3)
    assert shapes == block_segment_shapes(40, seed=3) and shapes != block_segment_shapes(40, seed=4)
    some_absent = set()
    for i, (log_ns, in_use) in enumerate(shapes):
        assert len(log_ns) == 9 and len(in_use) == 9
        for t, (lo, hi) in enumerate(B19807080_RANGES):
            assert lo <= log_ns[t] < hi                                   # scripts/prove_stdio.rs:89-101 (Rust ranges: hi exclusive)
            if not in_use[t]:
                assert log_ns[t] == lo and t in (1, 3, 4, 5, 8)           # only OPTIONAL_TABLE_INDICES may be absent
                some_absent.add(t)
        assert in_use[3] == in_use[4]                                     # Keccak and KeccakSponge go together (generation/mod.rs:588-593)
        assert in_use[8] == (i != len(shapes) - 1)                        # MemAfter empties in the block's last segment
    assert {1, 3, 4, 5, 8} <= some_absent
    assert all(max(l) <= 11 for l, _ in block_segment_shapes(10, max_log=11))


def test_table_ctl_specs_order_is_starkys():
    """per CTL, per challenge: the run of the table's looking entries, then the looked entry (cross_table_lookup_data)"""
    from zk_evm_amd.all_stark import AllStark, Table
    from zk_evm_amd.shard_prover import table_ctl_specs
    st = AllStark((1, 2, 3, 4))
    chal = [(11, 12), (21, 22)]
    spec = table_ctl_specs(st, Table.MemBefore, chal)
    # MemBefore: one looking entry in the memory CTL, then the looked side of mem_before -- each under both challenges
    assert [(b, g, len(e)) for b, g, e in spec] == [(11, 12, 1), (21, 22, 1), (11, 12, 1), (21, 22, 1)]
    spec = table_ctl_specs(st, Table.Keccak, chal)
    assert [(b, g, len(e)) for b, g, e in spec] == [(11, 12, 1), (21, 22, 1)] * 2          # looked by keccak_inputs and keccak_outputs
    ks = table_ctl_specs(st, Table.KeccakSponge, chal)
    assert max(len(e) for _, _, e in ks) == 136                           # its run of 136 memory reads is ONE z-data with helper columns
    n_z = {t: len(table_ctl_specs(st, t, chal)) for t in range(9)}
    from zk_evm_amd.segment import num_ctl_helpers_zs_all
    for t in range(9):
        assert n_z[t] == num_ctl_helpers_zs_all(st.cross_table_lookups, t, 2, 3)[1], t


def test_contract_line_is_small_strict_json_and_carries_the_contract():
    """r04 verdict, item 1: the driver could not parse the 22 KB line.  The line built from a canned full result (the r04z run)
    is < 4 KB, one ASCII line, strict JSON (no NaN / Infinity), and holds every field the contract names plus `roofline`
    and `cpu_baseline`; a result poisoned with NaN / inf / huge error strings / missing objects still yields such a line."""
    import copy
    import json
    import os
    from tools.benchlib import CONTRACT_LINE_LIMIT, contract_line, write_extra
    full = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "bench_full_result_r04z.json")))

    def strict(s):
        def bad(c):
            raise ValueError(c)
        return json.loads(s, parse_constant=bad)

    line = contract_line(full)
    assert len(line) < CONTRACT_LINE_LIMIT < 8000 and "\n" not in line and line.isascii()
    d = strict(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert abs(d["value"] - full["value"]) < 1e-8 * full["value"] and abs(d["ms_per_step"] - full["ms_per_step"]) < 1e-6
    assert set(d["config"]) == {"workload", "parallelism"} and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-6 and r["traffic"] > 0
    assert {"value", "unit", "cores", "kind", "sample"} <= set(d["cpu_baseline"])
    assert d["ntt"]["achieved_GBs"] > 0 and d["dist"]["backend"] == "nccl" and d["secondary"]["realistic_ms_per_proof"] > 0
    # r05: the baseline in the headline's unit -- a whole-segment sample scaled by committed cells (tools/bench_secondary.py)
    seg = copy.deepcopy(full)
    seg["cpu_baseline"] = {"value": 1.0 / 1363.2, "unit": "segment proofs/s", "cores": 16, "kind": "port", "sample": "s" * 900, "seconds": 1363.2,
                           "sample_seconds": 21.3, "sample_log_n": 14, "scale": 64.0, "shape": "9 tables x 2^14 rows measured",
                           "gpu_same_sample_s": 0.05, "proofs_identical": True, "cpu_model": "x", "poseidon_perms_per_s_per_core": 9.7e5}
    d = strict(contract_line(seg))
    cb = d["cpu_baseline"]
    assert cb["unit"] == d["unit"] == "segment proofs/s" and cb["proofs_identical"] is True and cb["scale"] == 64.0 and len(cb["sample"]) <= 400
    assert abs(cb["value"] * cb["seconds"] - 1.0) < 1e-5 and abs(cb["sample_seconds"] * cb["scale"] - cb["seconds"]) < 1e-6 * cb["seconds"]
    assert cb.get("cpu_model") == "x"
    # r06: a cached MEASUREMENT of the workload's own shape (tools/cpu_baseline_cache.py) says so on the line, with shape, cores, CPU model
    from tools.bench_secondary import cached_cpu_baseline
    hit = {"oracle_hash": "0123456789abcdef", "cpu_model": "AMD EPYC 9575F 64-Core Processor", "cores": 16, "log_ns": [20] * 9, "hasher": 0,
           "cdk_erigon": False, "cpu_seconds": 1402.7, "committed_cells": 4.47e9, "measured_at": "2026-09-30T12:00:00Z"}
    seg["cpu_baseline"] = cached_cpu_baseline(hit)
    cb = strict(contract_line(seg))["cpu_baseline"]
    assert cb["measured"] is True and cb["shape"] == "9 x 2^20 rows" and cb["cores"] == 16 and cb["cpu_model"].startswith("AMD EPYC")
    assert cb["unit"] == "segment proofs/s" and abs(cb["value"] * 1402.7 - 1.0) < 1e-6 and "MEASURED" in cb["sample"] and "scale" not in cb
    # poisoned
    bad = copy.deepcopy(full)
    bad["value"] = float("nan")
    bad["roofline"]["traffic"] = float("inf")
    bad["roofline"]["valu"] = None
    bad["realistic"] = {"error": "x" * 5000}
    bad["cpu_baseline"] = {"error": "y" * 5000}
    bad["dist"]["error"] = "z" * 5000
    bad["config"]["workload"] = "w" * 5000
    del bad["ntt"], bad["in_flight"]
    line = contract_line(bad)
    d = strict(line)
    assert len(line) < CONTRACT_LINE_LIMIT and d["value"] is None and d["roofline"]["traffic"] is None and "realistic" in d["secondary_failed"]
    assert contract_line({"metric": "m"}) and strict(contract_line({}))["roofline"]["bound"] == "hbm"
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        wrote = write_extra(bad, root=td)
        assert len(wrote) == 2 and strict(open(wrote[0]).read())["value"] is None


def test_cpu_baseline_cache_is_keyed_by_oracle_host_and_shape(tmp_path):
    """tools/cpu_baseline_cache.py: a measurement is used only for exactly the oracle sources, CPU model, core count, segment shape
    and hasher it was made with; storing replaces the entry of the same key."""
    from tools import cpu_baseline_cache as cbc
    path = str(tmp_path / "cache.json")
    e = {"oracle_hash": "aaaa", "cpu_model": "cpu A", "cores": 16, "log_ns": [20] * 9, "hasher": 0, "cdk_erigon": False, "cpu_seconds": 1400.0}
    cbc.store(e, path)
    cbc.store(dict(e, log_ns=[17, 14, 19, 17, 13, 16, 21, 19, 19], cpu_seconds=79.5), path)
    cbc.store(dict(e, cpu_seconds=1390.0), path)                        # same key: replaced
    entries = cbc.load(path)
    assert len(entries) == 2
    look = lambda **k: cbc.lookup(k.get("model", "cpu A"), k.get("cores", 16), k.get("log_ns", [20] * 9), k.get("hasher", 0), entries=entries, oracle_hash=k.get("oh", "aaaa"))
    assert look()["cpu_seconds"] == 1390.0
    assert look(log_ns=[17, 14, 19, 17, 13, 16, 21, 19, 19])["cpu_seconds"] == 79.5
    assert look(oh="bbbb") is None and look(model="cpu B") is None and look(cores=8) is None and look(hasher=1) is None and look(log_ns=[19] * 9) is None
    assert len(cbc.oracle_source_hash()) == 16 and cbc.oracle_source_hash() == cbc.oracle_source_hash()
