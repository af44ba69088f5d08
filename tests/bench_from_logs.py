"""From operation logs to a segment proof: time `tracegen.Traces.into_tables` (device witness generation, log upload
included) and `prove_with_traces` on the VALID 20 000-iteration hash-and-store workload of the test suite (all nine
tables live: Cpu 2^18, Keccak 2^19, KeccakSponge 2^15, Logic 2^17, BytePacking 2^15, Arithmetic 2^16, Memory 2^21).
The interpreter run that produces the logs is test infrastructure and not timed.  Usage: python -m tests.bench_from_logs"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import zk_evm_amd as zk
    import zk_evm_amd.segment as sg
    import zk_evm_amd.tracegen as tg
    from oracle import segment as oseg                      # public-value writes of the test workload (test infrastructure)
    from tests import consistent_segment as cs
    from tests.oracle_lib import load_oracle
    from zk_evm_amd.all_stark import AllStark
    oracle = load_oracle()
    n_it = 20000
    code, halt = cs.hashing_loop_program(n_it)
    run = cs.cpu_program_trace(oracle.keccak256, n=1 << 18, program=code, halt_pc=halt, return_run=True)
    pvd = cs.make_public_values(np.random.default_rng(88))
    m64 = (1 << 64) - 1
    before = [((0, cs.SEG_CODE, i), b) for i, b in enumerate(code)] + [((0, cs.SEG_SHIFT_TABLE, i), 1 << i) for i in range(256)]
    pub = [dict(filter=True, timestamp=2, ctx=0, seg=s, virt=i, is_read=False, value=v) for s, i, v in oseg.public_memory_writes(pvd, 1, len(code))]
    tr = tg.Traces()
    tr.cpu = torch.from_numpy(np.ascontiguousarray(run.t.T).view(np.int64))
    tr.memory_ops = np.array([[(1 if d["is_read"] else 0) | 2, d["timestamp"], d["ctx"], d["seg"], d["virt"]] +
                              [(d["value"] >> (64 * l)) & m64 for l in range(4)] for d in pub + run.mem_ops], dtype=np.uint64)
    ar = np.zeros((n_it, 18), dtype=np.uint64)
    ar[:, 0], ar[:, 2], ar[:, 6] = 2, [op[2] for op in run.arith], [op[3] for op in run.arith]
    tr.arithmetic_ops = ar
    # every log in the C ABI's packed record layout (what a Rust caller would hand over): the timed part is then
    # upload + kernels, not Python list handling
    tr.keccak_sponge_ops = run.sponge                               # (address, timestamp, bytes): variable-length inputs
    tr.byte_packing_ops = np.array([[1 if rd else 0, c, s, v, ts, len(d)] + [int.from_bytes(d[8 * k:8 * k + 8].ljust(8, b"\0"), "little") for k in range(4)]
                                    for rd, (c, s, v), ts, d in run.packing], dtype=np.uint64)
    eff = [cs.single_block_sponge_effects(d, ts) for _, ts, d in run.sponge]
    tr.keccak_inputs = (np.array([e[0][0] for e in eff], dtype=np.uint64), np.array([e[0][1] for e in eff], dtype=np.uint64))
    tr.logic_ops = np.array([[k] + [(a >> (64 * l)) & m64 for l in range(4)] + [(b >> (64 * l)) & m64 for l in range(4)]
                             for e in eff for k, a, b in e[1]], dtype=np.uint64)
    bef = np.array([[c, s, v] + [(val >> (64 * l)) & m64 for l in range(4)] for (c, s, v), val in before], dtype=np.uint64)
    st, cfg = AllStark((halt, 0, 777777, 888888)), zk.StarkConfig()
    from tests.test_gpu_segment import to_public_values
    gen, prove = [], []
    for _ in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        dev, _ = tr.into_tables(st, bef, [], cfg)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        sg.prove_with_traces(st, cfg, dev, [True] * 9, to_public_values(pvd))
        torch.cuda.synchronize(); t2 = time.perf_counter()
        gen.append(t1 - t0); prove.append(t2 - t1)
    cells = sum(int(t.shape[0]) * int(t.shape[1]) for t in dev)
    print(json.dumps({"workload": "20000-iteration hash-and-store loop, valid witness, nine tables live",
                      "table_heights_log2": [int(t.shape[1]).bit_length() - 1 for t in dev], "trace_GB": cells * 8 / 1e9,
                      "into_tables_ms": round(1e3 * min(gen[1:]), 2), "prove_ms": round(1e3 * min(prove[1:]), 2),
                      "note": "logs handed over as packed arrays in the C-ABI record layouts (KeccakSponge as a tuple list); "
                              "into_tables = upload from pageable memory + generator kernels + the Cpu transpose"}))


if __name__ == "__main__":
    main()
