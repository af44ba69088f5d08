"""Soak with CHANGING shapes: a long-running worker proves segments of different table heights back to back.  The arena
must not creep (fragmentation) and every shape must reproduce its own proof bit for bit.
Usage: python tools/soak_shapes.py [rounds]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import zk_evm_amd as zk
    import zk_evm_amd.segment as sg
    from bench import REALISTIC_LOG_NS, synthetic_segment_traces
    from tools.soak_segment import digest
    from zk_evm_amd.all_stark import AllStark
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    dev = torch.device("cuda:0")
    ctx = zk.Context(0)
    ctx.use_torch_current_stream()
    shapes = [REALISTIC_LOG_NS, [16] * 9, [18, 12, 17, 15, 12, 14, 19, 16, 17], [14] * 9, [19, 14, 19, 16, 13, 16, 20, 18, 18]]
    traces = [synthetic_segment_traces(s, dev, seed=7 + i) for i, s in enumerate(shapes)]
    st, cfg = AllStark((1, 2, 3, 4)), zk.StarkConfig()
    first, reserved, t0 = {}, [], time.perf_counter()
    for r in range(rounds):
        for i, tr in enumerate(traces):
            d = digest(sg.prove_with_traces(st, cfg, tr, [True] * 9, sg.PublicValues(), ctx=ctx))
            assert first.setdefault(i, d) == d, (r, i)
        m = ctx.mem_stats()
        assert m["in_use"] == 0
        reserved.append(m["reserved"])
    torch.cuda.synchronize()
    print(json.dumps({"rounds": rounds, "shapes": len(shapes), "proofs": rounds * len(shapes), "seconds": time.perf_counter() - t0,
                      "arena_reserved_GB_first_round": reserved[0] / 1e9, "arena_reserved_GB_last_round": reserved[-1] / 1e9,
                      "arena_reserved_GB_max": max(reserved) / 1e9}))


if __name__ == "__main__":
    main()
