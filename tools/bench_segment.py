#!/usr/bin/env python3
"""Exploratory: time one full segment proof (all nine tables, real CTL wiring) at given sizes on one GPU."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch


def synthetic_segment_traces(log_ns, dev, seed=1):
    """Random traces with every CTL / lookup filter column binary (one-hot flags), generated in HBM."""
    from zk_evm_amd.all_stark import TABLE_COLUMNS
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    out = []
    for t, (c, l) in enumerate(zip(TABLE_COLUMNS, log_ns)):
        n = 1 << l
        tr = torch.randint(-(1 << 63), (1 << 63) - 1, (c, n), dtype=torch.int64, device=dev, generator=g)
        def binary(cols):
            for k in cols:
                tr[k] = torch.randint(0, 2, (n,), dtype=torch.int64, device=dev, generator=g)
        def one_hot(cols, extra=1):
            pick = torch.randint(0, len(cols) + extra, (n,), device=dev, generator=g)
            for i, k in enumerate(cols):
                tr[k] = (pick == i).to(torch.int64)
        if t == 0: one_hot(list(range(17)))
        elif t == 1: one_hot(list(range(1, 33)))
        elif t == 2:
            one_hot(list(range(6, 24)))
            binary(list(range(24, 33)) + [41, 54, 67, 80])
        elif t == 3: binary([0, 23])
        elif t == 4:
            kind = torch.randint(0, 3, (n,), device=dev, generator=g)
            ln = torch.randint(0, 136, (n,), device=dev, generator=g)
            tr[0] = (kind == 1).to(torch.int64)
            for i in range(136):
                tr[6 + i] = ((kind == 2) & (ln <= i)).to(torch.int64)
        elif t == 5: one_hot([0, 1, 2])
        elif t == 6:
            binary([0, 22, 24, 26])
            one_hot([15, 16], 2)
            f = torch.randint(0, 2, (n,), dtype=torch.int64, device=dev, generator=g)
            tr[1] = f          # timestamp in {0,1}, timestamp_inv = timestamp: filter_mem_before = 1 - t*t_inv
            tr[2] = f
        else: binary([0])
        out.append(tr)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-n", type=int, default=16)
    ap.add_argument("--log-ns", type=str, default="")
    ap.add_argument("--steps", type=int, default=1)
    a = ap.parse_args()
    import zk_evm_amd as zk
    import zk_evm_amd.segment as sg
    from zk_evm_amd.all_stark import AllStark
    dev = torch.device("cuda:0")
    log_ns = [int(x) for x in a.log_ns.split(",")] if a.log_ns else [a.log_n] * 9
    traces = synthetic_segment_traces(log_ns, dev)
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    print("free after trace gen GB", torch.cuda.mem_get_info()[0] / 1e9, "torch reserved", torch.cuda.memory_reserved() / 1e9, flush=True)
    print("trace GB", sum(t.numel() for t in traces) * 8 / 1e9, flush=True)
    cfg = zk.StarkConfig.standard_fast_config()
    alls = AllStark((1, 2, 3, 4))
    for it in range(a.steps + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        timing = {} if it == a.steps else None
        pr = sg.prove_with_traces(alls, cfg, traces, [True] * 9, sg.PublicValues(), timing=timing)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(json.dumps({"iter": it, "seconds": dt, "timing": timing, "peak_alloc_GB": torch.cuda.max_memory_allocated() / 1e9,
                          "free_GB": torch.cuda.mem_get_info()[0] / 1e9}), flush=True)
        del pr


if __name__ == "__main__":
    main()
