#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X STARK commitment path.

Workload (BASELINE.json configs[1]): one ArithmeticStark-shaped trace, 116 columns x 2^20 rows,
`PolynomialBatch::from_values(trace, rate_bits=1, blinding=false, cap_height=4)` with the Poseidon
hasher -- iNTT, coset LDE to 2^21, Poseidon leaf hashing, Merkle cap -- i.e. the "compute trace
commitment" scope of the reference (evm_arithmetization/src/prover.rs:92-111).  A "step" is one
such commit over a synthetic trace already resident in HBM.

Multi-GPU: trace segments / tables are independent units, so each rank commits its
own trace with no data-path collective ("scaling": "weak"); value = commits of all ranks / max time.

Prints ONE JSON line on rank 0.  `roofline` is for the dominant kernel (Poseidon leaf hashing),
timed with HIP events on the kernel's own stream inside the timed region; `cpu_baseline` is the
CPU oracle (OpenMP over columns / leaves, the axes rayon uses in the reference) on a bounded
sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--cols", type=int, default=116)
    ap.add_argument("--log-n", type=int, default=20)
    ap.add_argument("--hasher", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-log-n", type=int, default=16)
    ap.add_argument("--proof-steps", type=int, default=2,
                    help="also time N full ArithmeticStark table proofs (0 disables)")
    return ap.parse_args()


def cpu_baseline(cols, log_n, sample_log_n, hasher):
    """Time the oracle's from_values on a bounded sample (cols x 2^sample_log_n) and extrapolate
    linearly in rows to the full workload (slightly optimistic for the CPU: NTT is n log n)."""
    import numpy as np
    from tests.oracle_lib import load_oracle, splitmix64
    o = load_oracle()
    n = 1 << sample_log_n
    vals = np.stack([splitmix64(0x6FEB51B7EC230F25 + c, n) for c in range(cols)])
    o.commit_values(vals[:, : 1 << 10].copy(), want_leaves=False, hasher=hasher)  # warm
    t0 = time.perf_counter()
    reps = 0
    while True:
        o.commit_values(vals, rate_bits=1, cap_height=4, hasher=hasher, want_leaves=False)
        reps += 1
        el = time.perf_counter() - t0
        if el > 10.0 or reps >= 5:
            break
    per_sample = el / reps
    scale = float(1 << (log_n - sample_log_n))
    return {
        "value": 1.0 / (per_sample * scale),
        "unit": "commits/s",
        "cores": int(o.lib.orc_num_threads()),
        "kind": "port",
        "sample": f"oracle from_values on {cols} x 2^{sample_log_n} rows ({reps} reps, "
                  f"{per_sample:.3f} s each), scaled x{int(scale)} rows to 2^{log_n}",
        "seconds_per_full_commit_est": per_sample * scale,
    }


def table_proof_bench(ctx, dev, log_n, steps):
    """Secondary measurement (not `value`): one full ArithmeticStark TABLE proof = starky
    prove_with_commitment (logUp helper columns, CTL partial sums, auxiliary commit, quotient with
    the complete Arithmetic AIR, quotient commit, openings, FRI with the production parameters)
    on top of the trace commit.  Synthetic trace: one-hot op flags, 16-bit limbs, real range-counter
    and frequency columns, so the lookup argument is the real one."""
    import numpy as np
    import torch
    import zk_evm_amd as zk
    import zk_evm_amd.prover as zp
    from zk_evm_amd.stark import Column, Filter, Lookup, ctl_partial_sums
    n = 1 << log_n
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    trace = torch.zeros((116, n), dtype=torch.int64, device=dev)
    which = torch.randint(0, 18, (n,), device=dev, generator=g)
    for i in range(17):
        trace[i] = (which == i).to(torch.int64)
    trace[18:114] = torch.randint(0, 1 << 16, (96, n), dtype=torch.int64, device=dev, generator=g)
    trace[114] = torch.clamp(torch.arange(n, device=dev), max=65535)
    trace[115, : 1 << 16] = torch.bincount(trace[18:114].reshape(-1), minlength=1 << 16)
    lookup = Lookup(Column.singles(range(18, 114)), Column.single(114), Column.single(115), [Filter() for _ in range(96)])
    cols = [Column.single(17)]
    for reg in (18, 34, 50, 66):
        cols += [Column.linear_combination([(reg + 2 * k, 1), (reg + 2 * k + 1, 1 << 16)]) for k in range(8)]
    ctl_entry = [(cols, Filter.new_simple(Column.sum(range(17))))]
    cfg = zk.StarkConfig.standard_fast_config()
    times = []
    stages = {}
    for it in range(steps + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tb = zk.PolynomialBatch.from_values(trace, 1, False, 4, ctx=ctx)
        ch = zk.Challenger(0)
        ch.observe_cap(tb.merkle_tree.cap)
        chal = [(ch.get_challenge(), ch.get_challenge()) for _ in range(cfg.num_challenges)]
        t1 = time.perf_counter()
        zd = []
        for b, gm in chal:
            zd.append(zp.CtlZData(b, gm, ctl_entry, ctl_partial_sums(trace, ctl_entry, b, gm, 3, ctx=ctx)))
        t2 = time.perf_counter()
        pr = zp.prove_with_commitment(zp.AIR_ARITHMETIC, cfg, trace, tb, [lookup], zd, chal, ch)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        tb.free()
        if it:  # first iteration is warm-up
            times.append(t3 - t0)
            for k, v in (("trace_commit", t1 - t0), ("ctl_data", t2 - t1), ("prove_with_commitment", t3 - t2)):
                stages[k] = stages.get(k, 0.0) + v * 1e3 / steps
    sec = sum(times) / len(times)
    return {"what": f"ArithmeticStark table proof, 2^{log_n} rows, standard_fast_config (2 challenges, 84 queries, 16 PoW bits)",
            "proofs_per_s": 1.0 / sec, "ms_per_proof": sec * 1e3, "steps": steps, "stages_ms": stages,
            "proof_words": int(pr.opening_proof.size)}


def main():
    a = parse()
    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    torch.cuda.set_device(local)
    dev = torch.device(f"cuda:{local}")

    import zk_evm_amd
    from zk_evm_amd import PolynomialBatch
    ctx = zk_evm_amd.Context(local)
    ctx.use_torch_current_stream()

    n = 1 << a.log_n
    N = n << 1
    g = torch.Generator(device=dev)
    g.manual_seed(0x5EED + rank)
    # synthetic trace, uniform u64 bit patterns (non-canonical representatives included), in HBM
    hi = torch.randint(0, 1 << 32, (a.cols, n), dtype=torch.int64, device=dev, generator=g)
    lo = torch.randint(0, 1 << 32, (a.cols, n), dtype=torch.int64, device=dev, generator=g)
    trace = (hi << 32) | lo
    del hi, lo

    def step():
        b = PolynomialBatch.from_values(trace, 1, False, 4, hasher=a.hasher, ctx=ctx)
        t = ctx.last_timings()
        b.free()
        return t

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    stage = {"ifft": 0.0, "lde": 0.0, "leaf_hash": 0.0, "tree": 0.0}
    for _ in range(a.steps):
        t = step()
        for k in stage:
            stage[k] += t[k]
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    if rank == 0:
        ms_per_step = 1e3 * elapsed / a.steps
        value = world * a.steps / elapsed
        for k in stage:
            stage[k] /= a.steps
        # dominant kernel: poseidon_hash_rows_kernel (one launch per commit). Algorithmic bytes:
        # read the LDE once (8*C*N) + write N 32-byte digests.
        dom_bytes = 8.0 * a.cols * N + 32.0 * N
        dom_ms = stage["leaf_hash"]
        achieved = dom_bytes / (dom_ms * 1e-3) / 1e9
        perms = N * ((a.cols + 7) // 8) if a.cols > 4 else 0
        # whole-commit algorithmic bytes: 32*C*n + 128*n
        commit_bytes = 32.0 * a.cols * n + 128.0 * n
        ntt_bytes = 40.0 * a.cols * n
        ntt_ms = stage["ifft"] + stage["lde"]
        # HBM traffic and VALU instruction counts of the dominant kernel come from separate
        # rocprofv3 --pmc passes (tools/collect_pmc.sh), summarised in profiles/pmc_latest.json;
        # they only apply to the default workload they were collected on.
        traffic = None
        valu = None
        pmc_path = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc_path) and a.cols == 116 and a.log_n == 20 and a.hasher == 0:
            try:
                pmc = json.load(open(pmc_path))
                traffic = pmc.get("leaf_hash_hbm_bytes_per_launch")
                insts = pmc.get("leaf_hash_valu_wave_insts_per_launch")
                if insts:
                    # integer-issue roofline: every useful integer VALU op on gfx950 issues at
                    # ~4 cycles per wave64 per SIMD (profiles/r01_ubench_valu_issue_rates.txt)
                    peak = 1024 * 2.4e9 / 4.0
                    ach = insts / (dom_ms * 1e-3)
                    valu = {"wave_insts_per_launch": insts, "achieved_wave_insts_per_s": ach,
                            "peak_wave_insts_per_s": peak, "frac": ach / peak,
                            "assumes": "1024 SIMDs x 2.4 GHz / 4 cycles per integer VALU wave-instruction",
                            "source": pmc.get("source")}
            except Exception:
                traffic = None
        out = {
            "metric": "ArithmeticStark-shaped 2^20-row trace commits/sec (Goldilocks iNTT + coset LDE + "
                      "Poseidon Merkle cap; BASELINE configs[1], the commit stage of segment STARK proofs/sec)",
            "value": value,
            "unit": "commits/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {"workload": f"PolynomialBatch::from_values {a.cols} cols x 2^{a.log_n} rows, "
                                   f"rate_bits 1, cap_height 4, hasher {'poseidon' if a.hasher == 0 else 'keccak25'}",
                       "parallelism": f"{world} independent traces (one per GPU), no collective"},
            "roofline": {"bound": "hbm", "kernel": "poseidon_hash_rows_kernel" if a.hasher == 0 else "keccak_hash_rows_kernel",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "ms_per_launch": dom_ms, "algorithmic_bytes": dom_bytes,
                         "note": "kernel is integer-ALU bound (Poseidon), see DESIGN.md; "
                                 "permutations/s = %.3e" % (perms / (dom_ms * 1e-3) if dom_ms else 0),
                         "valu": valu},
            "stages_ms": stage,
            "ntt": {"achieved_GBs": ntt_bytes / (ntt_ms * 1e-3) / 1e9, "algorithmic_bytes": ntt_bytes,
                    "frac_of_hbm_peak": ntt_bytes / (ntt_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
            "commit": {"achieved_GBs": commit_bytes / (ms_per_step * 1e-3) / 1e9,
                       "algorithmic_bytes": commit_bytes},
        }
        if a.proof_steps > 0 and a.cols == 116 and a.hasher == 0:
            try:
                out["table_proof"] = table_proof_bench(ctx, dev, a.log_n, a.proof_steps)
            except Exception as e:
                out["table_proof"] = {"error": repr(e)}
        if not a.no_cpu_baseline and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline(a.cols, a.log_n, min(a.cpu_sample_log_n, a.log_n), a.hasher)
            except Exception as e:  # the oracle is only a reported baseline; never fatal
                out["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
