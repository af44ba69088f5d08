#!/usr/bin/env python3
"""bench.py -- headline benchmark: segment STARK proofs/sec on MI355X.

Default workload (BASELINE.json `metric`, configs[2]): ONE full segment proof = the reference's
`prove_with_traces` (evm_arithmetization/src/prover.rs:72-194) over all nine AllStark tables
(Arithmetic 116, BytePacking 71, Cpu 85, Keccak 2431, KeccakSponge 438, Logic 523, Memory 30, MemBefore 12,
MemAfter 12 columns), every table at 2^20 rows, with the real all_stark.rs CTL wiring (10 CTLs, 176 Memory
lookers) and range-check lookups, `StarkConfig::standard_fast_config` (2 challenges, rate_bits 1, cap_height 4,
16 PoW bits, 84 FRI queries), Poseidon hasher.  A "step" is one such proof over synthetic traces already
resident in HBM (31.2 GB): 9 trace commitments, CTL columns, and per table lookup columns, auxiliary
commitment, quotient with the table's full AIR, quotient commitment, openings and FRI.

Also measured in the same run (secondary objects of the same JSON line):
  * `commit_config1`: BASELINE configs[1], the single ArithmeticStark 116 x 2^20 trace commitment
    (`PolynomialBatch::from_values`), with stage timings, NTT GB/s and the Poseidon VALU-issue fraction;
  * `segment_timing`: wall time per stage / table of one extra, synchronised, untimed proof.

`--workload commit` makes the configs[1] commit the timed step instead.

Multi-GPU: segments are independent units (fresh Challenger per segment), so each rank proves
its own segment with no data-path collective ("scaling": "weak"); value = segments of all ranks / max time.

Prints ONE JSON line on rank 0.  `roofline` is for the dominant kernel (Poseidon leaf hashing,
poseidon_hash_rows_kernel: ~60 % of the segment), timed with HIP events on the kernel's own stream inside the
timed region and summed over its launches (one per commitment); `cpu_baseline` is the CPU oracle (OpenMP over
columns / leaves, the axes rayon uses in the reference) on a bounded sample.
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
    ap.add_argument("--gpus", type=int, default=1,
                    help="ranks = GPUs of this node; with N > 1 and no WORLD_SIZE in the environment bench.py launches "
                         "the N ranks itself (one process per GPU, RCCL); under torchrun it joins the given world")
    ap.add_argument("--dist-backend", choices=["nccl", "gloo"], default=os.environ.get("ZK_BENCH_BACKEND", "nccl"),
                    help="nccl = RCCL (one rank per GPU); gloo lets several ranks share one GPU (launcher test)")
    ap.add_argument("--devices", type=str, default=os.environ.get("ZK_BENCH_DEVICES", ""),
                    help="comma-separated device index per local rank (default: rank r -> device r)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the process group (and run barrier / MAX all-reduce / cap all-gather through it) even "
                         "with ONE rank: executes the RCCL path of an N-GPU run on a single-GPU box")
    ap.add_argument("--no-dist-selftest", action="store_true",
                    help="N = 1 only: do not join a one-rank process group (the `dist` object of the line)")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=["segment", "commit"], default="segment")
    ap.add_argument("--commit-steps", type=int, default=5, help="configs[1] commits timed as a secondary object")
    ap.add_argument("--cols", type=int, default=116)
    ap.add_argument("--log-n", type=int, default=20)
    ap.add_argument("--hasher", type=int, default=0)
    ap.add_argument("--cdk-erigon", action="store_true",
                    help="the ten-table cdk_erigon feature set (86-column Cpu, Poseidon table, 13 CTLs) instead of eth_mainnet")
    ap.add_argument("--log-ns", type=str, default="",
                    help="nine comma-separated table heights (log2) instead of --log-n for all; 'realistic' = the upper "
                         "ends of the per-table ranges of the reference's scripts/prove_stdio.rs:89-101")
    ap.add_argument("--in-flight", type=int, default=2,
                    help="also report W segments in flight per GPU as a secondary object (1 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the h2d / realistic secondary objects")
    ap.add_argument("--no-pmc", action="store_true",
                    help="do not collect the dominant kernel's HBM-traffic / VALU counters with rocprofv3 --pmc child passes "
                         "(the committed profiles/pmc_latest.json is quoted instead, marked as such)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-sample-log-n", type=int, default=18)
    ap.add_argument("--cpu-table-log-n", type=int, default=20,
                    help="height of the ArithmeticStark table proven on the CPU for cpu_baseline (0 = skip, fall back to "
                         "the commit-sample extrapolation)")
    return ap.parse_args()


# Arithmetic, BytePacking, Cpu, Keccak, KeccakSponge, Logic, Memory, MemBefore, MemAfter (scripts/prove_stdio.rs:89-101)
REALISTIC_LOG_NS = [17, 14, 19, 17, 13, 16, 21, 19, 19]


def cpu_baseline(cols, log_n, sample_log_n, hasher, max_reps=5):
    """Time the oracle's from_values on a bounded sample (cols x 2^sample_log_n) and extrapolate
    linearly in rows to the full workload (slightly optimistic for the CPU: NTT is n log n)."""
    import ctypes
    import math
    import numpy as np
    from tests.oracle_lib import load_oracle, splitmix64
    o = load_oracle()
    # threads actually available to this process: affinity mask and cgroup CPU quota, not just the core count OpenMP
    # sees (running 128 threads inside a smaller quota makes the baseline look worse than the hardware is)
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else int(o.lib.orc_num_threads())
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            cores = max(1, min(cores, math.ceil(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    cores = min(cores, int(o.lib.orc_num_threads()))
    try:
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(cores)
    except OSError:
        pass
    n = 1 << sample_log_n
    vals = np.stack([splitmix64(0x6FEB51B7EC230F25 + c, n) for c in range(cols)])
    o.commit_values(vals[:, : 1 << 10].copy(), want_leaves=False, hasher=hasher)  # warm
    t0 = time.perf_counter()
    reps = 0
    while True:
        o.commit_values(vals, rate_bits=1, cap_height=4, hasher=hasher, want_leaves=False)
        reps += 1
        el = time.perf_counter() - t0
        if el > 10.0 or reps >= max_reps:
            break
    per_sample = el / reps
    scale = float(1 << (log_n - sample_log_n))
    return {
        "value": 1.0 / (per_sample * scale),
        "unit": "commits/s",
        "cores": cores,
        "kind": "port",
        "sample": f"oracle from_values on {cols} x 2^{sample_log_n} rows ({reps} reps, "
                  f"{per_sample:.3f} s each), scaled x{int(scale)} rows to 2^{log_n}",
        "seconds_per_full_commit_est": per_sample * scale,
    }


def arithmetic_table_trace(dev, log_n):
    """ArithmeticStark-shaped table for the table-proof comparison: one-hot operation flags, 16-bit limbs, the real
    range-counter and frequency columns (arithmetic_stark.rs:130-156), so the table's own logUp argument and its CTL
    (looked side of CTL 0, all_stark.rs:176-181) are exactly the reference's."""
    import torch
    n = 1 << log_n
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    trace = torch.zeros((116, n), dtype=torch.int64, device=dev)
    which = torch.randint(0, 18, (n,), device=dev, generator=g)
    for i in range(17):
        trace[i] = (which == i).to(torch.int64)
    trace[17] = torch.randint(0, 256, (n,), dtype=torch.int64, device=dev, generator=g)        # opcode
    trace[18:114] = torch.randint(0, 1 << 16, (96, n), dtype=torch.int64, device=dev, generator=g)
    trace[114] = torch.clamp(torch.arange(n, device=dev), max=65535)
    trace[115, : 1 << 16] = torch.bincount(trace[18:114].reshape(-1), minlength=1 << 16)
    return trace


def gpu_table_proof(ctx, trace, all_stark, cfg, reps):
    """One ArithmeticStark table proof on the GPU: from_values + transcript + CTL data + prove_single_table
    (= the reference's keccak_benchmark shape, keccak_stark.rs:692-760: `from_values` and `prove_single_table` timed
    together).  -> (seconds per proof, stage seconds, last proof)."""
    import torch
    import zk_evm_amd as zk
    import zk_evm_amd.prover as zp
    from zk_evm_amd.all_stark import Table
    from zk_evm_amd.stark import ctl_partial_sums
    looked = all_stark.cross_table_lookups[0].looked_table
    assert looked.table == Table.Arithmetic
    entry = [(looked.columns, looked.filter)]
    times, stages, pr = [], {}, None
    for it in range(reps + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tb = zk.PolynomialBatch.from_values(trace, 1, False, 4, ctx=ctx)
        ch = zk.Challenger(0)
        ch.observe_cap(tb.merkle_tree.cap)
        chal = [(ch.get_challenge(), ch.get_challenge()) for _ in range(cfg.num_challenges)]
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        zd = [zp.CtlZData(b, gm, entry, ctl_partial_sums(trace, entry, b, gm, 3, ctx=ctx)) for b, gm in chal]
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        pr = zp.prove_single_table(zp.AIR_ARITHMETIC, cfg, trace, tb, all_stark.lookups[Table.Arithmetic], zd, chal, ch)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        tb.free()
        if it:  # first iteration is warm-up
            times.append(t3 - t0)
            for k, v in (("trace commitment", t1 - t0), ("ctl columns", t2 - t1), ("prove_with_commitment", t3 - t2)):
                stages[k] = stages.get(k, 0.0) + v / reps
    return sum(times) / len(times), stages, pr


def cpu_table_proof_baseline(ctx, dev, log_n, gpu_reps=3):
    """`cpu_baseline`: ONE whole ArithmeticStark table proof MEASURED on the host -- trace commitment, logUp helper
    columns, CTL columns, auxiliary commitment, quotient (the complete Arithmetic AIR, 707 constraints, + lookup + CTL
    checks), quotient commitment, openings, FRI with standard_fast_config -- by the CPU oracle (C + OpenMP over columns /
    leaves / rows, the axes rayon uses in the reference), next to the same proof of the same trace on the GPU, and the
    two proofs compared word for word."""
    import ctypes as C
    import math
    import platform
    import numpy as np
    import zk_evm_amd as zk
    import tests.oracle_lib as ol
    from oracle import airs as oairs
    from oracle import all_stark as oas
    from oracle import fast_stark as FS
    from oracle import stark as OS
    from zk_evm_amd.all_stark import AllStark
    o = ol.load_oracle()
    ol.setup_fri_api(o)
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else int(o.lib.orc_num_threads())
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            cores = max(1, min(cores, math.ceil(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    cores = min(cores, int(o.lib.orc_num_threads()))
    try:
        C.CDLL("libgomp.so.1").omp_set_num_threads(cores)
    except OSError:
        pass
    model = platform.processor() or "unknown"
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    trace = arithmetic_table_trace(dev, log_n)
    cfg = zk.StarkConfig.standard_fast_config()
    gpu_s, gpu_stages, gp = gpu_table_proof(ctx, trace, AllStark((1, 2, 3, 4)), cfg, gpu_reps)
    host = trace.cpu().numpy().view(np.uint64)
    del trace
    # ---- the CPU proof, measured once ----
    reg = oas.Registry(False)
    ocfg = ol.make_cfg()
    stages = {}
    t0 = time.perf_counter()
    commit = o.commit_values(host, rate_bits=1, cap_height=4, hasher=0)
    och = ol.new_challenger(o, 0)
    o.lib.orc_challenger_observe_cap(C.byref(och), commit["cap"], 16)
    chal = [OS.GrandProductChallenge(o.lib.orc_challenger_get(C.byref(och)), o.lib.orc_challenger_get(C.byref(och)))
            for _ in range(ocfg.num_challenges)]
    stages["trace commitment"] = time.perf_counter() - t0
    looked = reg.ctls[0].looked_table
    zds = [OS.CtlZData(ch, [(looked.columns, looked.filter)], 0) for ch in chal]
    init = np.zeros(12, dtype=np.uint64)
    o.lib.orc_challenger_compact(C.byref(och), init)
    cp = FS.prove_with_commitment(o, ol, ocfg, oairs.AIRS[5][0], host, commit, reg.lookups[0], zds,
                                  [(c.beta, c.gamma) for c in chal], och, timing=stages)
    cpu_s = time.perf_counter() - t0
    same = (np.array_equal(gp.trace_cap, commit["cap"]) and np.array_equal(gp.auxiliary_polys_cap, cp["aux_cap"])
            and np.array_equal(gp.quotient_polys_cap, cp["quotient_cap"])
            and np.array_equal(gp.openings.reshape(-1), cp["openings"]) and np.array_equal(gp.opening_proof, cp["fri"]))
    return {
        "value": 1.0 / cpu_s, "unit": "ArithmeticStark table proofs/s (116 columns x 2^%d rows)" % log_n, "cores": cores,
        "kind": "port", "cpu_model": model, "omp_num_threads": cores,
        "sample": "ONE whole ArithmeticStark table proof, 116 x 2^%d rows, standard_fast_config (2 challenges, 84 queries, "
                  "16 PoW bits), measured end to end, not scaled: from_values + logUp (96 columns) + CTL + auxiliary "
                  "commitment + quotient (707 AIR constraints + lookup / CTL checks) + quotient commitment + openings + "
                  "FRI; oracle = C/OpenMP restatement (NTT, Poseidon, Merkle, FRI) with the constraint program traced from "
                  "the Python restatement and interpreted per row" % log_n,
        "seconds": cpu_s, "stages_s": {k: round(v, 3) for k, v in stages.items()},
        "gpu_same_proof": {"seconds": gpu_s, "proofs_per_s": 1.0 / gpu_s, "stages_s": {k: round(v, 4) for k, v in gpu_stages.items()},
                           "ratio_to_this_oracle": cpu_s / gpu_s,
                           "ratio_note": "against THIS repository's oracle (textbook C/OpenMP NTT + Poseidon and a tape interpreter "
                                         "for the constraints), not against plonky2's AVX2 / rayon prover: a statement that the two "
                                         "proofs are the same work, not a speed claim"},
        "proofs_identical": bool(same),
    }


def synthetic_segment_traces(log_ns, dev, seed=1, cdk_erigon=False):
    """Random traces in HBM for the nine tables with every CTL / lookup *filter* column binary (one-hot op
    flags etc.): the helper-column kernels reject non-binary filters exactly like starky's debug assert.  Values are
    otherwise uniform 64-bit patterns (non-canonical representatives included).  cdk_erigon: ten tables (86-column
    Cpu, Poseidon)."""
    import torch
    from zk_evm_amd.all_stark import AllStark
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    out = []
    x = 1 if cdk_erigon else 0
    for t, (c, l) in enumerate(zip(AllStark((0, 0, 0, 0), cdk_erigon).table_columns, log_ns)):
        n = 1 << l
        tr = torch.randint(-(1 << 63), (1 << 63) - 1, (c, n), dtype=torch.int64, device=dev, generator=g)

        def binary(cols):
            for k in cols:
                tr[k] = torch.randint(0, 2, (n,), dtype=torch.int64, device=dev, generator=g)

        def one_hot(cols, extra=1):
            pick = torch.randint(0, len(cols) + extra, (n,), device=dev, generator=g)
            for i, k in enumerate(cols):
                tr[k] = (pick == i).to(torch.int64)
        if t == 0:
            one_hot(list(range(17)))                      # Arithmetic op flags + IS_RANGE_CHECK
        elif t == 1:
            one_hot(list(range(1, 33)))                   # BytePacking index_len
        elif t == 2:
            one_hot(list(range(6, 24 + x)))               # Cpu op flags
            binary(list(range(24 + x, 33 + x)) + [41 + x, 54 + x, 67 + x, 80 + x])
        elif t == 3:
            binary([0, 23])                               # Keccak first / last round flags
        elif t == 4:                                      # KeccakSponge: none / full block / final block of length ln
            kind = torch.randint(0, 3, (n,), device=dev, generator=g)
            ln = torch.randint(0, 136, (n,), device=dev, generator=g)
            tr[0] = (kind == 1).to(torch.int64)
            for i in range(136):
                tr[6 + i] = ((kind == 2) & (ln <= i)).to(torch.int64)
        elif t == 5:
            one_hot([0, 1, 2])                            # Logic ops
        elif t == 6:                                      # Memory
            binary([0, 22, 24, 26])
            one_hot([15, 16], 2)
            f = torch.randint(0, 2, (n,), dtype=torch.int64, device=dev, generator=g)
            tr[1] = f                                     # timestamp = timestamp_inv in {0,1}: mem_before filter binary
            tr[2] = f
        elif t == 9:                                      # Poseidon (cdk_erigon)
            one_hot(list(range(6, 14)))
            binary([319, 320, 321])
        else:
            binary([0])                                   # MemBefore / MemAfter filter
        out.append(tr)
    return out


def segment_committed_cells(log_ns, cdk_erigon=False):
    """(columns x rows) the segment commits: trace + auxiliary (lookup + CTL) + 4 quotient chunks per table."""
    from zk_evm_amd import all_stark as A
    from zk_evm_amd.segment import num_ctl_helpers_zs_all
    st = A.AllStark((0, 0, 0, 0), cdk_erigon)
    ctls = st.cross_table_lookups
    cells = 0
    for t in range(st.num_tables):
        aux = sum(num_ctl_helpers_zs_all(ctls, t, 2, 3)[:2]) + 2 * sum(l.num_helper_columns(3) for l in A.table_lookups(t))
        cells += (st.table_columns[t] + aux + 4) << log_ns[t]
    return cells


def measure_commit(ctx, dev, a, rank, steps, warmup):
    """BASELINE configs[1]: `steps` commits of one cols x 2^log_n trace; returns (elapsed_s, stage ms, trace)."""
    import torch
    from zk_evm_amd import PolynomialBatch
    n = 1 << a.log_n
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
    return trace, step


def commit_report(a, stage, ms_per_step):
    """Roofline pieces of one cols x 2^log_n commit from its HIP-event stage times (ms)."""
    n = 1 << a.log_n
    N = n << 1
    # dominant kernel: poseidon_hash_rows_kernel (one launch per commit). Algorithmic bytes:
    # read the LDE once (8*C*N) + write N 32-byte digests.
    dom_bytes = 8.0 * a.cols * N + 32.0 * N
    dom_ms = stage["leaf_hash"]
    achieved = dom_bytes / (dom_ms * 1e-3) / 1e9
    perms = N * ((a.cols + 7) // 8) if a.cols > 4 else 0
    commit_bytes = 32.0 * a.cols * n + 128.0 * n          # whole-commit algorithmic bytes
    ntt_bytes = 40.0 * a.cols * n
    ntt_ms = stage["ifft"] + stage["lde"]
    # HBM traffic and VALU instruction counts of the dominant kernel come from separate rocprofv3 --pmc passes
    # (tools/collect_pmc.sh), summarised in profiles/pmc_latest.json; they apply to the 116 x 2^20 Poseidon launch.
    traffic = None
    valu = None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    if os.path.exists(pmc_path) and a.cols == 116 and a.log_n == 20 and a.hasher == 0:
        try:
            pmc = json.load(open(pmc_path))
            traffic = pmc.get("leaf_hash_hbm_bytes_per_launch")
            insts = pmc.get("leaf_hash_valu_wave_insts_per_launch")
            if insts:
                # integer-issue roofline: every useful integer VALU op on gfx950 issues at ~4 cycles per wave64
                # per SIMD (profiles/r01_ubench_valu_issue_rates.txt)
                peak = 1024 * 2.4e9 / 4.0
                ach = insts / (dom_ms * 1e-3)
                valu = {"wave_insts_per_launch": insts, "achieved_wave_insts_per_s": ach,
                        "peak_wave_insts_per_s": peak, "frac": ach / peak,
                        "assumes": "1024 SIMDs x 2.4 GHz / 4 cycles per integer VALU wave-instruction",
                        "source": pmc.get("source"), "source_commit": pmc.get("git_commit"),
                        "measured_in_this_run": False}
        except Exception:
            traffic = None
    roof = {"bound": "hbm", "kernel": "poseidon_hash_rows_kernel" if a.hasher == 0 else "keccak_hash_rows_kernel",
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic, "ms_per_launch": dom_ms, "algorithmic_bytes": dom_bytes,
            "note": "kernel is integer-ALU bound (Poseidon), see DESIGN.md; permutations/s = %.3e"
                    % (perms / (dom_ms * 1e-3) if dom_ms else 0),
            "valu": valu}
    extra = {"stages_ms": stage,
             "ntt": {"achieved_GBs": ntt_bytes / (ntt_ms * 1e-3) / 1e9, "algorithmic_bytes": ntt_bytes,
                     "frac_of_hbm_peak": ntt_bytes / (ntt_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
             "commit": {"achieved_GBs": commit_bytes / (ms_per_step * 1e-3) / 1e9, "algorithmic_bytes": commit_bytes}}
    return roof, extra


def segments_in_flight(ctx, workers, per_worker, arena_peak, all_stark, cfg, traces, in_use, cdk_erigon):
    """W segments in flight on this GPU through the product scheduler (zk_evm_amd/scheduler.py: one worker thread +
    Context + HIP stream per slot, one shared job queue); every job proves the resident traces."""
    import torch
    import zk_evm_amd.segment as sg
    from zk_evm_amd.scheduler import SegmentJob, SegmentScheduler
    ctx.mem_trim()                                    # the main ctx hands its idle slabs back; every worker grows its own
    free, total = torch.cuda.mem_get_info()
    need = workers * arena_peak
    if need > 0.9 * free:
        return {"skipped": f"{workers} arenas of {arena_peak / 1e9:.0f} GB do not fit in the {free / 1e9:.0f} GB free"}

    def job():
        return SegmentJob(lambda dev: traces, in_use, sg.PublicValues(burn_addr=1 if cdk_erigon else None))
    el = 0.0
    with SegmentScheduler(all_stark, cfg, [ctx.device], workers) as sch:
        for n in (2, per_worker):                      # warm-up (arena growth), then the timed round
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            sch.map([job() for _ in range(workers * n)])
            el = time.perf_counter() - t0
        errors = [e for st in sch.stats for e in st.errors]
    if errors:
        return {"error": errors[0]}
    return {"workers_per_gpu": workers, "proofs": workers * per_worker, "value": workers * per_worker / el,
            "unit": "segment proofs/s", "note": "SegmentScheduler: one Context + stream + worker thread per in-flight "
                                                "segment, one job queue, shared resident inputs"}


def realistic_profile(ctx, dev, a, all_stark, cfg, steps=4, in_flight=3):
    """Secondary object: the `north_star` shape -- per-table heights at the upper ends of the reference's own ranges
    (scripts/prove_stdio.rs:89-101: Arithmetic 2^17, BytePacking 2^14, Cpu 2^19, Keccak 2^17, KeccakSponge 2^13, Logic
    2^16, Memory 2^21, MemBefore / MemAfter 2^19) -- one segment at a time, and `in_flight` segments per GPU through the
    product scheduler."""
    import torch
    import zk_evm_amd.segment as sg
    from zk_evm_amd.scheduler import SegmentJob, SegmentScheduler
    n_tab = all_stark.num_tables
    log_ns = REALISTIC_LOG_NS + [14] * (n_tab - 9)
    traces = synthetic_segment_traces(log_ns, dev, seed=11, cdk_erigon=a.cdk_erigon)
    in_use = [True] * n_tab

    def pv():
        return sg.PublicValues(burn_addr=1 if a.cdk_erigon else None)
    sg.prove_with_traces(all_stark, cfg, traces, in_use, pv(), ctx=ctx)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        sg.prove_with_traces(all_stark, cfg, traces, in_use, pv(), ctx=ctx)
    torch.cuda.synchronize()
    single = (time.perf_counter() - t0) / steps
    peak = ctx.mem_stats()["peak_in_use"]
    out = {"log_ns": log_ns, "steps": steps, "single": {"value": 1.0 / single, "unit": "segment proofs/s", "ms_per_proof": 1e3 * single},
           "trace_GB": 8.0 * sum(c << l for c, l in zip(all_stark.table_columns, log_ns)) / 1e9}
    try:
        with SegmentScheduler(all_stark, cfg, [ctx.device], in_flight) as sch:
            mk = lambda: SegmentJob(lambda d: traces, in_use, pv())
            sch.map([mk() for _ in range(2 * in_flight)])                # warm-up: every worker grows its arena
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            sch.map([mk() for _ in range(steps * in_flight)])
            el = time.perf_counter() - t0
        out["in_flight"] = {"workers_per_gpu": in_flight, "value": steps * in_flight / el, "unit": "segment proofs/s"}
    except Exception as e:
        out["in_flight"] = {"error": repr(e)}
    # ---- one segment carried through its recursion layer (fixed_recursive_verifier.rs:2053-2160, 3167-3179) -----------------
    # prove_segment = the STARK, then per table a StarkWrapperCircuit proof and its shrink() chain down to 2^13 rows, then the
    # root circuit.  Modelled as 35 PLONK proofs: per table one wrapper proof at 2^14 rows and two shrinking proofs at 2^13
    # (27), the root at 2^14 and seven more 2^13 steps for the larger tables.  The chains of different tables are
    # independent, a chain's own steps are serial: step k of all nine tables is ONE zk_plonk_prove_batch call (synthetic
    # circuits carrying all fourteen gate kinds; witness generation is the Rust side's and is not in this number).
    try:
        import zk_evm_amd
        import zk_evm_amd.plonk as zp
        g = torch.Generator(device=dev)
        g.manual_seed(123)
        circ = {}
        ctx2 = zk_evm_amd.Context(ctx.device)                              # the recursion layer's own context and stream
        st2 = torch.cuda.Stream(device=dev)
        for lb in (13, 14):
            cs, wires = plonk_synthetic_circuit(dev, lb, g)
            torch.cuda.synchronize()
            with torch.cuda.stream(st2):
                circ[lb] = (zp.CircuitData(zp.CircuitConfig(), lb, PLONK_RECURSION_GATES, 4, cs, PLONK_K_IS, [1, 2, 3, 4], 123, ctx=ctx2), wires)
        plan = [(14, 9), (13, 9), (13, 9), (13, 7), (14, 1)]               # (circuit rows, proofs in the batch), in chain order

        def recursion():
            n = 0
            with torch.cuda.stream(st2):                                   # (thread-local: whichever thread runs this)
                for lb, k in plan:
                    cd, wires = circ[lb]
                    cd.prove_batch([wires] * k, [[5, 6, 7]] * k, in_flight=min(k, 6))
                    n += k
            return n

        def stark():
            sg.prove_with_traces(all_stark, cfg, traces, in_use, pv(), ctx=ctx)
        stark(); n_rec = recursion()                                         # noqa: E702  (warm: worker contexts, arenas)
        torch.cuda.synchronize()
        reps = 3
        t0 = time.perf_counter()
        for _ in range(reps):
            stark()
        t_stark = (time.perf_counter() - t0) / reps
        t0 = time.perf_counter()
        for _ in range(reps):
            recursion()
        t_rec = (time.perf_counter() - t0) / reps
        # pipelined: the STARK of segment k + 1 (this thread, ctx) beside the recursion of segment k (a second thread)
        import threading
        t0 = time.perf_counter()
        th = None
        for _ in range(reps + 1):
            stark()
            if th is not None:
                th.join()
            th = threading.Thread(target=recursion)
            th.start()
        th.join()
        t_pipe = (time.perf_counter() - t0) / (reps + 1)
        out["segment_with_recursion"] = {
            "plonk_proofs_per_segment": n_rec, "stark_ms": 1e3 * t_stark, "recursion_ms": 1e3 * t_rec,
            "serial": {"value": 1.0 / (t_stark + t_rec), "unit": "segments/s"},
            "pipelined": {"value": 1.0 / t_pipe, "unit": "segments/s",
                          "note": "the next segment's STARK runs beside this segment's recursion proofs (two host threads)"},
            "note": "realistic table heights; 35 synthetic-circuit PLONK proofs per segment in five zk_plonk_prove_batch calls "
                    "(chain order); circuit witness generation (Rust) not included"}
        for cd, _ in circ.values():
            cd.free()
        ctx2.close()
    except Exception as e:
        out["segment_with_recursion"] = {"error": repr(e)}
    del traces
    torch.cuda.empty_cache()
    return out


def h2d_profile(dev, trace_bytes, step_s, step_fn=None):
    """Secondary object: host->device bandwidth measured here (1 GiB, pageable and pinned) and what uploading the step's
    traces costs -- `value` itself starts with the traces resident in HBM (bench contract).  With `step_fn`, the overlapped
    case is MEASURED: a second stream uploads one segment's worth of trace bytes from pinned host memory into a second
    device buffer while `step_fn` proves the resident segment."""
    import torch
    n = 1 << 27                                             # 1 GiB of int64
    dst = torch.empty(n, dtype=torch.int64, device=dev)
    res = {}
    pinned = None
    for kind in ("pageable", "pinned"):
        try:
            src = torch.ones(n, dtype=torch.int64)
            if kind == "pinned":
                src = src.pin_memory()
                pinned = src
            dst.copy_(src)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                dst.copy_(src, non_blocking=True)
            torch.cuda.synchronize()
            res[kind + "_GBs"] = 3 * 8.0 * n / (time.perf_counter() - t0) / 1e9
            del src
        except Exception as e:
            res[kind + "_error"] = repr(e)
    del dst
    bw = max([v for k, v in res.items() if k.endswith("_GBs")] or [0.0])
    if bw > 0:
        up = trace_bytes / 1e9 / bw
        res.update(trace_GB=trace_bytes / 1e9, upload_s=up,
                   serial_upload_then_prove={"value": 1.0 / (step_s + up), "unit": "segment proofs/s"},
                   overlapped_upload_modelled={"value": 1.0 / max(step_s, up), "unit": "segment proofs/s",
                                               "note": "arithmetic only: 1 / max(proof time, upload time)"},
                   note="the eight non-Cpu tables can be generated on the device from operation logs (zk_*_generate_trace), "
                        "which leaves only the Cpu rows and the logs on PCIe")
    if step_fn is not None and pinned is not None:
        try:
            total = int(trace_bytes) // 8
            second = torch.empty(total, dtype=torch.int64, device=dev)          # where segment k+1's traces land
            upl = torch.cuda.Stream(device=dev)

            def upload():
                with torch.cuda.stream(upl):
                    for off in range(0, total, n):
                        m = min(n, total - off)
                        second[off: off + m].copy_(pinned[:m], non_blocking=True)
            reps = 3
            upload(); step_fn(); upl.synchronize(); torch.cuda.synchronize()  # noqa: E702  (warm)
            t0 = time.perf_counter()
            for _ in range(reps):
                upload()
                step_fn()
                upl.synchronize()
            torch.cuda.synchronize()
            el = (time.perf_counter() - t0) / reps
            res["overlapped_upload"] = {
                "value": 1.0 / el, "unit": "segment proofs/s", "s_per_segment": el, "measured": True,
                "note": "measured: %.1f GB from pinned host memory on a second stream into a second device buffer while the "
                        "resident segment is proven (%d repetitions); proof alone %.3f s, upload alone %.3f s"
                        % (trace_bytes / 1e9, reps, step_s, trace_bytes / 1e9 / bw)}
            del second
        except Exception as e:
            res["overlapped_upload"] = {"error": repr(e)}
    return res


def from_logs_profile(ctx, dev, all_stark, cfg, reps=3):
    """Secondary object: operation logs -> witness tables ON THE DEVICE -> segment proof, i.e. the
    path that replaces the 3.9 GB trace upload of `realistic` by the upload of the interpreter's compact logs
    (`witness/traces.rs:135-262` `Traces::into_tables`).  Synthetic logs in the C ABI's packed record layouts, sized so
    the tables come out at the `prove_stdio.rs` heights (Arithmetic 2^17, BytePacking 2^14, Cpu 2^19, Keccak 2^17,
    KeccakSponge 2^13, Logic 2^16, Memory 2^21, MemBefore 2^19); the Cpu rows are the interpreter's own output and are
    uploaded as they are.  Logs are random, not an execution: the generators' and the prover's work does not depend on it."""
    import numpy as np
    import torch
    import zk_evm_amd.segment as sg
    import zk_evm_amd.tracegen as tg
    rng = np.random.default_rng(7)
    u64 = lambda *shape: rng.integers(0, 1 << 64, size=shape, dtype=np.uint64)
    tr = tg.Traces()
    n_ar = 120000                                           # one-row kinds only: 120 000 rows -> 2^17
    ar = np.zeros((n_ar, 18), dtype=np.uint64)
    ar[:, 0] = rng.choice([tg.ARITH_ADD, tg.ARITH_MUL, tg.ARITH_SUB, tg.ARITH_LT, tg.ARITH_GT], size=n_ar)
    ar[:, 2:10] = u64(n_ar, 8)
    tr.arithmetic_ops = ar
    n_bp = 15000
    bp = np.zeros((n_bp, 10), dtype=np.uint64)
    bp[:, 0] = rng.integers(0, 2, size=n_bp)
    bp[:, 2], bp[:, 3], bp[:, 4], bp[:, 5] = 1, rng.integers(0, 1 << 16, size=n_bp), np.arange(2, 2 + n_bp), 32
    bp[:, 6:10] = u64(n_bp, 4)
    tr.byte_packing_ops = bp
    n_cpu_cols = all_stark.table_columns[2]
    cpu = u64(1 << 19, n_cpu_cols) >> np.uint64(1)
    pick = rng.integers(0, 19, size=1 << 19)                # CTL filter columns binary, as in synthetic_segment_traces
    for i, k in enumerate(range(6, 24)):
        cpu[:, k] = pick == i
    for k in list(range(24, 33)) + [41, 54, 67, 80]:
        cpu[:, k] = rng.integers(0, 2, size=1 << 19)
    tr.cpu = torch.from_numpy(cpu.view(np.int64))
    n_k = 5400                                              # 24 rows per permutation -> 2^17
    tr.keccak_inputs = (u64(n_k, 25), np.arange(2, 2 + n_k, dtype=np.uint64))
    tr.keccak_sponge_ops = [((0, 2, int(a)), 2 + i, rng.bytes(int(l))) for i, (a, l) in
                            enumerate(zip(rng.integers(0, 1 << 16, size=3500), rng.integers(1, 270, size=3500)))]
    n_lg = 60000
    lg = np.zeros((n_lg, 9), dtype=np.uint64)
    lg[:, 0] = rng.integers(0, 3, size=n_lg)
    lg[:, 1:9] = u64(n_lg, 8)
    tr.logic_ops = lg
    n_bef, n_ops = 400000, 1400000                          # Memory table: initial values + operations + gap rows -> 2^21
    bef = np.zeros((n_bef, 7), dtype=np.uint64)
    bef[:, 1], bef[:, 2] = np.arange(n_bef) // 100000, np.arange(n_bef) % 100000
    bef[:, 3:7] = u64(n_bef, 4)
    mo = np.zeros((n_ops, 9), dtype=np.uint64)
    mo[:, 0] = rng.integers(0, 2, size=n_ops).astype(np.uint64) | np.uint64(2)
    mo[:, 1] = 2 + np.arange(n_ops) // 4
    mo[:, 3], mo[:, 4] = rng.integers(0, 4, size=n_ops), rng.integers(0, 100000, size=n_ops)
    mo[:, 5:9] = u64(n_ops, 4)
    tr.memory_ops = mo
    log_bytes = ar.nbytes + bp.nbytes + tr.cpu.numel() * 8 + tr.keccak_inputs[0].nbytes + lg.nbytes + bef.nbytes + mo.nbytes + \
        sum(len(d) for _, _, d in tr.keccak_sponge_ops)
    gen, prove, tables = [], [], None
    for _ in range(reps):
        tables = None
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tables, _ = tr.into_tables(all_stark, bef, [], cfg, device=ctx.device, ctx=ctx, packed_final=True)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        sg.prove_with_traces(all_stark, cfg, tables, [True] * 9, sg.PublicValues(), ctx=ctx)
        torch.cuda.synchronize()
        gen.append(t1 - t0)
        prove.append(time.perf_counter() - t1)
    g, pr = min(gen[1:]), min(prove[1:])
    heights = [int(t.shape[1]).bit_length() - 1 for t in tables]
    cells = sum(int(t.shape[0]) * int(t.shape[1]) for t in tables)
    return {"table_heights_log2": heights, "log_GB": log_bytes / 1e9, "cpu_rows_GB": tr.cpu.numel() * 8 / 1e9,
            "trace_GB": cells * 8 / 1e9, "into_tables_ms": 1e3 * g, "prove_ms": 1e3 * pr,
            "value": 1.0 / (g + pr), "unit": "segment proofs/s",
            "note": "logs (pageable host memory, C-ABI record layouts) -> zk_*_generate_trace / zk_memory_trace_* on the device "
                    "-> prove_with_traces, serial; the Cpu table's rows are uploaded and transposed, every other table is built "
                    "in HBM from its log"}


# the gate set of the recursion circuits, as zk_plonk_gate records (kind, param, selector column,
# selector group): fourteen kinds sorted by (degree, id), four selector groups under max degree 9
PLONK_RECURSION_GATES = [(0, 0, 0, 0, 7), (1, 2, 0, 0, 7), (12, 0, 0, 0, 7), (2, 0, 0, 0, 7), (6, 63, 0, 0, 7), (8, 32, 0, 0, 7),
                         (7, 43, 0, 0, 7), (4, 10, 1, 7, 11), (3, 20, 1, 7, 11), (5, 13, 1, 7, 11), (9, 66, 1, 7, 11),
                         (11, 4 | 4 << 8 | 2 << 16, 2, 11, 13), (13, 4 | 6 << 8, 2, 11, 13), (10, 0, 3, 13, 14)]
PLONK_K_IS = [pow(14293326489335486720, i, 0xFFFFFFFF00000001) for i in range(80)]   # get_unique_coset_shifts(80)


def plonk_synthetic_circuit(dev, lb, g):
    """(constants ++ sigmas [4 selectors + 2 constants + 80][n], wires [135][n]) of a synthetic 2^lb-row circuit over
    PLONK_RECURSION_GATES: every row one of the fourteen gates at random (selector columns = the row's gate index in its
    group's column, UNUSED_SELECTOR elsewhere), everything else uniform."""
    import torch
    gates, n_sel, n = PLONK_RECURSION_GATES, 4, 1 << lb
    cs = torch.randint(-(1 << 63), (1 << 63) - 1, (n_sel + 2 + 80, n), dtype=torch.int64, device=dev, generator=g)
    # rows per gate kind ~ a recursive STARK / PLONK verifier circuit (an ESTIMATE from the builder calls under
    # recursive_verifier.rs:336-349 and plonky2's FRI verifier gadget: Merkle paths and challenger = PoseidonGate rows
    # dominate, then extension arithmetic for the alpha-reductions, bit decompositions, random accesses, one coset
    # interpolation per fold).  The prover's time does not depend on these frequencies -- plonky2 and this library evaluate
    # every gate kind of the circuit at every point and apply the selector filter -- only on WHICH kinds are present.
    census = {0: 2, 1: 1, 12: 1, 2: 0.1, 6: 5, 8: 3, 7: 3, 4: 12, 3: 8, 5: 3, 9: 1, 11: 5, 13: 2, 10: 54}
    w = torch.tensor([census[q[0]] for q in gates], dtype=torch.float32, device=dev)
    gate_of_row = torch.multinomial(w, n, replacement=True, generator=g).to(torch.int64)
    sel_of_gate = torch.tensor([q[2] for q in gates], dtype=torch.int64, device=dev)
    for sidx in range(n_sel):
        cs[sidx] = torch.where(sel_of_gate[gate_of_row] == sidx, gate_of_row, torch.full_like(gate_of_row, 0xFFFFFFFF))
    wires = torch.randint(-(1 << 63), (1 << 63) - 1, (135, n), dtype=torch.int64, device=dev, generator=g)
    return cs, wires


def plonk_recursion_profile(ctx, dev, with_cpu, sizes=(12, 13, 14), reps=8):
    """Secondary object: the recursion layer's PLONK proofs -- `CircuitConfig::
    standard_recursion_config()` (135 wires, 80 routed, FRI rate_bits 3, 28 queries, 16 PoW bits), circuits of 2^12 ..
    2^14 rows (THRESHOLD_DEGREE_BITS = 13, fixed_recursive_verifier.rs:69), `reps` proofs per size = the chain of
    `shrink()` proofs the reference runs per table.  Synthetic circuit data (random constants / sigmas / wires with valid
    selector values: the prover's work does not depend on satisfiability).  With `with_cpu` the oracle's restatement of
    plonky2's prove() is timed once at 2^13 on the same data and the two proofs are compared word for word."""
    import numpy as np
    import torch
    import zk_evm_amd.plonk as zp
    P = 0xFFFFFFFF00000001
    gates, n_sel, n_gate_constraints = PLONK_RECURSION_GATES, 4, 123   # PoseidonGate's 123 constraints are the maximum
    out = {"config": "standard_recursion_config, fourteen gate kinds {Noop, Constant, PoseidonMds, PublicInput, BaseSum, "
                     "ReducingExtension, Reducing, ArithmeticExtension, Arithmetic, MulExtension, Exponentiation, RandomAccess, "
                     "CosetInterpolation, Poseidon}, rows dealt to the kinds by an estimated verifier-circuit census (54 % "
                     "Poseidon; the prover's cost depends on which kinds are present, not on their row counts)",
           "proofs_per_size": reps, "sizes": {}}
    g = torch.Generator(device=dev)
    g.manual_seed(99)
    k_is = PLONK_K_IS
    for lb in sizes:
        cs, wires = plonk_synthetic_circuit(dev, lb, g)
        cd = zp.CircuitData(zp.CircuitConfig(), lb, gates, n_sel, cs, k_is, [1, 2, 3, 4], n_gate_constraints, ctx=ctx)
        pis = [5, 6, 7]
        pr = cd.prove(wires, pis)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            pr = cd.prove(wires, pis)
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / reps
        out["sizes"]["2^%d" % lb] = {"ms_per_proof": 1e3 * el, "proofs_per_s": 1.0 / el,
                                     "stages_ms": {k: round(v, 3) for k, v in pr.stage_ms.items()},
                                     "proof_words": int(pr.opening_proof.size)}
        if with_cpu and lb == 13:
            try:
                import tests.oracle_lib as ol
                from oracle import plonk as PK
                o = ol.load_oracle()
                ol.setup_fri_api(o)
                host = cs.cpu().numpy().view(np.uint64) % np.uint64(P)
                og = sorted([PK.NoopGate(), PK.ConstantGate(2), PK.PublicInputGate(), PK.ArithmeticGate(20),
                             PK.ArithmeticExtensionGate(10), PK.MulExtensionGate(13), PK.BaseSumGate(63), PK.ReducingGate(43),
                             PK.ReducingExtensionGate(32), PK.ExponentiationGate(66), PK.PoseidonGate(),
                             PK.RandomAccessGate(4, 4, 2), PK.PoseidonMdsGate(), PK.CosetInterpolationGate(4, 8)],
                            key=lambda q: (q.degree, q.id))
                assert [(q.KIND, q.PARAM) for q in og] == [(q[0], q[1]) for q in gates]
                circ = PK.Circuit(PK.CircuitConfig(), lb, og, [q[2] for q in gates], sorted({(q[3], q[4]) for q in gates}),
                                  n_sel, np.ascontiguousarray(host[:n_sel + 2]), np.ascontiguousarray(host[n_sel + 2:]), k_is,
                                  [1, 2, 3, 4])
                PK.commit_circuit(o, circ)
                tm = {}
                t0 = time.perf_counter()
                ep = PK.prove(o, ol, circ, wires.cpu().numpy().view(np.uint64), pis, timing=tm)
                cpu_s = time.perf_counter() - t0
                out["cpu_2^13"] = {"seconds": cpu_s, "stages_s": {k: round(v, 3) for k, v in tm.items()}, "kind": "port",
                                   "cores": ol.usable_cores(), "speedup": cpu_s / el,
                                   "proofs_identical": bool(np.array_equal(ep["fri"], pr.opening_proof) and
                                                            np.array_equal(ep["openings"], pr.openings.reshape(-1)))}
            except Exception as e:
                out["cpu_2^13"] = {"error": repr(e)}
        if lb == 13:
            # the per-table shrink chains of one segment are independent of each other: W proofs in flight on this GPU,
            # one worker thread + Context + HIP stream + CircuitData each (the scheduler's slot model)
            try:
                import threading
                import zk_evm_amd

                def run_in_flight(W, per):
                    errs = []
                    bar = threading.Barrier(W + 1)

                    def worker(k):
                        try:
                            st = torch.cuda.Stream()
                            with torch.cuda.stream(st):
                                c2 = zk_evm_amd.Context(ctx.device)
                                d2 = zp.CircuitData(zp.CircuitConfig(), lb, gates, n_sel, cs, k_is, [1, 2, 3, 4], n_gate_constraints, ctx=c2)
                                d2.prove(wires, pis)
                                bar.wait()
                                for _ in range(per):
                                    d2.prove(wires, pis)
                                st.synchronize()
                                bar.wait()
                                d2.free()
                                c2.close()
                        except Exception as e:           # pragma: no cover
                            errs.append(repr(e))
                            bar.abort()
                    th = [threading.Thread(target=worker, args=(k,)) for k in range(W)]
                    for t in th:
                        t.start()
                    bar.wait()
                    t0 = time.perf_counter()
                    bar.wait()
                    elw = time.perf_counter() - t0
                    for t in th:
                        t.join()
                    if errs:
                        return {"error": errs[0]}
                    return {"workers_per_gpu": W, "proofs_per_s": W * per / elw, "ms_per_proof_effective": 1e3 * elw / (W * per)}
                runs = [run_in_flight(W, 2 * reps) for W in (4, 8)]
                ok = [r for r in runs if "error" not in r]
                out["in_flight_2^13"] = dict(max(ok, key=lambda r: r["proofs_per_s"]), tried=runs) if ok else runs[0]
            except Exception as e:
                out["in_flight_2^13"] = {"error": repr(e)}
            # the same from ONE caller: zk_plonk_prove_batch keeps the proofs in flight inside the library
            try:
                K, best = 48, None
                tried = []
                for W in (4, 6, 8):
                    cd.prove_batch([wires] * W, [pis] * W, in_flight=W)          # worker contexts, arenas
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    prs = cd.prove_batch([wires] * K, [pis] * K, in_flight=W)
                    elb = time.perf_counter() - t0
                    same = all(np.array_equal(q.opening_proof, pr.opening_proof) for q in prs)
                    r = {"in_flight": W, "proofs": K, "proofs_per_s": K / elb, "ms_per_proof_effective": 1e3 * elb / K,
                         "proofs_identical_to_single": bool(same)}
                    tried.append(r)
                    if best is None or r["proofs_per_s"] > best["proofs_per_s"]:
                        best = r
                out["batch_2^13"] = dict(best, tried=tried, note="one call of zk_plonk_prove_batch from one thread")
            except Exception as e:
                out["batch_2^13"] = {"error": repr(e)}
        cd.free()
        del cs, wires
    return out


KERNEL_CLASSES = (   # (class, substring of the rocprofv3 kernel name)
    ("leaf_hash", "hash_rows_kernel<false>"), ("leaf_hash_coop", "hash_rows_coop_kernel"),
    ("ntt_coeffs_to_values", "ntt_pass_kernel<true"), ("ntt_values_to_coeffs", "ntt_pass_kernel<false"),
    ("merkle_levels", "merkle_level"), ("fri_combine", "fri_combine_kernel"), ("openings", "eval_columns_partial_kernel"),
    ("helper_columns", "helper_cols_kernel"), ("lookup_singles", "lookup_singles_kernel"),
    ("quotient_checks", "quotient_checks_kernel"))


def kernel_class(name):
    import re
    m = re.search(r"quotient_kernel(?:_heavy)?<(\w+)", name)
    if m:
        return "quotient:" + m.group(1)
    if "quotient_arith_kernel" in name:            # the LDS-tiled form of the Arithmetic AIR (arith_quotient.cuh)
        return "quotient:AirArithmetic"
    for cls, sub in KERNEL_CLASSES:
        if sub in name:
            return cls
    return None


def collect_kernel_counters(a, timeout_s=300, passes=("trace", "fetch", "write"), keep_dir=None):
    """Per-kernel-class time and counters of ONE segment of this run's workload, measured now: child passes of this same
    script (`--pmc-child`) under rocprofv3 -- `trace`: --kernel-trace only (durations, unperturbed by counter collection);
    `fetch`: --pmc FETCH_SIZE SQ_INSTS_VALU GRBM_GUI_ACTIVE; `write`: --pmc WRITE_SIZE (FETCH_SIZE and WRITE_SIZE cannot
    share a pass; counters only, no sys / hip / memory tracing).  Returns {class: {launches, ms, fetch_kib, write_kib,
    valu_wave_insts, gui_active}} summed over the segment's launches, or None if rocprofv3 is absent or a pass fails.
    FETCH_SIZE / WRITE_SIZE are rocprofv3's KiB as reported (the caller applies the gfx950 read correction)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    table, seq = {}, {}
    spec = {"trace": [], "fetch": ["FETCH_SIZE", "SQ_INSTS_VALU", "GRBM_GUI_ACTIVE"], "write": ["WRITE_SIZE"]}
    key = {"FETCH_SIZE": "fetch_kib", "WRITE_SIZE": "write_kib", "SQ_INSTS_VALU": "valu_wave_insts", "GRBM_GUI_ACTIVE": "gui_active"}
    for name in passes:
        d = tempfile.mkdtemp(prefix="zkpmc_", dir="/tmp")
        try:
            cmd = [exe, "--kernel-trace"] + (["--pmc", *spec[name]] if spec[name] else []) + [
                "--output-format", "csv", "-d", d, "-o", name, "--", sys.executable, os.path.abspath(__file__), "--pmc-child",
                "--no-pmc", "--hasher", str(a.hasher), "--log-n", str(a.log_n)]
            if a.log_ns:
                cmd += ["--log-ns", a.log_ns]
            if a.cdk_erigon:
                cmd += ["--cdk-erigon"]
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), timeout=timeout_s, capture_output=True)
            if r.returncode != 0:
                return None
            if name == "trace":
                for path in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
                    rows = sorted(csv.DictReader(open(path)), key=lambda r: float(r["Start_Timestamp"]))
                    for row in rows:
                        cls = kernel_class(row.get("Kernel_Name", ""))
                        if cls:
                            ms = (float(row["End_Timestamp"]) - float(row["Start_Timestamp"])) * 1e-6
                            e = table.setdefault(cls, {"launches": 0, "ms": 0.0})
                            e["launches"] += 1
                            e["ms"] += ms
                            if cls.startswith("quotient"):          # per dispatch, in launch order (= table order)
                                seq.setdefault("ms", []).append((cls, ms))
            else:
                for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                    for row in csv.DictReader(open(path)):
                        cls = kernel_class(row.get("Kernel_Name", ""))
                        if cls and row["Counter_Name"] in key:
                            e = table.setdefault(cls, {"launches": 0, "ms": 0.0})
                            e[key[row["Counter_Name"]]] = e.get(key[row["Counter_Name"]], 0.0) + float(row["Counter_Value"])
                            if row["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
                                e["n_" + key[row["Counter_Name"]]] = e.get("n_" + key[row["Counter_Name"]], 0) + 1
                                if cls.startswith("quotient"):
                                    seq.setdefault(key[row["Counter_Name"]], []).append(
                                        (int(row.get("Dispatch_Id", 0)), cls, float(row["Counter_Value"])))
            if keep_dir:
                shutil.copytree(d, os.path.join(keep_dir, name), dirs_exist_ok=True)
        except Exception:
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    if table and seq:
        table["_quotient_sequence"] = {"ms": seq.get("ms", []),
                                       "fetch_kib": [(c, v) for _, c, v in sorted(seq.get("fetch_kib", []))],
                                       "write_kib": [(c, v) for _, c, v in sorted(seq.get("write_kib", []))]}
    return table or None


def kernel_counter_report(kc, log_ns, all_stark, cfg, cdk_erigon):
    """`kernel_counters` of the bench line: per kernel class of one segment -- launches, ms, HBM traffic (FETCH_SIZE both
    as reported and with the guide's x2 read correction, + WRITE_SIZE) next to the ALGORITHMIC bytes where DESIGN.md
    defines them, and cycles per wave-instruction.  A reader sees traffic / algorithmic per stage without the CSVs."""
    import zk_evm_amd.segment as sg
    names = all_stark.table_names
    n_aux = {}
    for t in range(all_stark.num_tables):
        h, z, _ = sg.num_ctl_helpers_zs_all(all_stark.cross_table_lookups, t, cfg.num_challenges, all_stark.constraint_degree)
        lk = sum(cfg.num_challenges * (-(-len(l.columns) // (all_stark.constraint_degree - 1)) + 1) for l in all_stark.lookups[t])
        n_aux[t] = lk + h + z
    air_of = {"Arithmetic": "AirArithmetic", "BytePacking": "AirBytePacking", "Cpu": "AirCpuT", "Keccak": "AirKeccak",
              "KeccakSponge": "AirKeccakSponge", "Logic": "AirLogic", "Memory": "AirMemory", "Poseidon": "AirPoseidon"}
    alg = {}
    for t, nm in enumerate(names):
        cls = "quotient:" + air_of.get(nm, "AirMemContinuation")
        c = all_stark.table_columns[t]
        # reads (C + A) LDE columns once at each of the 2n coset points, writes 2 challenge values per point
        alg[cls] = alg.get(cls, 0.0) + 8.0 * (c + n_aux[t]) * (2 << log_ns[t]) + 16.0 * (2 << log_ns[t])
    rep = {}
    qseq = kc.pop("_quotient_sequence", None)
    for cls, e in sorted(kc.items(), key=lambda kv: -kv[1].get("ms", 0.0)):
        r = {"launches": e["launches"], "ms": e["ms"]}
        if e.get("n_fetch_kib") and e.get("n_write_kib"):
            r["fetch_bytes_reported"] = e["fetch_kib"] * 1024.0
            r["write_bytes"] = e["write_kib"] * 1024.0
            r["traffic_bytes"] = (2.0 * e["fetch_kib"] + e["write_kib"]) * 1024.0
            if e["ms"] > 0:
                r["traffic_GBs"] = r["traffic_bytes"] / e["ms"] / 1e6
        if e.get("valu_wave_insts") and e.get("gui_active"):
            r["cycles_per_wave_instruction"] = e["gui_active"] / 8.0 * 1024.0 / e["valu_wave_insts"]
        rep[cls] = r
    # The quotient of a table is TWO launches -- its AIR kernel, then the lookup / CTL checks kernel (when it has any) -- in
    # table order; their traffic together is compared with the table's algorithmic bytes 8 (C + A) 2n + 16 * 2n.
    if qseq and qseq["ms"]:
        def per_table(items):
            out, cur = [], None
            for cls, v in items:
                if cls != "quotient_checks":
                    cur = [cls, v, 0.0]
                    out.append(cur)
                elif cur is not None:
                    cur[2] += v
            return out
        ms, fe, wr = per_table(qseq["ms"]), per_table(qseq["fetch_kib"]), per_table(qseq["write_kib"])
        live = [t for t in range(all_stark.num_tables)]
        tabs = {}
        if len(ms) == len(live) and len(fe) == len(live) and len(wr) == len(live):
            for k, t in enumerate(live):
                c = all_stark.table_columns[t]
                algb = 8.0 * (c + n_aux[t]) * (2 << log_ns[t]) + 16.0 * (2 << log_ns[t])
                traffic = (2.0 * (fe[k][1] + fe[k][2]) + wr[k][1] + wr[k][2]) * 1024.0
                tabs[names[t]] = {"air_kernel": ms[k][0], "air_ms": ms[k][1], "checks_ms": ms[k][2], "algorithmic_bytes": algb,
                                  "traffic_bytes": traffic, "traffic_over_algorithmic": traffic / algb,
                                  "reported_over_algorithmic": ((fe[k][1] + fe[k][2]) + wr[k][1] + wr[k][2]) * 1024.0 / algb}
            rep["quotient_per_table"] = tabs
            rep["quotient_ms_total"] = sum(m[1] + m[2] for m in ms)
    rep["_note"] = ("one segment of this workload under rocprofv3, this run: `ms` from a --kernel-trace-only pass; traffic_bytes = "
                    "2 x FETCH_SIZE (gfx950 read correction, MI355X_MICROARCH.md) + WRITE_SIZE; fetch_bytes_reported is FETCH_SIZE "
                    "as rocprofv3 prints it; algorithmic_bytes (quotients) = 8 (C + A) 2n read + 16 * 2n written")
    return rep


def dist_selftest(rank, world, backend):
    """Run the tensor collectives of zk_evm_amd/collectives.py + sharding.gather_caps once through the live process group
    and check what comes back.  Under `--dist-backend nccl` this is RCCL moving device tensors."""
    import numpy as np
    from zk_evm_amd import collectives as co
    from zk_evm_amd.sharding import gather_caps
    res = {"backend": backend, "world": world, "payload_device": str(co.device_for())}
    try:
        n_tab = 9
        mine = {t: np.full((16, 4), 1000 * t + 7, dtype=np.uint64) for t in range(n_tab) if t % world == rank}
        caps = gather_caps(mine, n_tab, 16)
        assert all(int(caps[t][3, 2]) == 1000 * t + 7 for t in range(n_tab))
        co.agree(None, "selftest")
        st = co.broadcast_words(np.arange(32, dtype=np.uint64) + 5 if rank == world - 1 else None, 32, world - 1)
        assert int(st[31]) == 36
        parts = co.gather_varlen_words(np.arange(10 + rank, dtype=np.uint64) * (rank + 1))
        if rank == 0:
            assert [p.size for p in parts] == [10 + r for r in range(world)] and all(int(p[-1]) == (9 + r) * (r + 1) for r, p in enumerate(parts))
        res["ok"] = True
        res["collectives"] = ["all_gather (caps)", "all_reduce MAX (status)", "broadcast (challenger state)", "gather (proof words)"]
    except Exception as e:
        res["ok"] = False
        res["error"] = repr(e)
    return res


def self_launch(a) -> int:
    """`python bench.py --gpus N` with no launcher around it: start the N ranks here -- one process per GPU, rank r on
    device r (or --devices), rendezvous on 127.0.0.1 -- and pass rank 0's JSON line through.  The children are this
    same script with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, i.e. exactly what `python -m
    torch.distributed.run --nproc-per-node N bench.py --gpus N` gives them."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), LOCAL_WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        while procs and rc == 0:
            for p in list(procs):
                try:
                    p.wait(timeout=0.5)
                except subprocess.TimeoutExpired:
                    continue
                procs.remove(p)
                rc = rc or p.returncode
    finally:
        for p in procs:                                   # a rank failed (or we were interrupted): stop the others
            p.kill()
            p.wait()
    return rc


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a))
    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    if a.devices:
        local_dev = [int(x) for x in a.devices.split(",")][local]
    else:
        local_dev = local
    # A single rank joins a process group of one as well (unless --no-dist-selftest / a non-default workload): the N = 1
    # line then exercises the same init / barrier / all-reduce / collectives as the N > 1 run, on RCCL.  At world size 1 a
    # failure to initialise is recorded in the line instead of ending the run.
    use_dist = world > 1 or a.force_dist or (a.workload == "segment" and not a.no_dist_selftest and not a.pmc_child)
    dist_error = None
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1 and "MASTER_PORT" not in os.environ:
            import socket
            sk = socket.socket()
            sk.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
            sk.close()
        try:
            if a.dist_backend == "nccl":
                torch.cuda.set_device(local_dev)
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local_dev}"))
            else:
                dist.init_process_group("gloo", rank=rank, world_size=world)
        except Exception as e:
            if world > 1 or a.force_dist:
                raise
            use_dist, dist_error = False, repr(e)
    torch.cuda.set_device(local_dev)
    dev = torch.device(f"cuda:{local_dev}")
    local = local_dev

    import zk_evm_amd
    ctx = zk_evm_amd.Context(local)
    ctx.use_torch_current_stream()
    hname = "poseidon" if a.hasher == 0 else "keccak25"

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if use_dist:
            tt = torch.tensor([x], dtype=torch.float64, device=dev if a.dist_backend == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            return float(tt.item())
        return x

    def timed_commits(steps, warmup):
        trace, step = measure_commit(ctx, dev, a, rank, steps, warmup)
        for _ in range(warmup):
            step()
        barrier()
        t0 = time.perf_counter()
        stage = {"ifft": 0.0, "lde": 0.0, "leaf_hash": 0.0, "tree": 0.0}
        for _ in range(steps):
            t = step()
            for k in stage:
                stage[k] += t[k]
        barrier()
        elapsed = max_over_ranks(time.perf_counter() - t0)
        for k in stage:
            stage[k] /= steps
        del trace
        return elapsed, stage

    out = None
    if a.workload == "commit":
        elapsed, stage = timed_commits(a.steps, a.warmup)
        if rank == 0:
            ms_per_step = 1e3 * elapsed / a.steps
            roof, extra = commit_report(a, stage, ms_per_step)
            out = {
                "metric": "ArithmeticStark-shaped 2^20-row trace commits/sec (Goldilocks iNTT + coset LDE + "
                          "Poseidon Merkle cap; BASELINE configs[1], the commit stage of segment STARK proofs/sec)",
                "value": world * a.steps / elapsed, "unit": "commits/s", "n_gpus": world, "steps": a.steps,
                "warmup": a.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "u64", "data": "synthetic",
                "config": {"workload": f"PolynomialBatch::from_values {a.cols} cols x 2^{a.log_n} rows, "
                                       f"rate_bits 1, cap_height 4, hasher {hname}",
                           "parallelism": f"{world} independent traces (one per GPU), no collective"},
                "roofline": roof}
            out.update(extra)
            if not a.no_cpu_baseline and world == 1:
                try:
                    out["cpu_baseline"] = cpu_baseline(a.cols, a.log_n, min(a.cpu_sample_log_n, a.log_n), a.hasher)
                except Exception as e:  # the oracle is only a reported baseline; never fatal
                    out["cpu_baseline"] = {"error": repr(e)}
    else:
        import zk_evm_amd.segment as sg
        from zk_evm_amd.all_stark import AllStark
        n_tab = 10 if a.cdk_erigon else 9
        log_ns = [a.log_n] * n_tab
        if a.log_ns:
            log_ns = (REALISTIC_LOG_NS + [14] * (n_tab - 9)) if a.log_ns == "realistic" else [int(x) for x in a.log_ns.split(",")]
            assert len(log_ns) == n_tab, "--log-ns takes one height per table"
        uniform = len(set(log_ns)) == 1
        traces = synthetic_segment_traces(log_ns, dev, seed=1 + rank, cdk_erigon=a.cdk_erigon)
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        cfg = zk_evm_amd.StarkConfig(hasher=a.hasher)   # == standard_fast_config() with the chosen hasher
        all_stark = AllStark((1, 2, 3, 4), a.cdk_erigon)   # kernel-label constants of the Cpu AIR: arbitrary for timing
        in_use = [True] * n_tab
        TABLE_COLUMNS = all_stark.table_columns

        def step(timing=None):
            return sg.prove_with_traces(all_stark, cfg, traces, in_use, sg.PublicValues(burn_addr=1 if a.cdk_erigon else None),
                                        ctx=ctx, timing=timing)
        if a.pmc_child:                       # one segment under rocprofv3 --pmc (collect_pmc_in_run), nothing printed
            step()
            torch.cuda.synchronize()
            return
        for _ in range(a.warmup):
            step()
        barrier()
        ctx.commit_totals(reset=True)
        ctx.side_commit_totals(reset=True)
        t0 = time.perf_counter()
        for _ in range(a.steps):
            proof = step()
        barrier()
        elapsed = max_over_ranks(time.perf_counter() - t0)
        tot = ctx.commit_totals(reset=True)
        side = ctx.side_commit_totals(reset=True)
        mem = ctx.mem_stats()
        if rank == 0:
            ms_per_step = 1e3 * elapsed / a.steps
            leaf_ms = tot["leaf_hash"]
            achieved = tot["leaf_hash_bytes"] / (leaf_ms * 1e-3) / 1e9
            ntt_ms = tot["ifft"] + tot["lde"]
            proof_words = sum(int(p.proof.opening_proof.size) for p in proof.multi_proof.stark_proofs if p is not None)
            timing = {}
            step(timing)                     # one extra, synchronised, untimed proof for the stage breakdown
            trace_bytes = 8.0 * sum(c << l for c, l in zip(TABLE_COLUMNS, log_ns))
            # HBM bytes per leaf-hash launch (mean over the 27 launches of a segment) from the rocprofv3 --pmc passes
            # on this same workload, summarised in profiles/pmc_latest.json["segment"]; only valid for the default shape
            seg_traffic, seg_valu = None, None
            try:
                pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json"))).get("segment")
                if pm and log_ns == [20] * 9 and a.hasher == 0:
                    seg_traffic = pm["leaf_hash_hbm_bytes_per_launch"]
                    ach = pm["leaf_hash_valu_wave_insts_per_launch"] / (leaf_ms / max(tot["commits"], 1) * 1e-3)
                    seg_valu = {"wave_insts_per_launch": pm["leaf_hash_valu_wave_insts_per_launch"],
                                "achieved_wave_insts_per_s": ach, "peak_wave_insts_per_s": 1024 * 2.4e9 / 4.0,
                                "frac": ach / (1024 * 2.4e9 / 4.0),
                                "assumes": "1024 SIMDs x 2.4 GHz / 4 cycles per wave-instruction (carry / mad / select class); "
                                           "v_mov / v_add_u32-class ops issue at ~2.4 cycles (profiles/r01_ubench_valu_issue_rates.txt), "
                                           "so a mix with many movs can exceed 1.0: the SIMDs are issue-saturated either way",
                                "source": pm["source"], "source_commit": pm.get("git_commit"),
                                "measured_in_this_run": False}
            except Exception:
                pass
            cells = segment_committed_cells(log_ns, a.cdk_erigon)
            out = {
                "metric": "segment STARK proofs/sec (2^20-row traces, all nine AllStark tables)" if log_ns == [20] * 9 else
                          "segment STARK proofs/sec (table heights 2^%s%s)" % (",".join(map(str, log_ns)), ", cdk_erigon" if a.cdk_erigon else ""),
                "value": world * a.steps / elapsed, "unit": "segment proofs/s", "n_gpus": world, "steps": a.steps,
                "warmup": a.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "u64", "data": "synthetic",
                "config": {"workload": f"prove_with_traces: full AllStark segment proof (BASELINE configs[2]), 9 tables x "
                                       f"2^{log_ns[0] if uniform else log_ns} rows ({sum(TABLE_COLUMNS)} trace columns, {trace_bytes / 1e9:.1f} GB), "
                                       f"10 CTLs + lookups, standard_fast_config, hasher {hname}",
                           "parallelism": f"{world} independent segments (one per GPU), no collective",
                           "committed_cells": cells, "proof_words": proof_words},
                "roofline": {"bound": "hbm", "limiting_resource": "integer VALU issue (the `valu` object), not HBM: `frac` is the "
                                                                    "contract's HBM fraction, `valu.frac` says how good the kernel is",
                             "kernel": "poseidon_hash_rows_kernel" if a.hasher == 0 else "keccak_hash_rows_kernel",
                             "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                             "traffic": seg_traffic,
                             "traffic_source": "rocprofv3 --pmc passes committed under profiles/ (pmc_latest.json), not collected "
                                               "in this run" if seg_traffic else None,
                             "launches": tot["commits"], "ms_per_launch": leaf_ms / max(tot["commits"], 1),
                             "ms_per_step": leaf_ms / a.steps, "share_of_step": leaf_ms / a.steps / ms_per_step,
                             "algorithmic_bytes": tot["leaf_hash_bytes"] / max(tot["commits"], 1),
                             "permutations_per_launch": tot["leaf_hash_perms"] / max(tot["commits"], 1),
                             "valu": seg_valu,
                             "note": "summed over the %d leaf-hash launches of the timed region (one per commitment: "
                                     "9 trace + 9 auxiliary + 9 quotient per segment); integer-VALU bound, not HBM bound "
                                     "(DESIGN.md): permutations/s = %.3e; traffic / valu from the --pmc passes on this workload "
                                     "(profiles/pmc_latest.json)"
                                     % (tot["commits"], tot["leaf_hash_perms"] / (leaf_ms * 1e-3))},
                "commit_stages_ms_per_step": {k: tot[k] / a.steps for k in ("ifft", "lde", "leaf_hash", "tree")},
                "commit_stages_note": "HIP events around each stage of the main-lane commitments; since r03t the levels of <= 2^17 "
                                      "nodes of a trace commitment's tree run on the ctx's tail stream under the NEXT commitment's "
                                      "NTT, so `tree` is an event-to-event span across two streams (an upper bound), and the stages "
                                      "no longer add up to the wall time of the commit phase (segment_timing_s has that)",
                "ntt": {"achieved_GBs": tot["ntt_bytes"] / (ntt_ms * 1e-3) / 1e9,
                        "algorithmic_bytes_per_step": tot["ntt_bytes"] / a.steps,
                        "frac_of_hbm_peak": tot["ntt_bytes"] / (ntt_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
                "side_lane": {"commits_per_step": side["commits"] / a.steps,
                              "ms_per_step": {k: side[k] / a.steps for k in ("ifft", "lde", "leaf_hash", "tree")},
                              "leaf_hash_bytes_per_step": side["leaf_hash_bytes"] / a.steps,
                              "note": "the auxiliary commitments (and trace commitments of tables <= 2^16 rows) run on the ctx's "
                                      "low-priority side stream, overlapped with the main stream's per-table chain; their "
                                      "event-to-event times include the sharing of the chip, so they are reported here and kept "
                                      "out of `roofline`, `ntt` and `commit_stages_ms_per_step` (main-lane launches only)"},
                "segment_timing_s": timing,
                "arena": {k: v / 1e9 for k, v in mem.items()},
            }
        if world == 1 and a.in_flight > 1:
            # Secondary object, never `value`: W segments in flight on this GPU (one worker thread + Context + stream
            # each, the same resident read-only inputs): what a deployment with W workers per GPU gets.  Skipped unless
            # W arenas fit beside the inputs; any failure only drops the object.
            try:
                out["in_flight"] = segments_in_flight(ctx, a.in_flight, max(2, a.steps), mem["peak_in_use"], all_stark, cfg,
                                                      traces, in_use, a.cdk_erigon)
            except Exception as e:
                out["in_flight"] = {"error": repr(e)}
        del traces
        torch.cuda.empty_cache()
        if a.commit_steps > 0 and a.hasher == 0:
            # BASELINE configs[1] in the same run (every rank runs it so the ranks stay in step)
            a_cols, a_logn = a.cols, a.log_n
            elapsed_c, stage_c = timed_commits(a.commit_steps, 2)
            if rank == 0:
                roof_c, extra_c = commit_report(a, stage_c, 1e3 * elapsed_c / a.commit_steps)
                out["commit_config1"] = {"workload": f"PolynomialBatch::from_values {a_cols} cols x 2^{a_logn} rows",
                                         "commits_per_s": world * a.commit_steps / elapsed_c,
                                         "ms_per_commit": 1e3 * elapsed_c / a.commit_steps, "roofline": roof_c}
                out["commit_config1"].update(extra_c)
        if rank == 0 and world == 1 and not a.no_secondary:
            try:
                if log_ns == [20] * n_tab:
                    traces = synthetic_segment_traces(log_ns, dev, seed=1 + rank, cdk_erigon=a.cdk_erigon)   # (freed above)
                    out["h2d"] = h2d_profile(dev, trace_bytes, ms_per_step / 1e3, step)
                    del traces
                    torch.cuda.empty_cache()
                else:
                    out["h2d"] = h2d_profile(dev, trace_bytes, ms_per_step / 1e3)
            except Exception as e:
                out["h2d"] = {"error": repr(e)}
            if log_ns == [20] * n_tab and a.hasher == 0:
                try:
                    ctx.mem_trim()
                    out["realistic"] = realistic_profile(ctx, dev, a, all_stark, cfg)
                except Exception as e:
                    out["realistic"] = {"error": repr(e)}
                if not a.cdk_erigon:
                    try:
                        ctx.mem_trim()
                        out["from_logs"] = from_logs_profile(ctx, dev, all_stark, cfg)
                    except Exception as e:
                        out["from_logs"] = {"error": repr(e)}
                try:
                    out["plonk_recursion"] = plonk_recursion_profile(ctx, dev, not a.no_cpu_baseline)
                except Exception as e:
                    out["plonk_recursion"] = {"error": repr(e)}
        if rank == 0 and world == 1 and not a.no_pmc:
            # counters of every hot kernel class measured in this run (child passes under rocprofv3); on any failure the
            # committed profile's numbers stay, marked measured_in_this_run: false
            try:
                ctx.mem_trim()
                torch.cuda.empty_cache()
                kc = collect_kernel_counters(a)
            except Exception:
                kc = None
            if kc:
                out["kernel_counters"] = kernel_counter_report(kc, log_ns, all_stark, cfg, a.cdk_erigon)
                lh = kc.get("leaf_hash")
                if lh and lh.get("n_fetch_kib") and lh.get("n_write_kib") and lh.get("valu_wave_insts"):
                    roof = out["roofline"]
                    n = lh["n_fetch_kib"]
                    roof["traffic"] = (2.0 * lh["fetch_kib"] / n + lh["write_kib"] / lh["n_write_kib"]) * 1024.0
                    roof["traffic_source"] = ("rocprofv3 --pmc FETCH_SIZE (x2, gfx950 correction) + WRITE_SIZE, child passes of one "
                                              "segment each in this run, mean over %d leaf-hash launches" % n)
                    insts = lh["valu_wave_insts"] / n
                    simd_cycles = lh["gui_active"] / n / 8.0 * 1024.0       # GRBM_GUI_ACTIVE is summed over the 8 XCDs
                    cpi = simd_cycles / insts
                    roof["valu"] = {
                        "wave_insts_per_launch": insts, "cycles_per_wave_instruction": cpi,
                        "frac": 2.0 / cpi,
                        "frac_is": "wave-instructions issued / issue slots, one wave64 VALU op per 2 cycles per SIMD being the floor "
                                   "(MI355X_MICROARCH.md); the kernel's own mix is dominated by v_mad_u64_u32 / carry-chain / "
                                   "v_cndmask ops that issue at ~4.3 cycles each (profiles/r01_ubench_valu_issue_rates.txt), so a "
                                   "cycles_per_wave_instruction of 3.5-3.8 is an issue-saturated SIMD",
                        "source": "rocprofv3 --pmc SQ_INSTS_VALU and GRBM_GUI_ACTIVE (/ 8 XCDs x 1024 SIMDs) of the same launches, "
                                  "this run", "measured_in_this_run": True}
                    # all 27 leaf-hash launches of the profiled segment (main and side lane): total instructions / total
                    # permutations = N * ceil(cols / 8) per commitment with more than 4 columns
                    seg_perms = 0.0
                    for t in range(all_stark.num_tables):
                        h, z, _ = sg.num_ctl_helpers_zs_all(all_stark.cross_table_lookups, t, cfg.num_challenges, all_stark.constraint_degree)
                        lk = sum(cfg.num_challenges * l.num_helper_columns(all_stark.constraint_degree) for l in all_stark.lookups[t])
                        for c in (all_stark.table_columns[t], lk + h + z):
                            if c > 4:
                                seg_perms += (2 << log_ns[t]) * ((c + 7) // 8)
                    if seg_perms:
                        roof["valu"]["instructions_per_permutation"] = lh["valu_wave_insts"] * 64.0 / seg_perms
                ntt = [kc.get("ntt_coeffs_to_values"), kc.get("ntt_values_to_coeffs")]
                if all(k and k.get("n_fetch_kib") and k.get("n_write_kib") for k in ntt):
                    tr = sum((2.0 * k["fetch_kib"] + k["write_kib"]) * 1024.0 for k in ntt)
                    alg = out["ntt"]["algorithmic_bytes_per_step"]
                    out["ntt"].update(traffic_bytes_per_step=tr, traffic_over_algorithmic=tr / alg,
                                      achieved_GBs_on_traffic=tr / (out["commit_stages_ms_per_step"]["ifft"] +
                                                                    out["commit_stages_ms_per_step"]["lde"]) / 1e6,
                                      traffic_source="FETCH_SIZE x2 + WRITE_SIZE of every ntt_pass_kernel launch of one segment, "
                                                     "this run; the time is the un-profiled timed region's")
                    out["ntt"]["frac_of_hbm_peak_on_traffic"] = out["ntt"]["achieved_GBs_on_traffic"] / HBM_PEAK_GBS
        if rank == 0 and not a.no_cpu_baseline and world == 1:
            extrap = None
            try:
                sl = a.cpu_sample_log_n
                cb = cpu_baseline(116, sl, sl, a.hasher, max_reps=2)
                sample_cells = 116 << sl
                sec = (1.0 / cb["value"]) * cells / sample_cells
                extrap = {
                    "value": 1.0 / sec, "unit": "segment proofs/s", "cores": cb["cores"], "kind": "port",
                    "sample": cb["sample"].split(", scaled")[0] + f"; scaled by committed cells ({cells} / {sample_cells}) to "
                              "the segment's 27 commitments -- COMMIT PHASE ONLY, an extrapolation and an upper bound on the "
                              "CPU rate",
                    "seconds_per_segment_est": sec}
            except Exception as e:  # the oracle is only a reported baseline; never fatal
                extrap = {"error": repr(e)}
            if a.cpu_table_log_n > 0 and a.hasher == 0:
                try:
                    ctx.mem_trim()
                    out["cpu_baseline"] = cpu_table_proof_baseline(ctx, dev, a.cpu_table_log_n)
                    out["cpu_baseline"]["segment_commit_phase_extrapolation"] = extrap
                except Exception as e:
                    out["cpu_baseline"] = extrap or {}
                    out["cpu_baseline"]["table_proof_error"] = repr(e)
            else:
                out["cpu_baseline"] = extrap
    if use_dist and (world == 1 or a.force_dist or os.environ.get("ZK_BENCH_DIST_SELFTEST") == "1"):
        # the collectives of the product's multi-GPU paths on this backend (RCCL under nccl): cap all-gather, status
        # all-reduce, challenger-state broadcast, variable-length gather -- every rank takes part, rank 0 reports.  In a
        # multi-rank run only on request: the scaling line needs nothing but the barrier and the MAX all-reduce above, and
        # must not depend on anything else.
        selftest = dist_selftest(rank, world, a.dist_backend)
        if rank == 0 and out is not None:
            out["dist"] = selftest
    if rank == 0 and out is not None and dist_error:
        out["dist"] = {"ok": False, "error": dist_error, "backend": a.dist_backend, "world": world}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
