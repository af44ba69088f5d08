"""Shared pieces of bench.py and tools/bench_secondary.py: the synthetic workloads of BASELINE.json's configs (traces
already resident in HBM), the roofline arithmetic of a commitment, and the rocprofv3 child passes that measure the
per-kernel-class counters of one segment.  Nothing here is timed by the driver: bench.py owns the timed region."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
BENCH_PY = os.path.join(ROOT, "bench.py")

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)

# Arithmetic, BytePacking, Cpu, Keccak, KeccakSponge, Logic, Memory, MemBefore, MemAfter (scripts/prove_stdio.rs:89-101)
REALISTIC_LOG_NS = [17, 14, 19, 17, 13, 16, 21, 19, 19]


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
                # per SIMD (profiles/archive/r01_ubench_valu_issue_rates.txt)
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


KERNEL_CLASSES = (   # (class, substring of the rocprofv3 kernel name)
    ("leaf_hash", "hash_rows_kernel<false>"), ("leaf_hash_coop", "hash_rows_coop_kernel"),
    ("ntt_coeffs_to_values", "ntt_pass_kernel<true"), ("ntt_values_to_coeffs", "ntt_pass_kernel<false"),
    # r05, csrc/ntt_swap.cuh (template arguments print as <true, 9> / <false, 10>; the wave kernels carry their direction in the name)
    ("ntt_coeffs_to_values", "ntt_strided_swap_kernel<true"), ("ntt_values_to_coeffs", "ntt_strided_swap_kernel<false"),
    ("ntt_coeffs_to_values", "ntt_contig_wave_kernel_dit"), ("ntt_values_to_coeffs", "ntt_contig_wave_kernel_dif"),
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
                "--output-format", "csv", "-d", d, "-o", name, "--", sys.executable, BENCH_PY, "--pmc-child",
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



# ------------------------------------------------------------------------------------------------------------------------
# BASELINE configs[3] at shape level: one block's job list
# Per-table degree-bit ranges the reference sets for witness_b19807080 (scripts/prove_stdio.rs:89-101; Rust ranges, upper
# end exclusive), in table order Arithmetic, BytePacking, Cpu, Keccak, KeccakSponge, Logic, Memory, MemBefore, MemAfter.
B19807080_RANGES = [(16, 18), (8, 15), (9, 20), (7, 18), (8, 14), (5, 17), (17, 22), (16, 20), (7, 20)]


def block_segment_shapes(n_segments, seed=19807080, max_log=None):
    """[(log_ns, table_in_use)] for the segments of ONE block: heights drawn inside the witness_b19807080 ranges, the
    optional tables absent in some segments the way generation/mod.rs:586-631 decides it (Keccak and KeccakSponge together
    when a segment hashes nothing, Logic, BytePacking; MemAfter in the block's last segment) -- an absent table is still
    committed, at its minimum height (prover.rs:90-111 commits every trace).  `max_log` clips the heights (tests).  The real
    witness needs the Rust interpreter; this reproduces what the prover sees of it: a list of differently shaped segments."""
    import numpy as np
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n_segments):
        log_ns = [int(rng.integers(lo, hi)) for lo, hi in B19807080_RANGES]
        in_use = [True] * 9
        if rng.random() < 0.25:
            in_use[3] = in_use[4] = False
        if rng.random() < 0.2:
            in_use[5] = False
        if rng.random() < 0.2:
            in_use[1] = False
        if i == n_segments - 1:
            in_use[8] = False
        for t in range(9):
            if not in_use[t]:
                log_ns[t] = B19807080_RANGES[t][0]
            if max_log is not None:
                log_ns[t] = min(log_ns[t], max_log)
        out.append((log_ns, in_use))
    return out


def block_jobs(shapes, cdk_erigon=False, seed0=4000):
    """SegmentJob list for `shapes`: job i materialises its synthetic traces on the proving device when a worker takes it
    (the scheduler's `load(device)` contract), seeded by the job index, so every process builds the same traces."""
    import zk_evm_amd.segment as sg
    from zk_evm_amd.scheduler import SegmentJob

    def loader(i, log_ns):
        return lambda dev: synthetic_segment_traces(log_ns, dev, seed=seed0 + i, cdk_erigon=cdk_erigon)
    return [SegmentJob(loader(i, log_ns), in_use, sg.PublicValues(burn_addr=1 if cdk_erigon else None), tag=i)
            for i, (log_ns, in_use) in enumerate(shapes)]


# ------------------------------------------------------------------------------------------------------------------------
# the contract line: small, boring, ASCII, one line -- everything else goes to bench_extra.json (r04 verdict, item 1: the
# 22 KB line of r04 did not parse on the driver's side)
CONTRACT_LINE_LIMIT = 4000


def _num(x, digits=6):
    """a finite float rounded to `digits` significant figures, an int as it is, anything else (NaN, inf, None, text) -> None"""
    import math
    if isinstance(x, bool):
        return x
    if isinstance(x, int):
        return x
    if isinstance(x, float) and math.isfinite(x):
        return float("%.*g" % (digits, x))
    return None


def _pick(d, *path):
    for k in path:
        if not isinstance(d, dict) or k not in d:
            return None
        d = d[k]
    return d


def contract_line(out, extra_path="bench_extra.json"):
    """The ONE line bench.py prints: the contract's fields, the `roofline` and `cpu_baseline` objects, and a handful of scalars
    lifted out of the secondaries -- every value a finite number, a short string, a bool or null.  `out` is the full result
    dict (which goes to `extra_path` unchanged).  Raises if the line would not round-trip through a strict JSON parser."""
    r = out.get("roofline") or {}
    roof = {"bound": r.get("bound", "hbm"), "kernel": r.get("kernel"), "achieved": _num(r.get("achieved")),
            "peak": _num(r.get("peak")), "unit": r.get("unit", "GB/s"), "frac": _num(r.get("frac")),
            "traffic": _num(r.get("traffic"), 10), "algorithmic_bytes": _num(r.get("algorithmic_bytes"), 10),
            "ms_per_launch": _num(r.get("ms_per_launch")), "launches": _num(r.get("launches")),
            "share_of_step": _num(r.get("share_of_step")),
            "limited_by": "integer VALU issue" if _pick(r, "valu") is not None or r.get("limiting_resource") else None,
            "valu_cycles_per_wave_instruction": _num(_pick(r, "valu", "cycles_per_wave_instruction")),
            "valu_frac": _num(_pick(r, "valu", "frac")),
            "counters_measured_in_this_run": _pick(r, "valu", "measured_in_this_run")}
    line = {k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                    "scaling", "vs_baseline", "dtype", "data")}
    for k in ("value", "ms_per_step"):
        line[k] = _num(line[k], 9)
    cfg = out.get("config") or {}
    line["config"] = {"workload": str(cfg.get("workload", ""))[:260], "parallelism": str(cfg.get("parallelism", ""))[:160]}
    line["roofline"] = roof
    cb = out.get("cpu_baseline")
    if isinstance(cb, dict):
        if "error" in cb and "value" not in cb:
            line["cpu_baseline"] = {"error": str(cb["error"])[:160]}
        else:
            line["cpu_baseline"] = {"value": _num(cb.get("value")), "unit": cb.get("unit"), "cores": _num(cb.get("cores")),
                                    "kind": cb.get("kind"), "sample": str(cb.get("sample", ""))[:400]}
            for k in ("measured", "cpu_model", "seconds", "sample_seconds", "scale", "scale_rule", "proofs_identical", "shape", "gpu_same_sample_s"):
                if k in cb:
                    line["cpu_baseline"][k] = _num(cb[k]) if not isinstance(cb[k], (str, bool, list)) else cb[k]
    n = out.get("ntt") or {}
    if n:
        line["ntt"] = {"achieved_GBs": _num(n.get("achieved_GBs")), "frac": _num(n.get("frac_of_hbm_peak")),
                       "traffic_over_algorithmic": _num(n.get("traffic_over_algorithmic"))}
        if n.get("lane_swap_plans") is not None:          # the ctx's plan table: transform shapes on the lane-swap kernels, tree tops batched
            line["ntt"]["lane_swap_plans"] = _num(n.get("lane_swap_plans"))
            line["ntt"]["tree_tops_batched"] = bool(n.get("tree_tops_batched"))
            line["ntt"]["plans_source"] = n.get("plans_source")
            if n.get("forced"):
                line["ntt"]["forced"] = n.get("forced")
    d = out.get("dist") or {}
    if d:
        line["dist"] = {k: d.get(k) for k in ("backend", "world", "ok", "selftest_ok", "fallback", "tried", "payload_device",
                                               "rccl_single_piece_above_1GiB_intact", "pieces_of_256MiB_intact",
                                               "comm_c_api_ok", "comm_transport", "rccl_large_piece_intact_self",
                                               "rccl_large_piece_intact_peer", "library_pieces_intact") if k in d}
        if d.get("error"):
            line["dist"]["error"] = str(d["error"])[:200]
    if out.get("per_rank_ms_per_step") is not None:
        line["per_rank_ms_per_step"] = [_num(x) for x in out["per_rank_ms_per_step"]][:16]
    sc = {"realistic_ms_per_proof": _pick(out, "realistic", "single", "ms_per_proof"),
          "realistic_proofs_per_s": _pick(out, "realistic", "single", "value"),
          "realistic_in_flight_proofs_per_s": _pick(out, "realistic", "in_flight", "value"),
          "in_flight_proofs_per_s": _pick(out, "in_flight", "value"),
          "commit_config1_per_s": _pick(out, "commit_config1", "commits_per_s"),
          "commit_config1_ntt_GBs": _pick(out, "commit_config1", "ntt", "achieved_GBs"),
          "block_replay_segments_per_s": _pick(out, "block_replay", "value"),
          "from_logs_proofs_per_s": _pick(out, "from_logs", "value"),
          "plonk_2p13_batch_proofs_per_s": _pick(out, "plonk_recursion", "batch_2^13", "proofs_per_s"),
          "quotient_ms_total": _pick(out, "kernel_counters", "quotient_ms_total"),
          "ntt_coeffs_to_values_cycles_per_instruction": _pick(out, "kernel_counters", "ntt_coeffs_to_values", "cycles_per_wave_instruction"),
          "h2d_pinned_GBs": _pick(out, "h2d", "pinned_GBs"),
          "overlapped_upload_proofs_per_s": _pick(out, "h2d", "overlapped_upload", "value")}
    line["secondary"] = {k: _num(v) for k, v in sc.items() if _num(v) is not None}
    failed = sorted(k for k, v in out.items() if isinstance(v, dict) and "error" in v and "value" not in v and k != "cpu_baseline")
    if failed:
        line["secondary_failed"] = failed[:12]
    line["extra"] = extra_path
    s = json.dumps(line, allow_nan=False, ensure_ascii=True, separators=(", ", ": "))
    assert "\n" not in s and len(s) < CONTRACT_LINE_LIMIT, "contract line too long: %d bytes" % len(s)
    json.loads(s)
    return s


def write_extra(out, root=ROOT, name="bench_extra.json"):
    """the full result (stage tables, kernel counters, every secondary) next to bench.py and, when the scratch directory
    exists or can be made, under gpurun_out/ so that it travels back from the GPU box"""
    def clean(x):
        import math
        if isinstance(x, dict):
            return {str(k): clean(v) for k, v in x.items()}
        if isinstance(x, (list, tuple)):
            return [clean(v) for v in x]
        if isinstance(x, float) and not math.isfinite(x):
            return None
        return x
    txt = json.dumps(clean(out), allow_nan=False, indent=1)
    written = []
    for d in (root, os.path.join(root, "gpurun_out")):
        try:
            os.makedirs(d, exist_ok=True)
            with open(os.path.join(d, name), "w") as f:
                f.write(txt + "\n")
            written.append(os.path.join(d, name))
        except OSError:
            pass
    return written
