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

After the timed region (never inside it) the line collects secondary objects: `segment_timing_s` (one extra synchronised
proof), `kernel_counters` (rocprofv3 child passes of one segment), and -- each in ITS OWN child process with a time limit,
`tools/bench_secondary.py` -- `commit_config1` (BASELINE configs[1]), `in_flight`, `h2d`, `realistic`, `from_logs`,
`plonk_recursion`, `block_replay`, `cpu_baseline`.  A secondary that fails or hangs costs its own limit and leaves
`{"error": ...}` in its slot; the throughput line always prints.

`--workload commit` makes the configs[1] commit the timed step instead.

Multi-GPU: segments are independent units (fresh Challenger per segment), so each rank proves
its own segment with no data-path collective ("scaling": "weak"); value = segments of all ranks / max time.

Prints ONE JSON line on rank 0.  `roofline` is for the dominant kernel (Poseidon leaf hashing,
poseidon_hash_rows_kernel: ~60 % of the segment), timed with HIP events on the kernel's own stream inside the
timed region and summed over its launches (one per commitment); `cpu_baseline` is the CPU oracle (C / OpenMP over
columns / leaves / rows, the axes rayon uses in the reference) proving ONE WHOLE nine-table segment on a bounded sample of
the workload (every table at 2^15 rows, ~15-25 s of CPU work on the box's 16 cores), scaled by committed cells to the workload's heights: the
unit is `value`'s, "segment proofs/s" (r04 verdict, item 7; the single-table comparison is `--secondary cpu_table`).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from tools.benchlib import (HBM_PEAK_GBS, REALISTIC_LOG_NS, collect_kernel_counters, commit_report,  # noqa: E402,F401
                            contract_line, kernel_counter_report, measure_commit, segment_committed_cells,
                            synthetic_segment_traces, write_extra)


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
    ap.add_argument("--mode", choices=["segments", "table_parallel", "table_parallel_keccak_rows"], default="segments",
                    help="how N ranks share the work: `segments` (default; SURVEY 8(e) level 1: one independent segment per GPU "
                         "and step, weak scaling, no data-path collective) or the LATENCY modes -- ONE segment per step proven by "
                         "all ranks together, strong scaling: `table_parallel` (level 2: a table per rank, caps all-gathered, the "
                         "challenger state broadcast along the chain) and `table_parallel_keccak_rows` (level 2 with the "
                         "2431-column Keccak table row-sharded over all ranks, level 3: all-to-all, sub-root all-gather, sharded "
                         "quotient / openings / FRI combination)")
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
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary objects (tools/bench_secondary.py children)")
    ap.add_argument("--secondary", type=str, default="all",
                    help="comma-separated subset of the secondaries to run (default: all that apply to the workload)")
    ap.add_argument("--secondary-budget-s", type=float, default=900.0,
                    help="wall-clock budget for ALL secondary children together; each also has its own limit")
    ap.add_argument("--dist-timeout-s", type=float, default=180.0, help="time limit of the RCCL probe collectives")
    ap.add_argument("--no-pmc", action="store_true",
                    help="do not collect the dominant kernel's HBM-traffic / VALU counters with rocprofv3 --pmc child passes "
                         "(the committed profiles/pmc_latest.json is quoted instead, marked as such)")
    ap.add_argument("--extra-name", type=str, default="bench_extra.json",
                    help="file name (next to bench.py and under gpurun_out/) of the FULL result: stage tables, kernel counters, "
                         "every secondary object -- stdout carries only the small contract line")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-sample-log-n", type=int, default=18)
    ap.add_argument("--cpu-table-log-n", type=int, default=20,
                    help="height of the ArithmeticStark table proven on the CPU by --secondary cpu_table (0 = the commit-sample "
                         "extrapolation only)")
    ap.add_argument("--cpu-segment-sample-log-n", type=int, default=15,
                    help="cpu_baseline: height of every table of the whole segment proven on the CPU (the bounded sample; scaled "
                         "by committed cells to the workload)")
    return ap.parse_args()



# ------------------------------------------------------------------------------------------------------------------------
# secondaries: one child process each (tools/bench_secondary.py), a limit each, one budget for all of them
SECONDARY_LIMITS_S = {"commit_config1": 120, "in_flight": 240, "h2d": 240, "realistic": 300, "block_replay": 300,
                      "from_logs": 180, "plonk_recursion": 300, "cpu_baseline": 420, "cpu_segment": 1500, "cpu_table": 420}
EXPLICIT_ONLY = {"cpu_segment", "cpu_table"}          # minutes of CPU time: only when named in --secondary


def run_secondary(name, a, device, deadline, extra=()):
    """Run one secondary in its own process group and return its JSON object; {"error": ...} on a non-zero exit, an
    unparsable answer or the time limit (the child's whole process group is killed by its pgid -- never by pattern)."""
    import signal
    import subprocess
    limit = min(float(SECONDARY_LIMITS_S[name]), deadline - time.monotonic())
    if limit < 10.0:
        return {"error": "skipped: the secondary budget (--secondary-budget-s) is spent"}
    cmd = [sys.executable, os.path.join(ROOT, "tools", "bench_secondary.py"), "--name", name, "--device", str(device),
           "--hasher", str(a.hasher), "--cols", str(a.cols), "--log-n", str(a.log_n), "--steps", str(a.steps),
           "--commit-steps", str(a.commit_steps), "--in-flight", str(a.in_flight),
           "--cpu-sample-log-n", str(a.cpu_sample_log_n), "--cpu-table-log-n", str(a.cpu_table_log_n),
           "--cpu-segment-sample-log-n", str(a.cpu_segment_sample_log_n), *extra]
    if a.log_ns:
        cmd += ["--log-ns", a.log_ns]
    if a.cdk_erigon:
        cmd += ["--cdk-erigon"]
    if a.no_cpu_baseline:
        cmd += ["--no-cpu-baseline"]
    t0 = time.monotonic()
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
    try:
        so, se = p.communicate(timeout=limit)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(p.pid, signal.SIGKILL)
        except OSError:
            pass
        p.communicate()
        return {"error": "time limit of %.0f s reached; child killed" % limit}
    lines = [ln for ln in so.strip().splitlines() if ln.startswith("{")]
    if p.returncode != 0 or not lines:
        return {"error": "exit code %d: %s" % (p.returncode, (se or so).strip()[-400:])}
    try:
        out = json.loads(lines[-1])
    except ValueError as e:
        return {"error": "unparsable answer: %r" % e}
    if isinstance(out, dict):
        out["wall_s"] = round(time.monotonic() - t0, 1)
    return out


def dist_selftest(rank, world, backend, group=None, ctx=None):
    """Run the tensor collectives of zk_evm_amd/collectives.py + sharding.gather_caps once through the live process group
    and check what comes back.  Under `nccl` this is RCCL moving device tensors."""
    import numpy as np
    from zk_evm_amd import collectives as co
    from zk_evm_amd.sharding import gather_caps
    res = {"backend": backend, "world": world, "payload_device": str(co.device_for(group))}
    try:
        n_tab = 9
        mine = {t: np.full((16, 4), 1000 * t + 7, dtype=np.uint64) for t in range(n_tab) if t % world == rank}
        caps = gather_caps(mine, n_tab, 16, group)
        assert all(int(caps[t][3, 2]) == 1000 * t + 7 for t in range(n_tab))
        co.agree(None, "selftest", group)
        st = co.broadcast_words(np.arange(32, dtype=np.uint64) + 5 if rank == world - 1 else None, 32, world - 1, group)
        assert int(st[31]) == 36
        parts = co.gather_varlen_words(np.arange(10 + rank, dtype=np.uint64) * (rank + 1), 0, group)
        if rank == 0:
            assert [p.size for p in parts] == [10 + r for r in range(world)] and all(int(p[-1]) == (9 + r) * (r + 1) for r, p in enumerate(parts))
        res["ok"] = True
        res["collectives"] = ["all_gather (caps)", "all_reduce MAX (status)", "broadcast (challenger state)", "gather (proof words)"]
        # the library's own communicator (csrc/comm_host.inc: RCCL's C API under nccl, the host-staged transport otherwise) -- what
        # zk_prove_table_sharded / zk_prove_segment_table_parallel run on -- made from this group and exercised once
        try:
            if ctx is None:
                raise RuntimeError("no zk_ctx (CPU-side self-test)")
            import ctypes as C
            import torch
            from zk_evm_amd.comm import Comm
            cm = Comm.from_group(ctx, group)
            got = cm.all_gather_words(np.arange(4, dtype=np.uint64) + 10 * rank)
            res["comm_transport"] = cm.transport
            res["comm_c_api_ok"] = bool(all(int(got[r, 3]) == 3 + 10 * r for r in range(world)))
            if backend == "nccl" and world == 1:
                # RCCL 2.26 of this image returned corrupted data for a send / recv above 2^30 bytes (tools/rccl_repro.py), so the
                # library cuts its exchanges into pieces of 256 MiB; both reported, so that a fixed RCCL shows up
                import torch.distributed as dist
                a = torch.arange(160 << 20, dtype=torch.int64, device=co.device_for(group))          # 1.25 GiB
                b, c = torch.zeros_like(a), torch.zeros_like(a)
                dist.all_to_all([b], [a], group=group)
                nbytes = (C.c_size_t * 1)(a.numel() * 8)
                ctx.use_torch_current_stream()
                ctx.check(ctx.lib.zk_comm_all_to_all_device(cm.handle, (C.c_void_p * 1)(a.data_ptr()), nbytes, (C.c_void_p * 1)(c.data_ptr()), nbytes))
                torch.cuda.synchronize()
                res["rccl_single_piece_above_1GiB_intact"] = bool(torch.equal(a, b))
                res["pieces_of_256MiB_intact"] = bool(torch.equal(a, c))
                del a, b, c
            cm.close()
        except Exception as e:                   # (a report, not part of the self-test's verdict)
            res["comm_c_api_ok"] = False
            res["comm_error"] = repr(e)[:300]
    except Exception as e:
        res["ok"] = False
        res["error"] = repr(e)
    return res


class RunGroup:
    """The process group(s) of one bench run.  What the timed region needs from the other ranks is a barrier on both sides
    and ONE max-over-ranks of a float -- it must never be the reason the line does not print (r03 verdict, weak 5: a
    failing `init_process_group("nccl")` at world > 1 used to end the run).  So:

      * a `gloo` group over 127.0.0.1 is the safety net and carries host-side bookkeeping (per-rank times);
      * with `--dist-backend nccl` an RCCL group is created beside it and PROBED (one all-reduce of device tensors under a
        time limit, outcome agreed over gloo): if every rank's probe passed, the barrier and the MAX all-reduce of the
        timed region run on RCCL, as the contract says; if any failed, every rank falls back to gloo and the line says
        so (`dist: {"ok": false, "error": ..., "fallback": "gloo"}`);
      * if even gloo cannot be initialised the ranks run unsynchronised and the line says that."""

    def __init__(self, a, rank, world, local_dev, want):
        self.rank, self.world, self.a = rank, world, a
        self.gloo = self.nccl = None
        self.info = {"backend": "none", "world": world, "ok": world == 1}
        self.dist = None
        if not want:
            self.info["note"] = "no process group requested"
            return
        import datetime
        import torch
        import torch.distributed as dist
        self.dist, self.torch = dist, torch
        self.dev = torch.device(f"cuda:{local_dev}")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1 and "MASTER_PORT" not in os.environ:
            import socket
            sk = socket.socket()
            sk.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
            sk.close()
        try:
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=1200))
            self.gloo = dist.group.WORLD
            self.info.update(backend="gloo", ok=True)
        except Exception as e:
            self.info.update(ok=False, error="gloo init: " + repr(e), fallback="unsynchronised ranks")
            return
        if a.dist_backend != "nccl":
            return
        # RCCL beside it -- but first PROBED IN A CHILD PROCESS per rank.  A communicator that cannot come up does not
        # always raise: `ncclCommInitRank` can block inside the library for as long as a peer is alive (seen on the gpurun
        # box with two ranks on one device: the process-group timeout never fired, the gloo agreement below waited its full
        # 600 s and the run died).  A child that hangs is killed by its pid after --dist-timeout-s; this process never
        # enters RCCL unless every rank's child got an all-reduce through.
        err, grp = self._probe_in_child(local_dev, a.dist_timeout_s), None
        flag = torch.tensor([0 if err is None else 1], dtype=torch.int64)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.gloo)
        if int(flag.item()) != 0:
            self.info.update(backend="gloo", ok=False, fallback="gloo", tried="nccl",
                             error=err or "the RCCL probe failed on another rank")
            return
        # Blocking waits turn a collective that never completes into an exception after the group's time limit instead of a
        # watchdog abort of the process.
        os.environ["TORCH_NCCL_BLOCKING_WAIT"] = "1"
        os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "0")
        try:
            torch.cuda.set_device(local_dev)
            to = datetime.timedelta(seconds=a.dist_timeout_s)
            try:
                grp = dist.new_group(backend="nccl", timeout=to, device_id=self.dev)
            except TypeError:
                grp = dist.new_group(backend="nccl", timeout=to)
            t = torch.ones(1, dtype=torch.float64, device=self.dev)
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=grp, async_op=True).wait()
            torch.cuda.synchronize()
            if int(t.item()) != world:
                err = "RCCL probe all-reduce returned %r, expected %d" % (t.item(), world)
        except Exception as e:
            err = repr(e)
        # every rank learns whether EVERY rank's probe passed (gloo): all use RCCL, or all fall back
        flag = torch.tensor([0 if err is None else 1], dtype=torch.int64)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.gloo)
        if int(flag.item()) == 0:
            self.nccl = grp
            self.info.update(backend="nccl", ok=True, probe="all_reduce SUM of device tensors over %d rank(s), first in a child "
                                                            "process per rank, then in this one" % world)
        else:
            self.info.update(backend="gloo", ok=False, fallback="gloo", tried="nccl",
                             error=err or "the RCCL probe failed on another rank")

    PROBE = ("import os, sys, datetime, torch, torch.distributed as dist\n"
             "d = int(sys.argv[1]); torch.cuda.set_device(d)\n"
             "dist.init_process_group('nccl', rank=int(os.environ['RANK']), world_size=int(os.environ['WORLD_SIZE']),\n"
             "                        timeout=datetime.timedelta(seconds=float(sys.argv[2])), device_id=torch.device('cuda', d))\n"
             "t = torch.ones(1, dtype=torch.float64, device='cuda')\n"
             "dist.all_reduce(t); torch.cuda.synchronize()\n"
             "assert int(t.item()) == int(os.environ['WORLD_SIZE']), t\n"
             "print('RCCL_PROBE_OK', flush=True)\n"
             "os._exit(0)\n")

    def _probe_in_child(self, local_dev, limit_s):
        """None if a child of this rank brought up an RCCL group with the other ranks' children (their own rendezvous port,
        agreed over gloo) and got one all-reduce through; otherwise what went wrong.  Never blocks longer than limit_s."""
        import signal
        import socket
        import subprocess
        torch, dist = self.torch, self.dist
        port = torch.zeros(1, dtype=torch.int64)
        if self.rank == 0:
            sk = socket.socket()
            sk.bind(("127.0.0.1", 0))
            port[0] = sk.getsockname()[1]
            sk.close()
        dist.broadcast(port, src=0, group=self.gloo)
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(int(port.item())), RANK=str(self.rank),
                   WORLD_SIZE=str(self.world), LOCAL_RANK=str(self.rank), HSA_ENABLE_IPC_MODE_LEGACY="0")
        try:
            p = subprocess.Popen([sys.executable, "-c", self.PROBE, str(local_dev), str(limit_s)], env=env,
                                 stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
        except OSError as e:
            return "cannot start the probe: %r" % e
        try:
            so, se = p.communicate(timeout=limit_s)
        except subprocess.TimeoutExpired:
            try:
                os.killpg(p.pid, signal.SIGKILL)
            except OSError:
                pass
            p.communicate()
            return "the RCCL probe process did not finish within %.0f s (killed)" % limit_s
        if p.returncode == 0 and "RCCL_PROBE_OK" in so:
            return None
        return "the RCCL probe process failed (exit code %s): %s" % (p.returncode, (se or so).strip()[-300:])

    @property
    def backend(self):
        return "nccl" if self.nccl is not None else ("gloo" if self.gloo is not None else "none")

    def group(self):
        return self.nccl if self.nccl is not None else self.gloo

    def barrier(self):
        import torch
        if self.nccl is not None:
            self.dist.barrier(group=self.nccl, device_ids=[self.dev.index])
        elif self.gloo is not None:
            self.dist.barrier(group=self.gloo)
        if torch.cuda.is_available():
            torch.cuda.synchronize()

    def max_over_ranks(self, x):
        import torch
        if self.nccl is not None:
            tt = torch.tensor([x], dtype=torch.float64, device=self.dev)
            self.dist.all_reduce(tt, op=self.dist.ReduceOp.MAX, group=self.nccl)
            return float(tt.item())
        if self.gloo is not None:
            tt = torch.tensor([x], dtype=torch.float64)
            self.dist.all_reduce(tt, op=self.dist.ReduceOp.MAX, group=self.gloo)
            return float(tt.item())
        return x

    def per_rank(self, x):
        """every rank's value on every rank (host side, gloo): a straggler is visible in the line"""
        import torch
        if self.gloo is None:
            return [x]
        tt = torch.tensor([x], dtype=torch.float64)
        parts = [torch.zeros(1, dtype=torch.float64) for _ in range(self.world)]
        self.dist.all_gather(parts, tt, group=self.gloo)
        return [float(p.item()) for p in parts]

    def close(self):
        """Leave the group(s).  After a failed RCCL probe the communicator may be half built and its destructor can block:
        the ranks agree over gloo that they are done, and the process then exits without running it."""
        if self.gloo is None:
            return
        try:
            self.dist.barrier(group=self.gloo)
            if self.info.get("fallback"):
                sys.stdout.flush()
                sys.stderr.flush()
                os._exit(0)
            self.dist.destroy_process_group()
        except Exception:
            pass


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
    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    local_dev = [int(x) for x in a.devices.split(",")][local] if a.devices else local
    # A single rank joins a process group of one as well (unless --no-dist-selftest / a non-default workload): the N = 1
    # line then exercises the same init / probe / barrier / all-reduce / collectives as the N > 1 run, on RCCL.
    want = world > 1 or a.force_dist or (a.workload == "segment" and not a.no_dist_selftest and not a.pmc_child)
    torch.cuda.set_device(local_dev)
    rg = RunGroup(a, rank, world, local_dev, want)
    if a.force_dist and not rg.info.get("ok"):
        raise RuntimeError("--force-dist: %r" % (rg.info,))
    dev = torch.device(f"cuda:{local_dev}")
    local = local_dev

    import zk_evm_amd
    ctx = zk_evm_amd.Context(local)
    ctx.use_torch_current_stream()
    hname = "poseidon" if a.hasher == 0 else "keccak25"
    wanted = (set(SECONDARY_LIMITS_S) - EXPLICIT_ONLY) if a.secondary == "all" else {s for s in a.secondary.split(",") if s}
    unknown = wanted - set(SECONDARY_LIMITS_S)
    assert not unknown, "unknown secondaries: %s" % sorted(unknown)

    def timed_commits(steps, warmup):
        trace, step = measure_commit(ctx, dev, a, rank, steps, warmup)
        for _ in range(warmup):
            step()
        rg.barrier()
        t0 = time.perf_counter()
        stage = {"ifft": 0.0, "lde": 0.0, "leaf_hash": 0.0, "tree": 0.0}
        for _ in range(steps):
            t = step()
            for k in stage:
                stage[k] += t[k]
        rg.barrier()
        own = time.perf_counter() - t0
        elapsed = rg.max_over_ranks(own)
        for k in stage:
            stage[k] /= steps
        del trace
        return elapsed, own, stage

    out, own_elapsed = None, None
    secondaries = []                  # (name, extra args) run after the timed region on rank 0 of a one-rank run
    if a.workload == "commit":
        elapsed, own_elapsed, stage = timed_commits(a.steps, a.warmup)
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
                    from tools.bench_secondary import cpu_baseline
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

        latency_mode = a.mode != "segments"
        if latency_mode:
            # every rank generates the SAME segment (seed 1) and keeps what it owns: its tables (the library's own assignment)
            # and, for the row-sharded table, its row block
            from zk_evm_amd.sharding import assign_tables, prove_segment_table_parallel
            KECCAK = 3
            if rank != 0 or world > 1:
                del traces
                traces = synthetic_segment_traces(log_ns, dev, seed=1, cdk_erigon=a.cdk_erigon)
            wide = {}
            if a.mode == "table_parallel_keccak_rows":               # (one rank: the whole table as its only block)
                nb = traces[KECCAK].shape[1] // world
                wide = {KECCAK: traces[KECCAK][:, rank * nb:(rank + 1) * nb].contiguous()}
            solo = [t for t in range(n_tab) if t not in wide]
            mine = [solo[k] for k in assign_tables([(TABLE_COLUMNS[t], log_ns[t]) for t in solo], world)[rank]]
            traces = [tr if t in mine else None for t, tr in enumerate(traces)]
            torch.cuda.empty_cache()
            grp = rg.group() if rg.dist is not None and rg.gloo is not None else None

        def step(timing=None):
            pv = sg.PublicValues(burn_addr=1 if a.cdk_erigon else None)
            if latency_mode:
                return prove_segment_table_parallel(all_stark, cfg, traces, in_use, pv, group=grp, ctx=ctx, timing=timing,
                                                    row_sharded=wide or None)
            return sg.prove_with_traces(all_stark, cfg, traces, in_use, pv, ctx=ctx, timing=timing)
        if a.pmc_child:                       # one segment under rocprofv3 --pmc (collect_kernel_counters), nothing printed
            step()
            torch.cuda.synchronize()
            return
        for _ in range(a.warmup):
            step()
        rg.barrier()
        ctx.commit_totals(reset=True)
        ctx.side_commit_totals(reset=True)
        t0 = time.perf_counter()
        for _ in range(a.steps):
            proof = step()
        rg.barrier()
        own_elapsed = time.perf_counter() - t0
        elapsed = rg.max_over_ranks(own_elapsed)
        tot = ctx.commit_totals(reset=True)
        side = ctx.side_commit_totals(reset=True)
        mem = ctx.mem_stats()
        timing = {}
        if latency_mode:
            step(timing)                     # (the stage breakdown's extra proof: every rank is part of it in these modes)
        if rank == 0:
            ms_per_step = 1e3 * elapsed / a.steps
            leaf_ms = tot["leaf_hash"]
            achieved = tot["leaf_hash_bytes"] / (leaf_ms * 1e-3) / 1e9
            ntt_ms = tot["ifft"] + tot["lde"]
            proof_words = sum(int(p.proof.opening_proof.size) for p in proof.multi_proof.stark_proofs if p is not None)
            if not latency_mode:
                step(timing)                 # one extra, synchronised, untimed proof for the stage breakdown
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
                                           "v_mov / v_add_u32-class ops issue at ~2.4 cycles (profiles/archive/r01_ubench_valu_issue_rates.txt), "
                                           "so a mix with many movs can exceed 1.0: the SIMDs are issue-saturated either way",
                                "source": pm["source"], "source_commit": pm.get("git_commit"),
                                "measured_in_this_run": False}
            except Exception:
                pass
            cells = segment_committed_cells(log_ns, a.cdk_erigon)
            out = {
                "metric": "segment STARK proofs/sec (2^20-row traces, all nine AllStark tables)" if log_ns == [20] * 9 else
                          "segment STARK proofs/sec (table heights 2^%s%s)" % (",".join(map(str, log_ns)), ", cdk_erigon" if a.cdk_erigon else ""),
                "value": (1 if latency_mode else world) * a.steps / elapsed, "unit": "segment proofs/s", "n_gpus": world, "steps": a.steps,
                "warmup": a.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong" if latency_mode else "weak",
                "vs_baseline": None, "dtype": "u64", "data": "synthetic",
                "config": {"workload": f"prove_with_traces: full AllStark segment proof (BASELINE configs[2]), 9 tables x "
                                       f"2^{log_ns[0] if uniform else log_ns} rows ({sum(TABLE_COLUMNS)} trace columns, {trace_bytes / 1e9:.1f} GB), "
                                       f"10 CTLs + lookups, standard_fast_config, hasher {hname}",
                           "parallelism": (f"{world} independent segments (one per GPU), no collective" if not latency_mode else
                                           f"ONE segment per step over {world} ranks, mode {a.mode}: tables {mine} on rank 0"
                                           + (", Keccak's rows over all ranks" if wide else "")),
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
            # which of the equivalent kernels served this run: the ctx's plan table (data, not a trial; include/zkstark.h zk_ctx_set_plans)
            plans = ctx.get_plans()
            out["ntt"]["plans"] = plans
            out["ntt"]["plans_source"] = "ZK_NTT_SWAP_PLANS" if os.environ.get("ZK_NTT_SWAP_PLANS") else "compiled in"
            out["ntt"]["lane_swap_plans"] = sum(1 for it in plans.split(";") if it[:1] in ("v", "d") and it.endswith("=2"))
            out["ntt"]["tree_tops_batched"] = "T=1" in plans.split(";")
            out["ntt"]["forced"] = {k: os.environ[k] for k in ("ZK_NTT_SWAP", "ZK_TREE_BATCH", "ZK_NTT_COL_BATCH_MB") if os.environ.get(k)}
            default_shape = log_ns == [20] * n_tab
            if a.in_flight > 1:
                secondaries.append(("in_flight", ["--arena-peak", str(mem["peak_in_use"])]))
            if a.commit_steps > 0 and a.hasher == 0:
                secondaries.append(("commit_config1", []))
            secondaries.append(("h2d", ["--step-s", str(ms_per_step / 1e3)]))
            if default_shape and a.hasher == 0:
                secondaries.append(("realistic", []))
                secondaries.append(("block_replay", []))
                if not a.cdk_erigon:
                    secondaries.append(("from_logs", []))
                secondaries.append(("plonk_recursion", []))
        del traces
        torch.cuda.empty_cache()
        if rank == 0 and world == 1 and not a.no_pmc:
            # counters of every hot kernel class measured in this run (child passes under rocprofv3, each with its own time
            # limit); on any failure the committed profile's numbers stay, marked measured_in_this_run: false
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
                                   "v_cndmask ops that issue at ~4.3 cycles each (profiles/archive/r01_ubench_valu_issue_rates.txt), so a "
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
                ntt = [kc.get(k) for k in ("ntt_coeffs_to_values", "ntt_values_to_coeffs", "ntt_fused")]
                ntt = [k for k in ntt if k]
                if ntt and all(k.get("n_fetch_kib") and k.get("n_write_kib") for k in ntt):
                    tr = sum((2.0 * k["fetch_kib"] + k["write_kib"]) * 1024.0 for k in ntt)
                    alg = out["ntt"]["algorithmic_bytes_per_step"]
                    out["ntt"].update(traffic_bytes_per_step=tr, traffic_over_algorithmic=tr / alg,
                                      achieved_GBs_on_traffic=tr / (out["commit_stages_ms_per_step"]["ifft"] +
                                                                    out["commit_stages_ms_per_step"]["lde"]) / 1e6,
                                      traffic_source="FETCH_SIZE x2 + WRITE_SIZE of every NTT kernel launch of one segment, "
                                                     "this run; the time is the un-profiled timed region's")
                    out["ntt"]["frac_of_hbm_peak_on_traffic"] = out["ntt"]["achieved_GBs_on_traffic"] / HBM_PEAK_GBS
        if rank == 0 and world == 1 and "cpu_segment" in wanted and a.secondary != "all":
            secondaries.append(("cpu_segment", []))
        if rank == 0 and not a.no_cpu_baseline and world == 1:
            secondaries.insert(0, ("cpu_baseline", []))      # the contract's field first: it must never lose its time to the others
    # ---- every rank's own time (a straggler shows), the process-group record, the product collectives' self-test ----------
    per_rank = rg.per_rank(own_elapsed if own_elapsed is not None else 0.0)
    if rank == 0 and out is not None:
        out["per_rank_ms_per_step"] = [round(1e3 * x / a.steps, 3) for x in per_rank]
        out["dist"] = dict(rg.info)
    if rg.gloo is not None and (world == 1 or a.force_dist or os.environ.get("ZK_BENCH_DIST_SELFTEST") == "1"):
        # the collectives of the product's multi-GPU paths on this backend (RCCL under nccl): cap all-gather, status
        # all-reduce, challenger-state broadcast, variable-length gather -- every rank takes part, rank 0 reports.  In a
        # multi-rank run only on request: the scaling line needs nothing but the barrier and the MAX all-reduce above, and
        # must not depend on anything else.
        selftest = dist_selftest(rank, world, rg.backend, rg.group(), ctx)
        if rank == 0 and out is not None:
            out["dist"].update({k: v for k, v in selftest.items() if k not in ("ok", "error")})
            if not selftest.get("ok"):
                out["dist"]["selftest_error"] = selftest.get("error")
            out["dist"]["selftest_ok"] = bool(selftest.get("ok"))
    if world > 1 and rg.nccl is not None and os.environ.get("ZK_BENCH_RCCL_DRILL", "1") == "1":
        # r04 verdict, item 6: the first multi-GPU box answers whether RCCL's corruption of messages above 1 GiB (seen with one rank,
        # self copy) hits real peers -- one 1.27 GB send / recv and all-to-all between neighbours, and the library's 256 MiB pieces
        # The drill is the first code of this repository that moves gigabytes between two real RCCL peers.  It runs AFTER the
        # timed region and must never cost the scaling line: if it has not returned within ZK_BENCH_DRILL_TIMEOUT_S, every rank's
        # own watchdog ends its process -- rank 0 after printing the contract line, which is complete without the drill.
        import threading
        drill_lock, drill_done = threading.Lock(), [False]

        def drill_watchdog():
            with drill_lock:
                if drill_done[0]:
                    return
                if rank == 0 and out is not None:
                    out["dist"]["rccl_drill_error"] = "no answer within %s s: ended by the watchdog (the timed region was complete)" % drill_timeout
                    try:
                        write_extra(out, name=os.path.basename(a.extra_name))
                    except Exception:
                        pass
                    sys.stdout.write("\n" + contract_line(out, os.path.basename(a.extra_name)) + "\n")
                    sys.stdout.flush()
                os._exit(0)
        drill_timeout = float(os.environ.get("ZK_BENCH_DRILL_TIMEOUT_S", "120"))
        watchdog = threading.Timer(drill_timeout, drill_watchdog)
        watchdog.daemon = True
        watchdog.start()
        try:
            from tools.rccl_repro import run as rccl_drill
            drill = rccl_drill((1.27,), True, rg.nccl, ctx, rg.gloo)
            with drill_lock:
                drill_done[0] = True
            watchdog.cancel()
            if rank == 0 and out is not None:
                out["dist"].update({k: drill.get(k) for k in ("rccl_large_piece_intact_self", "rccl_large_piece_intact_peer", "library_pieces_intact")})
                out["rccl_drill"] = drill
        except Exception as e:
            with drill_lock:
                drill_done[0] = True
            watchdog.cancel()
            if rank == 0 and out is not None:
                out["dist"]["rccl_drill_error"] = repr(e)[:200]
    # ---- secondaries: after everything the contract names is in `out`, each in its own process under a time limit -------
    if rank == 0 and world == 1 and out is not None and not a.no_secondary and secondaries:
        ctx.mem_trim()
        torch.cuda.empty_cache()
        deadline = time.monotonic() + a.secondary_budget_s
        ran = {}
        for name, extra in secondaries:
            if name not in wanted:
                continue
            res = run_secondary(name, a, local, deadline, extra)
            ran[name] = res.get("wall_s") if isinstance(res, dict) else None
            out[name] = res
        out["secondary_wall_s"] = ran
    # RCCL leaves its version banner in the C runtime's stdout buffer until exit -- on every rank, and a launcher merges the
    # ranks' stdout: out with it now, everywhere, so that rank 0's JSON line is the LAST line
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    rg.barrier()
    if rank == 0:
        # the full result goes to bench_extra.json (and gpurun_out/); stdout gets the small contract line, on a line of its
        # own, last (tools/benchlib.contract_line; tests/test_bench_helpers.py pins its size and strict-JSON round trip)
        try:
            write_extra(out, name=os.path.basename(a.extra_name))
        except Exception as e:
            sys.stderr.write("bench_extra.json not written: %r\n" % (e,))
        sys.stdout.write("\n" + contract_line(out, os.path.basename(a.extra_name)) + "\n")
        sys.stdout.flush()
    rg.close()


if __name__ == "__main__":
    main()
