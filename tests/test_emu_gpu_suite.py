"""CPU (pytest -m "not gpu"): the GPU parity tests -- tests/test_gpu_*.py, unchanged -- against the CPU EMULATION BUILD of the library
(tests/emu/): libzkstark's own kernels and host code, compiled for the host from the sources where they lie, on a
stand-in HIP runtime (fibers per block, wave operations, streams in order or deferred).  What this gives the rows of SURVEY section 8
without hardware: zk_commit_* / zk_prove_* END TO END equal the oracle -- including the kernels and host paths written in rounds 5 and
6, which no GPU has run (lane-swap NTT kernels, column batches on two streams, batched tree tops, the plan table, the comm failure
protocol).  What it does not give: the gfx950 inline assembly (portable bodies are taken), timing, RCCL.

The slice run here is tests/emu/quick_slice.txt (~140 tests); tools/emu_full_suite.sh runs everything that fits a CPU, and
tools/emu_sanitizers.sh the same under ASan + UBSan and TSan (logs under profiles/)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ids(name):
    return [ln.strip() for ln in open(os.path.join(ROOT, "tests", "emu", name)) if "::" in ln and not ln.startswith("#")]


@pytest.fixture(scope="module")
def emu_env():
    sys.path.insert(0, ROOT)
    from tools.emu_survey import emu_env as make
    return make()


def _run(ids, env, timeout):
    r = subprocess.run([sys.executable, "-m", "pytest", *ids, "-q", "-p", "no:cacheprovider", "-x"], capture_output=True, text=True, cwd=ROOT,
                       env=env, timeout=timeout)
    tail = r.stdout[-3000:] + r.stderr[-1500:]
    assert r.returncode == 0, tail
    last = [ln for ln in r.stdout.splitlines() if " passed" in ln][-1]
    assert "failed" not in last and "error" not in last, tail
    return int(last.split(" passed")[0].split()[-1])


def test_the_emulation_build_has_the_librarys_c_abi(emu_env):
    """Same symbols as the real library: every prototype of include/zkstark.h resolves in libzkstark_emu.so (and the emulator's own
    controls are there)."""
    import ctypes as C
    sys.path.insert(0, ROOT)
    from zk_evm_amd._lib import SIGNATURES
    lib = C.CDLL(emu_env["ZK_STARK_LIB"])
    for name in SIGNATURES:
        getattr(lib, name)
    for name in ("hipemu_counters", "hipemu_fail_malloc_from", "hipemu_fail_launch_at", "hipMalloc", "hipStreamWaitEvent"):
        getattr(lib, name)


def test_gpu_tests_pass_on_the_emulation_build(emu_env):
    ids = _ids("quick_slice.txt")
    assert len(ids) >= 120
    assert _run(ids, emu_env, 1500) >= len(ids)


def test_gpu_tests_pass_with_deferred_streams(emu_env):
    """HIPEMU_ASYNC=1: no operation runs until the host waits for something that depends on it, and then only that -- a missing
    event wait between the lanes, a pinned buffer reused before its copy kernel ran, an arena block handed out under a pending
    kernel would change the words of a proof here."""
    ids = _ids("async_slice.txt")
    assert len(ids) >= 20
    assert _run(ids, dict(emu_env, HIPEMU_ASYNC="1"), 1500) >= len(ids)


def test_a_failing_rank_never_strands_its_peer(emu_env):
    """Two ranks on the host transport, zk_commit_rows_sharded; rank 1's device is full, or one of its kernel launches is refused, at
    seven different points between and around the collectives.  Every time: rank 1 returns its own error, rank 0 ZK_ERR_COMM, both at
    once (the 60 s transport limit is never waited for), and the SAME communicator then commits the table with the right cap
    (csrc/comm_host.inc "the failure protocol"; r05 advisor)."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(emu_env, HIPEMU_THREADS="2", ZK_COMM_TIMEOUT_S="60", ZK_FAIL_AT="-1,1,2,4,6,8,10,40")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "emu", "multirank_failure_driver.py"), str(r), "2", str(port)],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT, env=env) for r in range(2)]
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-1500:] for o in outs]
    res = [json.loads([ln for ln in o[0].splitlines() if ln.startswith("RESULT ")][-1][7:]) for o in outs]
    r0, r1 = (res[0], res[1]) if res[0]["rank"] == 0 else (res[1], res[0])
    assert r0["transport"] == "host"
    failed = 0
    for a, b in zip(r0["runs"], r1["runs"]):
        assert a["fail_at"] == b["fail_at"] and a["cap0"] == b["cap0"] == r0["runs"][0]["cap0"]      # the retry always succeeds, same cap
        if b["code"] != 0:
            failed += 1
            assert b["code"] in (-2, -3) and a["code"] == -6, (a, b)                                   # own error | ZK_ERR_COMM
            assert a["seconds"] < 20 and b["seconds"] < 20, (a, b)                                     # nobody sat out a time limit
        else:
            assert a["code"] == 0, (a, b)
    assert failed >= 6 and r0["runs"][-1]["code"] == 0                                                  # (launch 40 does not exist: a clean run)


_TUNER_CHILD = r"""
import ctypes as C, json, sys
import zk_evm_amd as zk
ctx = zk.Context(0)
lib = ctx.lib
lib.zki_ntt_tune_range.restype = C.c_int
lib.zki_ntt_tune_range.argtypes = [C.c_void_p] + [C.c_int] * 5 + [C.POINTER(C.c_int)]
lib.zki_tree_batch_trial.restype = C.c_int
lib.zki_tree_batch_trial.argtypes = [C.c_void_p, C.POINTER(C.c_uint), C.POINTER(C.c_int)]
lib.zki_ntt_tune_report.restype = C.c_size_t
lib.zki_ntt_tune_report.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
ctx.set_plans("")
d, td = C.c_int(-1), C.c_int(-1)
rc1 = lib.zki_ntt_tune_range(ctx.handle, 10, 15, 10, 12, 1, C.byref(d))
rc2 = lib.zki_tree_batch_trial(ctx.handle, (C.c_uint * 9)(4, 4, 4, 4, 4, 4, 5, 4, 4), C.byref(td))
buf = C.create_string_buffer(1 << 16)
lib.zki_ntt_tune_report(ctx.handle, buf, len(buf))
print("RESULT " + json.dumps({"rc": [rc1, rc2], "differ": [d.value, td.value], "plans": ctx.get_plans(), "report": buf.value.decode()}))
"""


def test_the_offline_tuners_trials_find_identical_outputs(emu_env):
    """What `zk_ntt_tune` does on a device, at sizes a CPU can carry: both NTT plans of every transform shape 2^10 .. 2^15 (values ->
    coefficients, coefficients -> values with and without a free stage) on pseudo-random columns, compared word for word; the
    column-batch forms of from_values (one stream, two streams) digest for digest; a nine-table segment proven with the tree tops per
    tree and batched, proof word for proof word.  No second form may differ, and the plan string the trials leave in the ctx parses."""
    import re
    r = subprocess.run([sys.executable, "-c", _TUNER_CHILD], capture_output=True, text=True, cwd=ROOT, timeout=1500,
                       env=dict(emu_env, ZK_NTT_TUNE_ELEMS_LOG="15", ZK_NTT_TUNE_COLS="3"))
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])
    assert out["rc"] == [0, 0] and out["differ"] == [0, 0], out
    assert re.fullmatch(r"([vd][0-9]+f[01]=[12];|b[0-9]+r1=[0-9]+x[12];|T=[01];)+", out["plans"]), out["plans"]
    lines = out["report"].splitlines()
    assert not any("DIFFER" in ln or "failed" in ln for ln in lines), out["report"]
    assert sum(ln.startswith("ntt plan") for ln in lines) >= 10 and sum(ln.startswith("ntt column batches") for ln in lines) == 3
    assert any(ln.startswith("tree tops") and "differing words 0" in ln for ln in lines)


def test_the_emulator_checks_what_it_claims(tmp_path):
    """tests/emu/hipemu_selftest.cpp: a kernel that reverses a block through LDS WITH its __syncthreads() is clean in every mode (streams in
    order / deferred behind an event wait, plain / ThreadSanitizer); the same kernel WITHOUT the barrier is reported as a data race by
    the ThreadSanitizer build (every fiber a thread of its own, only barriers order them) -- so "no report" on the library's kernels
    (tools/emu_sanitizers.sh) means something.  Wave shuffles and threads that leave before a barrier are checked on the way."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    from build_emu import CXX
    from translate import translate
    src = tmp_path / "st.cpp"
    src.write_text(translate(open(os.path.join(ROOT, "tests", "emu", "hipemu_selftest.cpp")).read()))
    base = [CXX, "-std=c++17", "-O1", "-g", "-pthread", "-I", os.path.join(ROOT, "tests", "emu", "hipemu")]
    emu_cpp = os.path.join(ROOT, "tests", "emu", "hipemu", "hipemu.cpp")
    for tag, flags, emuflags in (("plain", [], []), ("tsan", ["-fsanitize=thread"], ["-fno-sanitize=thread", "-DHIPEMU_TSAN=1"])):
        subprocess.run(base + flags + ["-c", str(src), "-o", str(tmp_path / ("st_%s.o" % tag))], check=True)
        subprocess.run(base + flags + emuflags + ["-c", emu_cpp, "-o", str(tmp_path / ("emu_%s.o" % tag))], check=True)
        subprocess.run(base + flags + [str(tmp_path / ("st_%s.o" % tag)), str(tmp_path / ("emu_%s.o" % tag)), "-o", str(tmp_path / ("st_" + tag))], check=True)
    for tag in ("plain", "tsan"):
        for mode in ("ok", "racy"):
            for asy in ("0", "1"):
                r = subprocess.run([str(tmp_path / ("st_" + tag)), mode], capture_output=True, text=True, timeout=300,
                                   env=dict(os.environ, HIPEMU_ASYNC=asy, HIPEMU_THREADS="2", TSAN_OPTIONS="exitcode=66"))
                reports = r.stderr.count("WARNING: ThreadSanitizer: data race")
                if mode == "ok":
                    assert r.returncode == 0 and reports == 0 and "0 wrong values" in r.stdout, (tag, asy, r.stdout, r.stderr[-1500:])
                elif tag == "tsan":
                    assert r.returncode == 66 and reports >= 1 and "reverse_without_barrier" in r.stderr, (asy, r.stderr[-1500:])


def test_bench_prints_its_contract_line_on_the_emulation_build(emu_env):
    """bench.py end to end (one tiny nine-table segment per step, no secondaries) with the emulation build as its device: the LAST
    line of stdout is the contract's strict-JSON object -- metric, value, unit, n_gpus, steps, warmup, ms_per_step, roofline,
    config.workload -- and the sidecar names the plan table the run used (compiled in, empty).  Not a measurement of anything: the
    point is that the driver's command parses (r04's record did not) on the code as it is now."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--log-n", "6", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-pmc",
                        "--no-secondary", "--commit-steps", "0", "--in-flight", "1", "--no-dist-selftest"],
                       capture_output=True, text=True, cwd=ROOT, env=emu_env, timeout=1200)
    assert r.returncode == 0, r.stderr[-2000:]
    line = r.stdout.strip().splitlines()[-1]

    def bad(c):
        raise ValueError(c)
    d = json.loads(line, parse_constant=bad)
    assert len(line) < 4096 and d["unit"] == "segment proofs/s" and d["n_gpus"] == 1 and d["steps"] == 1 and d["warmup"] == 0
    assert d["value"] > 0 and abs(d["value"] - 1e3 / d["ms_per_step"]) < 1e-6 * d["value"] and d["higher_is_better"] is True
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["peak"] == 8000.0 and "workload" in d["config"] and "model" not in d["config"]
    assert d["ntt"]["lane_swap_plans"] == 0 and d["ntt"]["tree_tops_batched"] is False and d["ntt"]["plans_source"] == "compiled in"
    extra = json.load(open(os.path.join(ROOT, "bench_extra.json")))
    assert extra["ntt"]["plans"] == "" and extra["ntt"]["forced"] == {}




def test_one_process_drives_two_devices(emu_env):
    """SURVEY 8(e) level 1 inside ONE process: `SegmentScheduler(devices=[0, 1])` -- a worker thread, a zk_ctx and a stream per device --
    on an emulated node with two devices (HIPEMU_DEVICES=2).  Every proof equals the one a single ctx makes, and both devices worked.
    (What this pins: nothing in the library is set up once per PROCESS where it must be once per device / ctx -- r06 moved three
    kernel-attribute calls from static flags into the ctx.)"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "emu", "two_devices_driver.py")], capture_output=True, text=True, cwd=ROOT, timeout=1200,
                       env=dict(emu_env, HIPEMU_DEVICES="2"))
    assert r.returncode == 0, r.stderr[-2500:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])
    assert out["same"] is True and sorted(out["per_device"]) == ["0", "1"] and sum(out["per_device"].values()) == 2, out


def test_a_failing_rank_never_strands_its_peer_inside_a_table_proof(emu_env):
    """The same drill one level up: zk_prove_table_sharded (a row-sharded table proof: about fifteen collectives -- all-to-alls,
    all-gathers of caps, carries, openings, FRI values) with one of rank 1's kernel launches refused at six points spread over the
    proof.  Every time both ranks are back within seconds -- rank 1 with its own error, rank 0 with ZK_ERR_COMM -- and the SAME
    communicator then produces the reference proof on both ranks."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(emu_env, HIPEMU_THREADS="3", ZK_COMM_TIMEOUT_S="60", ZK_FAIL_AT="1,12,30,60,80,5000")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "emu", "multirank_prove_failure_driver.py"), str(r), "2", str(port)],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT, env=env) for r in range(2)]
    outs = [p.communicate(timeout=900) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-1500:] for o in outs]
    res = [json.loads([ln for ln in o[0].splitlines() if ln.startswith("RESULT ")][-1][7:]) for o in outs]
    r0, r1 = (res[0], res[1]) if res[0]["rank"] == 0 else (res[1], res[0])
    assert r0["reference"] == r1["reference"]
    failed = 0
    for a, b in zip(r0["runs"], r1["runs"]):
        assert a["fail_at"] == b["fail_at"] and a["retry_same"] is True and b["retry_same"] is True, (a, b)
        if b["code"] != 0:
            failed += 1
            assert b["code"] == -3 and a["code"] == -6 and a["seconds"] < 20 and b["seconds"] < 20, (a, b)
        else:
            assert a["code"] == 0 and a["same_as_reference"] is True and b["same_as_reference"] is True, (a, b)
    assert failed >= 4 and r0["runs"][-1]["code"] == 0
