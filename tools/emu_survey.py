#!/usr/bin/env python3
"""Every GPU test (pytest -m gpu), one by one, against the CPU emulation build (tests/emu/), each in its own process
under a hard time limit -- a test that hangs or is simply too large for a CPU costs its limit, nothing more.  Writes a JSON map
test id -> {status, seconds, tail of the output if it failed}; tests/emu/quick_slice.txt (the slice `pytest -m "not gpu"` runs) is
cut from it.
    python tools/emu_survey.py [--limit 100] [--out /tmp/emu_survey.json] [--only <file with test ids>] [--async] [--workers 2]"""
import json
import os
import subprocess
import sys
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


# GPU tests the emulation cannot run at all: they start bench.py or RCCL (a real device and a real librccl), look for the real shared
# object in /proc/self/maps, or run the real zk_ntt_tune binary.  Reported as "n/a", never started.
NOT_EMULABLE = ("::test_bench_", "nccl", "rccl", "test_native_library_is_loaded", "test_first_multi_gpu_visit_script", "test_offline_tuner_finds_identical")


def emu_env(threads=None, async_streams=False, lib=None):
    from tests.emu import build_emu
    env = dict(os.environ, ZK_STARK_LIB=lib or build_emu.build(), HIPEMU_TORCH_SHIM="1",
               PYTHONPATH=os.path.join(ROOT, "tests", "emu", "site") + os.pathsep + ROOT)
    env.setdefault("ZK_COMM_TIMEOUT_S", "180")          # (host transport: a rank left alone by a killed test gives up soon)
    if threads:
        env["HIPEMU_THREADS"] = str(threads)
    if async_streams:
        env["HIPEMU_ASYNC"] = "1"
    return env


def main(argv):
    def opt(name, dflt):
        return argv[argv.index(name) + 1] if name in argv else dflt
    limit, out_path, workers = float(opt("--limit", 100)), opt("--out", "/tmp/emu_survey.json"), int(opt("--workers", 2))
    env = emu_env(threads=max(1, (os.cpu_count() or 2) // workers), async_streams="--async" in argv)
    if "--only" in argv:
        ids = [ln.strip() for ln in open(opt("--only", "")) if "::" in ln]
    else:
        r = subprocess.run([sys.executable, "-m", "pytest", "tests", "-m", "gpu", "--collect-only", "-q", "-p", "no:cacheprovider"],
                           capture_output=True, text=True, cwd=ROOT, env=env)
        ids = [ln.strip() for ln in r.stdout.splitlines() if "::" in ln]
    print(len(ids), "tests", flush=True)

    def run(tid):
        import signal
        if any(pat in tid for pat in NOT_EMULABLE):
            return tid, "n/a", 0.0, ""
        t0 = time.time()
        # its own session: a test that times out takes the ranks / child interpreters it started with it (a multi-rank test left
        # alone keeps spinning at a barrier for the transport's whole time limit)
        p = subprocess.Popen([sys.executable, "-m", "pytest", tid, "-q", "-x", "-p", "no:cacheprovider"], stdout=subprocess.PIPE,
                             stderr=subprocess.PIPE, text=True, cwd=ROOT, env=env, start_new_session=True)
        try:
            so, se = p.communicate(timeout=limit)
            status = "passed" if p.returncode == 0 else ("skipped" if " skipped" in so and "failed" not in so else "failed")
            tail = (so[-1500:] + se[-500:]) if status == "failed" else ""
        except subprocess.TimeoutExpired:
            status, tail = "timeout", ""
        finally:
            try:
                os.killpg(p.pid, signal.SIGKILL)
            except (ProcessLookupError, PermissionError):
                pass
            if p.poll() is None:
                p.communicate()
        return tid, status, round(time.time() - t0, 1), tail
    out = {}
    with ThreadPoolExecutor(max_workers=workers) as ex:
        for tid, status, dt, tail in ex.map(run, ids):
            out[tid] = {"status": status, "s": dt, "tail": tail}
            print(status, dt, tid, flush=True)
            with open(out_path, "w") as f:
                json.dump(out, f, indent=0)
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
