"""CPU-only: the NTT plan trials run in a process of their own (csrc/ntt_host.inc ntt_tune_isolated), and whatever happens to
that process -- verdicts, a crash, a hang, a bad exit, no helper at all -- the calling process gets an answer and keeps running.

The helper is replaced by shell scripts (ZK_NTT_TUNE_HELPER); no device is involved: zki_ntt_swap_verdict is the device-free part
of ntt_swap_decide."""
import os
import stat
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = textwrap.dedent("""
    import ctypes as C, json, sys
    from zk_evm_amd import build
    lib = C.CDLL(build.build())
    lib.zki_ntt_swap_verdict.restype = C.c_int
    lib.zki_ntt_swap_verdict.argtypes = [C.c_int] * 4
    lib.zki_ntt_tune_report.restype = C.c_size_t
    lib.zki_ntt_tune_report.argtypes = [C.c_char_p, C.c_size_t]
    libc = C.CDLL(None)
    libc.getenv.restype = C.c_char_p
    libc.getenv.argtypes = [C.c_char_p]
    shapes = [(0, 20, 0), (1, 21, 1), (1, 20, 0), (0, 13, 0), (1, 30, 1)]
    trees_first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    lib.zki_tree_batch_verdict.restype = C.c_int
    lib.zki_tree_batch_verdict.argtypes = [C.c_int]
    trees = lib.zki_tree_batch_verdict(0) if trees_first else None
    got = [lib.zki_ntt_swap_verdict(0, *s) for s in shapes]
    trees = lib.zki_tree_batch_verdict(0) if trees is None else trees
    lib.zki_ntt_batch_verdict.restype = C.c_int
    lib.zki_ntt_batch_verdict.argtypes = [C.c_int] * 3
    batches = [lib.zki_ntt_batch_verdict(0, l, 1) for l in (16, 17, 20, 21, 22)]
    again = [lib.zki_ntt_swap_verdict(0, *s) for s in shapes]
    buf = C.create_string_buffer(1 << 16)
    lib.zki_ntt_tune_report(buf, len(buf))
    print(json.dumps({"got": got, "again": again, "report": buf.value.decode(), "trees": trees, "batches": batches,
                      "env": (libc.getenv(b"ZK_NTT_SWAP_PLANS") or b"").decode()}))
""")


def _run(tmp_path, helper_body, extra_env=None, timeout=60, trees_first=0):
    env = {k: v for k, v in os.environ.items() if not k.startswith("ZK_NTT_")}
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    count = tmp_path / "calls"
    if count.exists():
        count.unlink()
    if helper_body is not None:
        h = tmp_path / "helper.sh"
        h.write_text("#!/bin/bash\necho x >> %s\n%s\n" % (count, helper_body))
        h.chmod(h.stat().st_mode | stat.S_IXUSR)
        env["ZK_NTT_TUNE_HELPER"] = str(h)
    else:
        env["ZK_NTT_TUNE_HELPER"] = str(tmp_path / "no_such_helper")
    env.update(extra_env or {})
    r = subprocess.run([sys.executable, "-c", CHILD, str(trees_first)], env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    import json
    out = json.loads(r.stdout.strip().splitlines()[-1])
    out["calls"] = len(count.read_text().split()) if count.exists() else 0
    return out


def test_helper_verdicts_are_used_once_and_exported(tmp_path):
    out = _run(tmp_path, 'echo "v20f0=2;d21f1=1;d20f0=2;"; echo "ntt plan values->coefficients 2^20 ... -> lane-swap"')
    assert out["got"] == [2, 1, 2, 1, -1]          # 2^13: not listed by the helper -> tile; 2^30: no second plan
    assert out["again"] == out["got"]
    assert out["calls"] == 1                        # one helper run per process and device
    assert "v20f0=2;" in out["env"] and "d21f1=1;" in out["env"]
    assert "lane-swap" in out["report"]


@pytest.mark.parametrize("trees_first", [0, 1])
def test_tree_batch_verdict_comes_from_the_same_helper_run(tmp_path, trees_first):
    out = _run(tmp_path, 'echo "v20f0=2;d21f1=1;T=1;"; echo "tree tops ... -> batched"', trees_first=trees_first)
    assert out["trees"] == 1 and out["got"][:2] == [2, 1]
    assert out["calls"] == 1                        # whichever decision is asked for first starts the helper; the other reuses it
    assert "T=1;" in out["env"]
    out = _run(tmp_path, 'echo "v20f0=2;"', trees_first=trees_first)      # a helper that skipped the tree trial: per tree
    assert out["trees"] == 0
    out = _run(tmp_path, 'echo "v20f0=2;T=1;"', extra_env={"ZK_TREE_BATCH": "0"})       # forced off / on: no question asked
    assert out["trees"] == 0
    out = _run(tmp_path, 'kill -SEGV $$', extra_env={"ZK_TREE_BATCH": "1"})
    assert out["trees"] == 1 and out["got"][0] == 1
    out = _run(tmp_path, 'kill -SEGV $$', extra_env={"ZK_NTT_SWAP_PLANS": "T=1;"}, trees_first=1)
    assert out["trees"] == 1 and out["got"][0] == 1 and "T=0" not in out["env"]      # an inherited verdict stands


def test_column_batch_verdicts_come_from_the_helper_too(tmp_path):
    out = _run(tmp_path, 'echo "v20f0=2;b17r1=0x1;b20r1=96x2;b21r1=96x1;T=0;"')
    # 2^16 and 2^22 rows have no trial (-1: all at once, nobody asked); 2^17: all at once; 2^20: 96 MiB on two streams; 2^21: on one
    assert out["batches"] == [-1, 0 * 4 + 1, 96 * 4 + 2, 96 * 4 + 1, -1]
    assert out["calls"] == 1
    out = _run(tmp_path, 'echo "v20f0=2;T=0;"')                       # a helper without batch verdicts: all at once
    assert out["batches"] == [-1, 1, 1, 1, -1]
    out = _run(tmp_path, "kill -SEGV $$")                             # a dead helper: all at once, and the children are told
    assert out["batches"] == [-1, 1, 1, 1, -1] and "b20r1=0x1;" in out["env"] and "b21r1=0x1;" in out["env"]
    out = _run(tmp_path, "kill -SEGV $$", extra_env={"ZK_NTT_COL_BATCH_MB": "64", "ZK_NTT_COL_BATCH_STREAMS": "2"})      # forced
    assert out["batches"] == [64 * 4 + 2] * 5
    out = _run(tmp_path, "kill -SEGV $$", extra_env={"ZK_NTT_TUNE_INPROC": "1"})                 # the helper itself: it runs the trial
    assert out["batches"] == [-1] * 5 and out["calls"] == 0


def test_helper_gets_inproc_switch_and_device(tmp_path):
    out = _run(tmp_path, 'echo "dev=$1 inproc=$ZK_NTT_TUNE_INPROC swap=$ZK_NTT_SWAP plans=$ZK_NTT_SWAP_PLANS" >> %s; echo "v20f0=2;"'
               % (tmp_path / "seen"), extra_env={"ZK_NTT_SWAP_PLANS": "d21f1=2;"})
    assert (tmp_path / "seen").read_text().strip() == "dev=0 inproc=1 swap=2 plans="
    assert out["got"][:2] == [2, 2]                 # the inherited verdict and the helper's
    assert out["env"].startswith("d21f1=2;")


@pytest.mark.parametrize("body,why", [
    ("kill -SEGV $$", "died of signal 11"),
    ("kill -ABRT $$", "died of signal 6"),
    ("exit 4", "exited with status 4"),
    ('echo "rm -rf /; not a verdict"', "not a verdict list"),
])
def test_dead_helper_means_tile_kernels_everywhere(tmp_path, body, why):
    out = _run(tmp_path, body)
    assert out["got"] == [1, 1, 1, 1, -1]
    assert why in out["report"] and "tile kernels for every shape" in out["report"]
    assert out["calls"] == 1
    # the children of this process must not try again: every shape is pinned to the tile kernels in the environment
    assert "v20f0=1;" in out["env"] and "d21f1=1;" in out["env"] and "d22f1=1;" in out["env"] and "T=0;" in out["env"]
    assert out["trees"] == 0


def test_hung_helper_is_killed(tmp_path):
    out = _run(tmp_path, "exec sleep 600", extra_env={"ZK_NTT_TUNE_TIMEOUT_S": "2"}, timeout=40)
    assert out["got"] == [1, 1, 1, 1, -1]
    assert "time limit" in out["report"]


def test_no_helper(tmp_path):
    out = _run(tmp_path, None)
    assert out["got"] == [1, 1, 1, 1, -1]
    assert "no helper at" in out["report"]


def test_inherited_verdicts_need_no_helper(tmp_path):
    out = _run(tmp_path, "kill -SEGV $$", extra_env={"ZK_NTT_SWAP_PLANS": "v20f0=2;d21f1=2;d20f0=1;v13f0=2;T=0;b17r1=0x1;b20r1=96x1;b21r1=0x1;"})
    assert out["got"] == [2, 2, 1, 2, -1]
    assert out["calls"] == 0


def test_inproc_mode_never_spawns(tmp_path):
    out = _run(tmp_path, "kill -SEGV $$", extra_env={"ZK_NTT_TUNE_INPROC": "1"})
    assert out["got"] == [0, 0, 0, 0, -1]          # 0 = "run the trial here" (what the helper process itself sees)
    assert out["trees"] == -1                       # no verdict, none to be had here: per tree
    assert out["calls"] == 0


def test_real_helper_is_found_next_to_the_library_and_its_failure_is_survived(tmp_path):
    """No ZK_NTT_TUNE_HELPER: the library locates zk_ntt_tune through its own path (dladdr).  Without a GPU the helper cannot create a
    context and exits 3 -- which is one more way for it to fail: tile kernels everywhere."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: the real helper would succeed (tests/test_gpu_tune.py)")
    env = {k: v for k, v in os.environ.items() if not k.startswith("ZK_NTT_")}
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    import json
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["got"] == [1, 1, 1, 1, -1] and out["trees"] == 0
    assert "exited with status 3" in out["report"], out["report"]


def test_helper_is_built_next_to_the_library():
    from zk_evm_amd import build
    build.build()
    assert os.access(build.TUNE, os.X_OK)
    assert os.path.dirname(build.TUNE) == os.path.dirname(build.OUT)
