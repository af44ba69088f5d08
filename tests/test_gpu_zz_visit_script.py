"""GPU: tools/first_multi_gpu_visit.sh on this box's one GPU.  Sorted after the parity tests on purpose (like test_gpu_zz_plans.py): the
script was written in a round without GPU access, and `pytest -x` should report the rows of SURVEY section 8 before it gets here."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu


def test_first_multi_gpu_visit_script_on_one_gpu(tmp_path):
    """tools/first_multi_gpu_visit.sh -- the one command for the first node with several GPUs -- on THIS box's single GPU: every step
    goes through RCCL on one rank (--force-dist), and the JSONL it writes has one parsed bench line per step with the fields the
    visit is about (the piece drill, the latency modes' proof check)."""
    import json
    import shutil
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ZK_VISIT_STEPS="2", ZK_VISIT_LOG_N="12", ZK_VISIT_FORCE_DIST="1", GRAFT_REPO_ROOT=root)
    r = subprocess.run(["bash", os.path.join(root, "tools", "first_multi_gpu_visit.sh"), "1", "visit_test"], capture_output=True, text=True,
                       cwd=root, env=env, timeout=1500)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    path = os.path.join(root, "gpurun_out", "visit_test_multi_gpu.jsonl")
    recs = [json.loads(ln) for ln in open(path)]
    assert [x["step"] for x in recs] == ["level1_gpus1", "level2_table_parallel", "level3_keccak_rows"], recs
    for x in recs:
        assert x["rc"] == 0 and x.get("value"), x
        assert (x.get("dist") or {}).get("backend") == "nccl", x
    for f in os.listdir(os.path.join(root, "gpurun_out")):
        if f.startswith("visit_test_"):
            os.remove(os.path.join(root, "gpurun_out", f))
