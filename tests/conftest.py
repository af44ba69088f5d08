import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from tests.oracle_lib import load_oracle
    return load_oracle()


@pytest.fixture(scope="session", autouse=True)
def _ntt_plans_settled_once():
    """On a GPU box: the library's NTT plan trials (a helper process, csrc/ntt_host.inc) run once per session and their verdicts
    go into os.environ, so that the dozens of child processes the GPU tests start do not each repeat them."""
    try:
        import torch
        if torch.cuda.is_available():
            from zk_evm_amd._lib import settle_ntt_plans
            settle_ntt_plans(0)
    except Exception as e:          # the tests themselves will say what is wrong with the library
        sys.stderr.write("conftest: NTT plans not settled: %r\n" % (e,))
    yield
