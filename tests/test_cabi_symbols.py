"""CPU-only: the C-ABI library loads and exports every symbol include/zkstark.h declares; the
Python binding lists exactly those symbols; compute calls fail loudly without a GPU."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "zkstark.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(zk_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def built_lib():
    from zk_evm_amd import build
    path = build.build()
    return C.CDLL(path)


def test_all_declared_symbols_exported(built_lib):
    syms = _declared_symbols()
    assert len(syms) >= 30
    missing = [s for s in syms if not hasattr(built_lib, s)]
    assert not missing, missing


def test_binding_covers_header():
    from zk_evm_amd._lib import SIGNATURES
    assert sorted(SIGNATURES) == _declared_symbols()


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import zk_evm_amd
    with pytest.raises(zk_evm_amd.ZkStarkError):
        zk_evm_amd.Context(0)
    import numpy as np
    with pytest.raises(zk_evm_amd.ZkStarkError):
        zk_evm_amd.PolynomialBatch.from_values(np.zeros((2, 8), dtype=np.uint64), 1, False, 2)


def test_product_never_imports_oracle():
    # the product path (zk_evm_amd/) must not reference the oracle in any way
    for dirpath, _, files in os.walk(os.path.join(ROOT, "zk_evm_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cuh", ".cpp", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "liboracle" not in txt and "oracle_lib" not in txt, f
                assert not re.search(r"^\s*(from|import)\s+oracle", txt, flags=re.M), f
                assert '#include "../../oracle' not in txt and "oracle/" not in txt.replace("oracle/goldilocks.h", ""), f


def test_host_argument_validation():
    import numpy as np
    import zk_evm_amd
    from zk_evm_amd.polynomial_batch import PolynomialBatch
    with pytest.raises(zk_evm_amd.ZkStarkError):
        PolynomialBatch._log2(12)
    with pytest.raises(zk_evm_amd.ZkStarkError):
        PolynomialBatch.from_values([np.zeros(8, np.uint64), np.zeros(4, np.uint64)], 1, False, 2)
    with pytest.raises(zk_evm_amd.ZkStarkError):
        PolynomialBatch.from_values(np.zeros((2, 8), np.uint64), 1, True, 2)


def test_c_callers_compile_against_the_headers(tmp_path):
    """The plain-C callers (tests/cabi/*.c) compile with gcc -Wall -Werror against include/zkstark.h and the generated
    include/zk_all_stark.h -- ABI drift shows up here without a GPU (they run in tests/test_gpu_cabi_harness.py)."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    assert gcc, "gcc is part of the image"
    for src, extra in (("harness.c", []), ("segment.c", ["-I", "/opt/rocm/include", "-D__HIP_PLATFORM_AMD__"])):
        subprocess.run([gcc, "-std=c11", "-O1", "-Wall", "-Werror", "-c", os.path.join(ROOT, "tests", "cabi", src),
                        "-I", os.path.join(ROOT, "include")] + extra + ["-o", str(tmp_path / (src + ".o"))], check=True)
