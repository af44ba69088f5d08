"""CPU: the lane-swap NTT kernels run FROM THEIR OWN SOURCE (zk_evm_amd/csrc/ntt_swap.cuh compiled by g++ against the shim under
tests/emu/: one OS thread per lane, __syncthreads a barrier over the workgroup, the v_permlane16/32_swap exchange and the wave-local
LDS synchronisation rendezvous of a wave's 64 threads) against the plain definition of each pass: strided passes with 7 .. 10 row
bits in both directions, the one-wave contiguous passes with every store factor, one and two cosets, with and without load factors.
Complements tests/test_ntt_swap_model.py (a Python restatement of the index algebra): this one executes the C++ that ships.  What
neither can check is the hardware's own semantics of the two swap instructions -- tests/test_gpu_commit.py does, on a GPU."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_kernels_compiled_for_the_cpu_match_the_pass_definitions(tmp_path):
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("no g++")
    exe = str(tmp_path / "ntt_swap_emu")
    r = subprocess.run([gxx, "-std=c++17", "-O1", "-pthread", "-Wno-attributes", "-DZK_NTT_EMULATE", "-I", os.path.join(ROOT, "tests", "emu"),
                        "-I", os.path.join(ROOT, "zk_evm_amd", "csrc"), os.path.join(ROOT, "tests", "emu", "ntt_swap_emu.cpp"), "-o", exe],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0 and "ALL OK" in r.stdout and "FAIL" not in r.stdout, r.stdout[-3000:]
    assert r.stdout.count("\nok ") + r.stdout.startswith("ok ") >= 17


def test_whole_transforms_through_the_new_plan_equal_the_tile_kernels(tmp_path):
    """tests/emu/ntt_plan_emu.cpp: tables built by the library's own table kernels, `from_values`' two transforms of a 2^17-row
    column run pass by pass through the LDS tile kernels (the code every GPU parity test has pinned since r01) and through the
    lane-swap kernels in the plan ntt_host.inc gives them -- strided R = 7, the one-wave contiguous kernels, the extension's two
    cosets with the second load-factor table -- compared word for word, and against direct evaluation of the polynomial.  About a
    minute of OS-thread rendezvous."""
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("no g++")
    exe = str(tmp_path / "ntt_plan_emu")
    r = subprocess.run([gxx, "-std=c++17", "-O1", "-pthread", "-Wno-attributes", "-DZK_NTT_EMULATE", "-I", os.path.join(ROOT, "tests", "emu"),
                        "-I", os.path.join(ROOT, "zk_evm_amd", "csrc"), os.path.join(ROOT, "tests", "emu", "ntt_plan_emu.cpp"), "-o", exe],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe, "quick"], capture_output=True, text=True, timeout=1800)
    assert r.returncode == 0 and "ALL OK" in r.stdout and "FAIL" not in r.stdout, r.stdout[-3000:]
