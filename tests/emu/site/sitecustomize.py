"""TEST INFRASTRUCTURE.  On PYTHONPATH only in processes started by the emulation tests (tests/test_emu_*.py, tools/emu_sanitizers.sh):
every Python child of such a process -- pytest workers, `python -m tests.fuzz_parity`, bench.py ranks, torch.multiprocessing spawns --
gets tests/emu/torch_shim.py installed before its own code runs, so that it, too, treats the CPU emulation build of libzkstark
(ZK_STARK_LIB) as its device.  Does nothing unless HIPEMU_TORCH_SHIM=1 and ZK_STARK_LIB names an emulation build."""
import os
import sys

if os.environ.get("HIPEMU_TORCH_SHIM") == "1" and "libzkstark_emu" in os.path.basename(os.environ.get("ZK_STARK_LIB", "")):
    _root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    if _root not in sys.path:
        sys.path.insert(0, _root)
    try:
        from tests.emu import torch_shim
        torch_shim.install()
    except Exception as e:          # never break an interpreter that has nothing to do with the emulation
        sys.stderr.write("tests/emu/site/sitecustomize.py: torch shim not installed: %r\n" % (e,))
