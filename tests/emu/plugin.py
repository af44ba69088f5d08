"""TEST INFRASTRUCTURE.  pytest plugin (`-p tests.emu.plugin`) for running GPU-marked tests against the CPU emulation build: installs
tests/emu/torch_shim.py before any test module is imported.  Used by tests/test_emu_gpu_suite.py (a child pytest) and
tools/emu_sanitizers.sh; refuses to load in a process whose library is the real one."""
from tests.emu import torch_shim

torch_shim.install()
