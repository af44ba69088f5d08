"""TEST INFRASTRUCTURE.  Lets the GPU parity tests (tests/test_gpu_*.py) and the Python mirror (zk_evm_amd/*.py) run UNCHANGED in a
process whose libzkstark is the CPU emulation build (ZK_STARK_LIB = tests/emu/build/libzkstark_emu.so, tests/emu/build_emu.py): in
that build a "device" pointer is a host pointer, so a CPU torch tensor's data_ptr() is a valid device buffer.  install() makes torch
agree: `.cuda()` / `.to("cuda")` copy on the CPU, `device="cuda:0"` allocates on the CPU, torch.cuda.* answers as one idle device
whose streams are hipemu's.  Nothing here is imported by the product, and nothing is installed unless ZK_STARK_LIB names an
emulation build."""
import contextlib
import ctypes as C
import os

_installed = False


def emulated() -> bool:
    return "libzkstark_emu" in os.path.basename(os.environ.get("ZK_STARK_LIB", ""))


def install():
    global _installed
    if _installed:
        return
    assert emulated(), "tests/emu/torch_shim.py is for processes running the emulation build (ZK_STARK_LIB)"
    import torch
    emu = C.CDLL(os.environ["ZK_STARK_LIB"])
    emu.hipDeviceSynchronize.restype = C.c_int
    emu.hipStreamCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
    emu.hipStreamSynchronize.argtypes = [C.c_void_p]

    def is_cuda_dev(d):
        if d is None:
            return False
        if isinstance(d, torch.device):
            return d.type == "cuda"
        if isinstance(d, int):
            return True
        return isinstance(d, str) and d.startswith("cuda")

    class Stream:
        def __init__(self, *a, **k):
            h = C.c_void_p()
            assert emu.hipStreamCreateWithFlags(C.byref(h), 1) == 0
            self.cuda_stream = h.value

        def synchronize(self):
            emu.hipStreamSynchronize(C.c_void_p(self.cuda_stream))

        def wait_stream(self, other):
            other.synchronize()

    class NullStream(Stream):
        def __init__(self):
            self.cuda_stream = 0

    cur = [NullStream()]

    @contextlib.contextmanager
    def stream_ctx(s):
        prev = cur[0]
        cur[0] = s if s is not None else prev
        try:
            yield
        finally:
            cur[0] = prev

    @contextlib.contextmanager
    def device_ctx(_d):
        yield

    tc = torch.cuda
    tc.is_available = lambda: True
    tc.device_count = lambda: 1
    tc.set_device = lambda d: None
    tc.current_device = lambda: 0
    tc.synchronize = lambda device=None: emu.hipDeviceSynchronize() and None
    tc.current_stream = lambda device=None: cur[0]
    tc.Stream = Stream
    tc.stream = stream_ctx
    tc.device = device_ctx
    tc.empty_cache = lambda: None
    tc.get_device_name = lambda d=None: "hipemu (CPU emulation)"

    T = torch.Tensor

    def sync_current():
        # what torch does before the host reads a CUDA tensor: the copy is ordered on the CURRENT stream and waited for.  With
        # deferred streams (HIPEMU_ASYNC=1) this is what makes the kernels the test enqueued on that stream run.
        emu.hipStreamSynchronize(C.c_void_p(cur[0].cuda_stream))

    def cpu(self, *a, **k):
        sync_current()
        return self.clone()
    T.cuda = lambda self, *a, **k: self.clone()
    T.cpu = cpu
    for name in ("item", "tolist", "numpy"):
        def make(orig):
            def f(self, *a, **k):
                sync_current()
                return orig(self, *a, **k)
            return f
        setattr(T, name, make(getattr(T, name)))
    orig_equal = torch.equal

    def equal(a, b):
        sync_current()
        return orig_equal(a, b)
    torch.equal = equal
    T.is_cuda = property(lambda self: True)
    orig_to = T.to

    def to(self, *a, **k):
        if a and is_cuda_dev(a[0]) and not isinstance(a[0], torch.dtype):
            a = a[1:]
            out = orig_to(self, *a, **k) if (a or k) else self
            return out.clone() if out is self else out
        if is_cuda_dev(k.get("device")):
            k = dict(k)
            k.pop("device")
            out = orig_to(self, *a, **k) if (a or k) else self
            return out.clone() if out is self else out
        return orig_to(self, *a, **k)
    T.to = to

    def strip(fn):
        def wrapped(*a, **k):
            if is_cuda_dev(k.get("device")):
                k = dict(k)
                k.pop("device")
            return fn(*a, **k)
        return wrapped
    for name in ("empty", "zeros", "ones", "full", "randint", "tensor", "arange", "zeros_like", "empty_like", "rand", "randn", "as_tensor"):
        setattr(torch, name, strip(getattr(torch, name)))
    orig_gen = torch.Generator
    torch.Generator = lambda device=None: orig_gen()
    _installed = True
