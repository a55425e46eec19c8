import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow_emu: long host-emulator run, enabled with SSN_SLOW=1")


@pytest.fixture(scope="session")
def emu_library():
    """Host-emulation build of the product's .hip sources (tests/emu) -- CPU tier only."""
    from action_detection_amd import _lib
    prebuilt = os.environ.get("SSN_EMU_LIB")      # a library built elsewhere (EMU_OUT of build_emu.sh)
    if prebuilt:
        return _lib.SsnLibrary(prebuilt, is_emulator=True)
    subprocess.check_call([os.path.join(ROOT, "tests", "emu", "build_emu.sh")], stdout=subprocess.DEVNULL)
    return _lib.SsnLibrary(os.path.join(ROOT, "tests", "emu", "libssn_emu.so"), is_emulator=True)


def _emu_violations(lib):
    import ctypes
    buf = ctypes.create_string_buffer(300)
    lib.cdll.ssn_emu_alloc_violations.restype = ctypes.c_long
    n = lib.cdll.ssn_emu_alloc_violations(buf, 300)
    return int(n), buf.value.decode()


@pytest.fixture
def emu(emu_library, monkeypatch):
    """The emulator as the active library -- with every planes tensor created during the test REGISTERED as an allocation, so that
    a buffer descriptor that starts inside one and claims bytes past its end (an over-read the GPU's range check would let through)
    fails the test (tests/emu/emu_globals.cpp)."""
    import ctypes
    import weakref
    from action_detection_amd import _lib, planes
    _lib.use_library_for_testing(emu_library)
    cd = emu_library.cdll
    cd.ssn_emu_alloc_register.argtypes = [ctypes.c_void_p, ctypes.c_long]
    cd.ssn_emu_alloc_unregister.argtypes = [ctypes.c_void_p]
    init = planes.PlaneTensor.__init__

    def registering_init(self, *a, **k):
        init(self, *a, **k)
        if not self.data.is_cuda:
            ptr = self.data.data_ptr()
            cd.ssn_emu_alloc_register(ptr, self.data.numel() * self.data.element_size())
            weakref.finalize(self.data, cd.ssn_emu_alloc_unregister, ptr)
    monkeypatch.setattr(planes.PlaneTensor, "__init__", registering_init)
    _emu_violations(emu_library)
    yield emu_library
    _lib.use_library_for_testing(None)
    n, msg = _emu_violations(emu_library)
    assert n == 0, "%d buffer descriptor(s) reached past their allocation; first: %s" % (n, msg)


@pytest.fixture(scope="session")
def hip_library():
    """The real gfx950 library; GPU tests must run through it (never the emulator)."""
    import torch
    import action_detection_amd as pkg
    from action_detection_amd import _lib
    assert torch.cuda.is_available(), "GPU test selected without a GPU"
    pkg.build()
    _lib.use_library_for_testing(None)
    lib = _lib.get_lib()
    assert not lib.is_emulator
    return lib


class Backend:
    def __init__(self, name, device):
        self.name = name
        self.device = device

    def put(self, t):
        return t.to(self.device) if t is not None else None

    @property
    def is_gpu(self):
        return self.name == "gpu"


@pytest.fixture(params=["emu", pytest.param("gpu", marks=pytest.mark.gpu)])
def backend(request):
    """Run a kernel test through the host emulator (CPU tier) or the real library (GPU tier)."""
    from action_detection_amd import _lib
    if request.param == "emu":
        request.getfixturevalue("emu")      # (installs the emulator + the allocation registry; its teardown checks the descriptors)
        yield Backend("emu", "cpu")
    else:
        request.getfixturevalue("hip_library")
        yield Backend("gpu", "cuda:0")
