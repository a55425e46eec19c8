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


@pytest.fixture
def emu(emu_library):
    from action_detection_amd import _lib
    _lib.use_library_for_testing(emu_library)
    yield emu_library
    _lib.use_library_for_testing(None)


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
        lib = request.getfixturevalue("emu_library")
        _lib.use_library_for_testing(lib)
        yield Backend("emu", "cpu")
        _lib.use_library_for_testing(None)
    else:
        request.getfixturevalue("hip_library")
        yield Backend("gpu", "cuda:0")
