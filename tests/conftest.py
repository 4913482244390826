import os
import sys

import pytest

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long CPU test")


def _has_gpu():
    if os.environ.get("ARK355_NO_TORCH"):          # diagnostic mode: plain HIP runtime, torch never imported
        return os.path.exists("/dev/kfd")
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def emul_lib():
    """The library's own sources compiled against the CPU HIP emulator (tests/emul) -- checker only."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "emul"))
    import build_emul
    from snark_amd._binding import Lib
    return Lib(build_emul.build())


@pytest.fixture(scope="session")
def emul_ctx(emul_lib):
    ctx = emul_lib.ctx_create(0)
    yield ctx
    emul_lib.ctx_destroy(ctx)


@pytest.fixture(scope="session")
def gpu_lib():
    if not _has_gpu():
        pytest.skip("no GPU")
    import snark_amd
    return snark_amd.lib()


@pytest.fixture(scope="session")
def gpu_ctx(gpu_lib):
    ctx = gpu_lib.ctx_create(0)
    yield ctx
    gpu_lib.ctx_destroy(ctx)
