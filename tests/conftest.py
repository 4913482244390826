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


class _PolicyShim:
    """monkeypatch.setenv-shaped access to a context's runtime policy (ark355_ctx_set_policy): the library reads the
    environment only when a context is created, and the test contexts are session-scoped."""

    def __init__(self, lib, ctx):
        self.lib, self.ctx, self.saved = lib, ctx, []

    def setenv(self, name, value):
        assert name.startswith("ARK355_")
        name = name[len("ARK355_"):]
        if name in ("SERIAL", "EPILOGUE_SYNC"):
            self.saved.append(("SCHED", self.lib.ctx_get_policy(self.ctx, "SCHED")))
        else:
            self.saved.append((name, self.lib.ctx_get_policy(self.ctx, name)))
        self.lib.ctx_set_policy(self.ctx, name, int(value))

    def restore(self):
        for name, old in reversed(self.saved):
            self.lib.ctx_set_policy(self.ctx, name, old)


@pytest.fixture
def emul_policy(emul_lib, emul_ctx):
    shim = _PolicyShim(emul_lib, emul_ctx)
    yield shim
    shim.restore()


@pytest.fixture
def gpu_policy(gpu_lib, gpu_ctx):
    shim = _PolicyShim(gpu_lib, gpu_ctx)
    yield shim
    shim.restore()
