import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # skip REASONS in the short summary of every run (`-rs`), also under the driver's plain `-x -q`: a self-skipping test
    # (the 2-GPU RCCL step on a one-GPU box) must say why in the log nobody watches
    if "s" not in (config.option.reportchars or ""):
        config.option.reportchars = (config.option.reportchars or "") + "s"


@pytest.fixture
def emulate_hip(monkeypatch):
    """Replace the ctypes kernel layer by its plain-PyTorch statement (tests/_emu.py) so host
    logic (autograd wiring, module graphs, step functions) can run on CPU.  Test-only."""
    import _emu
    from indm_amd import hip

    for name in dir(_emu):
        if name.startswith("_"):
            continue
        obj = getattr(_emu, name)
        if callable(obj) and hasattr(hip, name):
            monkeypatch.setattr(hip, name, obj)
    yield
