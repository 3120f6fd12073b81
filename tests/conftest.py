import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def pytest_terminal_summary(terminalreporter):
    """The largest per-row deviation from the oracle the parity sweeps saw in this session (tests/test_gpu_fuzz.py)."""
    mod = sys.modules.get("test_gpu_fuzz") or sys.modules.get("tests.test_gpu_fuzz")
    w = getattr(mod, "WORST_ROW", None) if mod else None
    if w and w["where"] is not None:
        terminalreporter.write_line("parity sweeps: worst single row vs the oracle %.3e (%s), bar %.0e" % (w["value"], w["where"], getattr(mod, "ROW_TOL", 0.0)))
