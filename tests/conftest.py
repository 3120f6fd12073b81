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
    full = sys.modules.get("test_gpu_full_size") or sys.modules.get("tests.test_gpu_full_size")
    for cfg, side, n, entries, worst, length, frob, per_class in (getattr(full, "WORST", None) or []):
        terminalreporter.write_line("full size %s, %s half: %d rows (%d entries) vs the oracle, worst row %.3e (a row of %d entries), "
                                    "relative Frobenius %.3e, bar %.0e per row; rows per length class %s"
                                    % (cfg, side, n, entries, worst, length, frob, getattr(full, "ROW_TOL", 0.0),
                                       ", ".join("%d-%s: %d" % (lo, "inf" if hi >= (1 << 40) else hi, c) for (lo, hi), c in per_class.items())))
