import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, GOLDEN):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """`-m gpu` tests need a GPU: skip (not fail) them when collected on a box without one."""
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return load


def rel_err(a, b):
    """max|a-b| / max|b|  (per-tensor normalised max error) on numpy/torch inputs."""
    import torch
    def cv(t):
        if isinstance(t, torch.Tensor):
            return t.detach().cpu().double()
        return torch.as_tensor(np.asarray(t)).double()
    a, b = cv(a), cv(b)
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def rel_l2(a, b):
    """||a-b||_2 / ||b||_2 - the "1e-3 rel" gate for tensors that pass through bf16 rounding points
    (isolated one-ulp rounding flips are unavoidable there and do not move the L2 error)."""
    import torch

    def cv(t):
        if isinstance(t, torch.Tensor):
            return t.detach().cpu().double()
        return torch.as_tensor(np.asarray(t)).double()
    a, b = cv(a), cv(b)
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def report(line):
    """Append a measured value to gpurun_out/parity_report.txt (kept under profiles/ per round) when that directory exists."""
    d = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, "parity_report.txt"), "a") as f:
            f.write(line + "\n")
