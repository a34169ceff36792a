import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


def load_golden(name):
    d = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return {k: torch.from_numpy(d[k]) for k in d.files}


@pytest.fixture
def golden():
    return load_golden


def rel_l2(a: torch.Tensor, b: torch.Tensor) -> float:
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float(torch.linalg.vector_norm(a - b) / (torch.linalg.vector_norm(b) + 1e-30))


def assert_close(a, b, rtol, what="", atol_frac=None):
    """|a-b| <= rtol*|b| + atol with atol = atol_frac*max|b| (default atol_frac = rtol):
    the north-star tolerance (rtol 2e-2 bf16 / 1e-4 fp32) plus the max-scaled absolute
    floor SURVEY 8(c) prescribes for near-zero elements."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    atol = (rtol if atol_frac is None else atol_frac) * float(b.abs().max())
    err = (a - b).abs()
    bad = err > (rtol * b.abs() + atol)
    assert not bool(bad.any()), (
        f"{what}: {int(bad.sum())}/{bad.numel()} out of tol; max err {float(err.max()):.3e} "
        f"(max|ref| {float(b.abs().max()):.3e}), rel-L2 {rel_l2(a, b):.3e}")
