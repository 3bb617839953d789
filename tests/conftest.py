import os
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent
GOLDEN = ROOT / "tests" / "golden"
for p in (ROOT, ROOT / "generative-turbulence_amd"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # GPU tests are skipped automatically where no GPU is visible (build container)
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


class Golden:
    """Lazy access to tests/golden/<name>.npz with '/'-separated keys."""

    def __init__(self, name):
        self.z = np.load(GOLDEN / f"{name}.npz")

    def __getitem__(self, key):
        return torch.from_numpy(np.asarray(self.z[key]))

    def keys(self, prefix=""):
        return [k for k in self.z.files if k.startswith(prefix)]

    def sub(self, prefix):
        """dict of tensors below prefix (prefix stripped)."""
        return {k[len(prefix):]: self[k] for k in self.keys(prefix)}


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = Golden(name)
        return cache[name]

    return get


def rel_l2(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def assert_grad_close(name, a, b, tol, noise_floor=1e-6):
    """rel-L2 check; gradients that are mathematically zero (a conv bias in front of a
    GroupNorm) only hold rounding noise in the reference, so they are checked absolutely."""
    if b.double().norm().item() < noise_floor:
        assert a.double().norm().item() < 10 * noise_floor, f"{name}: expected ~0, got {a.norm().item()}"
    else:
        err = rel_l2(a, b)
        assert err < tol, f"{name}: rel-L2 {err:.3e} >= {tol}"
