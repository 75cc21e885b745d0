import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
for p in (ROOT, GOLDEN):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = np.load(os.path.join(GOLDEN, name))
        return cache[name]
    return load


def rel_err(a, b):
    """max |a-b| / max(1, max|b|): the 1e-3 (fp32) bar of BASELINE.json:north_star, made scale-aware."""
    import torch
    a = torch.as_tensor(a).detach().to('cpu', torch.float64)
    b = torch.as_tensor(b).detach().to('cpu', torch.float64)
    return float((a - b).abs().max() / max(1.0, float(b.abs().max())))
