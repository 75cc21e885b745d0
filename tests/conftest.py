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


@pytest.fixture(autouse=True)
def _kernel_coverage(request):
    """Every `-m gpu` test runs inside the library's dispatch trace (include/istgcn.h: istgcn_trace); the kernels it launched
    go to gpurun_out/kernel_coverage.tsv as `test id<TAB>launches<TAB>kernel symbol`.  tools/kernel_coverage.py turns the file
    into the kernel -> tests table of DESIGN.md (which dispatch branch has which golden test; a kernel of the library that no
    test launches shows up as such)."""
    if request.node.get_closest_marker('gpu') is None:
        yield
        return
    import torch
    if not torch.cuda.is_available():
        yield
        return
    from istgcn_amd import ops
    with ops.trace() as tr:
        yield
    out = os.path.join(ROOT, 'gpurun_out')
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, 'kernel_coverage.tsv'), 'a') as f:
        for k, n in sorted(tr.kernels.items()):
            f.write('%s%s\t%d\t%s\n' % (request.node.nodeid, os.environ.get('ISTGCN_COVERAGE_TAG', ''), n, k))
