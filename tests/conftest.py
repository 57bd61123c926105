import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    # GPU tests are skipped (not failed) where no device exists, e.g. `pytest tests` in the build container.
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for it in items:
        if 'gpu' in it.keywords:
            it.add_marker(skip)


@pytest.fixture(autouse=True)
def _collect_garbage_between_gpu_tests(request):
    """Full-size tests hold tens of GB each; whatever a test leaves in reference cycles is collected before the next one."""
    yield
    if 'gpu' in request.keywords:
        import gc
        gc.collect()


@pytest.fixture(scope='session')
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return load


def rel_l2(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))
