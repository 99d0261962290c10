import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    """`gpu` tests need a real device: on a host without one they are skipped (never silently passed)."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no ROCm device visible (run on the GPU box: pytest -m gpu)')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


def load_golden(name):
    import torch
    return torch.load(os.path.join(GOLDEN, name), weights_only=False)
