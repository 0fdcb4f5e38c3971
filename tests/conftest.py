"""pytest configuration: marker registration + import paths.

`-m "not gpu"` : oracle vs golden vectors, host logic, C-ABI symbol checks (no GPU needed)
`-m gpu`       : parity tests proper -- HIP path through the C-ABI vs the oracle / golden fixtures
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'segmentation-networks-benchmark_amd')
for p in (PKG, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')
os.environ['SEGNB_TEST_HARNESS'] = '1'      # lets tests install the ABI emulator (segnb._native.set_backend_for_testing)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    # gpu-marked tests never silently pass without a device
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU visible')
    for it in items:
        if 'gpu' in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN
