import os, sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'needs_reference: needs /root/reference (build container only)')


def pytest_collection_modifyitems(config, items):
    have_ref = os.path.isdir('/root/reference/envs/h2000_v90')
    for it in items:
        if 'needs_reference' in it.keywords and not have_ref:
            it.add_marker(pytest.mark.skip(reason='/root/reference not present'))


@pytest.fixture(scope='session')
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)
    return load


@pytest.fixture(scope='session')
def engine():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    import serl_amd
    return serl_amd.RolloutEngine(0)
