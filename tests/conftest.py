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


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def assert_close(a, b, rel=1e-5, what=''):
    """The parity bar for floating-point results: max-norm relative 1e-5 per tensor
    and elementwise |a-b| <= rel*|b| + 1e-6."""
    a = np.asarray(a); b = np.asarray(b)
    assert a.shape == b.shape, '%s shape %s vs %s' % (what, a.shape, b.shape)
    if a.size == 0:
        return
    scale = max(np.abs(b).max(), 1e-30)
    err = np.abs(a - b).max()
    assert err <= rel * scale + 1e-7, '%s: max abs err %.3e vs scale %.3e' % (what, err, scale)
    assert np.all(np.abs(a - b) <= rel * np.abs(b) + 1e-6 * max(1.0, scale)), '%s: elementwise bound' % what
