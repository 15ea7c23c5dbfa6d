import os
import sys
import warnings
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
warnings.filterwarnings('ignore', category=UserWarning)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def lib():
    """The C-ABI library; GPU tests fail loudly if it is missing."""
    from interdiff_amd import _lib
    return _lib.load()
