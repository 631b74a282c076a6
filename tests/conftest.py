import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    """The multi-process GPU tests start child processes; they run FIRST, while this process has not initialised the GPU yet
    (a process that holds the device must not fork + exec on the GPU box)."""
    items.sort(key=lambda it: 0 if 'test_multirank_gpu' in it.nodeid else 1)


@pytest.fixture(scope='session')
def repo_root():
    return ROOT
