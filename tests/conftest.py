import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: CPU test of several minutes, skipped unless FG_SLOW=1 (run on its own: "
                                       "FG_SLOW=1 python -m pytest tests -m slow)")


def pytest_collection_modifyitems(config, items):
    if os.environ.get("FG_SLOW") == "1":
        return
    skip = pytest.mark.skip(reason="slow CPU test: set FG_SLOW=1")
    for item in items:
        if "slow" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def repo_root():
    return ROOT
