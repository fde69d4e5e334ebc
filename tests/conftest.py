import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (oracle/orp_oracle.c), built on demand with gcc."""
    from oracle import orp_oracle
    orp_oracle.build()
    return orp_oracle


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def pytest_collection_modifyitems(config, items):
    """ORP_TEST_ORDER=reverse runs the collected tests back to front (order-independence check of the GPU suite: no test
    may depend on allocator / cache state left by an earlier one; logs under profiles/r03_gputest_*.log)."""
    if os.environ.get("ORP_TEST_ORDER", "") == "reverse":
        items.reverse()
