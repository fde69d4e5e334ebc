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


# Lines a test wants in the run's log whatever its outcome (e.g. how many APAA quality values fell under the min-area-rect
# tie rule of rounds 3-5): tests append to REPORT, the summary hook prints them at the end of the session (also under -q).
REPORT = []


def pytest_terminal_summary(terminalreporter):
    if REPORT:
        terminalreporter.write_sep("-", "reported by tests")
        for line in REPORT:
            terminalreporter.write_line(line)
