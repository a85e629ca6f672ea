import os
import sys

import pytest

# The oracle is OpenMP code; a GPU box reports every host core in its affinity mask but a one-GPU job owns
# about 16 of them: cap the team before libgomp is loaded so the checker does not oversubscribe.
os.environ.setdefault("OMP_NUM_THREADS", str(min(16, len(os.sched_getaffinity(0)))))

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    import __graft_entry__ as ge
    return ge.load_package()


@pytest.fixture(scope="session")
def orc():
    from oracle import oracle
    oracle.build()
    return oracle
