import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_addoption(parser):
    parser.addoption("--soak", type=int, default=0,
                     help="tests/test_gpu_soak.py: run N more random cases (and N // 50 more concurrency rounds) than the suite's own few")
    parser.addoption("--late", type=int, default=0, help="tests/test_gpu_late.py: run N more late-quiz-state cases than the suite's own")
    parser.addoption("--late-first", type=int, default=1000, help="first seed of the --late range")
    parser.addoption("--soak-first", type=int, default=120, help="first seed of the --soak range (the suite's fixed seeds end below 120)")


@pytest.fixture(scope="session")
def soak(request):
    return request.config.getoption("--soak"), request.config.getoption("--soak-first")


@pytest.fixture(scope="session")
def late(request):
    return request.config.getoption("--late"), request.config.getoption("--late-first")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    import orclib

    return orclib.lib()


@pytest.fixture(scope="session")
def factory():
    """The product library.  GPU tests must run the HIP path: fail loudly if it is missing."""
    from probqa_amd import interop

    if not os.path.exists(interop.LIB_PATH):
        import __graft_entry__

        __graft_entry__.build()
    return interop.PqaEngineFactory()
