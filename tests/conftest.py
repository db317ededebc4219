import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def capi():
    from aaltoasr_amd import build, capi as A
    build.build()
    A.lib()
    return A


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def observed(label, value, bound):
    """Assert `value >= bound` and, with AASR_PRINT_OBSERVED=1, print what was observed (used to keep
    the thresholds of the LNA code-equality tests at what the hardware actually delivers)."""
    import os
    if os.environ.get("AASR_PRINT_OBSERVED") == "1":
        print("OBSERVED %s %.6f (bound %.4f)" % (label, value, bound))
    assert value >= bound, (label, value, bound)
