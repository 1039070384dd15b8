import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """The engine library is built in-tree and kept out of git: build it when a fresh checkout has none yet (hipcc
    cross-compiles without a GPU).  The product itself never builds or falls back at run time."""
    from sage_slam_amd import build as sage_build
    lib = os.path.join(ROOT, "sage_slam_amd", "libsage_ba.so")
    if not os.path.exists(lib) and os.path.exists(sage_build.HIPCC):
        sage_build.build(verbose=False)


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure): built on demand with gcc."""
    from oracle import oracle
    oracle.build()
    return oracle


# Lines the GPU tests want in the run's tail even when they pass (measured distances from the oracle): collected here and
# printed in the terminal summary -- `pytest -q` shows nothing of a passing test otherwise.
_SUMMARY_LINES = []


def summary_line(text: str) -> None:
    print(text)
    _SUMMARY_LINES.append(text)


def pytest_terminal_summary(terminalreporter):
    if _SUMMARY_LINES:
        terminalreporter.section("engine vs oracle (measured)")
        for ln in _SUMMARY_LINES:
            terminalreporter.write_line(ln)
