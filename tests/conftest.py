import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import binding
    binding.build()
    binding.lib()
    return binding


def pytest_terminal_summary(terminalreporter):
    """Largest achieved error per column kind over the GPU-vs-oracle comparisons of this run."""
    import util
    if not util.ACHIEVED:
        return
    terminalreporter.write_line("achieved maximum errors (error / tolerance of the comparison it came from):")
    for kind in sorted(util.ACHIEVED):
        err, tol = util.ACHIEVED[kind]
        terminalreporter.write_line("  %-16s %.3g / %.3g" % (kind, err, tol))
