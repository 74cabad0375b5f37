import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
if os.path.dirname(os.path.abspath(__file__)) not in sys.path:
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_report_header(config):
    """Which libm the oracle resolved and which parity bar follows from it (tests/_parity.py) -- printed on every run, CPU box and GPU box."""
    import _parity
    try:
        return "trayhip " + _parity.describe()
    except Exception as e:   # (never let the header break a run)
        return f"trayhip parity probe failed: {e}"


@pytest.fixture(scope="session")
def built():
    """libtrayhip.so + liboracle.so present (built in-tree; the GPU box uses the prebuilt files)."""
    import __graft_entry__ as g
    g.build()
    return True


@pytest.fixture(scope="session")
def assets(tmp_path_factory, built):
    from tray_rust_amd import scenes
    d = tmp_path_factory.mktemp("scenes")
    scenes.write_assets(str(d))
    return str(d)
