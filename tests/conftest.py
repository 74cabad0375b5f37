import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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
