import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _reset_table_configuration():
    """install_dropin(table_capacity=...) / ops.configure_table set process-wide state (like the reference's own module
    registration): every test starts from the default (ranked index for the implicit per-device workspaces)"""
    yield
    ops = sys.modules.get("cdv_slam_amd.ops")
    if ops is not None and getattr(ops, "_table_capacity", None) is not None:
        ops.configure_table(None)
