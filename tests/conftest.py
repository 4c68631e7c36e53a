import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    # the oracle is test infrastructure; build it when missing (plain gcc, a second or two)
    so = os.path.join(ROOT, "oracle", "libmm_oracle.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])


@pytest.fixture(scope="session")
def oracle():
    from tests import oracle_binding
    return oracle_binding.load()
