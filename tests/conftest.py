import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    """the CPU oracle (test infrastructure), built from oracle/*.c with gcc"""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    import oracle_lib
    return oracle_lib.Oracle(os.path.join(ROOT, "oracle", "_build", "liboracle.so"))


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
