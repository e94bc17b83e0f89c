import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "gpu_slow: GPU tests that start child processes with their own torch import (bench.py rehearsals, the 5-rank "
                                       "Engine run): part of -m gpu; deselect with -m 'gpu and not gpu_slow' for a quick pass")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
