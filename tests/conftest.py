"""pytest configuration: package path, markers, fixture loader."""

import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_ROOT = os.path.join(REPO, "bess-kge_amd")
for p in (PKG_ROOT, REPO):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN_DIR = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def load_golden(name: str):
    """Load tests/golden/<name>.npz (data produced by the reference)."""
    return np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def golden():
    return load_golden
