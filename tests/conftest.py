import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def scene_dir(tmp_path_factory):
    """Directory holding generated scene files (procedural PLY meshes are written on demand)."""
    return str(tmp_path_factory.mktemp("scenes"))


@pytest.fixture(scope="session")
def repo_root():
    return ROOT
