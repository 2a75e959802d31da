import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
DATA_DIR = os.path.join(GOLDEN_DIR, "data")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(GOLDEN_DIR, "golden.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def O():
    """the CPU oracle (test infrastructure)"""
    import oracle
    oracle.lib()
    return oracle


def data_path(name):
    return os.path.join(DATA_DIR, name + ".tsp")


@pytest.fixture(scope="session")
def instances(O):
    """name -> (xy, cost matrix) cache for the small fixtures"""
    cache = {}

    def get(name):
        if name not in cache:
            if name.startswith("n") and "_s" in name:
                n, seed = name[1:].split("_s")
                xy = O.random_points(int(n), int(seed))
            else:
                xy, _ = O.read_tsplib(data_path(name))
            cache[name] = (xy, O.cost_matrix(xy))
        return cache[name]
    return get
