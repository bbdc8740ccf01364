import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def read_index():
    return json.load(open(os.path.join(GOLD, "reads", "index.json")))


def load_read(key):
    """Fixture read -> (golden npz, ReadData, ReadTensors) through the product's host stage."""
    from nanoreviser_amd import hoststage as hs
    g = np.load(os.path.join(GOLD, "reads", key + ".npz"))
    rd = hs.collapse_events(g["ev_start"], g["ev_mean"], g["ev_stdv"], g["ev_model_state"],
                            g["ev_move"], g["raw_signal"])
    return g, rd, hs.read_tensors(rd)


@pytest.fixture(scope="session")
def reads(read_index):
    cache = {}

    def get(key):
        if key not in cache:
            cache[key] = load_read(key)
        return cache[key]
    get.keys = [e["key"] for e in read_index]
    return get


@pytest.fixture(scope="session")
def model_goldens():
    return np.load(os.path.join(GOLD, "model_goldens.npz"))


@pytest.fixture(scope="session")
def species_models():
    from nanoreviser_amd.weights import load_species
    return {sp: load_species(sp) for sp in ("ecoli", "human")}
