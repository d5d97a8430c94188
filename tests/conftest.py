import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def O():
    """The f64 CPU oracle (test infrastructure)."""
    from oracle import oracle_py

    oracle_py.lib()
    return oracle_py


@pytest.fixture(scope="session")
def cfg():
    from rustraytracer_amd import config

    return config


@pytest.fixture(scope="session")
def cornell(cfg):
    return cfg.load_scene(os.path.join(ROOT, "configs", "cornell_box", "scene.toml"))


@pytest.fixture(scope="session")
def smallpt(cfg):
    return cfg.load_scene(os.path.join(ROOT, "configs", "smallpt", "scene.toml"))


@pytest.fixture(scope="session")
def scenes(cornell, smallpt):
    return {"cornell_box": cornell, "smallpt": smallpt}


@pytest.fixture(scope="session")
def golden_hits():
    return np.load(os.path.join(GOLDEN, "primary_hits.npz"))


@pytest.fixture(scope="session")
def gpu():
    """(api module, Backend on device 0).  Fails loudly when the HIP library is missing."""
    from rustraytracer_amd import api

    return api, api.Backend(0)
