import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
BLOB = os.path.join(ROOT, "flybody_amd", "assets", "fly_flight.ffmb")
HAVE_REFERENCE = os.path.isdir("/root/reference/vnl_ray")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_quat():
    with np.load(os.path.join(GOLDEN, "quaternions.npz")) as f:
        return {k: f[k] for k in f.files}


@pytest.fixture(scope="session")
def golden_wbpg():
    with np.load(os.path.join(GOLDEN, "wbpg.npz")) as f:
        return {k: f[k] for k in f.files}


@pytest.fixture(scope="session")
def oracle_model():
    from oracle.oracle import OracleModel

    return OracleModel(BLOB)


@pytest.fixture(scope="session")
def wb_tables():
    from flybody_amd.tasks.synthetic import base_wing_pattern
    from flybody_amd.tasks.wbpg import build_tables

    return build_tables(base_wing_pattern())


@pytest.fixture(scope="session")
def ref_traj():
    from flybody_amd.tasks.synthetic import flight_trajectories
    from flybody_amd.tasks.trajectories import preprocess

    return preprocess(*flight_trajectories(8, 3006))
