import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

DATA = os.path.join(ROOT, "tests", "golden", "data")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def meshes(oracle):
    return {n: oracle.load_stl(os.path.join(DATA, "mesh", f"{n}.stl")) for n in ("ground", "ben")}


@pytest.fixture(scope="session")
def sensors(oracle):
    return {u: oracle.load_sensor(os.path.join(DATA, "config", f"hesai-pandar-XT-32-lidar_{u}.json"))
            for u in ("0000", "0001")}


@pytest.fixture(scope="session")
def capi():
    from lidarshooter_amd import capi as c
    return c


ENGINES = {"bvh": 1, "projection": 2}


def make_tracer(capi, sensor, engine=None, **kw):
    tr = capi.Tracer(sensor.vertical, sensor.h_begin, sensor.h_end, sensor.h_count, sensor.Rinv, sensor.t, **kw)
    if engine is not None:
        tr.setOption(capi.LS_OPT_ENGINE, ENGINES[engine])
    return tr


@pytest.fixture(params=["bvh", "projection"])
def engine(request):
    """Both closest-hit engines must give the oracle's answer."""
    return request.param
