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


def grazing_mesh(oracle, s, n=640, seed=77):
    """640 small triangles whose highest or lowest corner lies ON a ray of sensor `s`'s raster -- t * direction in float32, carried
    into the world frame (v_world = R v_sensor + t) -- and thin slivers between two neighbouring columns of one ring: whether such a
    ray hits is decided by the exact test's rounding, and a conservative footprint must contain it either way.
    -> (verts float32[3n, 3] in world coordinates, idx uint32[n, 3])"""
    import numpy as np
    V, H = s.V, s.H
    dirs = oracle.ray_dirs(s).astype(np.float64)
    R = s.R.reshape(3, 3).astype(np.float64)
    rng = np.random.default_rng(seed)
    tris_sensor = []
    for k in range(n):
        v, h = k % V, int(rng.integers(1, H - 2))
        d = dirs[v * H + h]
        r = float(rng.uniform(4.0, 70.0))
        apex = (np.float32(r) * d.astype(np.float32)).astype(np.float64)        # on the ray, as the kernels form t * direction
        side = np.cross(d, [0.0, 0.0, 1.0]); side /= np.linalg.norm(side)
        down = np.cross(side, d); down /= np.linalg.norm(down)                   # towards lower elevation, across the ray
        if down[2] > 0: down = -down
        w = r * float(rng.uniform(2e-4, 2e-2))
        sign = 1.0 if k % 3 else -1.0                                            # the corner on the ray is the highest / the lowest
        if k % 5 == 4:   # a sliver between this column's ray and the next one's, both corners on rays of the same ring
            other = (np.float32(r * float(rng.uniform(0.98, 1.02))) * dirs[v * H + h + 1].astype(np.float32)).astype(np.float64)
            third = 0.5 * (apex + other) + sign * down * w
            tris_sensor.append([apex, other, third])
        else:
            tris_sensor.append([apex, apex + sign * down * w + side * w * float(rng.uniform(0.2, 1.0)), apex + sign * down * w - side * w * float(rng.uniform(0.2, 1.0))])
    tri = np.array(tris_sensor, np.float64).reshape(-1, 3)
    verts = (tri @ R.T + s.t.astype(np.float64)).astype(np.float32)
    idx = np.arange(verts.shape[0], dtype=np.uint32).reshape(-1, 3)
    return verts, idx
