"""CPU tests of the C++ host mirror (LidarDevice, STL ingest) against the oracle and against the
reference's own gtest values (LidarDevice_test.cpp, EmbreeTracer_test.cpp:86-91).  No GPU calls."""
import os

import numpy as np
import pytest

from conftest import DATA


@pytest.fixture(scope="module")
def hostapi():
    from lidarshooter_amd import hostapi as h
    h.load()
    return h


def _cfg(uid):
    return os.path.join(DATA, "config", f"hesai-pandar-XT-32-lidar_{uid}.json")


def test_lidar_device_basics(hostapi):
    # LidarDevice_test.cpp:51-59
    d = hostapi.LidarDevice(_cfg("0000"))
    assert d.getSensorUid() == "lidar_0000"
    assert d.getTotalRays() == 150 * 32
    assert d.getTotalChannels() == 32 and d.getScanRayCount() == 150
    d2 = hostapi.LidarDevice(_cfg("0000"), "override")
    assert d2.getSensorUid() == "override"          # LidarDevice.cpp:498-500


def test_init_message(hostapi):
    # LidarDevice_test.cpp:61-76
    m = hostapi.LidarDevice(_cfg("0000")).initMessage(3)
    assert m == dict(seq=3, height=1, width=0, point_step=32, row_step=0, n_fields=5, is_bigendian=False,
                     is_dense=True, frame_id="PandarXT-32")


@pytest.mark.parametrize("uid", ["0000", "0001"])
def test_pose_and_tables_match_oracle(hostapi, oracle, sensors, uid):
    d = hostapi.LidarDevice(_cfg(uid))
    s = sensors[uid]
    R, Ri = d.rotation()
    assert np.array_equal(R, s.R) and np.array_equal(Ri, s.Rinv)
    desc = d.desc()
    assert np.array_equal(desc["vertical"], s.vertical)
    assert desc["h_count"] == s.h_count and desc["h_begin"] == s.h_begin and desc["h_end"] == s.h_end
    assert np.array_equal(desc["Rinv"], s.Rinv) and np.array_equal(desc["t"], s.t)
    assert d.step() == s.step()
    dirs = oracle.ray_dirs(s)
    for v, h in ((0, 0), (31, 149), (15, 75), (7, 1), (20, 148)):
        assert np.array_equal(d.rayDirection(v, h), dirs[v * s.H + h])
    p = np.array([1.0, -2.0, 3.0], np.float32)
    q = d.originToSensor(p)
    assert np.array_equal(q, oracle.transform_vertices(p[None, :], oracle.IDENTITY_AFFINE, s)[0])
    assert np.allclose(d.originToSensor(q, inverse=True), p, atol=1e-5)


def test_missing_config_raises(hostapi, capi):
    with pytest.raises(capi.LidarShooterHipError):
        hostapi.LidarDevice("/nonexistent/sensor.json")


@pytest.mark.parametrize("name,nv,nt", [("ground", 98, 162), ("ben", 2823, 5489)])
def test_stl_ingest(hostapi, meshes, name, nv, nt):
    # EmbreeTracer_test.cpp:86-91: 98 vertices / 162 triangles for ground.stl
    m = hostapi.PolygonMesh(os.path.join(DATA, "mesh", f"{name}.stl"))
    assert m.numPoints() == nv and m.numPolygons() == nt and m.pointStep() == 16
    assert np.array_equal(m.points(), meshes[name][0])
    assert np.array_equal(m.polygons(), meshes[name][1])


def test_trajectory_player(hostapi, oracle):
    # config/trajectory.json: 10 s of (0,5,0)/(0,0,1) then 5 s of (1,0,0)/(0,0,-1), 10 Hz messages
    path = os.path.join(DATA, "config", "trajectory.json")
    poses = hostapi.trajectory_play(path, 0.1)
    ref = oracle.play_trajectory(path, 0.1)
    assert poses.shape == (150, 6)
    assert np.array_equal(poses, ref)
    assert np.allclose(poses[0], [0, 5, 0, 0, 0, 1])            # first message: no rotation yet
    assert np.allclose(poses[99, 3:], [0, 0, 100]) and np.allclose(poses[-1, 3:], [0, 0, 50])
