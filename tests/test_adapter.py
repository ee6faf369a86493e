"""The ROS-typed adapter integration/HipTracer.hpp (lidarshooter::HipTracer : public ITracer), compiled against
integration/stubs and driven the way MeshProjector drives a tracer (MeshProjector.cpp:322-340, :446-464).

CPU part: the adapter compiles against the stand-in headers, and its sensor probe -- which may only use
LidarDevice's PUBLIC interface (LidarDevice.hpp:116-300: nextRay1, originToSensor[Inverse], getTotal*) --
recovers the oracle's ray tables and pose bit for bit.
GPU part: the reference's known answers through the adapter (EmbreeTracer_test.cpp:122-135,
OptixTracer_test.cpp:93-310: 1668 / 1781 / remove -> 1668 / empty -> 0), cloud bytes equal to the oracle's."""
import os
import subprocess

import numpy as np
import pytest

from conftest import DATA, ROOT

CFG = {u: os.path.join(DATA, "config", f"hesai-pandar-XT-32-lidar_{u}.json") for u in ("0000", "0001")}
STL = {n: os.path.join(DATA, "mesh", f"{n}.stl") for n in ("ground", "ben")}


@pytest.fixture(scope="module")
def adapterapi():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "integration")], stdout=subprocess.DEVNULL)
    from lidarshooter_amd import adapterapi as a
    return a


def test_adapter_compiles_against_stub_headers():
    # g++ -fsyntax-only of integration/HipTracer.hpp against integration/stubs (the shapes of ITracer.hpp:29-152,
    # LidarDevice.hpp:55-300, sensor_msgs::PointCloud2, pcl::PolygonMesh, Eigen, RTCGeometryType)
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "integration"), "syntax"], stdout=subprocess.DEVNULL)


def test_adapter_uses_only_public_lidar_device_members():
    # the stub LidarDevice declares nothing but the reference's public interface, so compiling against it proves
    # the adapter needs no new getter; spell the rule out against the text as well
    txt = open(os.path.join(ROOT, "integration", "HipTracer.hpp")).read()
    for name in ("verticalAngles", "horizontalBegin", "horizontalEnd", "rotationInverse", "translation()", "_channels", "_device."):
        assert name not in txt, f"adapter touches {name}, which lidarshooter's LidarDevice does not expose"


@pytest.mark.parametrize("uid", ["0000", "0001"])
def test_probe_recovers_oracle_tables_and_pose(adapterapi, oracle, sensors, uid):
    s = sensors[uid]
    p = adapterapi.probe_sensor(CFG[uid])
    assert (p["V"], p["H"]) == (s.V, s.H) and p["V"] * p["H"] == 4800          # LidarDevice_test.cpp:58
    st, ct, sp, cp = oracle.ray_tables(s)
    # the factor tables the kernels multiply: bit-equal to the oracle's libm tables
    for got, want in ((p["sin_theta"], st), (p["cos_theta"], ct), (p["sin_phi"], sp), (p["cos_phi"], cp)):
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    # every direction the reference would form (LidarDevice.cpp:310-316)
    d = oracle.ray_dirs(s).reshape(s.V, s.H, 3)
    assert np.array_equal((p["sin_theta"][:, None] * p["cos_phi"][None, :]).view(np.uint32), d[:, :, 0].view(np.uint32))
    assert np.array_equal((p["sin_theta"][:, None] * p["sin_phi"][None, :]).view(np.uint32), d[:, :, 1].view(np.uint32))
    # pose: exact (LidarDevice.cpp:383-401, :812-813)
    assert np.array_equal(p["Rinv"].view(np.uint32), s.Rinv.view(np.uint32))
    assert np.array_equal(p["t"].view(np.uint32), s.t.view(np.uint32))
    assert p["begin"] == s.h_begin and p["step"] == s.step()
    assert np.allclose(p["elevation"], s.vertical, atol=1e-4)                  # only feeds conservative bounds


def test_probe_syn128(adapterapi, oracle, sensors, tmp_path):
    from lidarshooter_amd import synth
    path = synth.write_sensor_json(CFG["0000"], str(tmp_path / "syn128.json"), synth.syn_vertical(128), 0.0, 360.0, 4096)
    s = oracle.load_sensor(path)
    assert (s.V, s.H) == (128, 4096)
    p = adapterapi.probe_sensor(path)
    st, ct, sp, cp = oracle.ray_tables(s)
    for got, want in ((p["sin_theta"], st), (p["cos_theta"], ct), (p["sin_phi"], sp), (p["cos_phi"], cp)):
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert p["begin"] == s.h_begin and p["step"] == s.step()


@pytest.mark.parametrize("begin,end,count", [(-180.0, 180.0, 360), (10.0, 350.0, 200), (360.0, 0.0, 90), (0.0, 90.0, 2)])
def test_probe_other_azimuth_ranges(adapterapi, oracle, tmp_path, begin, end, count):
    # rasters that do not start at 0, run clockwise, or have just two columns
    from lidarshooter_amd import synth
    vertical = np.array([30.0, 1.5, 0.0, -0.25, -45.0, -89.0], np.float32)
    path = synth.write_sensor_json(CFG["0001"], str(tmp_path / "odd.json"), vertical, begin, end, count)
    s = oracle.load_sensor(path)
    p = adapterapi.probe_sensor(path)
    d = oracle.ray_dirs(s).reshape(s.V, s.H, 3)
    assert np.array_equal((p["sin_theta"][:, None] * p["cos_phi"][None, :]).view(np.uint32), d[:, :, 0].view(np.uint32))
    assert np.array_equal((p["sin_theta"][:, None] * p["sin_phi"][None, :]).view(np.uint32), d[:, :, 1].view(np.uint32))
    assert np.array_equal(np.broadcast_to(p["cos_theta"][:, None], (s.V, s.H)).view(np.uint32), d[:, :, 2].view(np.uint32))


# ------------------------------------------------------------------------------------------------- GPU
def _points(cloud):
    return cloud["data"].reshape(-1, 32)


def _oracle_points(O, s, meshes, spec):
    """spec: [(geomID, mesh name, affine)]"""
    return O.trace_frame(s, [(g, *meshes[n], A) for g, n, A in spec])["points"]


@pytest.mark.gpu
def test_adapter_known_answers_and_remove_sequence(adapterapi, oracle, sensors, meshes):
    s = sensors["0000"]
    tr = adapterapi.AdapterTracer(CFG["0000"])
    tr.meshFromSTL("mesh", STL["ground"])
    tr.meshFromSTL("face", STL["ben"])
    assert (tr.L.lsa_mesh_vertices(tr.c, b"mesh"), tr.L.lsa_mesh_polygons(tr.c, b"mesh")) == (98, 162)   # EmbreeTracer_test.cpp:86-91
    # EmbreeTracer_test.cpp:122-135: ground alone, identity Affine3f -> 1668
    assert tr.addGeometry("mesh") == 0 and tr.getGeometryCount() == 1
    assert tr.updateGeometry("mesh", oracle.IDENTITY_AFFINE) == 0
    assert tr.commitScene() == 0
    assert tr.traceScene(1) == 0
    c = tr.cloud()
    assert c["width"] * c["height"] == 1668
    # LidarDevice_test.cpp:61-76 header + EmbreeTracer.cpp:364
    assert (c["height"], c["point_step"], c["row_step"], c["seq"], c["n_fields"]) == (1, 32, 0, 1, 5)
    assert not c["is_bigendian"] and c["is_dense"] and c["data"].size == 1668 * 32
    assert np.array_equal(_points(c), _oracle_points(oracle, s, {"mesh": meshes["ground"]}, [(0, "mesh", oracle.IDENTITY_AFFINE)]))
    # OptixTracer_test.cpp:122-169: + ben through the (translation, rotation) overload with zero displacement -> 1781
    assert tr.addGeometry("face") == 1 and tr.getGeometryCount() == 2
    tr.updateGeometry("mesh")
    tr.updateGeometry("face")
    assert tr.commitScene() == 0 and tr.traceScene(2) == 0
    c = tr.cloud()
    assert c["width"] == 1781 and c["seq"] == 2
    both = {"mesh": meshes["ground"], "face": meshes["ben"]}
    assert np.array_equal(_points(c), _oracle_points(oracle, s, both, [(0, "mesh", oracle.IDENTITY_AFFINE), (1, "face", oracle.IDENTITY_AFFINE)]))
    # OptixTracer_test.cpp:267-290: remove "face"; EmbreeTracer.cpp:252 re-commits inside removeGeometry, so a trace
    # without another commit sees the remaining mesh
    assert tr.removeGeometry("face") == 1 and tr.getGeometryCount() == 1
    assert tr.traceScene(3) == 0
    c = tr.cloud()
    assert c["width"] == 1668 and c["data"].size == 1668 * 32
    assert tr.removeGeometry("nope") == -1                                     # EmbreeTracer.cpp:224-225
    # OptixTracer_test.cpp:292-310: empty scene -> commit and trace return -1, zero points
    assert tr.removeGeometry("mesh") == 0 and tr.getGeometryCount() == 0
    assert tr.commitScene() == -1
    assert tr.traceScene(4) == -1
    c = tr.cloud()
    assert c["width"] == 0 and c["data"].size == 0
    tr.close()


@pytest.mark.gpu
def test_adapter_second_sensor_and_moving_mesh(adapterapi, oracle, sensors, meshes):
    # BASELINE.json configs[2]: lidar_0001 with its own pose; ben displaced and rotated per frame
    s = sensors["0001"]
    tr = adapterapi.AdapterTracer(CFG["0001"])
    tr.meshFromSTL("mesh", STL["ground"])
    tr.meshFromSTL("face", STL["ben"])
    tr.addGeometry("mesh")
    tr.addGeometry("face")
    for frame, (lin, ang) in enumerate([((0, 0, 0), (0, 0, 0)), ((1.5, -2.0, 0.25), (0.0, 0.0, 0.6)), ((-3.0, 4.0, 0.5), (0.1, -0.2, 1.9))]):
        tr.setDisplacement("face", lin, ang)
        tr.updateGeometry("mesh")
        tr.updateGeometry("face")
        assert tr.commitScene() == 0 and tr.traceScene(frame) == 0
        A = oracle.affine_from_components(np.array(lin, np.float32), np.array(ang, np.float32))
        want = _oracle_points(oracle, s, {"mesh": meshes["ground"], "face": meshes["ben"]},
                              [(0, "mesh", oracle.IDENTITY_AFFINE), (1, "face", A)])
        c = tr.cloud()
        assert c["width"] == want.shape[0] and np.array_equal(_points(c), want)
        if frame == 0:
            assert c["width"] == 1769
    tr.close()


@pytest.mark.gpu
def test_adapter_refuses_a_mesh_with_a_wild_index(adapterapi, oracle, sensors, meshes):
    """A pcl::PolygonMesh whose polygons name a vertex its cloud does not hold: commitScene throws BadGeometryException
    (Exceptions.hpp:143-157; the C shim returns -200 for it) instead of faulting the GPU; the tracer behind the ITracer::Ptr
    stays usable -- the same name with a sound mesh traces the reference's 1668 points."""
    s = sensors["0000"]
    gv, gt = meshes["ground"]
    bad = gt.copy()
    bad[3, 2] = gv.shape[0] + 100
    tr = adapterapi.AdapterTracer(CFG["0000"])
    tr.meshFromArrays("mesh", gv, bad, point_step=16)
    tr.addGeometry("mesh")
    tr.updateGeometry("mesh")
    assert tr.commitScene() == -200
    assert "BadGeometry" in tr.L.lsa_last_error().decode() and "vertex index %d" % (gv.shape[0] + 100) in tr.L.lsa_last_error().decode()
    assert "(type 0)" in tr.L.lsa_last_error().decode()          # RTC_GEOMETRY_TYPE_TRIANGLE: the refused geometry's own type
    assert tr.commitScene() == -200                              # it stays refused
    assert tr.removeGeometry("mesh") == 0
    # ... a QUAD geometry with a wild index is reported as a quad geometry (EmbreeTracer.cpp:179-198 accepts RTC_GEOMETRY_TYPE_QUAD)
    qv = np.array([(-30, -30, 0), (30, -30, 0), (30, 30, 0), (-30, 30, 0)], np.float32)
    tr.meshFromArrays("plate", qv, np.array([(0, 1, 2, 9)], np.uint32), point_step=16)
    tr.addGeometry("plate", geometry_type=1)
    tr.updateGeometry("plate")
    assert tr.commitScene() == -200 and "(type 1)" in tr.L.lsa_last_error().decode() and "'plate'" in tr.L.lsa_last_error().decode()
    assert tr.removeGeometry("plate") == 0
    tr.meshFromArrays("mesh", gv, gt, point_step=16)
    tr.addGeometry("mesh")
    tr.updateGeometry("mesh")
    assert tr.commitScene() == 0 and tr.traceScene(0) == 0
    want = oracle.trace_frame(s, [(0, gv, gt, oracle.IDENTITY_AFFINE)])["points"]
    assert np.array_equal(_points(tr.cloud()), want) and want.shape[0] == 1668
    tr.close()


@pytest.mark.gpu
def test_adapter_mesh_policies(adapterapi, oracle, sensors, meshes):
    """UploadAlways (default) sees an in-place edit of the cloud (MeshProjector.cpp:306-307) with the same header.
    SkipUnchanged is a contract: the same buffer, size, header.seq and header.stamp stand for the same vertices -> a
    transform-only update; an edit is announced by the header or by invalidateMesh(), and the tracer does not sample the
    data to second-guess it (an unannounced in-place edit is, by contract, not looked at)."""
    s = sensors["0000"]
    gv, gt = meshes["ground"]
    tr = adapterapi.AdapterTracer(CFG["0000"])
    tr.meshFromArrays("mesh", gv, gt, point_step=16)
    tr.addGeometry("mesh")

    def frame(i):
        tr.updateGeometry("mesh")
        assert tr.commitScene() == 0 and tr.traceScene(i) == 0
        return _points(tr.cloud())

    want0 = oracle.trace_frame(s, [(0, gv, gt, oracle.IDENTITY_AFFINE)])["points"]
    assert np.array_equal(frame(0), want0)
    assert tr.uploadCounts() == (1, 0)
    lifted = gv.copy()
    lifted[:, 2] += np.float32(0.5)
    want1 = oracle.trace_frame(s, [(0, lifted, gt, oracle.IDENTITY_AFFINE)])["points"]
    tr.setVertices("mesh", lifted, seq=0)                       # same buffer, same header: only the bytes differ
    assert np.array_equal(frame(1), want1)
    assert tr.uploadCounts() == (2, 0)
    # ---- SkipUnchanged
    tr.setSkipUnchanged(True)
    assert np.array_equal(frame(2), want1)                      # same buffer and header as the upload of frame 1
    assert np.array_equal(frame(3), want1)
    assert tr.uploadCounts() == (2, 2)
    tr.setDisplacement("mesh", (0.5, 0.25, -0.125), (0.0, 0.0, 0.3))   # pose-only change: no vertex traffic
    A = oracle.affine_from_components(np.array((0.5, 0.25, -0.125), np.float32), np.array((0.0, 0.0, 0.3), np.float32))
    assert np.array_equal(frame(4), oracle.trace_frame(s, [(0, lifted, gt, A)])["points"])
    assert tr.uploadCounts() == (2, 3)
    tr.setVertices("mesh", gv, seq=7)                           # a new mesh message: header.seq differs -> upload
    want_gv = oracle.trace_frame(s, [(0, gv, gt, A)])["points"]
    assert np.array_equal(frame(5), want_gv)
    assert tr.uploadCounts() == (3, 3)
    # an in-place edit that leaves the header alone is announced with invalidateMesh (the one line a maintainer adds
    # to MeshProjector::affineMeshCallback); until then the contract says "same vertices"
    tr.setVertices("mesh", lifted, seq=7)
    assert np.array_equal(frame(6), want_gv) and tr.uploadCounts() == (3, 4)
    tr.invalidateMesh("mesh")
    assert np.array_equal(frame(7), oracle.trace_frame(s, [(0, lifted, gt, A)])["points"])
    assert tr.uploadCounts() == (4, 4)
    # an UNSTAMPED cloud (seq and stamp zero) cannot announce anything through its header: for those a 64-vertex sample is
    # compared as a safety net -- an in-place edit of the whole cloud is seen without invalidateMesh()
    tr.setVertices("mesh", gv, seq=0)                           # header 7 -> 0: upload
    assert np.array_equal(frame(8), want_gv) and tr.uploadCounts() == (5, 4)
    assert np.array_equal(frame(9), want_gv) and tr.uploadCounts() == (5, 5)
    tr.setVertices("mesh", lifted, seq=0)                       # same buffer, same (empty) header, other bytes
    assert np.array_equal(frame(10), oracle.trace_frame(s, [(0, lifted, gt, A)])["points"]) and tr.uploadCounts() == (6, 5)
    tr.close()


@pytest.mark.gpu
def test_adapter_per_name_getters(adapterapi, meshes):
    """EmbreeTracer's getters through the adapter, with the reference's behaviour for unknown names: getGeometryId
    returns -1 (EmbreeTracer.cpp:82-89), getVertexCount / getElementCount / getGeometryType throw TraceException codes
    1 / 4 / 8 (:369-379, :405-415, :103-113); test/EmbreeTracer_test.cpp:99-120."""
    gv, gt = meshes["ground"]
    tr = adapterapi.AdapterTracer(CFG["0000"])
    tr.meshFromArrays("mesh", gv, gt)
    quads = np.array([[0, 1, 2, 3]], np.uint32)
    tr.meshFromArrays("plate", gv[:4], quads)
    assert tr.addGeometry("mesh") == 0 and tr.addGeometry("plate", 1) == 1
    assert tr.getter("id", "mesh") == 0 and tr.getter("id", "plate") == 1          # AddGeometryId
    assert tr.getter("type", "mesh") == 0 and tr.getter("type", "plate") == 1      # GeometryType: RTC_GEOMETRY_TYPE_TRIANGLE / QUAD
    assert tr.getter("vertices", "mesh") == 98 and tr.getter("elements", "mesh") == 162
    assert tr.getter("vertices", "plate") == 4 and tr.getter("elements", "plate") == 1
    assert tr.getter("id", "nope") == -1
    assert tr.getter("vertices", "nope") == -1001
    assert tr.getter("elements", "nope") == -1004
    assert tr.getter("type", "nope") == -1008
    assert tr.removeGeometry("mesh") == 0                                          # DeleteGeometryId
    assert tr.getter("id", "mesh") == -1 and tr.getter("type", "mesh") == -1008
    tr.close()


@pytest.mark.gpu
def test_adapter_syn128_over_grid(adapterapi, oracle, tmp_path):
    """A dense sensor through the adapter: 128 x 1024 rays over a 200 x 100-cell grid (40 000 triangles), against
    the oracle's BVH tracer; pcl::PointXYZ-sized vertex records (16 bytes)."""
    from lidarshooter_amd import synth
    path = synth.write_sensor_json(CFG["0000"], str(tmp_path / "syn.json"), synth.syn_vertical(128), 0.0, 360.0, 1024)
    s = oracle.load_sensor(path)
    v, t = synth.grid_mesh(200, 100)
    tr = adapterapi.AdapterTracer(path)
    tr.meshFromArrays("grid", v, t, point_step=16)
    assert tr.addGeometry("grid") == 0
    tr.updateGeometry("grid")
    assert tr.commitScene() == 0 and tr.traceScene(9) == 0
    want = oracle.trace_frame(s, [(0, v, t, oracle.IDENTITY_AFFINE)], use_bvh=True)["points"]
    c = tr.cloud()
    assert c["width"] == want.shape[0] > 10000 and np.array_equal(_points(c), want)
    spf = tr.frameLoop(5)                                        # MeshProjector::traceAffineMesh x5 in C++
    assert spf > 0
    assert np.array_equal(_points(tr.cloud()), want) and tr.cloud()["seq"] == 5
    tr.close()


@pytest.mark.gpu
def test_adapter_quad_mesh(adapterapi, oracle, sensors):
    # addGeometry(name, RTC_GEOMETRY_TYPE_QUAD, ...) with a pcl::PolygonMesh of four-index polygons (MeshTransformer.cpp:521-538)
    from test_gpu_dropin import _quad_grid
    s = sensors["0000"]
    v, q = _quad_grid(40, 30)
    tr = adapterapi.AdapterTracer(CFG["0000"])
    tr.meshFromArrays("relief", v, q, point_step=16)
    assert tr.addGeometry("relief", geometry_type=1) == 0
    tr.updateGeometry("relief")
    assert tr.commitScene() == 0 and tr.traceScene(3) == 0
    want = oracle.trace_frame(s, [(0, v, q, oracle.IDENTITY_AFFINE)])["points"]
    c = tr.cloud()
    assert c["width"] == want.shape[0] > 500 and np.array_equal(_points(c), want)
    # a triangle mesh registered as quads is refused like the reference does (BadGeometryException there)
    tr.meshFromArrays("wrong", v, oracle.quads_to_triangles(q), point_step=16)
    assert tr.addGeometry("wrong", geometry_type=1) == 1
    with pytest.raises(Exception, match="element vertex count"):
        tr.updateGeometry("wrong")
    tr.close()


@pytest.mark.gpu
def test_adapter_follows_the_sensor(adapterapi, oracle, sensors, meshes, tmp_path):
    """ITracer::setSensorConfig (ITracer.hpp:129, ITracer.cpp:48) and a LidarDevice initialised again in place
    (LidarDevice.hpp:116-117) take effect at the next traceScene, as with EmbreeTracer, whose traceScene reads its
    LidarDevice every frame (EmbreeTracer.cpp:299-307).  Round 3's adapter probed the sensor once, in its constructor,
    and went on tracing the old tables with the new device's header (missing 3)."""
    from lidarshooter_amd import synth
    both = {"mesh": meshes["ground"], "face": meshes["ben"]}
    spec = [(0, "mesh", oracle.IDENTITY_AFFINE), (1, "face", oracle.IDENTITY_AFFINE)]
    tr = adapterapi.AdapterTracer(CFG["0000"])
    tr.meshFromSTL("mesh", STL["ground"])
    tr.meshFromSTL("face", STL["ben"])
    assert tr.addGeometry("mesh") == 0 and tr.addGeometry("face") == 1

    def frame(i):
        tr.updateGeometry("mesh")
        tr.updateGeometry("face")
        assert tr.commitScene() == 0 and tr.traceScene(i) == 0
        return tr.cloud()

    c = frame(0)
    assert c["width"] == 1781 and tr.sensorProbeCount() == 1
    c = frame(1)
    assert c["width"] == 1781 and tr.sensorProbeCount() == 1                      # an unchanged sensor is not walked again
    # ---- another device object through ITracer::setSensorConfig: lidar_0001's pose and uid
    tr.setSensorConfig(CFG["0001"])
    c = frame(2)
    assert c["width"] == 1769 and tr.sensorProbeCount() == 2                       # lidar_0001 x ground+ben
    assert np.array_equal(_points(c), _oracle_points(oracle, sensors["0001"], both, spec))
    # ---- the same setSensorConfig without a new updateGeometry / commitScene in between: the library re-commits itself
    tr.setSensorConfig(CFG["0000"])
    assert tr.traceScene(3) == 0
    c = tr.cloud()
    assert c["width"] == 1781 and np.array_equal(_points(c), _oracle_points(oracle, sensors["0000"], both, spec))
    # ---- the SAME device object, initialised again with another raster (64 channels x 600 columns) and lidar_0000's pose
    path = synth.write_sensor_json(CFG["0000"], str(tmp_path / "wide.json"), synth.syn_vertical(64), 0.0, 360.0, 600)
    tr.reinitializeSensor(path)
    c = frame(4)
    s2 = oracle.load_sensor(path)
    want = _oracle_points(oracle, s2, both, spec)
    assert c["width"] == want.shape[0] > 2000 and np.array_equal(_points(c), want)
    assert tr.sensorProbeCount() == 4
    # ---- invalidateSensor: the sensor is walked again at the next trace whatever the fingerprint says; same answer
    tr.invalidateSensor()
    c = frame(5)
    assert tr.sensorProbeCount() == 5 and np.array_equal(_points(c), want)
    tr.close()


@pytest.mark.gpu
def test_adapter_grazing_corners_through_probed_tables(adapterapi, oracle, tmp_path):
    """the adapter hands the library factor tables it PROBED out of a live LidarDevice (nextRay1 walks), with an
    elevation estimate good to 1e-4 degrees -- half the elevation slack of the footprint bounds.  The library now derives the
    elevations from the tables itself; triangles whose corner lies ON a ray of the raster (conftest.grazing_mesh), rings on and
    around the horizon included, come out as the oracle's cloud through ITracer::Ptr."""
    from conftest import grazing_mesh
    from lidarshooter_amd import synth
    for name, vertical in (("syn", synth.syn_vertical(128)), ("horizon", np.linspace(-0.5, 0.5, 65, dtype=np.float32))):
        path = synth.write_sensor_json(CFG["0000"], str(tmp_path / f"{name}.json"), vertical, 0.0, 360.0, 1024)
        s = oracle.load_sensor(path)
        verts, idx = grazing_mesh(oracle, s)
        tr = adapterapi.AdapterTracer(path)
        tr.meshFromArrays("graze", verts, idx, point_step=16)
        assert tr.addGeometry("graze") == 0
        for k, A in enumerate((oracle.IDENTITY_AFFINE, oracle.affine_from_components(np.array((0.3, -0.2, 0.05), np.float32), np.array((0.0, 0.0, 0.7), np.float32)))):
            assert tr.updateGeometry("graze", A) == 0 and tr.commitScene() == 0 and tr.traceScene(k) == 0
            ref = oracle.trace_frame(s, [(0, verts, idx, A)])
            assert ref["points"].shape[0] > 300
            assert np.array_equal(_points(tr.cloud()), ref["points"])
        tr.close()


@pytest.mark.gpu
def test_adapter_width_is_unique_hits_where_the_embree_backend_would_duplicate(adapterapi, oracle, sensors, meshes, tmp_path):
    """A raster whose ceil(rays / 16) is not a multiple of 4 -- 32 x 151 = 4832 rays, channel 0 looking down: the reference's Embree
    backend re-traces rays 0 .. 31 from the wrapped iterator and appends their hits again (EmbreeTracer.cpp:304-307,
    LidarDevice.cpp:829-835; oracle.reference_width predicts its `width`).  HipTracer traces every ray once: its cloud is the
    oracle's, its width the number of UNIQUE hits -- the OptiX backend's behaviour (OptixTracer.cpp:895-942) -- and the difference
    is exactly the predicted 32.  INTEGRATION.md, "Differences a user can see"."""
    from lidarshooter_amd import synth
    base = sensors["0000"]
    path = synth.write_sensor_json(CFG["0000"], str(tmp_path / "odd.json"), base.vertical[::-1].copy(), float(base.h_begin), float(base.h_end), 151)
    s = oracle.load_sensor(path)
    assert s.total_rays == 4832 and len(oracle.reference_packet_walk(s.total_rays)) == 304
    tr = adapterapi.AdapterTracer(path)
    tr.meshFromSTL("mesh", STL["ground"])
    assert tr.addGeometry("mesh") == 0 and tr.updateGeometry("mesh", oracle.IDENTITY_AFFINE) == 0 and tr.commitScene() == 0
    assert tr.traceScene(0) == 0
    c = tr.cloud()
    ref = oracle.trace_frame(s, [(0, *meshes["ground"], oracle.IDENTITY_AFFINE)])
    assert np.array_equal(_points(c), ref["points"])
    unique = int((ref["gid"] != oracle.INVALID).sum())
    assert c["width"] == unique and oracle.reference_width(s, ref["gid"] != oracle.INVALID) == unique + 32
    tr.close()
