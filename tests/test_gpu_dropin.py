"""GPU tests of the drop-in (ITracer-shaped, host-buffer) path of the C ABI and of the handle's failure reporting:
asynchronous staged uploads, host-visible outputs, table-based creation (what the ROS adapter uses), remove ->
commit semantics (EmbreeTracer.cpp:252), the device status word, the per-slot ordering of library mesh copies."""
import numpy as np
import pytest

from conftest import make_tracer
from test_gpu_parity import _syn_sensor

pytestmark = pytest.mark.gpu


def _hits_array(hits):
    return np.stack([hits["ray"], hits["geom"], hits["prim"], hits["t"].view(np.uint32)], axis=1)


def _tracer_from_tables(capi, oracle, s):
    import ctypes as C
    st, ct, sp, cp = oracle.ray_tables(s)
    elev = np.ascontiguousarray(s.vertical, np.float32)
    tb = capi.SensorTables()
    keep = [np.ascontiguousarray(a, np.float32) for a in (st, ct, elev, sp, cp)]
    f32p = C.POINTER(C.c_float)
    tb.sin_theta, tb.cos_theta, tb.elevation_deg, tb.sin_phi, tb.cos_phi = [a.ctypes.data_as(f32p) for a in keep]
    tb.n_vertical, tb.h_count = s.V, s.H
    tb.h_begin_deg, tb.h_step_deg = float(s.h_begin), float(s.step())
    tb.Rinv = (C.c_float * 9)(*[float(x) for x in s.Rinv])
    tb.t = (C.c_float * 3)(*[float(x) for x in s.t])
    tr = capi.Tracer.__new__(capi.Tracer)
    tr.L = capi.load()
    h = C.c_void_p()
    assert tr.L.ls_tracer_create_tables(C.byref(tb), 0, C.byref(h)) == 0
    tr.h, tr.V, tr.H, tr.az0, tr.naz = h, s.V, s.H, 0, s.H
    return tr


@pytest.mark.parametrize("uid", ["0000", "0001"])
def test_create_from_tables_equals_create_from_desc(oracle, capi, sensors, meshes, uid, engine):
    # ls_tracer_create_tables (the adapter's entry: tables recovered through LidarDevice's public interface) gives the
    # oracle's frame, like ls_tracer_create
    from conftest import ENGINES
    s = sensors[uid]
    tr = _tracer_from_tables(capi, oracle, s)
    tr.setOption(capi.LS_OPT_ENGINE, ENGINES[engine])
    for name, key in (("ground", "ground"), ("face", "ben")):
        assert tr.addGeometry(name, meshes[key][0].shape[0], meshes[key][1].shape[0]) >= 0
        tr.updateGeometry(name, oracle.IDENTITY_AFFINE, *meshes[key])
    assert tr.commitScene() == 0
    rc, pts, hits = tr.traceScene(0)
    ref = oracle.trace_frame(s, [(0, *meshes["ground"], oracle.IDENTITY_AFFINE), (1, *meshes["ben"], oracle.IDENTITY_AFFINE)])
    assert rc == 0 and np.array_equal(pts, ref["points"]) and np.array_equal(_hits_array(hits), ref["hits"])
    tr.close()


@pytest.mark.parametrize("host_output,readback_hits", [(1, 1), (1, 0), (0, 1), (0, 0), (2, 1), (2, 0)])
def test_host_output_modes(oracle, capi, sensors, meshes, host_output, readback_hits, engine):
    s = sensors["0000"]
    tr = make_tracer(capi, s, engine)
    tr.setOption(capi.LS_OPT_HOST_OUTPUT, host_output)
    tr.setOption(capi.LS_OPT_READBACK_HITS, readback_hits)
    tr.addGeometry("ground", *[a.shape[0] for a in meshes["ground"]])
    tr.addGeometry("face", *[a.shape[0] for a in meshes["ben"]])
    ref = oracle.trace_frame(s, [(0, *meshes["ground"], oracle.IDENTITY_AFFINE), (1, *meshes["ben"], oracle.IDENTITY_AFFINE)])
    for frame in range(3):   # repeated frames reuse the pinned buffers
        tr.updateGeometry("ground", oracle.IDENTITY_AFFINE, *meshes["ground"])
        tr.updateGeometry("face", oracle.IDENTITY_AFFINE, *meshes["ben"])
        assert tr.commitScene() == 0
        rc, pts, hits = tr.traceScene(frame)
        assert rc == 0 and np.array_equal(pts, ref["points"])
        if readback_hits:
            assert np.array_equal(_hits_array(hits), ref["hits"])
        else:
            assert hits.shape[0] == 0 and not tr.last_frame.hits and tr.last_frame.d_hits
        t, gid = tr.denseHits()                      # rebuilt from the records wherever they live
        assert np.array_equal(gid, ref["gid"]) and np.array_equal(t, ref["t"])
    tr.close()


def test_host_upload_returns_with_caller_memory_free(oracle, capi, sensors):
    """When ls_update_geometry returns, the caller may overwrite its buffers (MeshProjector.cpp:448-461 does),
    whichever way the bytes travel (straight from pageable memory by default; LS_UPLOAD_MODE=0: copy pool -> pinned
    staging -> chunked DMA).  A mesh of several 512 KB chunks; the arrays are scribbled over right after every call."""
    from lidarshooter_amd import synth
    s = sensors["0001"]
    v, t = synth.grid_mesh(400, 300)                 # 120 701 vertices x 16 B = 1.9 MB, 240 000 triangles = 2.9 MB
    padded = np.zeros((v.shape[0], 4), np.float32)   # pcl::PointXYZ records
    padded[:, :3] = v
    tr = make_tracer(capi, s, "projection")
    assert tr.info(capi.LS_INFO_HOST_THREADS) >= 1
    tr.addGeometry("grid", v.shape[0], t.shape[0])
    ref = oracle.trace_frame(s, [(0, v, t, oracle.IDENTITY_AFFINE)], use_bvh=True)
    for frame in range(4):
        vv, tt = padded.copy(), t.copy()
        tr.updateGeometry("grid", oracle.IDENTITY_AFFINE, vv, tt if frame == 0 else None, stride=16)
        vv[:] = np.float32(1e9)                      # the caller's memory is its own again
        tt[:] = 0
        assert tr.commitScene() == 0
        rc, pts, hits = tr.traceScene(frame)
        assert rc == 0 and np.array_equal(pts, ref["points"]) and np.array_equal(_hits_array(hits), ref["hits"])
    tr.close()


def test_remove_geometry_commits(oracle, capi, sensors, meshes, engine):
    # EmbreeTracer.cpp:252: removeGeometry commits, so a traceScene that follows traces what is left
    s = sensors["0000"]
    tr = make_tracer(capi, s, engine)
    tr.addGeometry("ground", *[a.shape[0] for a in meshes["ground"]])
    tr.addGeometry("face", *[a.shape[0] for a in meshes["ben"]])
    tr.updateGeometry("ground", oracle.IDENTITY_AFFINE, *meshes["ground"])
    tr.updateGeometry("face", oracle.IDENTITY_AFFINE, *meshes["ben"])
    assert tr.commitScene() == 0
    assert len(tr.traceScene(0)[1]) == 1781
    assert tr.removeGeometry("face") == 1
    rc, pts, hits = tr.traceScene(1)                 # no commitScene in between
    ref = oracle.trace_frame(s, [(0, *meshes["ground"], oracle.IDENTITY_AFFINE)])
    assert rc == 0 and len(pts) == 1668 and np.array_equal(pts, ref["points"]) and np.array_equal(_hits_array(hits), ref["hits"])
    assert tr.removeGeometry("ground") == 0          # the commit inside finds an empty scene: not an error of the removal
    rc, pts, _ = tr.traceScene(2)
    assert rc == -1 and len(pts) == 0
    tr.close()


def test_three_stream_mode_orders_every_slot_after_a_library_copy(oracle, capi, sensors, meshes):
    """LS_OPT_PIPELINE = 2 on the handle's own stream: ls_update_geometry_device enqueues a D2D copy on that stream
    with no host wait; ALL three slot streams (not only the first frame's) have to order themselves after it."""
    import torch
    s = sensors["0000"]
    dev = torch.device("cuda", 0)
    gv, gt = meshes["ground"]
    bv, bt = meshes["ben"]
    tr = make_tracer(capi, s, "projection")
    tr.setOption(capi.LS_OPT_PIPELINE, 2)
    mode = tr.info(capi.LS_INFO_PIPELINE_MODE)
    assert mode in (1, 2) and (mode == 2) == (tr.info(capi.LS_INFO_CONCURRENT_STREAMS) >= 3)
    tr.addGeometry("ground", gv.shape[0], gt.shape[0])
    tr.addGeometry("face", bv.shape[0], bt.shape[0])
    d_gt = torch.from_numpy(gt.view(np.int32)).to(dev)
    d_bt = torch.from_numpy(bt.view(np.int32)).to(dev)
    cap = s.V * s.H
    bufs = [(torch.zeros(32 * cap, dtype=torch.uint8, device=dev), torch.zeros(16 * cap, dtype=torch.uint8, device=dev),
             torch.zeros(4, dtype=torch.int32, device=dev)) for _ in range(3)]
    for rnd in range(6):
        shift = np.float32(0.75 * rnd)
        bv2 = (bv + np.array([shift, -shift, 0.1 * rnd], np.float32)).astype(np.float32)
        d_gv = torch.from_numpy(gv).to(dev)
        d_bv = torch.from_numpy(bv2).to(dev)
        torch.cuda.synchronize()
        tr.updateGeometryDevice("ground", oracle.IDENTITY_AFFINE, d_gv.data_ptr(), 12, d_gt.data_ptr())
        tr.updateGeometryDevice("face", oracle.IDENTITY_AFFINE, d_bv.data_ptr(), 12, d_bt.data_ptr())
        assert tr.commitScene() == 0
        for k in range(3):                           # three frames, one per slot stream, right behind the copies
            p, h, n = bufs[k]
            tr.setOutputBuffers(p.data_ptr(), h.data_ptr(), n.data_ptr(), cap)
            tr.traceSceneAsync(3 * rnd + k)
        tr.synchronize()
        ref = oracle.trace_frame(s, [(0, gv, gt, oracle.IDENTITY_AFFINE), (1, bv2, bt, oracle.IDENTITY_AFFINE)])
        for p, h, n in bufs:
            cnt = int(n[0].item())
            assert cnt == ref["points"].shape[0]
            assert np.array_equal(p.cpu().numpy()[:32 * cnt].reshape(cnt, 32), ref["points"])
    tr.close()


def test_stuck_chained_prefix_is_reported(oracle, capi, sensors, meshes):
    """LS_OPT_DEBUG_FAULT makes one pipelined frame publish a tag nobody waits for: its workgroups give up, raise the
    device status word, and the next host wait returns LS_ERR_HIP instead of an empty cloud with rc 0.  The handle
    keeps working afterwards."""
    s = sensors["0000"]
    tr = make_tracer(capi, s, "projection")
    tr.setOption(capi.LS_OPT_PIPELINE, 1)
    tr.addGeometry("ground", *[a.shape[0] for a in meshes["ground"]])
    tr.updateGeometry("ground", oracle.IDENTITY_AFFINE, *meshes["ground"])
    assert tr.commitScene() == 0
    tr.traceSceneAsync(0)
    tr.synchronize()                                  # a healthy frame: no error
    tr.setOption(capi.LS_OPT_DEBUG_FAULT, 1)
    tr.traceSceneAsync(1)                             # this frame's finish + pack will publish the wrong tag
    with pytest.raises(capi.LidarShooterHipError, match="chained prefix"):
        tr.synchronize()
    assert tr.info(capi.LS_INFO_DEVICE_STATUS) == 0   # read-and-clear happened with the error
    ref = oracle.trace_frame(s, [(0, *meshes["ground"], oracle.IDENTITY_AFFINE)])
    for k in range(3):                                # both key sets and the epoch tags are intact
        tr.traceSceneAsync(2 + k)
    tr.synchronize()
    rc, pts, hits = tr.traceScene(9)
    assert rc == 0 and np.array_equal(pts, ref["points"]) and np.array_equal(_hits_array(hits), ref["hits"])
    tr.close()


def _quad_grid(nx, ny, half=30.0, seed=3):
    """A relief of nx x ny quads (not planar: the two triangles of a quad differ) + the same surface as triangles."""
    xs, ys = np.linspace(-half, half, nx + 1), np.linspace(-half, half, ny + 1)
    X, Y = np.meshgrid(xs, ys, indexing="xy")
    rng = np.random.default_rng(seed)
    Z = 0.4 * np.sin(0.3 * X) * np.cos(0.2 * Y) + rng.uniform(-0.05, 0.05, X.shape)
    v = np.stack([X, Y, Z], -1).reshape(-1, 3).astype(np.float32)
    j, i = np.meshgrid(np.arange(ny), np.arange(nx), indexing="ij")
    a = (j * (nx + 1) + i).reshape(-1)
    q = np.stack([a, a + 1, a + nx + 2, a + nx + 1], -1).astype(np.uint32)     # v0 v1 v2 v3 around the cell
    return v, q


def test_quad_geometry(oracle, capi, sensors, meshes, engine):
    """RTC_GEOMETRY_TYPE_QUAD (EmbreeTracer.cpp:179-198, MeshTransformer.cpp:521-538): four indices per element, traced
    as Embree's triangle pair (v0,v1,v3), (v2,v3,v1); primID is the quad's index.  A quad relief next to the triangle
    mesh ben.stl, device and host index paths."""
    import torch
    s = sensors["0001"]
    v, q = _quad_grid(60, 40)
    bv, bt = meshes["ben"]
    tr = make_tracer(capi, s, engine)
    assert tr.addGeometry("relief", v.shape[0], q.shape[0], geometry_type=1) == 0
    assert tr.addGeometry("face", bv.shape[0], bt.shape[0]) == 1
    assert tr.getElementCount("relief") == q.shape[0] and tr.getVertexCount("relief") == v.shape[0]
    tr.updateGeometry("relief", oracle.IDENTITY_AFFINE, v, q)
    tr.updateGeometry("face", oracle.IDENTITY_AFFINE, bv, bt)
    assert tr.commitScene() == 0
    rc, pts, hits = tr.traceScene(0)
    ref = oracle.trace_frame(s, [(0, v, q, oracle.IDENTITY_AFFINE), (1, bv, bt, oracle.IDENTITY_AFFINE)])
    assert rc == 0 and len(pts) > 1000
    assert np.array_equal(pts, ref["points"]) and np.array_equal(_hits_array(hits), ref["hits"])
    on_quads = hits["geom"] == 0
    assert on_quads.sum() > 500 and int(hits["prim"][on_quads].max()) < q.shape[0]
    # the same quads as a plain triangle mesh give the same points and t, primID = 2 * quad (+1)
    tri = make_tracer(capi, s, engine)
    t2 = oracle.quads_to_triangles(q)
    tri.addGeometry("relief", v.shape[0], t2.shape[0])
    tri.addGeometry("face", bv.shape[0], bt.shape[0])
    tri.updateGeometry("relief", oracle.IDENTITY_AFFINE, v, t2)
    tri.updateGeometry("face", oracle.IDENTITY_AFFINE, bv, bt)
    assert tri.commitScene() == 0
    _, pts3, hits3 = tri.traceScene(0)
    assert np.array_equal(pts3, pts) and np.array_equal(hits3["prim"][on_quads] // 2, hits["prim"][on_quads])
    tri.close()
    # device-resident indices (converted into the library's triangle array), then a moved pose
    dev = torch.device("cuda", 0)
    dv = torch.from_numpy(v).to(dev)
    dq = torch.from_numpy(q.view(np.int32)).to(dev)
    A = oracle.affine_from_components(np.array((1.0, -2.0, 0.3), np.float32), np.array((0.02, 0.0, 0.4), np.float32))
    tr.updateGeometryDeviceShared("relief", A, dv.data_ptr(), 12, dq.data_ptr())
    assert tr.commitScene() == 0
    rc, pts, hits = tr.traceScene(1)
    ref = oracle.trace_frame(s, [(0, v, q, A), (1, bv, bt, oracle.IDENTITY_AFFINE)])
    assert np.array_equal(pts, ref["points"]) and np.array_equal(_hits_array(hits), ref["hits"])
    tr.close()


def test_bvh_refit(oracle, capi, sensors, meshes):
    """BVH engine: a commit after which only poses / vertices differ refits (OptixTracer.cpp:532-535 OPERATION_UPDATE):
    same Morton order and tree topology, all boxes recomputed; a topology change rebuilds.  Results equal the oracle either
    way, also when the refitted mesh has moved far from where it was sorted."""
    from lidarshooter_amd import synth
    s = sensors["0000"]
    gv, gt = synth.grid_mesh(120, 80, half=45.0, seed=2)
    bv, bt = meshes["ben"]
    tr = make_tracer(capi, s, "bvh")
    tr.setOption(capi.LS_OPT_BVH_INSTANCED, 0)           # the classic hierarchy in the sensor frame (the instanced one never refits)
    tr.addGeometry("ground", gv.shape[0], gt.shape[0])
    tr.addGeometry("face", bv.shape[0], bt.shape[0])
    tr.updateGeometry("ground", oracle.IDENTITY_AFFINE, gv, gt)
    tr.updateGeometry("face", oracle.IDENTITY_AFFINE, bv, bt)
    assert tr.commitScene() == 0 and tr.info(capi.LS_INFO_LAST_COMMIT_REFIT) == 0 and tr.info(capi.LS_INFO_BVH_INSTANCED) == 0
    poses = [((0, 0, 0), (0, 0, 0)), ((2.0, -1.0, 0.2), (0.0, 0.1, 0.8)), ((-25.0, 30.0, 1.0), (0.3, 0.0, -2.0)), ((40.0, 40.0, 5.0), (0.0, 0.0, 3.0))]
    for k, (lin, ang) in enumerate(poses):
        A = oracle.affine_from_components(np.array(lin, np.float32), np.array(ang, np.float32))
        gk = gv.copy()
        gk[:, 2] += np.float32(0.3 * k)                      # the ground's vertices change too (same indices)
        tr.updateGeometry("ground", oracle.IDENTITY_AFFINE, gk, None)
        tr.updateGeometryTransform("face", A)
        assert tr.commitScene() == 0
        assert tr.info(capi.LS_INFO_LAST_COMMIT_REFIT) == (1 if k else 1)   # the first loop commit already follows a build
        rc, pts, hits = tr.traceScene(k)
        ref = oracle.trace_frame(s, [(0, gk, gt, oracle.IDENTITY_AFFINE), (1, bv, bt, A)])
        assert rc == 0 and np.array_equal(pts, ref["points"]) and np.array_equal(_hits_array(hits), ref["hits"])
    # new indices for one mesh: a full build
    tr.updateGeometry("face", A, bv, bt[::-1].copy())
    assert tr.commitScene() == 0 and tr.info(capi.LS_INFO_LAST_COMMIT_REFIT) == 0
    rc, pts, hits = tr.traceScene(9)
    ref = oracle.trace_frame(s, [(0, gk, gt, oracle.IDENTITY_AFFINE), (1, bv, bt[::-1].copy(), A)])
    assert np.array_equal(pts, ref["points"]) and np.array_equal(_hits_array(hits), ref["hits"])
    # refit switched off
    tr.setOption(capi.LS_OPT_BVH_REFIT, 0)
    tr.updateGeometryTransform("face", oracle.IDENTITY_AFFINE)
    assert tr.commitScene() == 0 and tr.info(capi.LS_INFO_LAST_COMMIT_REFIT) == 0
    tr.close()


def test_bvh_refit_after_an_index_upload_under_another_engine(oracle, capi, sensors, meshes):
    """The classic hierarchy's refit reuses what the last build left on the device (sorted order, tree topology with the
    nodes' leaf ranges, the rebased index copy).  Indices uploaded while the PROJECTION engine is active (same triangle
    count, other triangles) must not be met by those leftovers when the BVH engine comes back: every frame equals the oracle."""
    from lidarshooter_amd import synth
    s = sensors["0000"]
    gv, gt = synth.grid_mesh(60, 40, half=40.0, seed=5)
    bv, bt = meshes["ben"]
    tr = make_tracer(capi, s, "bvh")
    tr.setOption(capi.LS_OPT_BVH_INSTANCED, 0)
    tr.addGeometry("ground", gv.shape[0], gt.shape[0])
    tr.addGeometry("face", bv.shape[0], bt.shape[0])
    tr.updateGeometry("ground", oracle.IDENTITY_AFFINE, gv, gt)
    tr.updateGeometry("face", oracle.IDENTITY_AFFINE, bv, bt)
    assert tr.commitScene() == 0
    A = oracle.affine_from_components(np.array((1.0, 2.0, 0.1), np.float32), np.array((0.0, 0.0, 0.5), np.float32))
    tr.updateGeometryTransform("face", A)
    assert tr.commitScene() == 0 and tr.info(capi.LS_INFO_LAST_COMMIT_REFIT) == 1
    rc, pts, hits = tr.traceScene(0)
    ref = oracle.trace_frame(s, [(0, gv, gt, oracle.IDENTITY_AFFINE), (1, bv, bt, A)])
    assert rc == 0 and np.array_equal(pts, ref["points"]) and np.array_equal(_hits_array(hits), ref["hits"])
    # projection engine: the ground gets other triangles over the same vertices (every other cell's diagonal pair dropped
    # for a copy of its neighbour: same count, different set)
    gt2 = gt.copy()
    gt2[0::4] = gt[1::4]
    tr.setOption(capi.LS_OPT_ENGINE, 2)
    tr.updateGeometry("ground", oracle.IDENTITY_AFFINE, gv, gt2)
    assert tr.commitScene() == 0
    rc, pts, hits = tr.traceScene(1)
    ref = oracle.trace_frame(s, [(0, gv, gt2, oracle.IDENTITY_AFFINE), (1, bv, bt, A)])
    assert rc == 0 and np.array_equal(pts, ref["points"]) and np.array_equal(_hits_array(hits), ref["hits"])
    # back to the BVH engine, poses only
    tr.setOption(capi.LS_OPT_ENGINE, 1)
    tr.updateGeometryTransform("face", oracle.IDENTITY_AFFINE)
    assert tr.commitScene() == 0
    rc, pts, hits = tr.traceScene(2)
    ref = oracle.trace_frame(s, [(0, gv, gt2, oracle.IDENTITY_AFFINE), (1, bv, bt, oracle.IDENTITY_AFFINE)])
    assert rc == 0 and np.array_equal(pts, ref["points"]) and np.array_equal(_hits_array(hits), ref["hits"])
    tr.updateGeometryTransform("face", A)
    assert tr.commitScene() == 0
    rc, pts, hits = tr.traceScene(3)
    ref = oracle.trace_frame(s, [(0, gv, gt2, oracle.IDENTITY_AFFINE), (1, bv, bt, A)])
    assert rc == 0 and np.array_equal(pts, ref["points"]) and np.array_equal(_hits_array(hits), ref["hits"])
    tr.close()


def test_bvh_instanced(oracle, capi, sensors, meshes):
    """BVH engine, instanced mode (the default): one hierarchy per geometry in mesh space, rays carried into it.  A commit
    after which only poses differ builds nothing; a vertex or topology change rebuilds that geometry alone; matrices that
    are not rigid work as long as they can be inverted, a singular one and a scene of more than 16 geometries take the
    classic path.  Every frame equals the oracle bit for bit."""
    from lidarshooter_amd import synth
    s = sensors["0000"]
    gv, gt = synth.grid_mesh(120, 80, half=45.0, seed=2)
    bv, bt = meshes["ben"]
    tr = make_tracer(capi, s, "bvh")
    tr.addGeometry("ground", gv.shape[0], gt.shape[0])
    tr.addGeometry("face", bv.shape[0], bt.shape[0])
    tr.updateGeometry("ground", oracle.IDENTITY_AFFINE, gv, gt)
    tr.updateGeometry("face", oracle.IDENTITY_AFFINE, bv, bt)
    assert tr.commitScene() == 0 and tr.info(capi.LS_INFO_BVH_INSTANCED) == 2

    def check(k, scene):
        rc, pts, hits = tr.traceScene(k)
        ref = oracle.trace_frame(s, scene)
        assert rc == 0 and np.array_equal(pts, ref["points"]) and np.array_equal(_hits_array(hits), ref["hits"])
        return len(pts)

    assert check(0, [(0, gv, gt, oracle.IDENTITY_AFFINE), (1, bv, bt, oracle.IDENTITY_AFFINE)]) > 1000
    # poses only: nothing is built, whatever the matrices (rigid, far away, scaled + sheared, mirrored)
    mats = [oracle.affine_from_components(np.array(lin, np.float32), np.array(ang, np.float32))
            for lin, ang in (((2.0, -1.0, 0.2), (0.0, 0.1, 0.8)), ((-25.0, 30.0, 1.0), (0.3, 0.0, -2.0)), ((40.0, 40.0, 5.0), (0.0, 0.0, 3.0)))]
    mats.append(np.array([1.5, 0.2, 0.0, 1.0, -0.1, 0.7, 0.3, -2.0, 0.0, 0.4, 2.0, 0.5], np.float32))
    mats.append(np.array([-1.0, 0.0, 0.0, 3.0, 0.0, 1.0, 0.0, 1.0, 0.0, 0.0, 1.0, 0.2], np.float32))
    G = oracle.affine_from_components(np.array((0.5, 0.25, -0.1), np.float32), np.array((0.01, -0.02, 0.3), np.float32))
    for k, A in enumerate(mats):
        tr.updateGeometryTransform("face", A)
        tr.updateGeometryTransform("ground", G if k % 2 else oracle.IDENTITY_AFFINE)
        assert tr.commitScene() == 0 and tr.info(capi.LS_INFO_BVH_INSTANCED) == 1
        check(k, [(0, gv, gt, G if k % 2 else oracle.IDENTITY_AFFINE), (1, bv, bt, A)])
    # new vertices for one mesh: that hierarchy is rebuilt
    gk = gv.copy()
    gk[:, 2] += np.float32(0.4)
    tr.updateGeometry("ground", oracle.IDENTITY_AFFINE, gk, None)
    assert tr.commitScene() == 0 and tr.info(capi.LS_INFO_BVH_INSTANCED) == 2
    check(20, [(0, gk, gt, oracle.IDENTITY_AFFINE), (1, bv, bt, mats[-1])])
    # new topology for the other
    tr.updateGeometry("face", mats[0], bv, bt[::-1].copy())
    assert tr.commitScene() == 0 and tr.info(capi.LS_INFO_BVH_INSTANCED) == 2
    check(21, [(0, gk, gt, oracle.IDENTITY_AFFINE), (1, bv, bt[::-1].copy(), mats[0])])
    # a leaf size change rebuilds everything; 3 triangles per leaf leaves short last leaves
    tr.setOption(capi.LS_OPT_LEAF_SIZE, 4)
    assert tr.commitScene() == 0 and tr.info(capi.LS_INFO_BVH_INSTANCED) == 2
    check(22, [(0, gk, gt, oracle.IDENTITY_AFFINE), (1, bv, bt[::-1].copy(), mats[0])])
    # a singular matrix (the mesh squashed flat): the classic path, same answer
    flat = np.array([1.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 0.0, -1.0], np.float32)
    tr.updateGeometryTransform("face", flat)
    assert tr.commitScene() == 0 and tr.info(capi.LS_INFO_BVH_INSTANCED) == 0
    check(23, [(0, gk, gt, oracle.IDENTITY_AFFINE), (1, bv, bt[::-1].copy(), flat)])
    # ... and back
    tr.updateGeometryTransform("face", mats[1])
    assert tr.commitScene() == 0 and tr.info(capi.LS_INFO_BVH_INSTANCED) == 2
    check(24, [(0, gk, gt, oracle.IDENTITY_AFFINE), (1, bv, bt[::-1].copy(), mats[1])])
    tr.removeGeometry("face")                           # (commits by itself, EmbreeTracer.cpp:252: the layout change rebuilt already)
    assert tr.commitScene() == 0 and tr.info(capi.LS_INFO_BVH_INSTANCED) == 1
    check(25, [(0, gk, gt, oracle.IDENTITY_AFFINE)])
    tr.close()
    # more geometries than a launch carries: classic
    tr = make_tracer(capi, s, "bvh")
    scene = []
    rng = np.random.default_rng(3)
    for i in range(18):
        v, t = synth.grid_mesh(8, 8, half=3.0, seed=i)
        A = oracle.affine_from_components(rng.uniform(-15, 15, 3).astype(np.float32), rng.uniform(-1, 1, 3).astype(np.float32))
        tr.addGeometry("m%d" % i, v.shape[0], t.shape[0])
        tr.updateGeometry("m%d" % i, A, v, t)
        scene.append((i, v, t, A))
    assert tr.commitScene() == 0 and tr.info(capi.LS_INFO_BVH_INSTANCED) == 0
    check(30, scene)
    for i in (17, 16):
        tr.removeGeometry("m%d" % i)
    assert tr.commitScene() == 0 and tr.info(capi.LS_INFO_BVH_INSTANCED) in (1, 2)
    check(31, scene[:16])
    tr.close()


def test_bvh_instanced_stress(oracle, capi, sensors):
    """Instanced hierarchies under hostile poses: soups of mixed triangle sizes whose mesh coordinates sit thousands of
    metres from their own origin (the matrix brings them back around the sensor), anisotropic scales up to 30 : 1, shears,
    mirrors, several geometries.  The ray map into mesh space and the widening of the boxes must never lose a hit: every
    frame equals the projection engine's (which the other tests tie to the oracle), and the first ones the oracle's."""
    import os
    from test_gpu_parity import _random_soup, _syn_sensor
    s0 = _syn_sensor(oracle, sensors, V=24, H=200)
    s = oracle.Sensor(uid="soup", vertical=s0.vertical, h_begin=s0.h_begin, h_end=s0.h_end, h_count=s0.h_count,
                      R=np.eye(3, dtype=np.float32).reshape(9), Rinv=np.eye(3, dtype=np.float32).reshape(9), t=np.zeros(3, np.float32))
    n_cases = int(os.environ.get("LS_STRESS_SEEDS", "24"))
    bvh, prj = make_tracer(capi, s, "bvh"), make_tracer(capi, s, "projection")
    names = []
    instanced = 0
    for case in range(n_cases):
        rng = np.random.default_rng(1000 + case)
        for tr in (bvh, prj):
            for nm in names:
                tr.removeGeometry(nm)
        names = []
        scene = []
        for gi in range(int(rng.integers(1, 4))):
            v, t = _random_soup(rng, int(rng.integers(50, 1500)), float(rng.uniform(2.0, 12.0)))
            # a well-conditioned but not rigid linear part: rotation * diag(scales) * rotation, sometimes mirrored / sheared
            q1, _ = np.linalg.qr(rng.normal(size=(3, 3)))
            q2, _ = np.linalg.qr(rng.normal(size=(3, 3)))
            sc = np.exp(rng.uniform(np.log(0.2), np.log(6.0), 3))
            lin = q1 @ np.diag(sc) @ q2
            if rng.random() < 0.3:
                lin = lin @ np.array([[1, rng.uniform(-0.8, 0.8), 0], [0, 1, 0], [0, 0, 1.0]])
            if rng.random() < 0.3:
                lin[:, 0] *= -1.0
            # the mesh as uploaded lives far from its own origin; the matrix puts it back around the sensor
            centre = rng.uniform(-1.0, 1.0, 3) * float(rng.choice([0.0, 50.0, 4000.0]))
            mesh = (np.linalg.inv(lin) @ (v.astype(np.float64).T)).T + centre
            A = np.concatenate([lin, (-lin @ centre)[:, None]], axis=1).astype(np.float32).reshape(12)
            mv = mesh.astype(np.float32)
            nm = "g%d" % gi
            names.append(nm)
            for tr in (bvh, prj):
                tr.addGeometry(nm, mv.shape[0], t.shape[0])
                tr.updateGeometry(nm, A, mv, t)
            scene.append((gi, mv, t, A))
        for tr in (bvh, prj):
            assert tr.commitScene() == 0
        instanced += 1 if bvh.info(capi.LS_INFO_BVH_INSTANCED) else 0
        rc1, p1, h1 = bvh.traceScene(case)
        rc2, p2, h2 = prj.traceScene(case)
        assert rc1 == 0 and rc2 == 0 and np.array_equal(p1, p2) and np.array_equal(_hits_array(h1), _hits_array(h2)), case
        if case < 4:
            ref = oracle.trace_frame(s, scene)
            assert np.array_equal(p1, ref["points"]) and np.array_equal(_hits_array(h1), ref["hits"])
            assert len(p1) > 200
    assert instanced >= n_cases - 2    # (a drawn matrix may exceed the conditioning limit: that scene takes the classic path)
    bvh.close()
    prj.close()


@pytest.mark.parametrize("engine_name", ["projection", "bvh"])
def test_two_step_trace_equals_one_step(oracle, capi, sensors, meshes, engine_name):
    """ls_trace_scene_begin / ls_trace_scene_expand (the hit count first, then the cloud expanded into the caller's memory
    half by half as it arrives) deliver exactly ls_trace_scene's points: the XT-32 scenes while a mesh moves, the empty
    scene, and -- BVH engine -- the fallback that completes the frame in the first step."""
    s = sensors["0001"]
    tr = make_tracer(capi, s, engine_name)
    rc, pts = tr.traceSceneTwoStep(0)
    assert rc == -1 and pts.shape == (0, 32)                 # nothing committed: OptixTracer.cpp:280-288
    tr.addGeometry("ground", *[a.shape[0] for a in meshes["ground"]])
    tr.addGeometry("face", *[a.shape[0] for a in meshes["ben"]])
    tr.updateGeometry("ground", oracle.IDENTITY_AFFINE, *meshes["ground"])
    tr.updateGeometry("face", oracle.IDENTITY_AFFINE, *meshes["ben"])
    for k in range(5):
        A = oracle.affine_from_components(np.array((0.6 * k, -0.4 * k, 0.05 * k), np.float32), np.array((0.0, 0.0, 0.2 * k), np.float32))
        tr.updateGeometryTransform("face", A)
        assert tr.commitScene() == 0
        rc, pts = tr.traceSceneTwoStep(k)
        ref = oracle.trace_frame(s, [(0, *meshes["ground"], oracle.IDENTITY_AFFINE), (1, *meshes["ben"], A)])
        assert rc == 0 and np.array_equal(pts, ref["points"])
        rc1, pts1, _ = tr.traceScene(k)                       # the one-step call still works in between
        assert rc1 == 0 and np.array_equal(pts1, ref["points"])
    # an expand without a begin is refused; a begin may be followed by another begin (the first frame is dropped)
    assert tr.L.ls_trace_scene_expand(tr.h, None) < 0
    import ctypes as C
    n = C.c_uint32()
    assert tr.L.ls_trace_scene_begin(tr.h, 7, C.byref(n)) == 0 and n.value == ref["points"].shape[0]
    rc, pts = tr.traceSceneTwoStep(8)
    assert rc == 0 and np.array_equal(pts, ref["points"])
    tr.close()


def test_two_step_trace_headline_size(oracle, capi, sensors):
    """The same at 128 x 4096 rays over 240 000 triangles (2 048 ray blocks: the first 1 024 are expanded while the rest
    arrive), several frames back to back, against the BVH oracle; odd destination alignment takes the plain-store path."""
    from lidarshooter_amd import synth
    v, t = synth.grid_mesh(400, 300)
    s = _syn_sensor(oracle, sensors, V=128, H=4096)
    tr = make_tracer(capi, s, "projection")
    tr.addGeometry("grid", v.shape[0], t.shape[0])
    tr.updateGeometry("grid", oracle.IDENTITY_AFFINE, v, t)
    assert tr.commitScene() == 0
    ref = oracle.trace_frame(s, [(0, v, t, oracle.IDENTITY_AFFINE)], use_bvh=True)
    for k in range(6):
        rc, pts = tr.traceSceneTwoStep(k)
        assert rc == 0 and np.array_equal(pts, ref["points"])
    import ctypes as C
    n = C.c_uint32()
    assert tr.L.ls_trace_scene_begin(tr.h, 9, C.byref(n)) == 0
    buf = np.zeros(int(n.value) * 32 + 64, np.uint8)
    off = 4 if buf.ctypes.data % 16 == 0 else (16 - buf.ctypes.data % 16) + 4   # 4 bytes past a 16-byte boundary
    assert tr.L.ls_trace_scene_expand(tr.h, buf.ctypes.data + off) == 0
    assert np.array_equal(buf[off:off + int(n.value) * 32].reshape(-1, 32), ref["points"])
    tr.close()


def test_two_step_trace_on_an_azimuth_shard(oracle, capi, sensors, meshes):
    """The two-step trace on a handle that traces a sector only (ls_tracer_set_shard): the (ray, t) records carry the ray's
    number on the FULL raster, the host rebuilds the points with the full tables -- the sector's cloud is the full cloud's
    points whose column lies in the sector, in the same order."""
    s = sensors["0001"]
    tr = make_tracer(capi, s, "projection")
    tr.addGeometry("ground", *[a.shape[0] for a in meshes["ground"]])
    tr.addGeometry("face", *[a.shape[0] for a in meshes["ben"]])
    tr.updateGeometry("ground", oracle.IDENTITY_AFFINE, *meshes["ground"])
    tr.updateGeometry("face", oracle.IDENTITY_AFFINE, *meshes["ben"])
    ref = oracle.trace_frame(s, [(0, *meshes["ground"], oracle.IDENTITY_AFFINE), (1, *meshes["ben"], oracle.IDENTITY_AFFINE)])
    cols = ref["hits"][:, 0] % s.H
    seen = 0
    for first, n in ((0, s.H // 3), (s.H // 3, s.H // 2), (s.H // 3 + s.H // 2, s.H - s.H // 3 - s.H // 2)):
        tr.setShard(first, n)
        assert tr.commitScene() == 0
        rc, pts = tr.traceSceneTwoStep(first)
        want = ref["points"][(cols >= first) & (cols < first + n)]
        assert rc == 0 and np.array_equal(pts, want), (first, n, pts.shape, want.shape)
        seen += pts.shape[0]
    assert seen == ref["points"].shape[0] and seen > 0
    tr.close()


@pytest.mark.parametrize("V,H,blocks", [(160, 4096, 2560), (161, 4090, 2573), (131, 4000, 2047), (67, 4001, 1048), (40, 4099, 641)])
def test_raster_beyond_2048_ray_blocks(oracle, capi, sensors, V, H, blocks):
    """160 x 4096 rays = 2 560 blocks of 256: k_pack asks for a thread's block counts eight at a time, which covers the
    2 048 blocks of the headline raster in one go -- here the loop comes round a second time, and the workgroup that publishes
    the two-step trace's hit count owns chunks of ten counts.  One-step on both engines, two-step, and three frames in flight,
    all against the BVH oracle.  The other rasters are for the frames in flight: there the finish and pack passes take 8 / 4 /
    2 rays per lane (from 2 048 / 1 024 / 512 ray blocks: k_project_finish_wide, k_pack_wide), and these block counts leave the
    last workgroup a part of its span, the last block a part of its rays."""
    import torch
    from lidarshooter_amd import synth
    v, t = synth.grid_mesh(300, 200)
    s = _syn_sensor(oracle, sensors, V=V, H=H)
    assert (s.V * s.H + 255) // 256 == blocks
    scene = [(0, v, t, oracle.IDENTITY_AFFINE)]
    if blocks == 2047:   # two geometries (a quad mesh among them: primID = triangle >> 1): the hit records' (geomID, primID) look-up
        qv = np.array([[-30, -30, 1.5], [30, -30, 1.5], [30, 30, 2.5], [-30, 30, 2.5], [0, 0, 6.0]], np.float32)
        qi = np.array([[0, 1, 2, 3], [0, 1, 4, 4]], np.uint32)
        scene.append((1, qv, qi, oracle.IDENTITY_AFFINE))
    ref = oracle.trace_frame(s, scene, use_bvh=True)
    assert ref["points"].shape[0] > 30000
    for engine_name in ("projection", "bvh"):
        tr = make_tracer(capi, s, engine_name)
        tr.addGeometry("grid", v.shape[0], t.shape[0])
        tr.updateGeometry("grid", oracle.IDENTITY_AFFINE, v, t)
        if len(scene) > 1:
            assert tr.addGeometry("quads", scene[1][1].shape[0], scene[1][2].shape[0], capi.LS_GEOMETRY_TYPE_QUAD) == 1
            tr.updateGeometry("quads", oracle.IDENTITY_AFFINE, scene[1][1], scene[1][2])
        assert tr.commitScene() == 0
        rc, pts, hits = tr.traceScene(0)
        assert rc == 0 and np.array_equal(pts, ref["points"]) and np.array_equal(_hits_array(hits), ref["hits"])
        rc, pts = tr.traceSceneTwoStep(1)
        assert rc == 0 and np.array_equal(pts, ref["points"])
        if engine_name == "projection":
            tr.setOption(capi.LS_OPT_PIPELINE, 2)
            assert tr.info(capi.LS_INFO_PIPELINE_MODE) == 2
            cap = s.V * s.H
            bufs = [(torch.zeros(32 * cap, dtype=torch.uint8, device="cuda:0"), torch.zeros(16 * cap, dtype=torch.uint8, device="cuda:0"),
                     torch.zeros(4, dtype=torch.int32, device="cuda:0")) for _ in range(3)]
            for i in range(9):
                p, h, n = bufs[i % 3]
                assert tr.commitScene() == 0
                tr.setOutputBuffers(p.data_ptr(), h.data_ptr(), n.data_ptr(), cap)
                tr.traceSceneAsync(i)
            tr.synchronize()
            for p, h, n in bufs:
                cnt = int(n[0].item())
                assert cnt == ref["points"].shape[0]
                assert np.array_equal(p.cpu().numpy()[:32 * cnt].reshape(cnt, 32), ref["points"])
                assert np.array_equal(h.cpu().numpy()[:16 * cnt].view(np.uint32).reshape(cnt, 4), ref["hits"])
        tr.close()
