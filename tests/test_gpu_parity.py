"""GPU parity tests: the HIP path, called through the C ABI (lidarshooter_amd/capi.py), against the
CPU oracle on identical inputs.  Bar (BASELINE.json north_star): identical hit triangle ids and
channel indices; hit t and XYZ within 1e-4 relative.  Because the kernels and the oracle use the
same float operation sequence, the tests first ask for bit-equality and report how far off t is
if that ever fails; ids must always be identical.

Test cases mirror the reference's gtests (EmbreeTracer_test.cpp, OptixTracer_test.cpp)."""
import hashlib
import os

import numpy as np
import pytest

from conftest import GOLDEN, make_tracer

pytestmark = pytest.mark.gpu

REL_TOL = 1e-4  # north_star tolerance on t / XYZ


def _add(tr, name, mesh):
    gid = tr.addGeometry(name, mesh[0].shape[0], mesh[1].shape[0])
    assert gid >= 0
    return gid


def _assert_parity(O, sensor, tr, meshes_list, pts, hits):
    """meshes_list: [(geomID, verts, tris, affine)] as given to the tracer."""
    ref = O.trace_frame(sensor, meshes_list)
    t, gid = tr.denseHits()
    assert np.array_equal(gid, ref["gid"]), "hit triangle ids / hit-miss set differ from the oracle"
    hit = ref["gid"] != O.INVALID
    rel = np.abs(t[hit] - ref["t"][hit]) / np.abs(ref["t"][hit])
    assert rel.size == 0 or float(rel.max()) <= REL_TOL
    assert np.array_equal(t, ref["t"]), f"t not bit-equal (max rel err {rel.max() if rel.size else 0})"
    # packed outputs
    assert pts.shape == ref["points"].shape
    assert np.array_equal(pts, ref["points"])
    got = np.stack([hits["ray"], hits["geom"], hits["prim"], hits["t"].view(np.uint32)], axis=1)
    assert np.array_equal(got, ref["hits"])
    return ref


@pytest.mark.parametrize("uid,with_ben,expected", [
    ("0000", False, 1668),   # EmbreeTracer_test.cpp:122-135 / OptixTracer_test.cpp:93-120
    ("0000", True, 1781),    # OptixTracer_test.cpp:122-169
    ("0001", False, 1633),
    ("0001", True, 1769),
])
def test_xt32_known_answers(oracle, capi, sensors, meshes, uid, with_ben, expected, engine):
    s = sensors[uid]
    tr = make_tracer(capi, s, engine)
    assert tr.getTotalRays() == 4800                      # LidarDevice_test.cpp:58
    ml = []
    g0 = _add(tr, "ground", meshes["ground"])
    ml.append((g0, *meshes["ground"], oracle.IDENTITY_AFFINE))
    if with_ben:
        g1 = _add(tr, "face", meshes["ben"])
        ml.append((g1, *meshes["ben"], oracle.IDENTITY_AFFINE))
    for gid, v, t, A in ml:
        assert tr.updateGeometry("ground" if gid == g0 else "face", A, v, t) == 0
    assert tr.commitScene() == 0
    rc, pts, hits = tr.traceScene(0)
    assert rc == 0
    assert len(pts) == expected                           # cloud->width * cloud->height
    _assert_parity(oracle, s, tr, ml, pts, hits)
    # golden vectors committed under tests/golden
    g = np.load(os.path.join(GOLDEN, "xt32_golden.npz"))
    k = f"lidar_{uid}_{'ground_ben' if with_ben else 'ground'}"
    t, gid = tr.denseHits()
    assert np.array_equal(gid, g[k + "_gid"]) and np.array_equal(t, g[k + "_t"])
    assert hashlib.sha256(pts.tobytes()).digest() == g[k + "_points_sha256"].tobytes()
    tr.close()


def test_geometry_bookkeeping(capi, sensors, meshes):
    # EmbreeTracer_test.cpp:86-120, OptixTracer_test.cpp:171-215
    tr = make_tracer(capi, sensors["0000"])
    gid = _add(tr, "mesh", meshes["ground"])
    assert tr.getGeometryCount() == 1
    assert tr.getVertexCount("mesh") == 98 and tr.getElementCount("mesh") == 162
    assert tr.getGeometryId("mesh") == gid
    assert tr.getGeometryType("mesh") == capi.LS_GEOMETRY_TYPE_TRIANGLE        # EmbreeTracer_test.cpp:116-120
    g2 = _add(tr, "face", meshes["ben"])
    assert g2 == gid + 1 and tr.getGeometryCount() == 2
    assert tr.addGeometry("plate", 4, 1, geometry_type=capi.LS_GEOMETRY_TYPE_QUAD) == 2
    assert tr.getGeometryType("plate") == capi.LS_GEOMETRY_TYPE_QUAD and tr.removeGeometry("plate") == 2
    assert tr.getGeometryType("plate") < 0 and tr.getGeometryType("nope") < 0  # EmbreeTracer.cpp:103-113 throws code 8 there
    assert tr.addGeometry("face", 3, 1) < 0                                   # duplicate key
    assert tr.addGeometry("grid", 4, 1, geometry_type=2) < 0                  # unsupported type (RTC_GEOMETRY_TYPE_GRID)
    assert tr.removeGeometry("mesh") == gid
    assert tr.getGeometryCount() == 1
    assert tr.removeGeometry("mesh") == -1                                    # EmbreeTracer.cpp:224-225
    g3 = _add(tr, "again", meshes["ground"])
    assert g3 == gid                                                          # lowest free id is reused
    assert tr.removeGeometry("face") == g2 and tr.removeGeometry("again") == g3
    assert tr.getGeometryCount() == 0
    assert tr.getGeometryId("nope") < 0
    tr.close()


def test_remove_and_retrace_sequence(oracle, capi, sensors, meshes, engine):
    # OptixTracer_test.cpp:217-311: 1781 -> remove "face" -> 1668 -> remove "ground" -> -1 / 0 points
    s = sensors["0000"]
    tr = make_tracer(capi, s, engine)
    _add(tr, "ground", meshes["ground"])
    _add(tr, "face", meshes["ben"])
    tr.updateGeometry("ground", oracle.IDENTITY_AFFINE, *meshes["ground"])
    tr.updateGeometry("face", oracle.IDENTITY_AFFINE, *meshes["ben"])
    assert tr.commitScene() == 0
    rc, pts, _ = tr.traceScene(0)
    assert rc == 0 and len(pts) == 1781
    assert tr.removeGeometry("face") == 1
    tr.updateGeometry("ground", oracle.IDENTITY_AFFINE, *meshes["ground"])
    assert tr.commitScene() == 0
    rc, pts, hits = tr.traceScene(1)
    assert rc == 0 and len(pts) == 1668
    _assert_parity(oracle, s, tr, [(0, *meshes["ground"], oracle.IDENTITY_AFFINE)], pts, hits)
    assert tr.removeGeometry("ground") == 0
    assert tr.commitScene() == -1                         # OptixTracer.cpp:266-267
    rc, pts, _ = tr.traceScene(2)
    assert rc == -1 and len(pts) == 0                     # OptixTracer.cpp:280-288
    assert tr.getGeometryCount() == 0
    tr.close()


def test_dual_sensor_shared_scene(oracle, capi, sensors, meshes, engine):
    # BASELINE.json configs[2]: lidar_0000 + lidar_0001, one tracer per sensor (mainwindow.cpp:258)
    trs = {u: make_tracer(capi, sensors[u], engine) for u in ("0000", "0001")}
    for u, tr in trs.items():
        _add(tr, "ground", meshes["ground"])
        _add(tr, "face", meshes["ben"])
        tr.updateGeometry("ground", oracle.IDENTITY_AFFINE, *meshes["ground"])
        tr.updateGeometry("face", oracle.IDENTITY_AFFINE, *meshes["ben"])
        assert tr.commitScene() == 0
    ml = [(0, *meshes["ground"], oracle.IDENTITY_AFFINE), (1, *meshes["ben"], oracle.IDENTITY_AFFINE)]
    for u, n in (("0000", 1781), ("0001", 1769)):
        rc, pts, hits = trs[u].traceScene(5)
        assert rc == 0 and len(pts) == n
        _assert_parity(oracle, sensors[u], trs[u], ml, pts, hits)
    for tr in trs.values():
        tr.close()


def test_moved_mesh_components(oracle, capi, sensors, meshes, engine):
    # updateGeometry(name, translation, rotation, mesh): EmbreeTracer.cpp:276-288, MeshTransformer.cpp:467-477
    s = sensors["0000"]
    lin, ang = [1.5, -2.0, 0.25], [0.1, -0.2, 0.7]
    tr = make_tracer(capi, s, engine)
    _add(tr, "ground", meshes["ground"])
    _add(tr, "face", meshes["ben"])
    tr.updateGeometry("ground", oracle.IDENTITY_AFFINE, *meshes["ground"])
    tr.updateGeometryComponents("face", lin, ang, *meshes["ben"])
    assert tr.commitScene() == 0
    rc, pts, hits = tr.traceScene(0)
    A = oracle.affine_from_components(lin, ang)
    _assert_parity(oracle, s, tr, [(0, *meshes["ground"], oracle.IDENTITY_AFFINE), (1, *meshes["ben"], A)], pts, hits)
    g = np.load(os.path.join(GOLDEN, "xt32_golden.npz"))
    assert np.array_equal(A, g["benmoved_affine"])
    t, gid = tr.denseHits()
    assert np.array_equal(gid, g["lidar_0000_ground_benmoved_gid"])
    assert np.array_equal(t, g["lidar_0000_ground_benmoved_t"])
    # transform-only update (AffineMesh pose change): back to identity without re-sending vertices
    tr.updateGeometryTransform("face", oracle.IDENTITY_AFFINE)
    assert tr.commitScene() == 0
    rc, pts, hits = tr.traceScene(1)
    assert len(pts) == 1781
    _assert_parity(oracle, s, tr, [(0, *meshes["ground"], oracle.IDENTITY_AFFINE), (1, *meshes["ben"], oracle.IDENTITY_AFFINE)], pts, hits)
    # and to a third pose, still without vertex traffic
    A2 = oracle.affine_from_components([-3.0, 4.0, 0.5], [0.0, 0.3, -1.1])
    tr.updateGeometryTransform("face", A2)
    assert tr.commitScene() == 0
    rc, pts, hits = tr.traceScene(2)
    _assert_parity(oracle, s, tr, [(0, *meshes["ground"], oracle.IDENTITY_AFFINE), (1, *meshes["ben"], A2)], pts, hits)
    tr.close()


def test_vertex_stride_and_transformed_vertices(oracle, capi, sensors, meshes, engine):
    s = sensors["0001"]
    v, t = meshes["ben"]
    rec = np.zeros((v.shape[0], 8), np.float32)           # 32-byte XYZIRPoint-like records
    rec[:, :3] = v
    rec[:, 3:] = 123.0
    tr = make_tracer(capi, s, engine)
    _add(tr, "face", meshes["ben"])
    A = oracle.affine_from_components([0.5, 0.25, -0.125], [0.3, 0.2, -0.1])
    tr.updateGeometry("face", A, rec, t, stride=32)
    assert tr.commitScene() == 0
    dv, dt = tr.downloadScene()
    assert np.array_equal(dv, oracle.transform_vertices(v, A, s))     # bit-exact vertex transform
    assert np.array_equal(dt, t)
    rc, pts, hits = tr.traceScene(0)
    _assert_parity(oracle, s, tr, [(0, v, t, A)], pts, hits)
    tr.close()


@pytest.mark.parametrize("leaf", [1, 2, 4, 8])
def test_leaf_sizes(oracle, capi, sensors, meshes, leaf):
    s = sensors["0000"]
    tr = make_tracer(capi, s, "bvh")
    tr.setOption(capi.LS_OPT_BVH_INSTANCED, 0)     # the downloaded structure and the visit counts are the classic hierarchy's
    tr.setOption(capi.LS_OPT_LEAF_SIZE, leaf)
    _add(tr, "ground", meshes["ground"])
    _add(tr, "face", meshes["ben"])
    tr.updateGeometry("ground", oracle.IDENTITY_AFFINE, *meshes["ground"])
    tr.updateGeometry("face", oracle.IDENTITY_AFFINE, *meshes["ben"])
    assert tr.commitScene() == 0
    rc, pts, hits = tr.traceScene(0)
    assert len(pts) == 1781
    _assert_parity(oracle, s, tr, [(0, *meshes["ground"], oracle.IDENTITY_AFFINE),
                                   (1, *meshes["ben"], oracle.IDENTITY_AFFINE)], pts, hits)
    # the BVH walked on the CPU (same arrays, same order) gives the same answer and the same counts
    tr.setOption(capi.LS_OPT_COUNT_VISITS, 1)
    tr.traceScene(1)
    nodes, tri, g = tr.downloadBvh()
    assert g == leaf
    t2, gid2, stats = oracle.fat_traverse_stats(nodes, tri, g, oracle.ray_dirs(s))
    t, gid = tr.denseHits()
    assert np.array_equal(gid2, gid) and np.array_equal(t2, t)
    assert tr.visitCounts() == (int(stats[0]), int(stats[1]))
    tr.close()


def test_bvh_structure(oracle, capi, sensors, meshes):
    """Every node's child boxes contain the grandchildren's; the leaves partition the triangles;
    a left-first depth-first walk meets the leaves in Morton order."""
    tr = make_tracer(capi, sensors["0000"], "bvh")
    tr.setOption(capi.LS_OPT_BVH_INSTANCED, 0)     # the debug download shows the classic sensor-frame hierarchy
    tr.setOption(capi.LS_OPT_LEAF_SIZE, 2)
    _add(tr, "face", meshes["ben"])
    tr.updateGeometry("face", oracle.IDENTITY_AFFINE, *meshes["ben"])
    assert tr.commitScene() == 0
    nodes, tri, g = tr.downloadBvh()
    nt = tri.shape[0]
    L = (nt + g - 1) // g
    assert nodes.shape[0] == L - 1
    assert sorted(tri["gid"].tolist()) == list(range(nt))
    LEAF = capi.LEAF_BIT
    seen_leaves, visited, stack = [], 0, [0]
    while stack:
        ref = stack.pop()
        if ref & LEAF:
            k = ref & ~LEAF
            seen_leaves.append(k)
            continue
        visited += 1
        nd = nodes[ref]
        for side, child in (("l", int(nd["left"])), ("r", int(nd["right"]))):
            lo, hi = nd[side + "lo"], nd[side + "hi"]
            if child & LEAF:
                k = child & ~LEAF
                v = tri["v0"][k * g:min(k * g + g, nt)]
                assert np.all(v >= lo) and np.all(v <= hi)
            else:
                c = nodes[child]
                assert np.all(np.minimum(c["llo"], c["rlo"]) >= lo) and np.all(np.maximum(c["lhi"], c["rhi"]) <= hi)
        stack.append(int(nd["right"]))
        stack.append(int(nd["left"]))
    assert visited == L - 1
    assert seen_leaves == list(range(L))                  # left-to-right = Morton order
    tr.close()


@pytest.mark.parametrize("n", [1, 63, 64, 65, 2047, 2048, 2049, 100003, (2 << 20) + 77])
def test_build_sort_is_a_stable_radix_sort(capi, sensors, n):
    """ls_sort.hip (the hand-written sort of the (re)build path) against numpy's stable argsort: 30-bit keys with the
    heavy ties a Morton order has (few distinct values in the top digits), tile edges (2 048-key tiles up to 2 M pairs,
    8 192 above), ragged tails; equal keys keep their input order."""
    rng = np.random.default_rng(n)
    tr = make_tracer(capi, sensors["0000"])
    for kind in ("random", "ties", "sorted", "one_value"):
        if kind == "random":
            keys = rng.integers(0, 1 << 30, n, dtype=np.uint32)
        elif kind == "ties":
            keys = (rng.integers(0, 7, n, dtype=np.uint32) << 24) | (rng.integers(0, 3, n, dtype=np.uint32) << 9)
        elif kind == "sorted":
            keys = np.sort(rng.integers(0, 1 << 30, n, dtype=np.uint32))[::-1].copy()
        else:
            keys = np.full(n, 0x2AAAAAAA, np.uint32)
        vals = np.arange(n, dtype=np.uint32)
        k, v = tr.sortPairs(keys, vals)
        order = np.argsort(keys, kind="stable")
        assert np.array_equal(k, keys[order]) and np.array_equal(v, vals[order]), kind
    tr.close()


def test_raygen_kernel(oracle, capi, sensors):
    # LidarDevice::allRaysGPU (LidarDeviceKernels.cu:25-126) vs the libm tables of the CPU path
    import torch
    s = sensors["0000"]
    tr = make_tracer(capi, s)
    n = tr.getTotalRays()
    d = torch.zeros(3, n, dtype=torch.float32, device="cuda:0")
    tr.generateRays(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr())
    tr.synchronize()
    assert np.array_equal(d.cpu().numpy().T, oracle.ray_dirs(s))
    tr.close()


@pytest.mark.parametrize("shard", [None, (37, 50)])
def test_raygen_kernel_reference_buffers(oracle, capi, sensors, shard):
    """allRaysGPUKernel's two outputs in the reference's own layout (LidarDeviceKernels.cu:38-51, Ray.hpp:16-35,
    Hit.hpp:16-29): Ray{origin 0, direction} 32 B and Hit{t = 1e16, intensity = 64, ring = channel} 24 B per ray,
    against the oracle's restatement; also one output at a time, and an azimuth shard (shard-compact order)."""
    import torch
    s = sensors["0001"]
    tr = make_tracer(capi, s)
    if shard:
        tr.setShard(*shard)
    n = tr.getTotalRays()
    want_r, want_h = oracle.all_rays_aos(s)
    if shard:
        keep = (np.arange(s.V)[:, None] * s.H + shard[0] + np.arange(shard[1])[None, :]).reshape(-1)
        want_r, want_h = want_r[keep], want_h[keep]
    assert n == want_r.shape[0]
    rays = torch.full((n * 32,), 0xAB, dtype=torch.uint8, device="cuda:0")
    hits = torch.full((n * 24,), 0xAB, dtype=torch.uint8, device="cuda:0")
    tr.generateRaysAos(rays.data_ptr(), hits.data_ptr())
    tr.synchronize()
    got_r = rays.cpu().numpy().view(capi.RAY_DTYPE)
    got_h = hits.cpu().numpy().view(capi.REFHIT_DTYPE)
    assert got_r.tobytes() == want_r.tobytes() and got_h.tobytes() == want_h.tobytes()
    assert np.all(got_h["ring"] == np.repeat(np.arange(s.V), n // s.V)) and np.all(got_h["intensity"] == 64.0)
    rays.fill_(0)
    hits.fill_(0)
    tr.generateRaysAos(rays.data_ptr(), None)
    tr.generateRaysAos(None, hits.data_ptr())
    tr.synchronize()
    assert rays.cpu().numpy().tobytes() == want_r.tobytes() and hits.cpu().numpy().tobytes() == want_h.tobytes()
    with pytest.raises(capi.LidarShooterHipError):
        tr.generateRaysAos(None, None)
    tr.close()


def test_shards_union_equals_full(oracle, capi, sensors, meshes, engine):
    from lidarshooter_amd import synth
    s = sensors["0000"]
    ml = [(0, *meshes["ground"], oracle.IDENTITY_AFFINE), (1, *meshes["ben"], oracle.IDENTITY_AFFINE)]
    ref = oracle.trace_frame(s, ml)
    allhits = []
    for world in (3, 8):
        allhits.clear()
        for rank in range(world):
            tr = make_tracer(capi, s, engine)
            first, n = synth.shard_columns(s.H, world, rank)
            tr.setShard(first, n)
            assert tr.getTotalRays() == s.V * n
            _add(tr, "ground", meshes["ground"])
            _add(tr, "face", meshes["ben"])
            tr.updateGeometry("ground", oracle.IDENTITY_AFFINE, *meshes["ground"])
            tr.updateGeometry("face", oracle.IDENTITY_AFFINE, *meshes["ben"])
            assert tr.commitScene() == 0
            rc, pts, hits = tr.traceScene(0)
            allhits.append((pts, hits))
            tr.close()
        hits = np.concatenate([h for _, h in allhits])
        pts = np.concatenate([p for p, _ in allhits])
        order = np.argsort(hits["ray"], kind="stable")
        got = np.stack([hits["ray"], hits["geom"], hits["prim"], hits["t"].view(np.uint32)], axis=1)[order]
        assert np.array_equal(got, ref["hits"])
        assert np.array_equal(pts[order], ref["points"])


def _syn_sensor(oracle, sensors, V=64, H=512, begin=0.0, end=360.0, vertical=None):
    from lidarshooter_amd import synth
    base = sensors["0000"]
    vert = synth.syn_vertical(V) if vertical is None else np.asarray(vertical, np.float32)
    return oracle.Sensor(uid="syn", vertical=vert, h_begin=np.float32(begin), h_end=np.float32(end), h_count=H,
                         R=base.R, Rinv=base.Rinv, t=base.t)


def test_synthetic_medium_vs_oracle_and_bruteforce(oracle, capi, sensors, engine):
    """20k-triangle relief mesh, a 64x512 sensor: both engines == oracle (CPU BVH == CPU brute force)
    == the GPU exhaustive kernel."""
    from lidarshooter_amd import synth
    v, t = synth.grid_mesh(100, 100, half=40.0, seed=11)
    s = _syn_sensor(oracle, sensors)
    ml = [(0, v, t, oracle.IDENTITY_AFFINE)]
    ref = oracle.trace_frame(s, ml, use_bvh=True)
    assert int((ref["gid"] != oracle.INVALID).sum()) > 5000
    for leaf in ((1, 4) if engine == "bvh" else (1,)):
        tr = make_tracer(capi, s, engine)
        tr.setOption(capi.LS_OPT_LEAF_SIZE, leaf)
        tr.addGeometry("g", v.shape[0], t.shape[0])
        tr.updateGeometry("g", oracle.IDENTITY_AFFINE, v, t)
        assert tr.commitScene() == 0
        rc, pts, hits = tr.traceScene(0)
        _assert_parity(oracle, s, tr, ml, pts, hits)
        bt, bg = tr.bruteForce()
        assert np.array_equal(bg, ref["gid"]) and np.array_equal(bt, ref["t"])
        tr.close()


def _random_soup(rng, n, scale):
    """Triangles of mixed sizes all around the sensor: some straddle the vertical axis, some the
    azimuth wrap, some are slivers, some huge."""
    c = rng.normal(0.0, scale, size=(n, 1, 3))
    size = np.exp(rng.uniform(np.log(0.02), np.log(scale), size=(n, 1, 1)))
    tri = c + rng.normal(0.0, 1.0, size=(n, 3, 3)) * size
    verts = tri.reshape(-1, 3).astype(np.float32)
    idx = np.arange(3 * n, dtype=np.uint32).reshape(n, 3)
    return verts, idx


@pytest.mark.parametrize("case", ["full_circle", "partial_negative_begin", "reversed_sweep", "unsorted_channels",
                                  "steep_channels"])
def test_random_soup_both_engines(oracle, capi, sensors, engine, case):
    """Exhaustive answer on arbitrary geometry and odd sensor rasters: the footprint bounds of the
    projection engine (and the BVH boxes) must never lose a hit."""
    rng = np.random.default_rng({"full_circle": 1, "partial_negative_begin": 2, "reversed_sweep": 3,
                                 "unsorted_channels": 4, "steep_channels": 5}[case])
    kw = dict(V=24, H=200)
    if case == "partial_negative_begin":
        kw.update(begin=-170.0, end=35.0)
    elif case == "reversed_sweep":
        kw.update(begin=300.0, end=-60.0)
    elif case == "unsorted_channels":
        kw.update(vertical=rng.permutation(np.linspace(-40.0, 35.0, 24)))
    elif case == "steep_channels":
        kw.update(vertical=np.linspace(-90.0, 90.0, 24))
    s = _syn_sensor(oracle, sensors, **kw)
    # the sensor frame is what matters: identity pose puts the soup right around the origin
    s = oracle.Sensor(uid="soup", vertical=s.vertical, h_begin=s.h_begin, h_end=s.h_end, h_count=s.h_count,
                      R=np.eye(3, dtype=np.float32).reshape(9), Rinv=np.eye(3, dtype=np.float32).reshape(9),
                      t=np.zeros(3, np.float32))
    v, t = _random_soup(rng, 3000, 8.0)
    # a few hand-made nasties: a triangle containing the vertical axis above and below, one through
    # the azimuth wrap, a big far wall, a needle
    extra = np.array([[[-1, -1, 3], [2, -1, 3], [-1, 2, 3]], [[-1, -1, -2], [2, -1, -2], [-1, 2, -2]],
                      [[-5, -0.5, -1], [-5, 0.5, -1], [-5, 0, 2]], [[40, -60, -20], [40, 60, -20], [40, 0, 50]],
                      [[3, 3, -1], [3.0001, 3, 1], [3, 3.0001, 0]]], np.float32)
    v = np.concatenate([v, extra.reshape(-1, 3)])
    t = np.concatenate([t, (np.arange(15, dtype=np.uint32) + 9000).reshape(5, 3)])
    ml = [(0, v, t, oracle.IDENTITY_AFFINE)]
    ref = oracle.trace_frame(s, ml)
    assert int((ref["gid"] != oracle.INVALID).sum()) > 1000
    tr = make_tracer(capi, s, engine)
    tr.addGeometry("soup", v.shape[0], t.shape[0])
    tr.updateGeometry("soup", oracle.IDENTITY_AFFINE, v, t)
    assert tr.commitScene() == 0
    rc, pts, hits = tr.traceScene(0)
    _assert_parity(oracle, s, tr, ml, pts, hits)
    tr.close()


def test_many_geometries_one_frame(oracle, capi, sensors, engine):
    """37 geometries of very different sizes (3 ... 5000 triangles), some moved by an affine: the
    projection engine hands them to its kernel 16 per launch, each cut into waves by its own size;
    geomID / primID bookkeeping and the closest-hit fold across geometries must equal the oracle."""
    rng = np.random.default_rng(37)
    s = _syn_sensor(oracle, sensors, V=24, H=200)
    s = oracle.Sensor(uid="many", vertical=s.vertical, h_begin=s.h_begin, h_end=s.h_end, h_count=s.h_count,
                      R=np.eye(3, dtype=np.float32).reshape(9), Rinv=np.eye(3, dtype=np.float32).reshape(9),
                      t=np.zeros(3, np.float32))
    tr = make_tracer(capi, s, engine)
    ml = []
    for g in range(37):
        n = [1, 7, 64, 65, 300, 5000, 17][g % 7]
        v, t = _random_soup(rng, n, 6.0 + g % 5)
        A = oracle.IDENTITY_AFFINE if g % 3 else oracle.affine_from_components(
            rng.normal(0, 2.0, 3).astype(np.float32), rng.normal(0, 0.7, 3).astype(np.float32))
        gid = tr.addGeometry(f"g{g}", v.shape[0], t.shape[0])
        assert gid == g
        tr.updateGeometry(f"g{g}", A, v, t)
        ml.append((gid, v, t, A))
    assert tr.commitScene() == 0
    rc, pts, hits = tr.traceScene(0)
    ref = _assert_parity(oracle, s, tr, ml, pts, hits)
    assert len(set(ref["hits"][:, 1].tolist())) >= 3       # the cloud mixes several geometries
    if engine == "projection":
        # the same scene with two frames in flight: three launches per frame here (16 + 16 + 5 geometries),
        # the previous frame's finish + pack ride in the first of them
        import torch
        tr.setOption(capi.LS_OPT_PIPELINE, 1)
        cap = s.V * s.H
        bufs = [(torch.zeros(32 * cap, dtype=torch.uint8, device="cuda:0"), torch.zeros(16 * cap, dtype=torch.uint8, device="cuda:0"),
                 torch.zeros(4, dtype=torch.int32, device="cuda:0")) for _ in range(2)]
        for i in range(5):
            p, h, n = bufs[i & 1]
            tr.setOutputBuffers(p.data_ptr(), h.data_ptr(), n.data_ptr(), cap)
            tr.traceSceneAsync(10 + i)
        tr.synchronize()
        for p, h, n in bufs:
            k = int(n[0].item())
            assert k == ref["points"].shape[0]
            assert np.array_equal(p.cpu().numpy()[:32 * k].reshape(k, 32), ref["points"])
            assert np.array_equal(h.cpu().numpy()[:16 * k].view(np.uint32).reshape(k, 4), ref["hits"])
    tr.close()


@pytest.mark.parametrize("wide", [1, 0])
@pytest.mark.parametrize("leaf", [1, 8])
def test_bvh_instanced_single_leaf_geometries(oracle, capi, sensors, leaf, wide):
    """Instanced BVH path (16 geometries or fewer) where geometries that are ONE leaf -- a lone triangle; with leaf
    size 8 also a two-triangle quad and a 7-triangle soup -- do not come first in geomID order: a lane that leaves a
    bigger hierarchy with an empty stack enters them as `cur = leaf`, and must test that leaf with ITS records and
    transform, not the previous geometry's (at_entry was captured after the node branch)."""
    rng = np.random.default_rng(101)
    s = _syn_sensor(oracle, sensors, V=32, H=256)
    tr = make_tracer(capi, s, "bvh")
    tr.setOption(capi.LS_OPT_LEAF_SIZE, leaf)
    tr.setOption(capi.LS_OPT_BVH_WIDE, wide)       # the four-wide walk (round 6, default) and the binary one
    ml = []
    big_v, big_t = _random_soup(rng, 400, 9.0)
    # a wall of single triangles / small patches right around the sensor, so that most rays hit one of them
    def near_tri(k):
        a = 2.0 * np.pi * k / 5.0
        c = np.array([3.0 * np.cos(a), 3.0 * np.sin(a), 0.0], np.float32)
        v = (c + rng.normal(0, 1.6, (3, 3))).astype(np.float32)
        return v, np.array([[0, 1, 2]], np.uint32)
    quad_v = np.array([[2, -2, -1.5], [2, 2, -1.5], [2, 2, 1.5], [2, -2, 1.5]], np.float32)
    quad_t = np.array([[0, 1, 2], [0, 2, 3]], np.uint32)
    specs = [("big0", big_v, big_t), ("tri1", *near_tri(0)), ("tri2", *near_tri(1)), ("quad3", quad_v, quad_t),
             ("soup4", *_random_soup(rng, 7, 4.0)), ("tri5", *near_tri(2)), ("big6", *_random_soup(rng, 120, 7.0)),
             ("tri7", *near_tri(3))]
    for g, (name, v, t) in enumerate(specs):
        A = oracle.IDENTITY_AFFINE if g % 2 else oracle.affine_from_components(
            rng.normal(0, 0.5, 3).astype(np.float32), rng.normal(0, 0.4, 3).astype(np.float32))
        assert tr.addGeometry(name, v.shape[0], t.shape[0]) == g
        tr.updateGeometry(name, A, v, t)
        ml.append((g, v, t, A))
    assert tr.commitScene() == 0
    assert tr.info(capi.LS_INFO_BVH_INSTANCED) == 2
    rc, pts, hits = tr.traceScene(0)
    ref = _assert_parity(oracle, s, tr, ml, pts, hits)
    geoms_hit = set(ref["hits"][:, 1].tolist())
    assert {1, 2, 3, 5, 7} & geoms_hit, "the single-leaf geometries must actually be hit for the test to mean anything"
    assert len(geoms_hit) >= 5
    tr.close()


def test_engines_agree_full_size(oracle, capi, sensors):
    """BASELINE.json's headline workload (128 x 4096 rays over 1M triangles): the two engines and the
    CPU BVH oracle agree bit for bit on every ray, and on a 256-column sector the GPU exhaustive
    kernel (every ray x every triangle) agrees too."""
    from lidarshooter_amd import synth
    v, t = synth.syn_1m()
    s = _syn_sensor(oracle, sensors, V=128, H=4096)
    res = {}
    for eng in ("bvh", "projection"):
        tr = make_tracer(capi, s, eng)
        tr.addGeometry("g", v.shape[0], t.shape[0])
        tr.updateGeometry("g", oracle.IDENTITY_AFFINE, v, t)
        assert tr.commitScene() == 0
        rc, pts, hits = tr.traceScene(0)
        res[eng] = (tr.denseHits(), pts, hits)
        if eng == "projection":
            tr.setShard(1000, 256)
            tr.commitScene()
            tr.traceScene(1)
            st, sg = tr.denseHits()
            bt, bg = tr.bruteForce()
            assert np.array_equal(sg, bg) and np.array_equal(st, bt)
        tr.close()
    (tb, gb), pb, hb = res["bvh"]
    (tp, gp), pp, hp = res["projection"]
    assert np.array_equal(gb, gp) and np.array_equal(tb, tp)
    assert np.array_equal(pb, pp) and np.array_equal(hb, hp)
    ref = oracle.trace_frame(s, [(0, v, t, oracle.IDENTITY_AFFINE)], use_bvh=True, nthreads=16)
    assert np.array_equal(ref["gid"], gp) and np.array_equal(ref["t"], tp)
    assert np.array_equal(ref["points"], pp)
    n_hit = int((gp != oracle.INVALID).sum())
    assert 200000 < n_hit < 300000


def test_host_mirror_hiptracer(oracle, sensors, meshes):
    """The C++ host class (HipTracer : ITracer surface) end to end, reading the shipped JSON and STL
    files itself: the reference's TraceSceneCloud test (EmbreeTracer_test.cpp:122-135) and the
    two-mesh / remove sequence (OptixTracer_test.cpp:122-169, 217-311)."""
    from conftest import DATA
    from lidarshooter_amd import hostapi
    dev = hostapi.LidarDevice(os.path.join(DATA, "config", "hesai-pandar-XT-32-lidar_0000.json"))
    ground = hostapi.PolygonMesh(os.path.join(DATA, "mesh", "ground.stl"))
    ben = hostapi.PolygonMesh(os.path.join(DATA, "mesh", "ben.stl"))
    tr = hostapi.HipTracer(dev)
    gid = tr.addGeometry("mesh", ground.numPoints(), ground.numPolygons())
    assert gid == 0 and tr.getGeometryCount() == 1
    assert tr.getVertexCount("mesh") == 98 and tr.getElementCount("mesh") == 162
    assert tr.getGeometryId("mesh") == gid
    assert tr.getVertexCount("unknown") == -100            # TraceException (EmbreeTracer.cpp:369-439)
    assert tr.addGeometry("q", 4, 1, geometry_type=2) == 0  # an unsupported type: EmbreeTracer.cpp:200-201 returns false
    assert tr.updateGeometry("mesh", oracle.IDENTITY_AFFINE, ground) == 0
    assert tr.commitScene() == 0
    assert tr.traceScene(0) == 0
    c = tr.getTraceCloud()
    assert c["width"] * c["height"] == 1668 and c["point_step"] == 32 and c["seq"] == 0
    ref = oracle.trace_frame(sensors["0000"], [(0, *meshes["ground"], oracle.IDENTITY_AFFINE)])
    assert np.array_equal(c["data"], ref["points"])
    assert tr.addGeometry("face", ben.numPoints(), ben.numPolygons()) == 1
    tr.updateGeometry("mesh", oracle.IDENTITY_AFFINE, ground)
    tr.updateGeometryComponents("face", [0, 0, 0], [0, 0, 0], ben)
    assert tr.commitScene() == 0 and tr.traceScene(1) == 0
    c = tr.getTraceCloud()
    assert c["width"] == 1781 and c["seq"] == 1
    ref = oracle.trace_frame(sensors["0000"], [(0, *meshes["ground"], oracle.IDENTITY_AFFINE),
                                                (1, *meshes["ben"], oracle.IDENTITY_AFFINE)])
    assert np.array_equal(c["data"], ref["points"])
    h = tr.getHits()
    assert np.array_equal(np.stack([h["ray"], h["geom"], h["prim"], h["t"].view(np.uint32)], axis=1), ref["hits"])
    assert tr.removeGeometry("face") == 1 and tr.removeGeometry("mesh") == 0
    assert tr.commitScene() == -1 and tr.traceScene(2) == -1
    assert tr.getTraceCloud()["width"] == 0 and tr.getGeometryCount() == 0
    tr.close()


def test_trajectory_animated_frames(oracle, capi, sensors, meshes, engine):
    """BASELINE.json configs[4]'s moving instance: ben driven by config/trajectory.json through the
    reference's pose rule (AffineMesh.cpp:107-128), one updateGeometry(name, translation, rotation,
    mesh) + commit + trace per frame (MeshProjector.cpp:446-464); every frame equals the oracle."""
    from conftest import DATA
    s = sensors["0001"]
    poses = oracle.play_trajectory(os.path.join(DATA, "config", "trajectory.json"), 0.1)
    tr = make_tracer(capi, s, engine)
    _add(tr, "ground", meshes["ground"])
    _add(tr, "face", meshes["ben"])
    seen = set()
    for frame in (0, 1, 2, 3, 5, 8, 13, 40, 99, 100, 149):
        lin, ang = poses[frame, :3] * np.float32(0.05), poses[frame, 3:]     # keep ben inside the scene
        tr.updateGeometry("ground", oracle.IDENTITY_AFFINE, *meshes["ground"])
        tr.updateGeometryComponents("face", lin, ang, *meshes["ben"])
        assert tr.commitScene() == 0
        rc, pts, hits = tr.traceScene(frame)
        A = oracle.affine_from_components(lin, ang)
        _assert_parity(oracle, s, tr, [(0, *meshes["ground"], oracle.IDENTITY_AFFINE), (1, *meshes["ben"], A)], pts, hits)
        seen.add(len(pts))
    assert len(seen) > 3                                   # the cloud really changes from frame to frame
    tr.close()


def test_syn_10m_engines_agree(capi, oracle, sensors):
    """BASELINE.json configs[4] scene size: 9 998 244 triangles.  Both engines agree on every ray of
    the 128 x 4096 sensor; a 64-column sector is checked against the exhaustive GPU kernel."""
    from lidarshooter_amd import synth
    v, t = synth.syn_10m()
    assert t.shape[0] == 9998244
    s = _syn_sensor(oracle, sensors, V=128, H=4096)
    out = {}
    for eng in ("projection", "bvh"):
        tr = make_tracer(capi, s, eng)
        tr.addGeometry("g", v.shape[0], t.shape[0])
        tr.updateGeometry("g", oracle.IDENTITY_AFFINE, v, t)
        assert tr.commitScene() == 0
        rc, pts, hits = tr.traceScene(0)
        assert rc == 0
        out[eng] = (tr.denseHits(), pts, hits)
        if eng == "projection":
            tr.setShard(2040, 64)
            tr.commitScene()
            tr.traceScene(1)
            st, sg = tr.denseHits()
            bt, bg = tr.bruteForce()
            assert np.array_equal(sg, bg) and np.array_equal(st, bt)
        tr.close()
    (tp, gp), pp, hp = out["projection"]
    (tb, gb), pb, hb = out["bvh"]
    assert np.array_equal(gp, gb) and np.array_equal(tp, tb)
    assert np.array_equal(pp, pb) and np.array_equal(hp, hb)
    assert 200000 < len(pp) < 300000


def test_big_footprint_triangles(oracle, capi, sensors, engine):
    """Two triangles that each cover a large part of the raster (a 400 m ground quad under a 64 x 512
    sensor: > 8192 cells per triangle, the projection engine's big-footprint queue) plus a wall right
    in front of the sensor and small clutter."""
    s = _syn_sensor(oracle, sensors, V=64, H=512)
    s = oracle.Sensor(uid="big", vertical=s.vertical, h_begin=s.h_begin, h_end=s.h_end, h_count=s.h_count,
                      R=np.eye(3, dtype=np.float32).reshape(9), Rinv=np.eye(3, dtype=np.float32).reshape(9),
                      t=np.zeros(3, np.float32))
    quad = np.array([[-200, -200, -3], [200, -200, -3], [200, 200, -3], [-200, 200, -3]], np.float32)
    wall = np.array([[2, -30, -10], [2, 30, -10], [2, 0, 40]], np.float32)
    rng = np.random.default_rng(9)
    cv, ct = _random_soup(rng, 400, 6.0)
    v = np.concatenate([quad, wall, cv])
    t = np.concatenate([np.array([[0, 1, 2], [0, 2, 3], [4, 5, 6]], np.uint32), ct + np.uint32(7)])
    ml = [(0, v, t, oracle.IDENTITY_AFFINE)]
    ref = oracle.trace_frame(s, ml)
    assert int((ref["gid"] != oracle.INVALID).sum()) > 15000
    assert int((ref["gid"] < 3).sum()) > 8000              # the three big triangles own most of the hits
    tr = make_tracer(capi, s, engine)
    tr.addGeometry("m", v.shape[0], t.shape[0])
    tr.updateGeometry("m", oracle.IDENTITY_AFFINE, v, t)
    assert tr.commitScene() == 0
    for frame in range(3):                                  # the queue must re-arm between frames
        rc, pts, hits = tr.traceScene(frame)
        _assert_parity(oracle, s, tr, ml, pts, hits)
    tr.close()


def test_big_queue_chunks_and_overflow(oracle, capi, sensors, engine):
    """2600 triangles that each cover hundreds of raster cells: more than one 1024-entry cull chunk of
    the projection engine's big-footprint queue and more than its 2048 entries (the rest is expanded
    by the streaming kernel itself); three frames, so the queue re-arms with a full load."""
    s = _syn_sensor(oracle, sensors, V=64, H=512)
    s = oracle.Sensor(uid="bigq", vertical=s.vertical, h_begin=s.h_begin, h_end=s.h_end, h_count=s.h_count,
                      R=np.eye(3, dtype=np.float32).reshape(9), Rinv=np.eye(3, dtype=np.float32).reshape(9),
                      t=np.zeros(3, np.float32))
    rng = np.random.default_rng(2600)
    n = 2600
    c = rng.normal(0.0, 1.0, size=(n, 1, 3))
    c[:, :, 2] *= 0.2                                       # around the horizon, where the channels are
    dist = rng.uniform(8.0, 30.0, size=(n, 1, 1))
    c = c / np.linalg.norm(c, axis=2, keepdims=True) * dist
    tri = c + rng.normal(0.0, 1.0, size=(n, 3, 3)) * (0.12 * dist)   # ~14 degrees across: hundreds of cells, never near the origin
    v = tri.reshape(-1, 3).astype(np.float32)
    t = np.arange(3 * n, dtype=np.uint32).reshape(n, 3)
    ml = [(0, v, t, oracle.IDENTITY_AFFINE)]
    ref = oracle.trace_frame(s, ml)
    assert int((ref["gid"] != oracle.INVALID).sum()) > 20000
    assert len(np.unique(ref["gid"])) > 300                 # many different big triangles win somewhere
    tr = make_tracer(capi, s, engine)
    tr.addGeometry("m", v.shape[0], t.shape[0])
    tr.updateGeometry("m", oracle.IDENTITY_AFFINE, v, t)
    assert tr.commitScene() == 0
    for frame in range(3):
        rc, pts, hits = tr.traceScene(frame)
        _assert_parity(oracle, s, tr, ml, pts, hits)
    if engine == "projection":
        tr.setOption(capi.LS_OPT_COUNT_VISITS, 1)
        tr.traceScene(3)
        assert tr.visitStats()[1] > 128 * 2048              # the queue was full: every entry has > 128 cells
    tr.close()


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("LS_STRESS_SEEDS", "12")))))   # more seeds: LS_STRESS_SEEDS=300
def test_projection_footprints_never_lose_a_hit(oracle, capi, sensors, seed):
    """Randomised stress of the projection engine's conservative footprints against the exhaustive
    GPU kernel: random sensor rasters (channel sets, azimuth ranges, sweep direction, shard), random
    poses, triangle soups over six orders of magnitude in size and distance."""
    rng = np.random.default_rng(1000 + seed)
    V = int(rng.integers(1, 48))
    H = int(rng.integers(2, 400))
    vert = np.sort(rng.uniform(-89.0, 89.0, size=V))[::-1] if seed % 3 else rng.uniform(-60.0, 60.0, size=V)
    begin = float(rng.uniform(-360.0, 360.0))
    span = float(rng.choice([360.0, -360.0, rng.uniform(5.0, 359.0), -rng.uniform(5.0, 359.0), 720.0]))
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    R, Rinv = oracle.pose_from_quat(*q)
    s = oracle.Sensor(uid="rnd", vertical=vert.astype(np.float32), h_begin=np.float32(begin), h_end=np.float32(begin + span),
                      h_count=H, R=R, Rinv=Rinv, t=rng.normal(0.0, 3.0, size=3).astype(np.float32))
    scale = float(np.exp(rng.uniform(np.log(0.05), np.log(500.0))))
    v, t = _random_soup(rng, 4000, scale)
    tr = make_tracer(capi, s, "projection")
    if H > 8 and seed % 2:
        first = int(rng.integers(0, H - 4))
        tr.setShard(first, int(rng.integers(1, H - first + 1)))
    tr.addGeometry("soup", v.shape[0], t.shape[0])
    A = oracle.affine_from_components(rng.normal(0, scale, size=3), rng.uniform(-3, 3, size=3))
    tr.updateGeometry("soup", A, v, t)
    assert tr.commitScene() == 0
    rc, pts, hits = tr.traceScene(0)
    pt, pg = tr.denseHits()
    bt, bg = tr.bruteForce()
    assert np.array_equal(pg, bg) and np.array_equal(pt, bt)
    assert len(pts) == int((bg != oracle.INVALID).sum())
    tr.close()


def test_expand_gathered_hits(oracle, capi, sensors, meshes):
    """The receiving side of the multi-GPU all-gather: per-rank slots [count | hit records] are
    compacted into one cloud and the 32-byte points rebuilt from (ray, t) -- equals the full frame."""
    import torch
    from lidarshooter_amd import shards
    s = sensors["0000"]
    ml = [(0, *meshes["ground"], oracle.IDENTITY_AFFINE), (1, *meshes["ben"], oracle.IDENTITY_AFFINE)]
    ref = oracle.trace_frame(s, ml)
    world = 3
    cap = shards.slot_capacity(s.V, s.H, world)
    sb = shards.slot_bytes(cap)
    gathered = torch.zeros(world * sb, dtype=torch.uint8, device="cuda:0")
    scratch_pts = torch.zeros(32 * cap, dtype=torch.uint8, device="cuda:0")
    trs = []
    for rank in range(world):                              # each "rank" traces its sector straight into its slot
        tr = make_tracer(capi, s, "projection")
        first, n = shards.shard_columns(s.H, world, rank)
        tr.setShard(first, n)
        _add(tr, "ground", meshes["ground"])
        _add(tr, "face", meshes["ben"])
        tr.updateGeometry("ground", oracle.IDENTITY_AFFINE, *meshes["ground"])
        tr.updateGeometry("face", oracle.IDENTITY_AFFINE, *meshes["ben"])
        assert tr.commitScene() == 0
        base = gathered.data_ptr() + rank * sb
        tr.setOutputBuffers(scratch_pts.data_ptr(), base + shards.HEADER, base, cap)
        tr.traceSceneAsync(0)
        tr.synchronize()
        trs.append(tr)
    pts = torch.zeros(32 * cap * world, dtype=torch.uint8, device="cuda:0")
    hts = torch.zeros(16 * cap * world, dtype=torch.uint8, device="cuda:0")
    n = torch.zeros(4, dtype=torch.int32, device="cuda:0")
    trs[0].expandGatheredHits(gathered.data_ptr(), world, cap, pts.data_ptr(), hts.data_ptr(), n.data_ptr())
    trs[0].synchronize()
    k = int(n[0].item())
    assert k == 1781
    hits = hts.cpu().numpy()[:16 * k].view(np.uint32).reshape(k, 4)
    points = pts.cpu().numpy()[:32 * k].reshape(k, 32)
    order = np.argsort(hits[:, 0], kind="stable")
    assert np.array_equal(hits[order], ref["hits"]) and np.array_equal(points[order], ref["points"])
    assert np.array_equal(shards.decode_gathered(gathered.cpu().numpy(), world, cap).view(np.uint32).reshape(-1, 4), hits)
    for tr in trs:
        tr.close()


def test_cloud_to_world_two_sensors_merged(oracle, capi, sensors, meshes):
    """the traced clouds of lidar_0000 and lidar_0001 go from their sensor frames to
    the world frame on the device (CloudTransformer::applyInverseTransform + originToSensorInverse)
    and land back to back in one buffer; bytes equal the oracle's, in place and with a mesh-style affine."""
    import torch
    ml = [(0, *meshes["ground"], oracle.IDENTITY_AFFINE), (1, *meshes["ben"], oracle.IDENTITY_AFFINE)]
    cap = 2 * 4800
    merged = torch.zeros(32 * cap, dtype=torch.uint8, device="cuda:0")
    counts = torch.zeros(8, dtype=torch.int32, device="cuda:0")      # [total after sensor 0, total after sensor 1]
    expect, trs = [], []
    for k, uid in enumerate(("0000", "0001")):
        s = sensors[uid]
        tr = make_tracer(capi, s, "projection")
        _add(tr, "ground", meshes["ground"])
        _add(tr, "face", meshes["ben"])
        tr.updateGeometry("ground", oracle.IDENTITY_AFFINE, *meshes["ground"])
        tr.updateGeometry("face", oracle.IDENTITY_AFFINE, *meshes["ben"])
        assert tr.commitScene() == 0
        f = tr.traceSceneAsync(0)
        base = None if k == 0 else counts.data_ptr()
        tr.cloudToWorld(s.R, f.d_points32, f.d_n_points, merged.data_ptr(), cap, d_out_base=base,
                        d_out_total=counts.data_ptr() + 4 * k)
        tr.synchronize()
        expect.append(oracle.cloud_to_world(oracle.trace_frame(s, ml)["points"], s))
        trs.append(tr)
    n0, n1 = int(counts[0].item()), int(counts[1].item())
    assert (n0, n1 - n0) == (1781, 1769)
    got = merged.cpu().numpy()[:32 * n1].reshape(n1, 32)
    assert np.array_equal(got[:n0], expect[0]) and np.array_equal(got[n0:], expect[1])
    # the world-frame ground points are on the ground plane again (ground.stl is z = 0)
    z = got[:, 8:12].copy().view(np.float32).reshape(-1)
    assert np.abs(z).min() < 1e-4
    # in place on caller-owned buffers, with an affine in front (CloudTransformer's _transform) and a
    # capacity smaller than the cloud: the first 1000 records move, the others stay
    s, tr = sensors["0000"], trs[0]
    A = oracle.affine_from_components(np.array([1.5, -2.0, 0.25], np.float32), np.array([0.1, -0.2, 0.3], np.float32))
    pts = torch.zeros(32 * 4800, dtype=torch.uint8, device="cuda:0")
    hts = torch.zeros(16 * 4800, dtype=torch.uint8, device="cuda:0")
    cnt = torch.zeros(4, dtype=torch.int32, device="cuda:0")
    tr.setOutputBuffers(pts.data_ptr(), hts.data_ptr(), cnt.data_ptr(), 4800)
    tr.traceSceneAsync(1)
    tr.cloudToWorld(s.R, pts.data_ptr(), cnt.data_ptr(), pts.data_ptr(), 1000, affine=A, d_out_total=cnt.data_ptr() + 4)
    tr.synchronize()
    local = oracle.trace_frame(s, ml)["points"]
    want = oracle.cloud_to_world(local, s, A)
    got = pts.cpu().numpy()[:32 * 1781].reshape(1781, 32)
    assert cnt[:2].tolist() == [1781, 1000]
    assert np.array_equal(got[:1000], want[:1000]) and np.array_equal(got[1000:], local[1000:])
    for tr in trs:
        tr.close()


@pytest.mark.parametrize("mode", [1, 2, 12])
def test_pipelined_frames(oracle, capi, sensors, meshes, mode):
    """LS_OPT_PIPELINE = 1 (two frames in flight: the finish + pack workgroups of frame i ride in the launch
    of frame i+1) and = 2 (three frames in flight on three streams).  An animated scene (the moving mesh is
    copied into library-owned buffers every frame), caller-owned output buffers rotating over three sets;
    every frame is read once it is ordered on the handle's stream (after `mode` further calls, or the
    final flush) and equals the oracle; then the synchronous API and a shard change on the same handle."""
    import torch
    from conftest import DATA
    s = sensors["0001"]
    poses = oracle.play_trajectory(os.path.join(DATA, "config", "trajectory.json"), 0.1)
    tr = make_tracer(capi, s, "projection")
    graph = mode == 12                                          # three streams, every frame one graph launch (LS_OPT_FRAME_GRAPH)
    mode = 2 if graph else mode
    tr.setOption(capi.LS_OPT_PIPELINE, mode)
    tr.setOption(capi.LS_OPT_FRAME_GRAPH, 1 if graph else 0)
    _add(tr, "ground", meshes["ground"])
    _add(tr, "face", meshes["ben"])
    cap = s.V * s.H
    bufs = [(torch.zeros(32 * cap, dtype=torch.uint8, device="cuda:0"), torch.zeros(16 * cap, dtype=torch.uint8, device="cuda:0"),
             torch.zeros(4, dtype=torch.int32, device="cuda:0")) for _ in range(3)]
    frames = list(range(0, 24))
    expected = []

    def check(i):
        pts_t, hts_t, n_t = bufs[i % 3]
        n = int(n_t[0].item())
        ref = expected[i]
        assert n == ref["points"].shape[0]
        assert np.array_equal(pts_t.cpu().numpy()[:32 * n].reshape(n, 32), ref["points"])
        assert np.array_equal(hts_t.cpu().numpy()[:16 * n].view(np.uint32).reshape(n, 4), ref["hits"])

    stream = torch.cuda.Stream()
    tr.setStream(stream.cuda_stream)
    for i in frames:
        lin, ang = poses[4 * i, :3] * np.float32(0.05), poses[4 * i, 3:]
        A = oracle.affine_from_components(lin, ang)
        expected.append(oracle.trace_frame(s, [(0, *meshes["ground"], oracle.IDENTITY_AFFINE), (1, *meshes["ben"], A)]))
        tr.updateGeometry("ground", oracle.IDENTITY_AFFINE, *meshes["ground"])
        tr.updateGeometryComponents("face", lin, ang, *meshes["ben"])
        assert tr.commitScene() == 0
        p, h, n = bufs[i % 3]
        tr.setOutputBuffers(p.data_ptr(), h.data_ptr(), n.data_ptr(), cap)
        tr.traceSceneAsync(i)
        if mode == 1 and i:                                 # frame i-1 is ordered on the handle's stream now
            stream.synchronize()
            check(i - 1)
        if mode == 2 and i >= 2 and i % 3 == 2:             # three streams: a flush orders everything issued so far
            tr.flush()
            stream.synchronize()
            for k in (i - 2, i - 1, i):
                check(k)
    tr.flush()
    stream.synchronize()
    for i in frames[-mode:]:
        check(i)
    assert len({e["points"].shape[0] for e in expected}) > 3
    if graph:   # three graphs captured (one per stream of the rotation), every other frame a replay with the pose patched in
        assert tr.info(capi.LS_INFO_FRAME_GRAPH_STATE) == 1
        assert tr.info(capi.LS_INFO_FRAME_GRAPH_CAPTURES) == 3
        assert tr.info(capi.LS_INFO_FRAME_GRAPH_REPLAYS) == len(frames)
        assert tr.info(capi.LS_INFO_FRAME_GRAPH_PATCHES) >= len(frames) - 3
    # synchronous API on the pipelined handle, library-owned (twin) buffers, twice: both twins
    tr.setOutputBuffers(None, None, None, 0)
    ml = [(0, *meshes["ground"], oracle.IDENTITY_AFFINE), (1, *meshes["ben"], A)]
    for k in range(3):
        rc, pts, hits = tr.traceScene(100 + k)
        _assert_parity(oracle, s, tr, ml, pts, hits)
    # a shard change with frames in flight
    tr.traceSceneAsync(200)
    tr.setShard(10, 40)
    assert tr.commitScene() == 0
    rc, pts, hits = tr.traceScene(201)
    ref = expected[-1]
    cols = ref["hits"][:, 0] % s.H
    keep = (cols >= 10) & (cols < 50)
    assert np.array_equal(np.stack([hits["ray"], hits["geom"], hits["prim"], hits["t"].view(np.uint32)], axis=1), ref["hits"][keep])
    tr.close()


@pytest.mark.parametrize("mode", [1, 2, 12])
def test_pipeline_stress(oracle, capi, sensors, meshes, mode):
    """900 frames streamed without a host wait in either pipelined mode, the moving mesh cycling through
    four poses (in-place device meshes, transform only), outputs rotating over three caller-owned sets;
    every 150 frames everything is flushed and the three frames last issued are compared with the oracle."""
    import torch
    s = sensors["0000"]
    tr = make_tracer(capi, s, "projection")
    graph = mode == 12                                          # (12: three streams + LS_OPT_FRAME_GRAPH)
    mode = 2 if graph else mode
    tr.setOption(capi.LS_OPT_PIPELINE, mode)
    tr.setOption(capi.LS_OPT_FRAME_GRAPH, 1 if graph else 0)
    dev = torch.device("cuda", 0)
    d = {}
    for name, key in (("ground", "ground"), ("face", "ben")):
        v, t = meshes[key]
        d[name] = (torch.from_numpy(np.ascontiguousarray(v, np.float32)).to(dev),
                   torch.from_numpy(np.ascontiguousarray(t, np.uint32).view(np.int32)).to(dev))
        tr.addGeometry(name, v.shape[0], t.shape[0])
    poses = [oracle.affine_from_components(np.array(l, np.float32), np.array(a, np.float32))
             for l, a in (((0, 0, 0), (0, 0, 0)), ((1.5, -1.0, 0.2), (0.1, 0.0, 0.7)), ((-2.0, 0.5, 0.0), (0.0, 0.3, -1.1)),
                          ((0.3, 2.5, 0.4), (-0.2, 0.1, 2.0)))]
    refs = [oracle.trace_frame(s, [(0, *meshes["ground"], oracle.IDENTITY_AFFINE), (1, *meshes["ben"], A)]) for A in poses]
    assert len({r["points"].shape[0] for r in refs}) >= 3
    cap = s.V * s.H
    bufs = [(torch.zeros(32 * cap, dtype=torch.uint8, device=dev), torch.zeros(16 * cap, dtype=torch.uint8, device=dev),
             torch.zeros(4, dtype=torch.int32, device=dev)) for _ in range(3)]
    for i in range(900):
        tr.updateGeometryDeviceShared("ground", oracle.IDENTITY_AFFINE, d["ground"][0].data_ptr(), 12, d["ground"][1].data_ptr())
        tr.updateGeometryDeviceShared("face", poses[(i * 7) % 4], d["face"][0].data_ptr(), 12, d["face"][1].data_ptr())
        assert tr.commitScene() == 0
        p, h, n = bufs[i % 3]
        tr.setOutputBuffers(p.data_ptr(), h.data_ptr(), n.data_ptr(), cap)
        tr.traceSceneAsync(i)
        if i % 150 == 149:
            tr.synchronize()
            for k in (i - 2, i - 1, i):
                p, h, n = bufs[k % 3]
                ref = refs[(k * 7) % 4]
                cnt = int(n[0].item())
                assert cnt == ref["points"].shape[0], (k, cnt)
                assert np.array_equal(p.cpu().numpy()[:32 * cnt].reshape(cnt, 32), ref["points"])
                assert np.array_equal(h.cpu().numpy()[:16 * cnt].view(np.uint32).reshape(cnt, 4), ref["hits"])
    if graph:
        assert tr.info(capi.LS_INFO_FRAME_GRAPH_CAPTURES) == 3 and tr.info(capi.LS_INFO_FRAME_GRAPH_REPLAYS) == 900
    tr.close()


def test_frame_graph_follows_the_scene(oracle, capi, sensors, meshes):
    """LS_OPT_FRAME_GRAPH: what a cached frame graph must survive.  Same poses again: replays with NOTHING patched;
    a new pose: one node patched (k_project's arguments); another geometry set (a mesh added, later removed): the launch
    sequence may change, the graphs are captured anew; the option switched off and on again: plain launches in between.
    Every frame equals the oracle's."""
    import torch
    s = sensors["0000"]
    tr = make_tracer(capi, s, "projection")
    tr.setOption(capi.LS_OPT_PIPELINE, 2)
    if tr.info(capi.LS_INFO_PIPELINE_MODE) != 2:
        pytest.skip("fewer than three concurrent streams on this device")
    tr.setOption(capi.LS_OPT_FRAME_GRAPH, 1)
    _add(tr, "ground", meshes["ground"])
    tr.updateGeometry("ground", oracle.IDENTITY_AFFINE, *meshes["ground"])
    cap = s.V * s.H
    bufs = [(torch.zeros(32 * cap, dtype=torch.uint8, device="cuda:0"), torch.zeros(16 * cap, dtype=torch.uint8, device="cuda:0"),
             torch.zeros(4, dtype=torch.int32, device="cuda:0")) for _ in range(3)]
    issued = []

    def frame(ml):
        i = len(issued)
        p, h, n = bufs[i % 3]
        assert tr.commitScene() == 0
        tr.setOutputBuffers(p.data_ptr(), h.data_ptr(), n.data_ptr(), cap)
        tr.traceSceneAsync(i)
        issued.append(ml)

    def check_last_three():
        tr.synchronize()
        for k in range(len(issued) - 3, len(issued)):
            p, h, n = bufs[k % 3]
            ref = oracle.trace_frame(s, issued[k])
            cnt = int(n[0].item())
            assert cnt == ref["points"].shape[0], (k, cnt)
            assert np.array_equal(p.cpu().numpy()[:32 * cnt].reshape(cnt, 32), ref["points"])
            assert np.array_equal(h.cpu().numpy()[:16 * cnt].view(np.uint32).reshape(cnt, 4), ref["hits"])

    info = lambda: (tr.info(capi.LS_INFO_FRAME_GRAPH_CAPTURES), tr.info(capi.LS_INFO_FRAME_GRAPH_REPLAYS), tr.info(capi.LS_INFO_FRAME_GRAPH_PATCHES))
    g_only = [(0, *meshes["ground"], oracle.IDENTITY_AFFINE)]
    for _ in range(6):
        frame(g_only)
    check_last_three()
    assert info() == (3, 6, 0)                                   # three captures, three pure replays
    A = oracle.affine_from_components(np.array((1.0, 0.5, 0.1), np.float32), np.array((0.0, 0.0, 0.3), np.float32))
    tr.updateGeometryTransform("ground", A)
    moved = [(0, *meshes["ground"], A)]
    for _ in range(3):
        frame(moved)
    check_last_three()
    assert info() == (3, 9, 3), (info(), bin(tr.info(capi.LS_INFO_FRAME_GRAPH_LAST_PATCHED)))   # the pose went into each of the three graphs once
    for _ in range(3):
        frame(moved)
    assert info() == (3, 12, 3)
    _add(tr, "face", meshes["ben"])                              # another geometry set
    tr.updateGeometry("face", oracle.IDENTITY_AFFINE, *meshes["ben"])
    both = moved + [(1, *meshes["ben"], oracle.IDENTITY_AFFINE)]
    for _ in range(4):
        frame(both)
    check_last_three()
    c, r, _ = info()
    assert r == 16 and c in (3, 6)                               # (two geometries still fit one launch: the sequence may be the same)
    assert tr.removeGeometry("ground") == 0
    face_only = [(1, *meshes["ben"], oracle.IDENTITY_AFFINE)]
    for _ in range(3):
        frame(face_only)
    check_last_three()
    tr.setOption(capi.LS_OPT_FRAME_GRAPH, 0)
    r0 = info()[1]
    for _ in range(3):
        frame(face_only)
    check_last_three()
    assert info()[1] == r0                                       # plain launches
    tr.setOption(capi.LS_OPT_FRAME_GRAPH, 1)
    for _ in range(3):
        frame(face_only)
    check_last_three()
    assert info()[1] == r0 + 3 and tr.info(capi.LS_INFO_FRAME_GRAPH_STATE) == 1
    # the synchronous call on the same handle (it closes its graph itself and waits)
    tr.setOutputBuffers(None, None, None, 0)
    rc, pts, hits = tr.traceScene(99)
    _assert_parity(oracle, s, tr, face_only, pts, hits)
    tr.close()


@pytest.mark.parametrize("uid,expected,mode", [("0000", 1781, 0), ("0000", 1781, 1), ("0001", 1769, 2)])
def test_lsbench_cpp_harness(oracle, sensors, meshes, uid, expected, mode):
    """The C++ streaming harness (lidarshooter_amd/host/lsbench.cpp: host mirror + C ABI + /opt/rocm's HIP
    runtime, no Python or PyTorch in the process) on the reference's XT-32 scene, in every frame mode: the
    last of 300 streamed frames has the reference's known point count (OptixTracer_test.cpp:122-169)."""
    import json
    import subprocess
    from conftest import DATA
    exe = os.path.join(os.path.dirname(DATA), "..", "..", "lidarshooter_amd", "lsbench")
    exe = os.path.normpath(exe)
    assert os.path.exists(exe), "lsbench not built (python -c 'import __graft_entry__ as g; g.build()')"
    out = subprocess.run([exe, "--config", os.path.join(DATA, "config", f"hesai-pandar-XT-32-lidar_{uid}.json"),
                          "--mesh", "ground=" + os.path.join(DATA, "mesh", "ground.stl"),
                          "--mesh", "face=" + os.path.join(DATA, "mesh", "ben.stl"),
                          "--frames", "300", "--warmup", "20", "--pipeline", str(mode)],
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    rec = json.loads(out.stdout.strip().splitlines()[-1])
    assert rec["rays_per_frame"] == 4800 and rec["triangles"] == 162 + 5489
    assert rec["points_last_frame"] == expected
    # the bytes of the last streamed frame are the oracle's, not just their count
    ref = oracle.trace_frame(sensors[uid], [(0, *meshes["ground"], oracle.IDENTITY_AFFINE), (1, *meshes["ben"], oracle.IDENTITY_AFFINE)])
    assert rec["points_sha256"] == hashlib.sha256(ref["points"].tobytes()).hexdigest()
    assert rec["hits_sha256"] == hashlib.sha256(np.ascontiguousarray(ref["hits"], np.uint32).tobytes()).hexdigest()


def test_lsbench_raw_mesh_is_the_python_mesh(oracle, sensors, tmp_path):
    """lsbench --mesh-raw: the C++ harness traces a mesh of lidarshooter_amd/synth.py bit for bit (tools/dump_mesh.py writes
    it out; numpy's generator cannot be reproduced from C++) -- its cloud hashes to the oracle's for the same arrays, which is
    what makes lsbench's SYN-1M numbers comparable with bench.py's (--grid is a look-alike)."""
    import json
    import subprocess
    import sys
    from conftest import DATA, ROOT
    from lidarshooter_amd import synth
    raw = str(tmp_path / "grid.lsmesh")
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dump_mesh.py"), "grid:400x300", raw], check=True, capture_output=True)
    exe = os.path.join(ROOT, "lidarshooter_amd", "lsbench")
    out = subprocess.run([exe, "--config", os.path.join(DATA, "config", "hesai-pandar-XT-32-lidar_0001.json"), "--syn", "64", "512",
                          "--mesh-raw", "grid=" + raw, "--frames", "40", "--warmup", "5", "--pipeline", "2"],
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    rec = json.loads(out.stdout.strip().splitlines()[-1])
    v, t = synth.grid_mesh(400, 300)
    base = sensors["0001"]
    s = oracle.Sensor(uid="syn", vertical=synth.syn_vertical(64), h_begin=np.float32(0.0), h_end=np.float32(360.0), h_count=512,
                      R=base.R, Rinv=base.Rinv, t=base.t)
    ref = oracle.trace_frame(s, [(0, v, t, oracle.IDENTITY_AFFINE)], use_bvh=True)
    assert rec["triangles"] == t.shape[0] and rec["points_last_frame"] == ref["points"].shape[0]
    assert rec["points_sha256"] == hashlib.sha256(ref["points"].tobytes()).hexdigest()


def test_edge_cases(oracle, capi, sensors, engine):
    """Degenerate inputs: one-triangle scene (a BVH with no internal node), coincident triangles in two
    geometries (equal t: lowest geomID wins), a geometry with zero triangles next to a real one, a
    degenerate (zero-area) triangle, a one-column shard, a two-column one-channel sensor."""
    s0 = sensors["0000"]
    d = oracle.ray_dirs(s0)[31 * s0.H + 10].astype(np.float64)          # a downward ray of lidar_0000
    c = d * 20.0
    u = np.cross(d, [0, 0, 1.0]); u /= np.linalg.norm(u)
    w = np.cross(d, u)
    tri_sensor = np.array([c + 4 * u, c - 2 * u + 4 * w, c - 2 * u - 4 * w], np.float64)
    # put it in the world frame so that the library's transform brings it back: v_world = R * v_sensor + t
    R = s0.R.reshape(3, 3).astype(np.float64)
    tri = (tri_sensor @ R.T + s0.t.astype(np.float64)).astype(np.float32)
    idx = np.array([[0, 1, 2]], np.uint32)

    # one triangle
    tr = make_tracer(capi, s0, engine)
    tr.addGeometry("one", 3, 1)
    tr.updateGeometry("one", oracle.IDENTITY_AFFINE, tri, idx)
    assert tr.commitScene() == 0
    rc, pts, hits = tr.traceScene(0)
    ref = _assert_parity(oracle, s0, tr, [(0, tri, idx, oracle.IDENTITY_AFFINE)], pts, hits)
    assert len(pts) > 0
    # the same triangle again as a second geometry: every hit keeps geomID 0
    tr.addGeometry("two", 3, 1)
    tr.updateGeometry("one", oracle.IDENTITY_AFFINE, tri, idx)
    tr.updateGeometry("two", oracle.IDENTITY_AFFINE, tri, idx)
    assert tr.commitScene() == 0
    rc, pts2, hits2 = tr.traceScene(1)
    assert len(pts2) == len(pts) and np.all(hits2["geom"] == 0)
    _assert_parity(oracle, s0, tr, [(0, tri, idx, oracle.IDENTITY_AFFINE), (1, tri, idx, oracle.IDENTITY_AFFINE)], pts2, hits2)
    # remove geometry 0: now geomID 1 owns the hits
    assert tr.removeGeometry("one") == 0
    tr.updateGeometry("two", oracle.IDENTITY_AFFINE, tri, idx)
    assert tr.commitScene() == 0
    rc, pts3, hits3 = tr.traceScene(2)
    assert len(pts3) == len(pts) and np.all(hits3["geom"] == 1)
    # an empty geometry (0 vertices, 0 triangles) and a degenerate triangle change nothing
    tr.addGeometry("empty", 0, 0)
    tr.updateGeometry("empty", oracle.IDENTITY_AFFINE, np.zeros((0, 3), np.float32), np.zeros((0, 3), np.uint32))
    deg = np.array([tri[0], tri[0], tri[1]], np.float32)
    tr.addGeometry("degenerate", 3, 1)
    tr.updateGeometry("degenerate", oracle.IDENTITY_AFFINE, deg, idx)
    tr.updateGeometry("two", oracle.IDENTITY_AFFINE, tri, idx)
    assert tr.commitScene() == 0
    rc, pts4, hits4 = tr.traceScene(3)
    assert np.array_equal(pts4, pts3) and np.array_equal(hits4, hits3)
    # one-column shard
    tr.setShard(10, 1)
    assert tr.getTotalRays() == s0.V
    assert tr.commitScene() == 0
    rc, pts5, hits5 = tr.traceScene(4)
    assert np.array_equal(hits5["ray"] % s0.H, np.full(len(hits5), 10))
    assert len(pts5) == int((ref["hits"][:, 0] % s0.H == 10).sum())
    tr.close()

    # smallest raster the reference's formulas allow: 1 channel x 2 columns (step = end - begin)
    s1 = oracle.Sensor(uid="tiny", vertical=np.array([-30.0], np.float32), h_begin=np.float32(10.0), h_end=np.float32(50.0),
                       h_count=2, R=s0.R, Rinv=s0.Rinv, t=s0.t)
    from lidarshooter_amd import synth
    v, t = synth.grid_mesh(20, 20, half=40.0, seed=2)
    tr = make_tracer(capi, s1, engine)
    tr.addGeometry("g", v.shape[0], t.shape[0])
    tr.updateGeometry("g", oracle.IDENTITY_AFFINE, v, t)
    assert tr.commitScene() == 0
    rc, pts, hits = tr.traceScene(0)
    _assert_parity(oracle, s1, tr, [(0, v, t, oracle.IDENTITY_AFFINE)], pts, hits)
    tr.close()


def test_out_of_range_indices_are_refused(oracle, capi, sensors, meshes, engine):
    """A triangle that names a vertex its geometry does not have.  Embree reads it out of the shared host buffers unchecked
    (EmbreeTracer.cpp:140-176: undefined, usually survivable); every kernel here that gathers vertices would take a device
    memory fault.  So the upload is looked at (one max-reduction behind every index upload / hand-over) and the commit is
    REFUSED -- LS_ERR_OUT_OF_RANGE, nothing is built or launched over the mesh, a trace says "nothing committed" -- and the
    handle stays usable: new indices or the removal of the geometry clear the condition.  (The fault itself is never
    provoked: every case here must end in the error path.)"""
    import torch
    s = sensors["0000"]
    v, t = meshes["ground"]
    tr = make_tracer(capi, s, engine)
    assert tr.addGeometry("ground", v.shape[0], t.shape[0]) == 0
    for bad_value in (v.shape[0], v.shape[0] + 7, 0xFFFFFFF0):
        bad = t.copy()
        bad[50, 1] = bad_value
        assert tr.updateGeometry("ground", oracle.IDENTITY_AFFINE, v, bad) == 0     # (the upload itself is asynchronous)
        for _ in range(2):                                                           # the condition stays until new indices arrive
            with pytest.raises(capi.LidarShooterHipError, match="vertex index %d" % bad_value):
                tr.commitScene()
        rc, pts, hits = tr.traceScene(0)
        assert rc == -1 and len(pts) == 0                                            # nothing committed: nothing traced
    # a second, good geometry does not make the scene acceptable; removing the bad one does (ls_remove_geometry commits)
    bv, bt = meshes["ben"]
    assert tr.addGeometry("face", bv.shape[0], bt.shape[0]) == 1
    tr.updateGeometry("face", oracle.IDENTITY_AFFINE, bv, bt)
    with pytest.raises(capi.LidarShooterHipError, match="'ground'"):
        tr.commitScene()
    assert tr.removeGeometry("ground") == 0
    assert tr.commitScene() == 0
    rc, pts, hits = tr.traceScene(1)
    _assert_parity(oracle, s, tr, [(1, bv, bt, oracle.IDENTITY_AFFINE)], pts, hits)
    # the mesh again with its own indices: the reference's 1781 points
    assert tr.addGeometry("ground", v.shape[0], t.shape[0]) == 0
    tr.updateGeometry("ground", oracle.IDENTITY_AFFINE, v, t)
    assert tr.commitScene() == 0
    rc, pts, hits = tr.traceScene(2)
    assert len(pts) == 1781
    # the hand-over path (the caller's device buffers are read in place: every hand-over is looked at)
    dev = torch.device("cuda", 0)
    dv = torch.from_numpy(v).to(dev)
    bad = t.copy()
    bad[0, 0] = 1 << 20
    d_bad, d_good = torch.from_numpy(bad.view(np.int32)).to(dev), torch.from_numpy(t.view(np.int32)).to(dev)
    tr.updateGeometryDeviceShared("ground", oracle.IDENTITY_AFFINE, dv.data_ptr(), 12, d_bad.data_ptr())
    with pytest.raises(capi.LidarShooterHipError, match="vertex index 1048576"):
        tr.commitScene()
    tr.updateGeometryDeviceShared("ground", oracle.IDENTITY_AFFINE, dv.data_ptr(), 12, d_good.data_ptr())
    assert tr.commitScene() == 0
    rc, pts, hits = tr.traceScene(3)
    assert len(pts) == 1781
    # quads: the four indices of every element are looked at (through the triangle pair they become)
    quad_v = np.array([[30, -30, -2], [30, 30, -2], [-30, 30, -2], [-30, -30, -2]], np.float32)
    assert tr.addGeometry("plate", 4, 1, geometry_type=capi.LS_GEOMETRY_TYPE_QUAD) == 2
    tr.updateGeometry("plate", oracle.IDENTITY_AFFINE, quad_v, np.array([[0, 1, 2, 4]], np.uint32))
    with pytest.raises(capi.LidarShooterHipError, match="'plate'"):
        tr.commitScene()
    tr.updateGeometry("plate", oracle.IDENTITY_AFFINE, quad_v, np.array([[0, 1, 2, 3]], np.uint32))
    assert tr.commitScene() == 0
    tr.close()


def test_non_finite_and_huge_vertices(oracle, capi, sensors, meshes, engine):
    """NaN / infinite / 1e18 coordinates reach the tracer as they are (the reference checks nothing:
    MeshTransformer.cpp:142-205 multiplies whatever the cloud holds): the closest hits are the oracle's exhaustive ones,
    bit for bit -- a triangle with a non-finite corner is hit by nobody, its neighbours are unaffected; a 1e18-metre
    triangle under the sensor is hit like any other.  Also a quad with a repeated vertex (one of its two triangles is
    degenerate)."""
    s = sensors["0000"]
    v, t = meshes["ground"]
    cases = []
    nan_shared = v.copy(); nan_shared[40] = np.nan                       # a vertex seven triangles share
    cases.append(("nan shared", nan_shared, t))
    nan_alone = np.concatenate([v, np.full((1, 3), np.nan, np.float32)])  # a vertex nobody uses
    cases.append(("nan alone", nan_alone, t))
    inf_x = v.copy(); inf_x[41, 0] = np.inf
    cases.append(("inf", inf_x, t))
    minus_inf = v.copy(); minus_inf[12] = (-np.inf, np.inf, 0.0)
    cases.append(("-inf", minus_inf, t))
    huge = np.concatenate([v, np.array([[-1e18, -1e18, -3.0], [1e18, -1e18, -3.0], [0.0, 1e18, -3.0]], np.float32)])
    huge_t = np.concatenate([t, np.array([[v.shape[0], v.shape[0] + 1, v.shape[0] + 2]], np.uint32)])
    cases.append(("1e18", huge, huge_t))
    nan_component = v.copy(); nan_component[5, 2] = np.nan; nan_component[77, 1] = np.inf
    cases.append(("mixed", nan_component, t))
    seen_fewer = False
    clean = oracle.trace_frame(s, [(0, v, t, oracle.IDENTITY_AFFINE)])["points"].shape[0]
    for name, vv, tt in cases:
        for A in (oracle.IDENTITY_AFFINE, oracle.affine_from_components(np.array((1.0, -2.0, 0.1), np.float32), np.array((0.02, 0.0, 0.4), np.float32))):
            tr = make_tracer(capi, s, engine)
            assert tr.addGeometry("g", vv.shape[0], tt.shape[0]) == 0
            tr.updateGeometry("g", A, vv, tt)
            assert tr.commitScene() == 0, name
            rc, pts, hits = tr.traceScene(0)
            assert rc == 0, name
            ref = _assert_parity(oracle, s, tr, [(0, vv, tt, A)], pts, hits)
            seen_fewer = seen_fewer or ref["points"].shape[0] < clean
            tr.close()
    assert seen_fewer    # (the non-finite corners did take triangles out of the cloud)
    # a quad with a repeated vertex next to the ground
    quad_v = np.array([[30, -30, -2.5], [30, 30, -2.5], [-30, 30, -2.5], [-30, -30, -2.5]], np.float32)
    quads = np.array([[0, 1, 1, 3], [1, 2, 3, 3]], np.uint32)
    tr = make_tracer(capi, s, engine)
    assert tr.addGeometry("ground", v.shape[0], t.shape[0]) == 0
    assert tr.addGeometry("plate", 4, 2, geometry_type=capi.LS_GEOMETRY_TYPE_QUAD) == 1
    tr.updateGeometry("ground", oracle.IDENTITY_AFFINE, v, t)
    tr.updateGeometry("plate", oracle.IDENTITY_AFFINE, quad_v, quads)
    assert tr.commitScene() == 0
    rc, pts, hits = tr.traceScene(0)
    _assert_parity(oracle, s, tr, [(0, v, t, oracle.IDENTITY_AFFINE), (1, quad_v, quads, oracle.IDENTITY_AFFINE)], pts, hits)
    tr.close()


def test_triangles_whose_corner_lies_on_a_ray(oracle, capi, sensors, engine):
    """The footprint bounds of the projection engine are supersets with explicit slack; the tightest of them is the elevation
    slack of the channel tables (2e-4 degrees since round 5, ls_internal.h: kProjectElevMarginDeg).  Here every triangle has its
    highest or its lowest corner ON a ray of the raster -- t * direction in float32, then carried into the world frame and
    back by the library's transform, so the corner sits within a few 1e-7 rad of the ring, on either side -- and thin slivers
    hang between two neighbouring columns of one ring.  Whether such a ray hits is decided by the exact test's rounding; the
    footprint must contain it either way: ids and t equal to the oracle's exhaustive answer, on both engines."""
    base = sensors["0000"]
    from lidarshooter_amd import synth
    for V, H, vertical in ((128, 1024, synth.syn_vertical(128)),
                           # rings within half a degree of the horizon, one exactly on it: tan(elevation) ~ 0, where the band
                           # test's RELATIVE slack is worth nothing and the tables' margin alone keeps the footprint a superset
                           (65, 1024, np.linspace(-0.5, 0.5, 65, dtype=np.float32))):
        _corner_on_ray_case(oracle, capi, engine, base, V, H, np.asarray(vertical, np.float32))


def _corner_on_ray_case(oracle, capi, engine, base, V, H, vertical, through_tables=False):
    from conftest import grazing_mesh
    s = oracle.Sensor(uid="graze", vertical=vertical, h_begin=np.float32(0.0), h_end=np.float32(360.0), h_count=H,
                      R=base.R, Rinv=base.Rinv, t=base.t)
    verts, idx = grazing_mesh(oracle, s)
    if through_tables:
        # ls_tracer_create_tables with a SLOPPY elevation_deg: 0.004 degrees off, twenty times the elevation slack of the footprint
        # bounds.  The library derives the elevations from sin_theta / cos_theta itself, so nothing is lost.
        st, ct, sp, cp = oracle.ray_tables(s)
        sloppy = (vertical.astype(np.float64) + 0.004 * np.where(np.arange(V) % 2, 1.0, -1.0)).astype(np.float32)
        tr = capi.Tracer.fromTables(st, ct, sloppy, sp, cp, s.h_begin, s.step(), s.Rinv, s.t)
        tr.setOption(capi.LS_OPT_ENGINE, {"bvh": 1, "projection": 2}[engine])
    else:
        tr = make_tracer(capi, s, engine)
    assert tr.addGeometry("graze", verts.shape[0], idx.shape[0]) == 0
    for A in (oracle.IDENTITY_AFFINE, oracle.affine_from_components(np.array((0.3, -0.2, 0.05), np.float32), np.array((0.0, 0.0, 0.7), np.float32))):
        tr.updateGeometry("graze", A, verts, idx)
        assert tr.commitScene() == 0
        rc, pts, hits = tr.traceScene(0)
        assert rc == 0
        ref = _assert_parity(oracle, s, tr, [(0, verts, idx, A)], pts, hits)
        assert ref["points"].shape[0] > 300
    tr.close()


def test_triangles_whose_corner_lies_on_a_ray_through_given_tables(oracle, capi, sensors, engine):
    """The same grazing triangles through ls_tracer_create_tables -- the path the ROS-typed adapter takes -- with an elevation_deg
    that is 0.004 degrees off: since round 6 the library takes each channel's elevation from atan2(cos_theta, sin_theta) (the
    tables the kernels multiply), so a description that loose costs nothing; tables whose description is GROSSLY off (0.02 degrees,
    or an azimuth step that does not fit sin_phi / cos_phi) are refused instead of traced wrongly."""
    base = sensors["0000"]
    from lidarshooter_amd import synth
    for V, H, vertical in ((128, 1024, synth.syn_vertical(128)), (65, 1024, np.linspace(-0.5, 0.5, 65, dtype=np.float32))):
        _corner_on_ray_case(oracle, capi, engine, base, V, H, np.asarray(vertical, np.float32), through_tables=True)
    s = oracle.Sensor(uid="graze", vertical=synth.syn_vertical(16), h_begin=np.float32(0.0), h_end=np.float32(360.0), h_count=64,
                      R=base.R, Rinv=base.Rinv, t=base.t)
    st, ct, sp, cp = oracle.ray_tables(s)
    off = s.vertical.copy()
    off[5] += np.float32(0.02)
    with pytest.raises(capi.LidarShooterHipError):
        capi.Tracer.fromTables(st, ct, off, sp, cp, s.h_begin, s.step(), s.Rinv, s.t)
    with pytest.raises(capi.LidarShooterHipError):
        capi.Tracer.fromTables(st, ct, s.vertical, sp, cp, s.h_begin, np.float32(s.step() * 1.01), s.Rinv, s.t)
    tr = capi.Tracer.fromTables(st, ct, s.vertical, sp, cp, s.h_begin, s.step(), s.Rinv, s.t)
    with pytest.raises(capi.LidarShooterHipError, match="elevation_deg"):        # ... and by ls_tracer_set_sensor_tables, with the reason
        tr.setSensorTables(st, ct, off, sp, cp, s.h_begin, s.step(), s.Rinv, s.t)
    tr.close()


def test_argument_errors(capi, sensors):
    s = sensors["0000"]
    tr = make_tracer(capi, s)
    tr.addGeometry("m", 3, 1)
    v = np.zeros((3, 3), np.float32)
    with pytest.raises(capi.LidarShooterHipError):
        tr.updateGeometry("nope", capi.IDENTITY_AFFINE, v, np.zeros((1, 3), np.uint32))      # unknown name
    with pytest.raises(capi.LidarShooterHipError):
        tr.updateGeometry("m", capi.IDENTITY_AFFINE, np.zeros(12, np.uint8), None, stride=10)  # bad stride
    with pytest.raises(capi.LidarShooterHipError):
        tr.setShard(140, 20)                                                                  # beyond H = 150
    with pytest.raises(capi.LidarShooterHipError):
        tr.setOption(capi.LS_OPT_LEAF_SIZE, 3)
    with pytest.raises(capi.LidarShooterHipError):
        capi.Tracer(s.vertical, s.h_begin, s.h_end, 1, s.Rinv, s.t)                           # H < 2
    with pytest.raises(capi.LidarShooterHipError):
        capi.Tracer(s.vertical, s.h_begin, s.h_end, s.h_count, s.Rinv, s.t, device=99)        # no such GPU
    assert tr.commitScene() == -1 and tr.traceScene(0)[0] == -1                               # nothing uploaded yet
    tr.close()


@pytest.mark.parametrize("instanced", [1, 0])
def test_set_sensor_keeps_the_geometries(oracle, capi, sensors, meshes, engine, instanced):
    """ls_tracer_set_sensor (ITracer::setSensorConfig, ITracer.cpp:48): another raster and pose on a handle that keeps
    its geometries and its commit; both engines (the classic BVH lives in the sensor frame and is built again), with
    frames in flight when the sensor changes."""
    from lidarshooter_amd import synth
    s0, s1 = sensors["0000"], sensors["0001"]
    wide = oracle.Sensor(uid="wide", vertical=synth.syn_vertical(48), h_begin=np.float32(-180.0), h_end=np.float32(180.0), h_count=400,
                         R=s1.R, Rinv=s1.Rinv, t=s1.t)
    tr = make_tracer(capi, s0, engine)
    tr.setOption(capi.LS_OPT_BVH_INSTANCED, instanced)
    _add(tr, "ground", meshes["ground"])
    _add(tr, "face", meshes["ben"])
    A = oracle.affine_from_components(np.array((0.5, -1.0, 0.1), np.float32), np.array((0.0, 0.1, 0.8), np.float32))
    ml = [(0, *meshes["ground"], oracle.IDENTITY_AFFINE), (1, *meshes["ben"], A)]
    tr.updateGeometry("ground", oracle.IDENTITY_AFFINE, *meshes["ground"])
    tr.updateGeometry("face", A, *meshes["ben"])
    assert tr.commitScene() == 0
    rc, pts, hits = tr.traceScene(0)
    _assert_parity(oracle, s0, tr, ml, pts, hits)
    for k, s in enumerate((s1, wide, s0)):
        if engine == "projection":
            tr.setOption(capi.LS_OPT_PIPELINE, 2 if k == 1 else 0)
            tr.traceSceneAsync(10 + k)                            # a frame in flight when the sensor goes
        tr.setSensor(s.vertical, s.h_begin, s.h_end, s.h_count, s.Rinv, s.t)
        assert tr.getTotalRays() == s.V * s.H
        rc, pts, hits = tr.traceScene(20 + k)                     # no commit in between: the library committed again itself
        assert rc == 0
        _assert_parity(oracle, s, tr, ml, pts, hits)
    tr.close()


@pytest.mark.parametrize("leaf", [1, 2, 4])
def test_bvh_wide_and_binary_walks_give_the_oracles_answer(oracle, capi, sensors, meshes, leaf):
    """LS_OPT_BVH_WIDE (round 6): the instanced BVH engine walks four-wide nodes made of the binary hierarchy (k_widen: a node's
    slots are its grandchildren) -- half the trips per ray.  Same boxes, same leaves, same exact test: ids, t and points equal the
    oracle's with the option on and off, switching back and forth on one handle (the twins are made at the commit after the
    option changes), over the shipped scene with ben moving, after a vertex upload (a refit re-makes the twins) and over a
    grid whose hierarchy is deep enough for the stack to matter."""
    from lidarshooter_amd import synth
    s = sensors["0001"]
    tr = make_tracer(capi, s, "bvh")
    tr.setOption(capi.LS_OPT_LEAF_SIZE, leaf)
    _add(tr, "ground", meshes["ground"])
    _add(tr, "face", meshes["ben"])
    tr.updateGeometry("ground", oracle.IDENTITY_AFFINE, *meshes["ground"])
    bv, bt = meshes["ben"]
    for k, wide in enumerate((1, 0, 1, 1, 0)):
        tr.setOption(capi.LS_OPT_BVH_WIDE, wide)
        A = oracle.affine_from_components(np.array((0.5 * k, -0.2 * k, 0.03 * k), np.float32), np.array((0.0, 0.1 * k, 0.3 * k), np.float32))
        v = bv if k < 3 else (bv * np.float32(1.0 + 0.02 * k)).astype(np.float32)      # k >= 3: new vertices, same topology: a refit
        tr.updateGeometry("face", A, v, bt)
        assert tr.commitScene() == 0 and tr.info(capi.LS_INFO_BVH_INSTANCED) == 2
        rc, pts, hits = tr.traceScene(k)
        assert rc == 0
        _assert_parity(oracle, s, tr, [(0, *meshes["ground"], oracle.IDENTITY_AFFINE), (1, v, bt, A)], pts, hits)
    tr.close()
    s2 = _syn_sensor(oracle, sensors, V=64, H=512)
    gv, gt = synth.grid_mesh(300, 150)                       # 90 000 triangles
    res = {}
    for wide in (1, 0):
        tr = make_tracer(capi, s2, "bvh")
        tr.setOption(capi.LS_OPT_LEAF_SIZE, leaf)
        tr.setOption(capi.LS_OPT_BVH_WIDE, wide)
        tr.setOption(capi.LS_OPT_COUNT_VISITS, 1)
        tr.addGeometry("grid", gv.shape[0], gt.shape[0])
        tr.updateGeometry("grid", oracle.IDENTITY_AFFINE, gv, gt)
        assert tr.commitScene() == 0
        for f in range(3):   # a hierarchy of more than 65 536 leaves gets its twins at the second frame that finds it unchanged
            rc, pts, hits = tr.traceScene(f)
            _assert_parity(oracle, s2, tr, [(0, gv, gt, oracle.IDENTITY_AFFINE)], pts, hits)
            assert tr.info(capi.LS_INFO_BVH_WIDE) == (1 if wide and (f >= 1 or leaf > 1) else 0), (wide, f)
        res[wide] = tr.visitStats()
        tr.close()
    # (node fetches per frame: the wide walk makes fewer of them, and not more triangle tests than twice the binary walk's)
    assert res[1][0] < res[0][0] and res[1][1] <= 2 * res[0][1], res
