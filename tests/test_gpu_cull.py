"""Culling in the projection engine (LS_OPT_BLOCK_CULL: Morton-ordered mesh; a bound per 4 triangles and k_cull, which
drops groups no ring / no shard column can meet -- forced on here, auto takes it from 1.75 M triangles (measured crossover, round 6), and from 512 k under
an azimuth shard narrower than half a turn) must not change a
single bit: culled == unculled == BVH
engine on the headline-sized scene under moving transforms, vertex updates, azimuth shards and both frames-in-flight modes."""
import numpy as np
import pytest

from conftest import make_tracer
from test_gpu_parity import _syn_sensor

pytestmark = pytest.mark.gpu


def _frame(tr):
    assert tr.commitScene() == 0
    rc, pts, hits = tr.traceScene(0)
    assert rc == 0
    return tr.denseHits(), pts, hits


def _same(a, b):
    (ta, ga), pa, ha = a
    (tb, gb), pb, hb = b
    assert np.array_equal(ga, gb) and np.array_equal(ta, tb)
    assert np.array_equal(pa, pb) and np.array_equal(ha, hb)


def test_cull_equals_unculled_under_transforms_and_shards(oracle, capi, sensors):
    from lidarshooter_amd import synth
    v, t = synth.syn_1m()
    s = _syn_sensor(oracle, sensors, V=128, H=4096)
    # on: group culling forced; auto: the size rule (from 1.75 M triangles: not here; from 512 k under a narrow shard: here); off
    on, auto, off = make_tracer(capi, s, "projection"), make_tracer(capi, s, "projection"), make_tracer(capi, s, "projection")
    on.setOption(capi.LS_OPT_BLOCK_CULL, 1)
    off.setOption(capi.LS_OPT_BLOCK_CULL, 0)
    for tr in (on, auto, off):
        tr.addGeometry("g", v.shape[0], t.shape[0])
        tr.updateGeometry("g", oracle.IDENTITY_AFFINE, v, t)
    a, b = _frame(on), _frame(off)
    _same(a, b)
    _same(_frame(auto), b)
    assert 200000 < len(a[1]) < 300000
    # transform-only updates: the cached mesh-space bounds travel through this frame's matrix
    for lin, ang in (((3.0, -7.5, 0.4), (0.02, -0.03, 0.9)), ((-20.0, 11.0, -1.5), (0.3, 0.1, -2.2)), ((0.0, 0.0, 30.0), (1.2, 0.0, 0.0))):
        A = oracle.affine_from_components(np.array(lin, np.float32), np.array(ang, np.float32))
        for tr in (on, off):
            tr.updateGeometryTransform("g", A)
        _same(_frame(on), _frame(off))
    # a non-rigid matrix (the interface takes any Affine3f, ITracer.hpp:69)
    A = np.array([1.5, 0.2, 0.0, 1.0, -0.1, 0.7, 0.3, -2.0, 0.0, 0.4, 2.0, 0.5], np.float32)
    for tr in (on, off):
        tr.updateGeometryTransform("g", A)
    _same(_frame(on), _frame(off))
    # azimuth shards: a narrow sector, a half turn, the wrap-around columns
    for tr in (on, off):
        tr.updateGeometryTransform("g", oracle.IDENTITY_AFFINE)
    for first, n in ((1000, 256), (0, 2048), (3900, 196), (512, 512)):
        for tr in (on, auto, off):
            tr.setShard(first, n)
        ref = _frame(off)
        _same(_frame(on), ref)
        _same(_frame(auto), ref)
    # the auto rule (cull_enabled, ls_commit.cpp): a geometry of 512 k triangles or more is culled under a shard narrower than
    # half a turn (k_cull's counters run), not on the full turn or a half turn
    auto.setOption(capi.LS_OPT_COUNT_VISITS, 1)
    for (first, n), culls in (((512, 512), True), ((0, 4096), False), ((0, 2048), False), ((3900, 196), True)):
        auto.setShard(first, n)
        _frame(auto)
        survivors, bounds_read = auto.visitStats()[2:4]
        assert (survivors > 0 and bounds_read > 0) == culls, (first, n, survivors, bounds_read)
    auto.setOption(capi.LS_OPT_COUNT_VISITS, 0)
    # shards of a moved mesh, rigid or not
    seen = 0
    for A in (oracle.affine_from_components(np.array((12.0, -30.0, 0.8), np.float32), np.array((0.05, -0.02, 2.4), np.float32)),
              np.array([1.5, 0.2, 0.0, 1.0, -0.1, 0.7, 0.3, -2.0, 0.0, 0.4, 2.0, 0.5], np.float32)):
        for tr in (on, auto, off):
            tr.updateGeometryTransform("g", A)
        for first, n in ((0, 512), (1536, 512), (3584, 512), (2000, 100)):
            for tr in (on, auto, off):
                tr.setShard(first, n)
            ref = _frame(off)
            seen += len(ref[1])
            _same(_frame(on), ref)
            _same(_frame(auto), ref)
    assert seen > 10000
    # a vertex upload makes the bounds stale: the next commit rebuilds them
    v2 = v.copy()
    v2[:, 0] += 3.0
    for tr in (on, auto, off):
        tr.updateGeometry("g", oracle.IDENTITY_AFFINE, v2, None)
        tr.setShard(512, 512)
    ref = _frame(off)
    _same(_frame(on), ref)
    _same(_frame(auto), ref)
    on.close()
    auto.close()
    off.close()


def test_cull_follows_vertex_updates_and_small_meshes(oracle, capi, sensors, meshes):
    """A 600 000-triangle grid whose vertices change every frame (bounds recomputed per upload) next to the small
    ben.stl (no block data: its chunks pass through k_cull), against the BVH engine."""
    from lidarshooter_amd import synth
    v, t = synth.grid_mesh(1000, 300)
    bv, bt = meshes["ben"]
    s = _syn_sensor(oracle, sensors, V=128, H=2048)
    pr, bvh = make_tracer(capi, s, "projection"), make_tracer(capi, s, "bvh")
    pr.setOption(capi.LS_OPT_BLOCK_CULL, 1)
    for tr in (pr, bvh):
        tr.addGeometry("grid", v.shape[0], t.shape[0])
        tr.addGeometry("face", bv.shape[0], bt.shape[0])
    rng = np.random.default_rng(5)
    for k in range(4):
        vk = v.copy()
        vk[:, 2] += (0.5 * k + 0.2 * np.sin(0.21 * v[:, 0] + k)).astype(np.float32)
        if k == 3:
            vk = vk[:, [1, 0, 2]].copy()           # the same topology laid out differently: the old Morton order is stale but valid
        A = oracle.affine_from_components(rng.uniform(-2, 2, 3).astype(np.float32), rng.uniform(-0.3, 0.3, 3).astype(np.float32))
        for tr in (pr, bvh):
            tr.updateGeometry("grid", oracle.IDENTITY_AFFINE, vk, t if k == 0 else None)
            tr.updateGeometry("face", A, bv, bt)
        _same(_frame(pr), _frame(bvh))
    pr.close()
    bvh.close()


@pytest.mark.parametrize("mode", [1, 2])
def test_cull_with_frames_in_flight(oracle, capi, sensors, mode):
    """Streamed frames (two on one stream / three on three streams) with block culling and a pose that changes every
    frame: each frame's cloud equals the synchronous, unculled one for the same pose."""
    import torch
    from lidarshooter_amd import synth
    v, t = synth.syn_1m()
    s = _syn_sensor(oracle, sensors, V=128, H=4096)
    dev = torch.device("cuda", 0)
    dv = torch.from_numpy(v).to(dev)
    dt = torch.from_numpy(t.view(np.int32)).to(dev)
    poses = [oracle.affine_from_components(np.array((0.7 * k, -0.4 * k, 0.05 * k), np.float32), np.array((0.0, 0.01 * k, 0.2 * k), np.float32))
             for k in range(6)]
    ref_tr = make_tracer(capi, s, "projection")
    ref_tr.setOption(capi.LS_OPT_BLOCK_CULL, 0)
    ref_tr.addGeometry("g", v.shape[0], t.shape[0])
    ref_tr.updateGeometryDeviceShared("g", poses[0], dv.data_ptr(), 12, dt.data_ptr())
    refs = []
    for A in poses:
        ref_tr.updateGeometryTransform("g", A)
        refs.append(_frame(ref_tr)[1:])
    ref_tr.close()
    tr = make_tracer(capi, s, "projection")
    tr.setOption(capi.LS_OPT_BLOCK_CULL, 1)
    tr.setOption(capi.LS_OPT_PIPELINE, mode)
    tr.addGeometry("g", v.shape[0], t.shape[0])
    tr.updateGeometryDeviceShared("g", poses[0], dv.data_ptr(), 12, dt.data_ptr())
    cap = s.V * s.H
    bufs = [(torch.zeros(32 * cap, dtype=torch.uint8, device=dev), torch.zeros(16 * cap, dtype=torch.uint8, device=dev),
             torch.zeros(4, dtype=torch.int32, device=dev)) for _ in range(3)]
    n_frames = 18
    for i in range(n_frames):
        tr.updateGeometryTransform("g", poses[i % 6])
        assert tr.commitScene() == 0
        p, h, n = bufs[i % 3]
        tr.setOutputBuffers(p.data_ptr(), h.data_ptr(), n.data_ptr(), cap)
        tr.traceSceneAsync(i)
        if i % 3 == 2:
            tr.synchronize()
            for k in (i - 2, i - 1, i):
                p, h, n = bufs[k % 3]
                pts, hits = refs[k % 6]
                cnt = int(n[0].item())
                assert cnt == pts.shape[0]
                assert np.array_equal(p.cpu().numpy()[:32 * cnt].reshape(cnt, 32), pts)
                got = h.cpu().numpy()[:16 * cnt].view(np.uint32).reshape(cnt, 4)
                assert np.array_equal(got, np.stack([hits["ray"], hits["geom"], hits["prim"], hits["t"].view(np.uint32)], axis=1))
    tr.close()


@pytest.mark.parametrize("graph", [0, 1])
def test_streamed_shard_equals_the_synchronous_unculled_one(oracle, capi, sensors, graph):
    """What one of eight ranks runs (tools/shard_cost.py): an eighth-of-a-turn shard of SYN-128 over SYN-1M, three frames in
    flight, the auto rule's culling (k_cull's sector early out, survivors dealt to the waves), finish + pack as ONE launch
    with the chained prefix under plain launches / as two nodes inside captured frame graphs; a pose that changes every
    frame, the shard moved in mid-stream (the status words and their tag follow the new block count).  Every frame's hit
    records and points equal those of a synchronous, unculled frame of the same shard and pose."""
    import torch
    from lidarshooter_amd import synth
    v, t = synth.syn_1m()
    s = _syn_sensor(oracle, sensors, V=128, H=4096)
    dev = torch.device("cuda", 0)
    dv = torch.from_numpy(v).to(dev)
    dt = torch.from_numpy(t.view(np.int32)).to(dev)
    poses = [oracle.affine_from_components(np.array((0.9 * k, -0.3 * k, 0.04 * k), np.float32), np.array((0.0, 0.01 * k, 0.15 * k), np.float32))
             for k in range(5)]
    shards = ((2048, 512), (0, 512), (3584, 512), (1000, 300))
    ref_tr = make_tracer(capi, s, "projection")
    ref_tr.setOption(capi.LS_OPT_BLOCK_CULL, 0)
    ref_tr.addGeometry("g", v.shape[0], t.shape[0])
    ref_tr.updateGeometryDeviceShared("g", poses[0], dv.data_ptr(), 12, dt.data_ptr())
    refs = {}
    for first, n in shards:
        ref_tr.setShard(first, n)
        for k, A in enumerate(poses):
            ref_tr.updateGeometryTransform("g", A)
            refs[(first, n, k)] = _frame(ref_tr)[1:]
    ref_tr.close()
    tr = make_tracer(capi, s, "projection")
    tr.setOption(capi.LS_OPT_PIPELINE, 2)
    tr.setOption(capi.LS_OPT_FRAME_GRAPH, graph)
    tr.addGeometry("g", v.shape[0], t.shape[0])
    tr.updateGeometryDeviceShared("g", poses[0], dv.data_ptr(), 12, dt.data_ptr())
    cap = s.V * 512
    bufs = [(torch.zeros(32 * cap, dtype=torch.uint8, device=dev), torch.zeros(16 * cap, dtype=torch.uint8, device=dev),
             torch.zeros(4, dtype=torch.int32, device=dev)) for _ in range(3)]
    total = 0
    for first, n in shards:
        tr.setShard(first, n)
        for i in range(15):
            tr.updateGeometryTransform("g", poses[i % 5])
            assert tr.commitScene() == 0
            p, h, cnt = bufs[i % 3]
            tr.setOutputBuffers(p.data_ptr(), h.data_ptr(), cnt.data_ptr(), cap)
            tr.traceSceneAsync(i)
            if i % 3 == 2:
                tr.synchronize()
                for k in (i - 2, i - 1, i):
                    p, h, cnt = bufs[k % 3]
                    pts, hits = refs[(first, n, k % 5)]
                    c = int(cnt[0].item())
                    assert c == pts.shape[0], (first, n, k, c, pts.shape[0])
                    assert np.array_equal(p.cpu().numpy()[:32 * c].reshape(c, 32), pts)
                    got = h.cpu().numpy()[:16 * c].view(np.uint32).reshape(c, 4)
                    assert np.array_equal(got, np.stack([hits["ray"], hits["geom"], hits["prim"], hits["t"].view(np.uint32)], axis=1))
                    total += c
    assert total > 100000
    if graph:
        assert tr.info(capi.LS_INFO_FRAME_GRAPH_REPLAYS) >= 4 * 12
    tr.close()


def test_config5_as_composed_vs_oracle(oracle, capi, sensors, meshes):
    """BASELINE.json configs[4] as bench.py --workload cfg5 composes it: a 128 x 4096 sensor on the 20 m pose circle over
    SYN-10M (9 998 244 triangles, group culling on by the auto rule) + ben.stl moved by config/trajectory.json, at two
    trajectory frames: every ray against the CPU BVH oracle (ids and t bit for bit), on both engines for the first."""
    import os
    from conftest import DATA
    from lidarshooter_amd import synth
    v, t = synth.syn_10m()
    assert t.shape[0] == 9998244
    bv, bt = meshes["ben"]
    base = _syn_sensor(oracle, sensors, V=128, H=4096)
    ang = 2.0 * np.pi * 3 / 8.0                                           # rank 3's pose on the circle
    s = oracle.Sensor(uid="cfg5", vertical=base.vertical, h_begin=base.h_begin, h_end=base.h_end, h_count=base.h_count, R=base.R,
                      Rinv=base.Rinv, t=(base.t + np.array([20.0 * np.cos(ang), 20.0 * np.sin(ang), 0.0], np.float32)).astype(np.float32))
    poses = oracle.play_trajectory(os.path.join(DATA, "config", "trajectory.json"), 0.1)
    tr = make_tracer(capi, s, "projection")
    tr.addGeometry("ground", v.shape[0], t.shape[0])
    tr.addGeometry("face", bv.shape[0], bt.shape[0])
    tr.updateGeometry("ground", oracle.IDENTITY_AFFINE, v, t)
    tr.updateGeometry("face", oracle.IDENTITY_AFFINE, bv, bt)
    for k in (7, len(poses) - 1):
        A = oracle.affine_from_components(poses[k, :3] * np.float32(0.05), poses[k, 3:])
        tr.updateGeometryTransform("ground", oracle.IDENTITY_AFFINE)
        tr.updateGeometryTransform("face", A)
        got = _frame(tr)
        ref = oracle.trace_frame(s, [(0, v, t, oracle.IDENTITY_AFFINE), (1, bv, bt, A)], use_bvh=True, nthreads=16)
        (tt, gg), pts, hits = got
        assert np.array_equal(gg, ref["gid"]) and np.array_equal(tt, ref["t"])
        assert np.array_equal(pts, ref["points"])
        assert np.array_equal(np.stack([hits["ray"], hits["geom"], hits["prim"], hits["t"].view(np.uint32)], axis=1), ref["hits"])
        assert (hits["geom"] == 1).sum() > 0 and 200000 < len(pts) < 320000   # ben is in view, the ground fills the lower channels
    tr.close()


def test_cull_stress_poses(oracle, capi, sensors):
    """Group culling forced on, a 600 000-triangle relief under poses that stress its bound: the mesh stood on edge (its
    least-squares planes become walls: slopes are clamped, plain boxes), passing through the sensor's vertical axis and
    through the sensor itself (extent / distance >= 1/4: no first-order bound, groups are kept), scaled up and down by
    1000, mirrored, and sheared -- culled == unculled, bit for bit."""
    from lidarshooter_amd import synth
    v, t = synth.grid_mesh(1000, 300, half=40.0, seed=9, relief=1.5, noise=0.05)
    s = _syn_sensor(oracle, sensors, V=128, H=2048)
    on, off = make_tracer(capi, s, "projection"), make_tracer(capi, s, "projection")
    on.setOption(capi.LS_OPT_BLOCK_CULL, 1)
    off.setOption(capi.LS_OPT_BLOCK_CULL, 0)
    for tr in (on, off):
        tr.addGeometry("g", v.shape[0], t.shape[0])
        tr.updateGeometry("g", oracle.IDENTITY_AFFINE, v, t)
    tx, ty, tz = [float(x) for x in s.t]
    poses = [
        oracle.affine_from_components(np.array((0, 0, 0), np.float32), np.array((np.pi / 2, 0, 0), np.float32)),        # on edge
        oracle.affine_from_components(np.array((tx, ty, tz - 2.0), np.float32), np.array((0, 0, 0.3), np.float32)),     # under the sensor, axis pierces it
        oracle.affine_from_components(np.array((tx, ty, tz), np.float32), np.array((0.4, 0.2, 0), np.float32)),         # through the sensor origin
        oracle.affine_from_components(np.array((tx + 10, ty, tz + 30), np.float32), np.array((0, np.pi / 2, 1.0), np.float32)),
        np.array([1000, 0, 0, 0, 0, 1000, 0, 0, 0, 0, 1000, -20000], np.float32),                                      # huge
        np.array([1e-3, 0, 0, tx + 0.5, 0, 1e-3, 0, ty, 0, 0, 1e-3, tz - 0.2], np.float32),                             # tiny, next to the sensor
        np.array([-1, 0, 0, 5, 0, 1, 0.8, 0, 0, 0, 1, 0], np.float32),                                                  # mirrored + sheared
    ]
    n_hits = []
    for A in poses:
        for tr in (on, off):
            tr.updateGeometryTransform("g", A)
        a, b = _frame(on), _frame(off)
        _same(a, b)
        n_hits.append(len(a[1]))
    assert max(n_hits) > 50000 and len(set(n_hits)) > 3
    on.close()
    off.close()


def test_cull_when_nearly_everything_survives(oracle, capi, sensors):
    """A 540 000-triangle cylinder wall 10 m around the sensor, as tall as its field of view: every group of four triangles
    subtends more than the ring spacing, so the survivor list is nearly the whole mesh -- more than the culled launch's grid
    has workgroups for (it is sized for half of the groups to survive), and its waves come round again for the rest.  Also
    as a shard.  Against the BVH engine."""
    from lidarshooter_amd import synth
    g, t = synth.grid_mesh(900, 300, half=1.0, seed=11, relief=0.0, noise=0.0)
    s = _syn_sensor(oracle, sensors, V=128, H=512)
    rng = np.random.default_rng(3)
    ang = np.pi * g[:, 0].astype(np.float64) * (1.0 - 1e-4)          # the seam stays open by a hair: no coincident vertices
    rad = 10.0 + rng.uniform(-0.05, 0.05, g.shape[0])
    v = np.stack([float(s.t[0]) + rad * np.cos(ang), float(s.t[1]) + rad * np.sin(ang),
                  float(s.t[2]) - 1.0 + 3.6 * g[:, 1].astype(np.float64)], axis=1).astype(np.float32)
    pr, bvh = make_tracer(capi, s, "projection"), make_tracer(capi, s, "bvh")
    pr.setOption(capi.LS_OPT_BLOCK_CULL, 1)
    pr.setOption(capi.LS_OPT_COUNT_VISITS, 1)
    for tr in (pr, bvh):
        tr.addGeometry("wall", v.shape[0], t.shape[0])
        tr.updateGeometry("wall", oracle.IDENTITY_AFFINE, v, t)
    _same(_frame(pr), _frame(bvh))
    groups = (t.shape[0] + 3) // 4
    assert pr.visitStats()[2] > 0.6 * groups, "the scene is meant to keep most groups alive"
    pr.setOption(capi.LS_OPT_COUNT_VISITS, 0)
    for tr in (pr, bvh):
        tr.setShard(100, 200)
    _same(_frame(pr), _frame(bvh))
    pr.close()
    bvh.close()


def test_auto_rule_at_its_measured_threshold(oracle, capi, sensors):
    """cull_enabled (ls_commit.cpp) since round 6: the full raster is culled from 1.75 M triangles -- where the measured lines cross
    (tools/cull_crossover.sh: culling loses 0.8 us per frame at 1.5 M and wins 2.7 at 2 M) -- instead of round 5's 2 M.  A grid just
    below (1 700 416 triangles) is not culled, one just above (1 800 964) is (k_cull's counters run); the cloud is the unculled
    handle's either way."""
    from lidarshooter_amd import synth
    s = _syn_sensor(oracle, sensors, V=128, H=4096)
    for cells, culls in (((1304, 652), False), ((1342, 671), True)):
        v, t = synth.grid_mesh(*cells)
        assert (t.shape[0] >= 1750000) == culls
        auto, off = make_tracer(capi, s, "projection"), make_tracer(capi, s, "projection")
        off.setOption(capi.LS_OPT_BLOCK_CULL, 0)
        for tr in (auto, off):
            tr.addGeometry("g", v.shape[0], t.shape[0])
            tr.updateGeometry("g", oracle.IDENTITY_AFFINE, v, t)
        ref = _frame(off)
        _same(_frame(auto), ref)
        auto.setOption(capi.LS_OPT_COUNT_VISITS, 1)
        _frame(auto)
        survivors, bounds_read = auto.visitStats()[2:4]
        assert (survivors > 0 and bounds_read > 0) == culls, (t.shape[0], survivors, bounds_read)
        auto.setOption(capi.LS_OPT_COUNT_VISITS, 0)
        A = oracle.affine_from_components(np.array((4.0, -9.0, 0.3), np.float32), np.array((0.01, -0.02, 1.1), np.float32))
        for tr in (auto, off):
            tr.updateGeometryTransform("g", A)
        _same(_frame(auto), _frame(off))
        auto.close()
        off.close()
