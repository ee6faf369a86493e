"""(Test infrastructure.)  Soak of what a rank of a sharded group runs: a random walk over azimuth shards, poses, LS_OPT_BLOCK_CULL
(off / on / auto: from 512 k triangles under a narrow shard), frames in flight (one / riders / three streams), frames as captured
graphs, vertex uploads, index uploads -- now and then with a vertex index the mesh does not have, which the commit must refuse
and the next upload must cure -- on a 600 000-triangle relief (above the auto rule's threshold) next to ben.stl.  Streamed frames
(three per step, so that every slot of the rotation runs) are compared bit for bit -- points and hit records -- with a second
handle that traces the same shard synchronously, unculled, one frame in flight.  usage: soak_shard.py [seconds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from lidarshooter_amd import capi, synth
from oracle import oracle as O
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
data = os.path.join(ROOT, "tests", "golden", "data")
base = O.load_sensor(os.path.join(data, "config", "hesai-pandar-XT-32-lidar_0000.json"))
V, H = 64, 2048
s = O.Sensor(uid="soak", vertical=synth.syn_vertical(V), h_begin=np.float32(0.0), h_end=np.float32(360.0), h_count=H, R=base.R, Rinv=base.Rinv, t=base.t)
gv, gt = synth.grid_mesh(1000, 300, half=48.0, seed=9)        # 600 000 triangles
bv, bt = O.load_stl(os.path.join(data, "mesh", "ben.stl"))
dev = torch.device("cuda", 0)
def handle():
    tr = capi.Tracer(s.vertical, s.h_begin, s.h_end, s.h_count, s.Rinv, s.t, device=0)
    tr.setOption(capi.LS_OPT_ENGINE, 2)
    tr.addGeometry("ground", gv.shape[0], gt.shape[0]); tr.addGeometry("face", bv.shape[0], bt.shape[0])
    tr.updateGeometry("ground", capi.IDENTITY_AFFINE, gv, gt); tr.updateGeometry("face", capi.IDENTITY_AFFINE, bv, bt)
    return tr
tr, ref = handle(), handle()
ref.setOption(capi.LS_OPT_BLOCK_CULL, 0)
cap = V * H
bufs = [(torch.zeros(32 * cap, dtype=torch.uint8, device=dev), torch.zeros(16 * cap, dtype=torch.uint8, device=dev), torch.zeros(4, dtype=torch.int32, device=dev)) for _ in range(3)]
rng = np.random.default_rng(23)
t0 = time.time(); frames = 0; counts = {}; refused = 0; culled_frames = 0
cur_g = gv.copy()
A_g, A_b = capi.IDENTITY_AFFINE, capi.IDENTITY_AFFINE
while time.time() - t0 < budget:
    what = ["shard", "shard", "pose", "pose", "cull", "mode", "graph", "verts", "idx", "bad_idx"][int(rng.integers(0, 10))]
    counts[what] = counts.get(what, 0) + 1
    if what == "shard":
        n = int(rng.choice([1, 7, 64, 128, 256, 256, 512, 1024, 2048]))
        first = int(rng.integers(0, H - n + 1))
        for t in (tr, ref): t.setShard(first, n)
    elif what == "pose":
        A_b = O.affine_from_components(rng.uniform(-4, 4, 3).astype(np.float32), rng.uniform(-0.5, 0.5, 3).astype(np.float32))
        if rng.integers(0, 3) == 0:
            A_g = O.affine_from_components(rng.uniform(-10, 10, 3).astype(np.float32) * np.float32([1, 1, 0.02]), rng.uniform(-0.03, 0.03, 3).astype(np.float32) + np.float32([0, 0, rng.uniform(-3, 3)]))
    elif what == "cull":
        tr.setOption(capi.LS_OPT_BLOCK_CULL, int(rng.integers(0, 3)))
    elif what == "mode":
        tr.setOption(capi.LS_OPT_PIPELINE, int(rng.integers(0, 3)))
    elif what == "graph":
        tr.setOption(capi.LS_OPT_FRAME_GRAPH, int(rng.integers(0, 2)))
    elif what == "verts":
        cur_g = gv.copy(); cur_g[:, 2] += rng.uniform(-0.2, 0.2, gv.shape[0]).astype(np.float32)
        for t in (tr, ref): t.updateGeometry("ground", A_g, cur_g, None)
    elif what == "idx":
        cur_gt = gt[rng.permutation(gt.shape[0])].copy()
        for t in (tr, ref): t.updateGeometry("ground", A_g, cur_g, cur_gt)
    else:
        bad = gt.copy(); bad[int(rng.integers(0, gt.shape[0])), int(rng.integers(0, 3))] = gv.shape[0] + int(rng.integers(0, 1000))
        tr.updateGeometry("ground", A_g, cur_g, bad)
        try:
            tr.commitScene()
            raise SystemExit("a mesh with a wild index was committed")
        except capi.LidarShooterHipError as e:
            assert "vertex index" in str(e)
            refused += 1
        for t in (tr, ref): t.updateGeometry("ground", A_g, cur_g, gt)
    for t in (tr, ref):
        t.updateGeometryTransform("ground", A_g); t.updateGeometryTransform("face", A_b)
    assert ref.commitScene() == 0
    rc, rp, rh = ref.traceScene(frames)
    assert rc == 0
    want_h = np.stack([rh["ray"], rh["geom"], rh["prim"], rh["t"].view(np.uint32)], axis=1)
    for k in range(3):      # three streamed frames of the same scene: one per slot of the rotation
        assert tr.commitScene() == 0
        p, h, n = bufs[k]
        tr.setOutputBuffers(p.data_ptr(), h.data_ptr(), n.data_ptr(), cap)
        tr.traceSceneAsync(frames + k)
    tr.synchronize()
    for k in range(3):
        p, h, n = bufs[k]
        c = int(n[0].item())
        assert c == rp.shape[0], (frames, what, k, c, rp.shape[0])
        assert np.array_equal(p.cpu().numpy()[:32 * c].reshape(c, 32), rp), (frames, what, k)
        assert np.array_equal(h.cpu().numpy()[:16 * c].view(np.uint32).reshape(c, 4), want_h), (frames, what, k)
    tr.setOutputBuffers(None, None, None, 0)
    frames += 3
print("soak_shard: %d streamed frames in %.0f s, every one equal to the synchronous unculled trace of the same shard; %d wild-index meshes refused; steps: %s; "
      "frame graphs: %d captured, %d replayed, %d patched (%d patches waited for their stream)" %
      (frames, time.time() - t0, refused, counts, tr.info(capi.LS_INFO_FRAME_GRAPH_CAPTURES), tr.info(capi.LS_INFO_FRAME_GRAPH_REPLAYS),
       tr.info(capi.LS_INFO_FRAME_GRAPH_PATCHES), tr.info(capi.LS_INFO_FRAME_GRAPH_PATCH_WAITS)))
tr.close(); ref.close()
