"""(Test infrastructure.)  Soak of the sharded group's frame path on a one-rank communicator: per-set communicators, one
captured HIP graph per frame (patched, captured anew, discarded), optionally the sized gather -- while the scene walks at
random: poses, vertex uploads, a geometry that comes and goes (the hit count jumps: slots are resized, frames may be
reported truncated), the empty scene.  Every cloud that is delivered is compared bit for bit with a plain handle's
synchronous trace of the same scene.  usage: soak_group.py [seconds] [flags: 0 default, 2 no graph, 4 sized gather]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from lidarshooter_amd import capi, groupapi, synth
from oracle import oracle as O
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
flags = int(sys.argv[2]) if len(sys.argv) > 2 else 0
data = os.path.join(ROOT, "tests", "golden", "data")
s = O.load_sensor(os.path.join(data, "config", "hesai-pandar-XT-32-lidar_0000.json"))
gv, gt = synth.grid_mesh(60, 40, half=40.0, seed=4)
bv, bt = O.load_stl(os.path.join(data, "mesh", "ben.stl"))
def handle():
    tr = capi.Tracer(s.vertical, s.h_begin, s.h_end, s.h_count, s.Rinv, s.t, device=0)
    tr.setOption(capi.LS_OPT_ENGINE, 2)
    tr.addGeometry("face", bv.shape[0], bt.shape[0]); tr.updateGeometry("face", capi.IDENTITY_AFFINE, bv, bt)
    return tr
tr, ref = handle(), handle()
g = groupapi.Group(tr, 1, 0, groupapi.SHARDED, flags=flags)
rng = np.random.default_rng(23)
t0 = time.time(); frames = 0; delivered = 0; truncated = 0; counts = {}
have_ground = False
pending = []   # (frame, expected points, expected hits) of the frames still held by the three sets
def expect():
    if ref.commitScene() != 0: return None, None
    rc, p, h = ref.traceScene(0)
    return p, h
while time.time() - t0 < budget:
    what = ["pose", "pose", "pose", "pose", "verts", "ground", "empty"][int(rng.integers(0, 7))]
    counts[what] = counts.get(what, 0) + 1
    A = O.affine_from_components(rng.uniform(-3, 3, 3).astype(np.float32), rng.uniform(-0.5, 0.5, 3).astype(np.float32))
    if what == "pose":
        for t in (tr, ref): t.updateGeometryTransform("face", A)
    elif what == "verts":
        v2 = bv.copy(); v2[:, 2] += np.float32(rng.uniform(-0.2, 0.2))
        for t in (tr, ref): t.updateGeometry("face", A, v2, None)
    elif what == "ground":
        for t in (tr, ref):
            if have_ground: t.removeGeometry("ground")
            else:
                t.addGeometry("ground", gv.shape[0], gt.shape[0]); t.updateGeometry("ground", capi.IDENTITY_AFFINE, gv, gt)
        have_ground = not have_ground
    else:   # the scene is empty for a frame, then the face is back
        for t in (tr, ref): t.removeGeometry("face")
        if have_ground:
            for t in (tr, ref): t.removeGeometry("ground")
            have_ground = False
        rc_c = tr.commitScene(); ref.commitScene()
        assert rc_c == -1
        assert g.trace(frames) == -1
        pts, hits = g.download(frames)   # (an empty frame takes no turn in the rotation: the next frame has its set)
        assert pts.shape[0] == 0
        delivered += 1; frames += 1
        for t in (tr, ref):
            t.addGeometry("face", bv.shape[0], bt.shape[0]); t.updateGeometry("face", A, bv, bt)
    rc_c = tr.commitScene()
    assert rc_c == 0
    ep, eh = expect()
    assert g.trace(frames) == 0
    pending.append((frames, ep, eh)); frames += 1
    pending = pending[-3:]
    if rng.integers(0, 3) == 0:   # collect one of the frames the sets still hold
        f, ep, eh = pending[int(rng.integers(0, len(pending)))]
        try:
            pts, hits = g.download(f)
        except capi.LidarShooterHipError as e:
            assert "outgrew" in str(e), str(e)
            truncated += 1
            continue
        if ep is None:
            assert pts.shape[0] == 0, f
        else:
            assert np.array_equal(pts, ep), (f, what)
            assert np.array_equal(hits, np.ascontiguousarray(eh).view(np.uint32).reshape(-1, 4)), (f, what)   # (ray, geom, prim, bits of t)
        delivered += 1
print("soak_group flags %d: %d frames in %.0f s, %d clouds collected and equal, %d reported truncated; captures %d replays %d patches %d; steps %s" % (
    flags, frames, time.time() - t0, delivered, truncated, tr.info(capi.LS_INFO_FRAME_GRAPH_CAPTURES), tr.info(capi.LS_INFO_FRAME_GRAPH_REPLAYS),
    tr.info(capi.LS_INFO_FRAME_GRAPH_PATCHES), counts))
g.close(); tr.close(); ref.close()
