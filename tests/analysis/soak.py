"""(Test infrastructure: uses the oracle.)  Soak: tens of thousands of frames through the paths that hand data between threads, streams and the host -- the two-step
trace (host flags, worker pool, rebuild from (ray, t) records), frames in flight in every LS_OPT_PIPELINE mode, mesh poses
changing every frame -- each checked against the one-step synchronous call on the same handle.  usage: soak.py [seconds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from lidarshooter_amd import capi
from oracle import oracle as O
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
data = os.path.join(ROOT, "tests", "golden", "data")
s = O.load_sensor(os.path.join(data, "config", "hesai-pandar-XT-32-lidar_0001.json"))
ground = O.load_stl(os.path.join(data, "mesh", "ground.stl"))
ben = O.load_stl(os.path.join(data, "mesh", "ben.stl"))
tr = capi.Tracer(s.vertical, s.h_begin, s.h_end, s.h_count, s.Rinv, s.t, device=0)
tr.setOption(capi.LS_OPT_ENGINE, 2)
tr.addGeometry("ground", *[a.shape[0] for a in ground]); tr.addGeometry("face", *[a.shape[0] for a in ben])
tr.updateGeometry("ground", capi.IDENTITY_AFFINE, *ground); tr.updateGeometry("face", capi.IDENTITY_AFFINE, *ben)
rng = np.random.default_rng(5)
t0 = time.time(); frames = 0; checked = 0
while time.time() - t0 < budget:
    mode = int(rng.integers(0, 3))
    tr.setOption(capi.LS_OPT_PIPELINE, mode)
    tr.setOption(capi.LS_OPT_FRAME_GRAPH, int(rng.integers(0, 2)))   # (three-stream mode: frames as captured graphs, poses patched in)
    for k in range(200):
        A = O.affine_from_components(rng.uniform(-3, 3, 3).astype(np.float32), rng.uniform(-0.5, 0.5, 3).astype(np.float32))
        tr.updateGeometryTransform("face", A)
        assert tr.commitScene() == 0
        rc, pts = tr.traceSceneTwoStep(frames)
        frames += 1
        if k % 10 == 0:
            rc1, pts1, _ = tr.traceScene(frames)
            assert rc == 0 and rc1 == 0 and np.array_equal(pts, pts1), (frames, mode, pts.shape, pts1.shape)
            checked += 1
        else:
            tr.traceSceneAsync(frames)          # frames in flight between the synchronous ones
    if checked % 200 == 0:
        ref = O.trace_frame(s, [(0, *ground, O.IDENTITY_AFFINE), (1, *ben, A)])
        assert np.array_equal(pts, ref["points"])
print("frame graphs: captured %d, replayed %d, nodes patched %d" % (tr.info(capi.LS_INFO_FRAME_GRAPH_CAPTURES), tr.info(capi.LS_INFO_FRAME_GRAPH_REPLAYS),
                                                                     tr.info(capi.LS_INFO_FRAME_GRAPH_PATCHES)))
tr.synchronize(); tr.close()
print("soak ok: %d frames, %d compared with the one-step call, %.0f s" % (frames, checked, time.time() - t0))
