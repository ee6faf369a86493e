"""(Test infrastructure: compares the library with itself.)  Soak of the frames-in-flight path at rasters large enough for the
fat finish / pack passes (k_project_finish_wide, k_pack_wide: 2 / 4 / 8 rays per lane from 512 / 1 024 / 2 048 ray blocks): SYN sensors
of several sizes -- block counts that leave the last workgroup a part of its span -- over a 240 000-triangle relief and a moving
second mesh, three frames in flight (plain launches and frame graphs), a new pose every frame; every streamed frame's cloud
and hit records are compared with the synchronous one-step trace of the same pose on a second handle (one frame in flight:
the one-ray-per-lane kernels).  usage: soak_wide.py [seconds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from lidarshooter_amd import capi, synth
from oracle import oracle as O
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
data = os.path.join(ROOT, "tests", "golden", "data")
base = O.load_sensor(os.path.join(data, "config", "hesai-pandar-XT-32-lidar_0000.json"))
ben = O.load_stl(os.path.join(data, "mesh", "ben.stl"))
grid = synth.grid_mesh(400, 300)
rng = np.random.default_rng(11)
t0 = time.time(); frames = 0; rounds = 0; sizes = {}
rasters = [(128, 4096), (161, 4090), (131, 4000), (67, 4001), (40, 4099), (33, 2000)]
while time.time() - t0 < budget:
    V, H = rasters[rounds % len(rasters)]
    rounds += 1
    vert = synth.syn_vertical(V)
    trs = []
    for _ in range(2):
        tr = capi.Tracer(vert, np.float32(0.0), np.float32(360.0), H, base.Rinv, base.t, device=0)
        tr.setOption(capi.LS_OPT_ENGINE, 2)
        tr.addGeometry("grid", *[a.shape[0] for a in grid]); tr.addGeometry("face", *[a.shape[0] for a in ben])
        tr.updateGeometry("grid", capi.IDENTITY_AFFINE, *grid); tr.updateGeometry("face", capi.IDENTITY_AFFINE, *ben)
        trs.append(tr)
    fast, slow = trs
    fast.setOption(capi.LS_OPT_PIPELINE, 2)
    assert fast.info(capi.LS_INFO_PIPELINE_MODE) == 2
    fast.setOption(capi.LS_OPT_FRAME_GRAPH, int(rounds % 2))
    cap = V * H
    bufs = [(torch.zeros(32 * cap, dtype=torch.uint8, device="cuda:0"), torch.zeros(16 * cap, dtype=torch.uint8, device="cuda:0"),
             torch.zeros(4, dtype=torch.int32, device="cuda:0")) for _ in range(3)]
    for burst in range(4):
        poses = []
        for i in range(3):
            A = O.affine_from_components(rng.uniform(-6, 6, 3).astype(np.float32) * np.float32([1, 1, 0.1]), rng.uniform(-0.5, 0.5, 3).astype(np.float32))
            poses.append(A)
            p, h, n = bufs[i]
            fast.updateGeometryTransform("face", A)
            assert fast.commitScene() == 0
            fast.setOutputBuffers(p.data_ptr(), h.data_ptr(), n.data_ptr(), cap)
            fast.traceSceneAsync(frames + i)
        fast.synchronize()
        for i in range(3):
            slow.updateGeometryTransform("face", poses[i])
            assert slow.commitScene() == 0
            rc, pts, hits = slow.traceScene(frames + i)
            p, h, n = bufs[i]
            cnt = int(n[0].item())
            assert rc == 0 and cnt == pts.shape[0], (V, H, cnt, pts.shape)
            assert np.array_equal(p.cpu().numpy()[:32 * cnt].reshape(cnt, 32), pts), (V, H, frames + i)
            want = np.stack([hits["ray"], hits["geom"], hits["prim"], hits["t"].view(np.uint32)], axis=1)
            assert np.array_equal(h.cpu().numpy()[:16 * cnt].view(np.uint32).reshape(cnt, 4), want), (V, H, frames + i)
        frames += 3
    sizes[(V, H)] = sizes.get((V, H), 0) + 12
    for tr in trs:
        tr.synchronize(); tr.close()
print("soak_wide: %d streamed frames in %.0f s, every cloud and every hit record equal to the one-step trace of the same pose; rasters (frames): %s"
      % (frames, time.time() - t0, ", ".join("%dx%d=%d blocks (%d)" % (v, h, (v * h + 255) // 256, c) for (v, h), c in sizes.items())))
