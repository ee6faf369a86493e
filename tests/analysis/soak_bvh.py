"""(Test infrastructure.)  Soak of the BVH engine's build paths: a random walk over what can happen between two frames --
poses, vertex uploads with the same indices (refit), new indices (rebuild), leaf size, instanced / classic hierarchy, refit on /
off, engine switches with uploads in between -- on two meshes, every frame compared bit for bit with a second handle that
runs the projection engine on the same scene (tests/test_gpu_parity.py: the engines agree exactly).  usage: soak_bvh.py [seconds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from lidarshooter_amd import capi, synth
from oracle import oracle as O
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
data = os.path.join(ROOT, "tests", "golden", "data")
s = O.load_sensor(os.path.join(data, "config", "hesai-pandar-XT-32-lidar_0001.json"))
gv, gt = synth.grid_mesh(150, 100, half=45.0, seed=3)
bv, bt = O.load_stl(os.path.join(data, "mesh", "ben.stl"))
def handle(engine):
    tr = capi.Tracer(s.vertical, s.h_begin, s.h_end, s.h_count, s.Rinv, s.t, device=0)
    tr.setOption(capi.LS_OPT_ENGINE, engine)
    tr.addGeometry("ground", gv.shape[0], gt.shape[0]); tr.addGeometry("face", bv.shape[0], bt.shape[0])
    tr.updateGeometry("ground", capi.IDENTITY_AFFINE, gv, gt); tr.updateGeometry("face", capi.IDENTITY_AFFINE, bv, bt)
    return tr
bvh, ref = handle(1), handle(2)
rng = np.random.default_rng(11)
t0 = time.time(); frames = 0; counts = {}
cur_g, cur_gt, cur_bt = gv.copy(), gt.copy(), bt.copy()
while time.time() - t0 < budget:
    what = ["pose", "pose", "pose", "verts", "verts", "idx_ground", "idx_face", "leaf", "instanced", "refit", "engine_hop"][int(rng.integers(0, 11))]
    counts[what] = counts.get(what, 0) + 1
    A = O.affine_from_components(rng.uniform(-3, 3, 3).astype(np.float32), rng.uniform(-0.5, 0.5, 3).astype(np.float32))
    if what == "pose":
        for tr in (bvh, ref): tr.updateGeometryTransform("face", A)
    elif what == "verts":
        cur_g = gv.copy(); cur_g[:, 2] += rng.uniform(-0.3, 0.3, gv.shape[0]).astype(np.float32)
        for tr in (bvh, ref): tr.updateGeometry("ground", capi.IDENTITY_AFFINE, cur_g, None)
    elif what == "idx_ground":
        cur_gt = gt[rng.permutation(gt.shape[0])].copy()
        if rng.integers(0, 2): cur_gt[::7] = cur_gt[1::7][: cur_gt[::7].shape[0]]   # (other triangles, not only another order)
        for tr in (bvh, ref): tr.updateGeometry("ground", capi.IDENTITY_AFFINE, cur_g, cur_gt)
    elif what == "idx_face":
        cur_bt = bt[rng.permutation(bt.shape[0])].copy()
        for tr in (bvh, ref): tr.updateGeometry("face", A, bv, cur_bt)
    elif what == "leaf":
        bvh.setOption(capi.LS_OPT_LEAF_SIZE, int(rng.choice([1, 2, 4, 8])))
    elif what == "instanced":
        bvh.setOption(capi.LS_OPT_BVH_INSTANCED, int(rng.integers(0, 2)))
        bvh.setOption(capi.LS_OPT_BVH_WIDE, int(rng.integers(0, 2)))
    elif what == "refit":
        bvh.setOption(capi.LS_OPT_BVH_REFIT, int(rng.integers(0, 2)))
    else:   # the BVH handle runs a frame under the projection engine with an upload in between, then comes back
        bvh.setOption(capi.LS_OPT_ENGINE, 2)
        cur_gt = gt[rng.permutation(gt.shape[0])].copy()
        for tr in (bvh, ref): tr.updateGeometry("ground", capi.IDENTITY_AFFINE, cur_g, cur_gt)
        assert bvh.commitScene() == 0
        bvh.traceScene(frames)
        bvh.setOption(capi.LS_OPT_ENGINE, 1)
        for tr in (bvh, ref): tr.updateGeometryTransform("face", A)
    assert bvh.commitScene() == 0 and ref.commitScene() == 0
    rc1, p1, h1 = bvh.traceScene(frames)
    rc2, p2, h2 = ref.traceScene(frames)
    assert rc1 == rc2 == 0 and np.array_equal(p1, p2), (frames, what)
    for k in ("ray", "geom", "prim", "t"):
        assert np.array_equal(h1[k], h2[k]), (frames, what, k)
    frames += 1
print("soak_bvh: %d frames in %.0f s, every one equal to the projection engine's; steps: %s" % (frames, time.time() - t0, counts))
bvh.close(); ref.close()
