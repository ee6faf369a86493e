"""Upper bound for overlapping consecutive frames: N tracer handles (own streams), same scene,
frames issued round-robin; frames/s against one handle.  Experiment tooling."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gc
import numpy as np, torch
from lidarshooter_amd import capi
import bench

sensor, meshes = bench.build_workload(sys.argv[1] if len(sys.argv) > 1 else "syn128x1m")
gc.collect(); gc.freeze(); gc.disable()   # a full collection inside one of the runs would look like a slow configuration
dev = torch.device("cuda", 0)
d_meshes = [(n, torch.from_numpy(np.ascontiguousarray(v, np.float32)).to(dev),
             torch.from_numpy(np.ascontiguousarray(t, np.uint32).view(np.int32)).to(dev), v.shape[0], t.shape[0]) for n, v, t in meshes]
for nh in (2, 3, 4, 6, 2, 1):
    trs = []
    for k in range(nh):
        tr = capi.Tracer(sensor["vertical"], sensor["h_begin"], sensor["h_end"], int(sensor["h_count"]), sensor["Rinv"], sensor["t"])
        tr.setOption(capi.LS_OPT_ENGINE, 2)
        tr.setOption(capi.LS_OPT_PIPELINE, int(os.environ.get("PIPE", "0")))
        for n, dv, dt, nv, nt in d_meshes:
            tr.addGeometry(n, nv, nt)
        trs.append(tr)
    def frame(tr, i):
        for n, dv, dt, nv, nt in d_meshes:
            tr.updateGeometryDeviceShared(n, capi.IDENTITY_AFFINE, dv.data_ptr(), 12, dt.data_ptr())
        tr.commitScene(); tr.traceSceneAsync(i)
    for i in range(30):
        frame(trs[i % nh], i)
    for tr in trs: tr.synchronize()
    K = 600
    t0 = time.perf_counter()
    for i in range(K):
        frame(trs[i % nh], i)
    for tr in trs: tr.synchronize()
    el = time.perf_counter() - t0
    print("handles", nh, "us per frame %.2f" % (el / K * 1e6), "frames/s %.0f" % (K / el))
    for tr in trs: tr.close()
