"""Long streaming run: frames per second over windows, to expose host-side stalls (argv[3] == "nogc": with
CPython's cyclic collector frozen and off -- the one 30-40 ms pause of a run is its first full collection)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from lidarshooter_amd import capi
import bench
sensor, meshes = bench.build_workload(sys.argv[1] if len(sys.argv) > 1 else "syn128x1m")
pipe = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = torch.device("cuda", 0)
dm = [(n, torch.from_numpy(np.ascontiguousarray(v, np.float32)).to(dev), torch.from_numpy(np.ascontiguousarray(t, np.uint32).view(np.int32)).to(dev), v.shape[0], t.shape[0]) for n, v, t in meshes]
tr = capi.Tracer(sensor["vertical"], sensor["h_begin"], sensor["h_end"], int(sensor["h_count"]), sensor["Rinv"], sensor["t"])
tr.setOption(capi.LS_OPT_ENGINE, 2)
tr.setOption(capi.LS_OPT_PIPELINE, pipe)
for n, dv, dt, nv, nt in dm: tr.addGeometry(n, nv, nt)
def frame(i):
    for n, dv, dt, nv, nt in dm: tr.updateGeometryDeviceShared(n, capi.IDENTITY_AFFINE, dv.data_ptr(), 12, dt.data_ptr())
    tr.commitScene(); tr.traceSceneAsync(i)
import gc
if len(sys.argv) > 3 and sys.argv[3] == "nogc":
    gc.collect(); gc.freeze(); gc.disable()
for i in range(50): frame(i)
tr.synchronize()
W = 500
res = []
for w in range(16):
    t0 = time.perf_counter()
    worst = 0.0
    for i in range(W):
        t1 = time.perf_counter(); frame(i); worst = max(worst, time.perf_counter() - t1)
    tr.synchronize()
    res.append(((time.perf_counter() - t0) / W * 1e6, worst * 1e3))
print("pipeline", pipe, "us/frame per window of %d:" % W, [round(a, 1) for a, b in res])
print("worst single enqueue call per window, ms:", [round(b, 2) for a, b in res])
