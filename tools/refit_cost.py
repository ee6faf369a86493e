"""BVH engine, classic hierarchy, a REFIT per frame (poses restated, same topology): run under tools/rocprof_kernels.sh to get
the refit kernels' durations.  usage: refit_cost.py [frames]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from lidarshooter_amd import capi
import bench
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 50
sensor, meshes = bench.build_workload("syn128x1m")
dev = torch.device("cuda", 0)
tr = capi.Tracer(sensor["vertical"], sensor["h_begin"], sensor["h_end"], int(sensor["h_count"]), sensor["Rinv"], sensor["t"])
tr.setOption(capi.LS_OPT_ENGINE, 1)
tr.setOption(capi.LS_OPT_BVH_INSTANCED, 0)
keep = []
for n, v, t in meshes:
    dv = torch.from_numpy(np.ascontiguousarray(v, np.float32)).to(dev)
    dt = torch.from_numpy(np.ascontiguousarray(t, np.uint32).view(np.int32)).to(dev)
    keep.append((n, dv, dt))
    tr.addGeometry(n, v.shape[0], t.shape[0])
    tr.updateGeometryDeviceShared(n, capi.IDENTITY_AFFINE, dv.data_ptr(), 12, dt.data_ptr())
tr.commitScene()
import time
for f in range(frames):
    for n, dv, dt in keep: tr.updateGeometryTransform(n, capi.IDENTITY_AFFINE)
    tr.commitScene()
    tr.traceSceneAsync(f)
tr.synchronize()
print("last commit was a refit:", tr.info(capi.LS_INFO_LAST_COMMIT_REFIT))
t0 = time.perf_counter()
for f in range(frames):
    for n, dv, dt in keep: tr.updateGeometryTransform(n, capi.IDENTITY_AFFINE)
    tr.commitScene()
    tr.traceSceneAsync(f)
tr.synchronize()
print(f"refit + trace {(time.perf_counter() - t0) / frames * 1e3:.3f} ms per frame (host clock, {frames} frames)", flush=True)
tr.close()
