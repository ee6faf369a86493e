"""BVH engine, instanced hierarchies (the default): frames with the pose restated, nothing built -- the traversal kernel's
time by the host clock (run under tools/rocprof_kernels.sh for k_trace_inst's own).  Through capi only (LS_LIB_PATH safe).
usage: bvh_frame_cost.py [frames] [instanced: 1 | 0]"""
import os, sys, time, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from lidarshooter_amd import capi
import bench
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 100
instanced = int(sys.argv[2]) if len(sys.argv) > 2 else 1
sensor, meshes = bench.build_workload(os.environ.get("W", "syn128x1m"))
dev = torch.device("cuda", 0)
tr = capi.Tracer(sensor["vertical"], sensor["h_begin"], sensor["h_end"], int(sensor["h_count"]), sensor["Rinv"], sensor["t"])
tr.setOption(capi.LS_OPT_ENGINE, 1)
tr.setOption(capi.LS_OPT_BVH_INSTANCED, instanced)
if os.environ.get("LS_BENCH_BVH_WIDE") is not None:
    tr.setOption(capi.LS_OPT_BVH_WIDE, int(os.environ["LS_BENCH_BVH_WIDE"]))
keep = []
for n, v, t in meshes:
    dv = torch.from_numpy(np.ascontiguousarray(v, np.float32)).to(dev)
    dt = torch.from_numpy(np.ascontiguousarray(t, np.uint32).view(np.int32)).to(dev)
    keep.append((n, dv, dt))
    tr.addGeometry(n, v.shape[0], t.shape[0])
    tr.updateGeometryDeviceShared(n, capi.IDENTITY_AFFINE, dv.data_ptr(), 12, dt.data_ptr())
assert tr.commitScene() == 0
rc, pts, hits = tr.traceScene(0)
print("points", pts.shape[0], "sha256", hashlib.sha256(pts.tobytes()).hexdigest()[:8])
def frame(f):
    for n, dv, dt in keep: tr.updateGeometryTransform(n, capi.IDENTITY_AFFINE)
    tr.commitScene()
    tr.traceSceneAsync(f)
for f in range(20): frame(f)
tr.synchronize()
t0 = time.perf_counter()
for f in range(frames): frame(f)
tr.synchronize()
print(f"instanced {instanced}: {(time.perf_counter() - t0) / frames * 1e6:.1f} us per frame (host clock, {frames} frames)", flush=True)
tr.close()
