import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from lidarshooter_amd import capi
import bench
sensor, meshes = bench.build_workload("syn128x1m")
dev = torch.device("cuda", 0)
dm = [(n, torch.from_numpy(np.ascontiguousarray(v, np.float32)).to(dev), torch.from_numpy(np.ascontiguousarray(t, np.uint32).view(np.int32)).to(dev), v.shape[0], t.shape[0]) for n, v, t in meshes]
tr = capi.Tracer(sensor["vertical"], sensor["h_begin"], sensor["h_end"], int(sensor["h_count"]), sensor["Rinv"], sensor["t"])
tr.setOption(capi.LS_OPT_ENGINE, 2)
for n, dv, dt, nv, nt in dm: tr.addGeometry(n, nv, nt)
def frame(i):
    for n, dv, dt, nv, nt in dm: tr.updateGeometryDeviceShared(n, capi.IDENTITY_AFFINE, dv.data_ptr(), 12, dt.data_ptr())
    tr.commitScene(); tr.traceSceneAsync(i)
def run(tag, K=100):
    t0 = time.perf_counter()
    for i in range(K): frame(i)
    tr.synchronize()
    print(tag, "us/frame %.2f" % ((time.perf_counter() - t0) / K * 1e6))
run("np0")
for K in (100, 101, 102, 320, 321, 322):
    tr.setOption(capi.LS_OPT_PIPELINE, 1); run("pipe K=%d" % K, K); tr.setOption(capi.LS_OPT_PIPELINE, 0); run("  np after", 200)
