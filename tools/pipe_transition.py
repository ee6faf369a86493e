import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from lidarshooter_amd import capi
import bench
w = sys.argv[1] if len(sys.argv) > 1 else "xt32"
if w == "coarse":   # the dense sensor over the reference's coarse meshes: nearly every footprint is "big"
    sensor, _ = bench.build_workload("syn128x1m")
    _, meshes = bench.build_workload("xt32")
else:
    sensor, meshes = bench.build_workload(w)
dev = torch.device("cuda", 0)
dm = [(n, torch.from_numpy(np.ascontiguousarray(v, np.float32)).to(dev), torch.from_numpy(np.ascontiguousarray(t, np.uint32).view(np.int32)).to(dev), v.shape[0], t.shape[0]) for n, v, t in meshes]
tr = capi.Tracer(sensor["vertical"], sensor["h_begin"], sensor["h_end"], int(sensor["h_count"]), sensor["Rinv"], sensor["t"])
tr.setOption(capi.LS_OPT_ENGINE, 2)
for n, dv, dt, nv, nt in dm: tr.addGeometry(n, nv, nt)
def frame(i):
    for n, dv, dt, nv, nt in dm: tr.updateGeometryDeviceShared(n, capi.IDENTITY_AFFINE, dv.data_ptr(), 12, dt.data_ptr())
    tr.commitScene(); tr.traceSceneAsync(i)
def run(tag, K=300):
    tr.setOption(capi.LS_OPT_COUNT_VISITS, 1); frame(0); tr.synchronize(); vs = tr.visitStats()[1]; tr.setOption(capi.LS_OPT_COUNT_VISITS, 0)
    for i in range(20): frame(i)
    tr.synchronize()
    t0 = time.perf_counter()
    for i in range(K): frame(i)
    tr.synchronize()
    print(tag, "us/frame %.2f" % ((time.perf_counter() - t0) / K * 1e6), "tests", vs)
run("fresh non-pipelined")
tr.setOption(capi.LS_OPT_PIPELINE, 1); run("pipelined")
tr.setOption(capi.LS_OPT_PIPELINE, 0); run("non-pipelined after")
tr.setOption(capi.LS_OPT_TIMING, 2); run("timing=2"); tr.timings(); tr.setOption(capi.LS_OPT_TIMING, 0)
run("non-pipelined after timing")
tr.setOption(capi.LS_OPT_PIPELINE, 1); tr.setOption(capi.LS_OPT_TIMING, 2); run("pipe on + timing=2"); tr.timings(); tr.setOption(capi.LS_OPT_TIMING, 0)
tr.setOption(capi.LS_OPT_PIPELINE, 0); run("non-pipelined after pipe+timing")
print("---- per-frame after transition")
tr.setOption(capi.LS_OPT_PIPELINE, 1); run("pipelined again")
tr.setOption(capi.LS_OPT_PIPELINE, 0)
ts = []
for i in range(40):
    t0 = time.perf_counter(); frame(i); tr.synchronize(); ts.append((time.perf_counter() - t0) * 1e6)
print("sync frames us:", [round(x) for x in ts])
