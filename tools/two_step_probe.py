"""Where a two-step frame (ls_trace_scene_begin / ls_trace_scene_expand) spends its time on SYN-128 x SYN-1M, against the
one-step call + ls_expand_points."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from lidarshooter_amd import capi
import bench
sensor, meshes = bench.build_workload("syn128x1m")
dev = torch.device("cuda", 0)
tr = capi.Tracer(sensor["vertical"], sensor["h_begin"], sensor["h_end"], int(sensor["h_count"]), sensor["Rinv"], sensor["t"])
keep = []
for n, v, t in meshes:
    dv = torch.from_numpy(np.ascontiguousarray(v, np.float32)).to(dev)
    dt = torch.from_numpy(np.ascontiguousarray(t, np.uint32).view(np.int32)).to(dev)
    keep.append((dv, dt))
    tr.addGeometry(n, v.shape[0], t.shape[0])
    tr.updateGeometryDeviceShared(n, capi.IDENTITY_AFFINE, dv.data_ptr(), 12, dt.data_ptr())
tr.commitScene()
tr.setOption(capi.LS_OPT_HOST_OUTPUT, 2)
tr.setOption(capi.LS_OPT_READBACK_HITS, 0)
L, h = tr.L, tr.h
fr = capi.Frame()
dst = np.zeros(524288 * 32, np.uint8)
n = C.c_uint32()
N = 200
for _ in range(20):
    L.ls_trace_scene(h, 0, C.byref(fr)); L.ls_expand_points(dst.ctypes.data, fr.compact16, fr.n_points)
t0 = time.perf_counter()
for _ in range(N):
    L.ls_trace_scene(h, 0, C.byref(fr))
t1 = time.perf_counter()
for _ in range(N):
    L.ls_expand_points(dst.ctypes.data, fr.compact16, fr.n_points)
t2 = time.perf_counter()
for _ in range(N):
    L.ls_trace_scene(h, 0, C.byref(fr)); L.ls_expand_points(dst.ctypes.data, fr.compact16, fr.n_points)
t3 = time.perf_counter()
print("one step: trace %.1f us, expand %.1f us, both %.1f us" % ((t1 - t0) / N * 1e6, (t2 - t1) / N * 1e6, (t3 - t2) / N * 1e6))
for _ in range(20):
    L.ls_trace_scene_begin(h, 0, C.byref(n)); L.ls_trace_scene_expand(h, dst.ctypes.data)
tb = te = 0.0
for _ in range(N):
    a = time.perf_counter(); L.ls_trace_scene_begin(h, 0, C.byref(n)); b = time.perf_counter(); L.ls_trace_scene_expand(h, dst.ctypes.data); c = time.perf_counter()
    tb += b - a; te += c - b
print("two steps: begin %.1f us, expand %.1f us, both %.1f us (n = %d)" % (tb / N * 1e6, te / N * 1e6, (tb + te) / N * 1e6, n.value))
