"""Host enqueue cost of the headline frame loop over time, in batches of 50 frames without a wait in between (what the periodic\nslow windows of short timed windows are): usage enq_timeline.py [LS_OPT_PIPELINE]"""
import ctypes as C, os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from lidarshooter_amd import capi, hostapi
import bench
sensor, meshes = bench.build_workload("syn128x1m")
dev = torch.device("cuda", 0)
HL = hostapi.load()
f32p = C.POINTER(C.c_float)
HL.lsh_stream_frames.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(f32p), C.POINTER(C.c_uint), C.c_uint,
                                 C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_uint, C.c_uint, C.c_uint, C.c_uint]
ident = (C.c_float * 12)(*[float(x) for x in capi.IDENTITY_AFFINE])
V, H = len(sensor["vertical"]), int(sensor["h_count"]); cap = V * H
tr = capi.Tracer(sensor["vertical"], sensor["h_begin"], sensor["h_end"], H, sensor["Rinv"], sensor["t"])
tr.setOption(capi.LS_OPT_ENGINE, 2)
keep = []
for n, v, t in meshes:
    dv = torch.from_numpy(np.ascontiguousarray(v, np.float32)).to(dev); dt = torch.from_numpy(np.ascontiguousarray(t, np.uint32).view(np.int32)).to(dev)
    keep.append((dv, dt)); tr.addGeometry(n, v.shape[0], t.shape[0]); tr.updateGeometryDeviceShared(n, capi.IDENTITY_AFFINE, dv.data_ptr(), 12, dt.data_ptr())
tr.commitScene(); tr.setOption(capi.LS_OPT_PIPELINE, int(sys.argv[1]) if len(sys.argv) > 1 else 2)
outs = [torch.zeros(64 + 48 * cap, dtype=torch.uint8, device=dev) for _ in range(3)]
names = (C.c_char_p * len(meshes))(*[m[0].encode() for m in meshes]); aff = (f32p * len(meshes))(*[C.cast(ident, f32p) for _ in meshes]); na = (C.c_uint * len(meshes))(*[1 for _ in meshes])
P = (C.c_void_p * 3)(*[b.data_ptr() + 64 for b in outs]); Hh = (C.c_void_p * 3)(*[b.data_ptr() + 64 + 32 * cap for b in outs]); Cn = (C.c_void_p * 3)(*[b.data_ptr() for b in outs])
def run(first, n): assert HL.lsh_stream_frames(tr.h, names, aff, na, len(meshes), P, Hh, Cn, 3, cap, first, n) == 0
run(0, 600); tr.synchronize()
B = 50
ts = []
t0 = time.perf_counter()
for b in range(80):
    a = time.perf_counter(); run(b * B, B); ts.append((time.perf_counter() - a) / B * 1e6)
tr.synchronize()
tot = (time.perf_counter() - t0) / (80 * B) * 1e6
print("enqueue us/frame per batch of %d frames (no sync between): %s" % (B, " ".join("%.1f" % x for x in ts)))
print("overall %.2f us/frame" % tot)
