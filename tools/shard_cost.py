"""Per-rank cost of an azimuth shard (what one of N GPUs does per frame, without the collective): device-side stage
times (hipEvents: trace = k_cull + k_project, trace_aux = finish, pack) and the streamed frame time through the bare
C calls.  usage: shard_cost.py [LS_OPT_BLOCK_CULL value: 0 off, 1 on, 2 auto]"""
import ctypes as C
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from lidarshooter_amd import capi, shards
import bench
sensor, meshes = bench.build_workload(os.environ.get("W", "syn128x1m"))
dev = torch.device("cuda", 0)
dm = [(n, torch.from_numpy(np.ascontiguousarray(v, np.float32)).to(dev), torch.from_numpy(np.ascontiguousarray(t, np.uint32).view(np.int32)).to(dev), v.shape[0], t.shape[0]) for n, v, t in meshes]
tr = capi.Tracer(sensor["vertical"], sensor["h_begin"], sensor["h_end"], int(sensor["h_count"]), sensor["Rinv"], sensor["t"])
tr.setOption(capi.LS_OPT_ENGINE, 2)
for n, dv, dt, nv, nt in dm: tr.addGeometry(n, nv, nt)
for n, dv, dt, nv, nt in dm: tr.updateGeometryDeviceShared(n, capi.IDENTITY_AFFINE, dv.data_ptr(), 12, dt.data_ptr())
if len(sys.argv) > 1: tr.setOption(capi.LS_OPT_BLOCK_CULL, int(sys.argv[1]))
L, h = tr.L, tr.h
ident = (C.c_float * 12)(*[float(x) for x in capi.IDENTITY_AFFINE])
names = [n.encode() for n, *_ in dm]
fr = capi.Frame()
def frame(i):
    for nm in names: L.ls_update_geometry_transform(h, nm, ident)   # the mesh is unchanged: pose only
    L.ls_commit_scene(h); L.ls_trace_scene_async(h, i, C.byref(fr))
H = int(sensor["h_count"])
for world in ([int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else (1, 2, 4, 8)):
    for rank in sorted({0, world // 2}):
        first, n = shards.shard_columns(H, world, rank)
        tr.setShard(first, n)
        for i in range(300): frame(i)
        tr.synchronize()
        t0 = time.perf_counter()
        for i in range(500): frame(i)
        tr.synchronize()
        streamed = (time.perf_counter() - t0) / 500 * 1e6
        tr.setOption(capi.LS_OPT_TIMING, 1); tr.timings()
        for i in range(100): frame(i)
        tm = tr.timings(); tr.setOption(capi.LS_OPT_TIMING, 0)
        print("world %d rank %d: streamed %.2f us per frame; device: trace %.2f + finish %.2f + pack %.2f us" %
              (world, rank, streamed, tm["trace"] * 1e3, tm["trace_aux"] * 1e3, tm["pack"] * 1e3), flush=True)
