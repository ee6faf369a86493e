"""Would more than three frames in flight raise the headline's throughput?  N tracer handles on the same scene, each with its
three slot streams, each streamed from a thread of its own through the C++ frame loop (ctypes releases the GIL): aggregate
frames per second against one handle's.  usage: two_handles_probe.py [handles]"""
import ctypes as C, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from lidarshooter_amd import capi, hostapi
import bench
bench.pin_to_gpu_numa_node(0)
nh = int(sys.argv[1]) if len(sys.argv) > 1 else 2
sensor, meshes = bench.build_workload(os.environ.get("W", "syn128x1m"))
shard = [int(x) for x in os.environ["SHARD"].split(",")] if os.environ.get("SHARD") else None   # "first,n": an azimuth shard (what a rank of a group traces)
dev = torch.device("cuda", 0)
HL = hostapi.load()
f32p = C.POINTER(C.c_float)
HL.lsh_stream_frames.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(f32p), C.POINTER(C.c_uint), C.c_uint,
                                 C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_uint, C.c_uint, C.c_uint, C.c_uint]
ident = (C.c_float * 12)(*[float(x) for x in capi.IDENTITY_AFFINE])
V, H = len(sensor["vertical"]), int(sensor["h_count"])
cap = V * H
keep = []
for n, v, t in meshes:
    keep.append((n, torch.from_numpy(np.ascontiguousarray(v, np.float32)).to(dev), torch.from_numpy(np.ascontiguousarray(t, np.uint32).view(np.int32)).to(dev), v.shape[0], t.shape[0]))
names = (C.c_char_p * len(meshes))(*[m[0].encode() for m in meshes])
aff = (f32p * len(meshes))(*[C.cast(ident, f32p) for _ in meshes])
na = (C.c_uint * len(meshes))(*[1 for _ in meshes])
handles = []
for k in range(nh):
    tr = capi.Tracer(sensor["vertical"], sensor["h_begin"], sensor["h_end"], H, sensor["Rinv"], sensor["t"])
    tr.setOption(capi.LS_OPT_ENGINE, 2)
    if shard:
        tr.setShard(*shard)
    for n, dv, dt, nv, nt in keep:
        tr.addGeometry(n, nv, nt)
        tr.updateGeometryDeviceShared(n, capi.IDENTITY_AFFINE, dv.data_ptr(), 12, dt.data_ptr())
    tr.commitScene()
    torch.cuda.synchronize(dev)
    tr.setOption(capi.LS_OPT_PIPELINE, 2)
    outs = [torch.zeros(64 + 48 * cap, dtype=torch.uint8, device=dev) for _ in range(3)]
    P = (C.c_void_p * 3)(*[b.data_ptr() + 64 for b in outs])
    Hh = (C.c_void_p * 3)(*[b.data_ptr() + 64 + 32 * cap for b in outs])
    Cn = (C.c_void_p * 3)(*[b.data_ptr() for b in outs])
    handles.append((tr, outs, P, Hh, Cn))
def run(k, first, n):
    tr, outs, P, Hh, Cn = handles[k]
    assert HL.lsh_stream_frames(tr.h, names, aff, na, len(meshes), P, Hh, Cn, 3, cap, first, n) == 0
for k in range(nh): run(k, 0, 600)
torch.cuda.synchronize(dev)
K = 3000
for active in range(1, nh + 1):
    for rep in range(3):
        torch.cuda.synchronize(dev)
        ths = [threading.Thread(target=run, args=(k, 0, K)) for k in range(active)]
        t0 = time.perf_counter()
        for th in ths: th.start()
        for th in ths: th.join()
        torch.cuda.synchronize(dev)
        el = time.perf_counter() - t0
        print("%d handle(s) streaming at once: %.2f us per frame in aggregate (%d frames each)" % (active, el / (active * K) * 1e6, K), flush=True)
for tr, *_ in handles: tr.close()
