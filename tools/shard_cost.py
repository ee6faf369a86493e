"""Per-rank frame time of an azimuth shard (what one of N GPUs does per frame, without the collective)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from lidarshooter_amd import capi, shards
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
H = int(sensor["h_count"])
for world in (1, 2, 4, 8):
    for rank in sorted({0, world // 2}):
        first, n = shards.shard_columns(H, world, rank)
        tr.setShard(first, n)
        for i in range(300): frame(i)
        tr.synchronize()
        t0 = time.perf_counter()
        for i in range(500): frame(i)
        tr.synchronize()
        print("world %d rank %d: %.2f us per frame" % (world, rank, (time.perf_counter() - t0) / 500 * 1e6))
