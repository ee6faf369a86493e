"""GPU: what k_cull reads and keeps on a workload (LS_OPT_COUNT_VISITS): blocks, group bounds read, groups surviving.
usage: cull_stats.py [syn128x1m|syn128x10m] [cull option 0/1/2]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from lidarshooter_amd import capi
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "syn128x10m"
sensor, meshes = bench.build_workload(wl)
dev = torch.device("cuda", 0)
tr = capi.Tracer(sensor["vertical"], sensor["h_begin"], sensor["h_end"], int(sensor["h_count"]), sensor["Rinv"], sensor["t"])
tr.setOption(capi.LS_OPT_ENGINE, 2)
tr.setOption(capi.LS_OPT_BLOCK_CULL, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
keep = []
for n, v, t in meshes:
    dv = torch.from_numpy(np.ascontiguousarray(v, np.float32)).to(dev)
    dt = torch.from_numpy(np.ascontiguousarray(t, np.uint32).view(np.int32)).to(dev)
    keep.append((dv, dt))
    tr.addGeometry(n, v.shape[0], t.shape[0])
    tr.updateGeometryDeviceShared(n, capi.IDENTITY_AFFINE, dv.data_ptr(), 12, dt.data_ptr())
tr.setOption(capi.LS_OPT_COUNT_VISITS, 1)
tr.commitScene()
tr.traceSceneAsync(0)
tr.synchronize()
c = tr.visitStats()
nt = sum(t.shape[0] for _, _, t in meshes)
groups = (nt + 3) // 4
blocks = (groups + 63) // 64
print(f"{wl}: triangles {nt}, groups {groups}, blocks {blocks}; group bounds read {c[3]} ({c[3] / groups:.3f} of all, {c[3] // 64} blocks alive = {c[3] / 64 / blocks:.3f}); "
      f"groups surviving {c[2]} ({c[2] / groups:.3f}); candidate cell tests {c[1]}")
