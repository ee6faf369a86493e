"""Per-workgroup phase timeline of k_project from an LS_PROFILE build of the library
(build/var/lib_prof.so: hipcc ... -DLS_PROFILE ls_project.hip).  Timestamps are thread 0's, 100 MHz
wall clock.  Experiment tooling, not part of the product."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("LS_LIB_PATH", os.path.join(ROOT, "build", "var", "lib_prof.so"))
import torch  # noqa: F401,E402  (HIP runtime first)
from lidarshooter_amd import capi  # noqa: E402
import bench  # noqa: E402

sensor, meshes = bench.build_workload(sys.argv[1] if len(sys.argv) > 1 else "syn128x1m")
tr = capi.Tracer(sensor["vertical"], sensor["h_begin"], sensor["h_end"], int(sensor["h_count"]), sensor["Rinv"], sensor["t"])
tr.setOption(capi.LS_OPT_ENGINE, 2)
for name, v, t in meshes:
    tr.addGeometry(name, v.shape[0], t.shape[0])
    tr.updateGeometry(name, capi.IDENTITY_AFFINE, v, t)
for i in range(3):
    tr.commitScene()
    tr.traceScene(i)
L = capi.load()
out = np.zeros((16384, 8), np.uint64)
rc = L.ls_experiment_profile_dump(out.ctypes.data_as(C.c_void_p))
assert rc == 0, rc
p = out[out[:, 6] > 0].astype(np.float64)
t0 = p[:, 0].min()
T = (p[:, :7] - t0) / 100.0   # us
names = ["start", "tables+loads barrier", "phase 1 done", "after barrier", "phase 2 done", "after barrier", "end"]
print("workgroups", len(p), "kernel span us %.2f" % T[:, 6].max(), "survivors/wg mean %.1f max %d" % (p[:, 7].mean(), p[:, 7].max()))
for i, n in enumerate(names):
    col = T[:, i][p[:, i] > 0] if i else T[:, 0]
    print("%-22s p10 %6.2f p50 %6.2f p90 %6.2f max %6.2f" % (n, *np.percentile(col, [10, 50, 90, 100])))
d = np.diff(T, axis=1)
for i in range(6):
    print("delta %-30s p50 %6.2f p90 %6.2f max %6.2f" % (names[i] + " -> " + names[i + 1].split()[0], *np.percentile(d[:, i], [50, 90, 100])))
