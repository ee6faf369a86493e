import os, sys, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from lidarshooter_amd import capi, hostapi, synth
import bench
sensor, _ = bench.build_workload("syn128x1m")
DATA = bench.DATA
meshes = []
for n in ("ground", "ben"):
    m = hostapi.PolygonMesh(os.path.join(DATA, "mesh", n + ".stl")); meshes.append((n, m.points(), m.polygons()))
for eng in (2, 1):
    tr = capi.Tracer(sensor["vertical"], sensor["h_begin"], sensor["h_end"], int(sensor["h_count"]), sensor["Rinv"], sensor["t"])
    tr.setOption(capi.LS_OPT_ENGINE, eng)
    for name, v, t in meshes:
        tr.addGeometry(name, v.shape[0], t.shape[0]); tr.updateGeometry(name, capi.IDENTITY_AFFINE, v, t)
    tr.commitScene(); rc, pts, hits = tr.traceScene(0)
    tr.setOption(capi.LS_OPT_TIMING, 1); tr.timings()
    t0 = time.perf_counter()
    for i in range(50):
        tr.commitScene(); tr.traceSceneAsync(i)
    tr.synchronize(); el = (time.perf_counter() - t0) / 50
    print("engine", eng, "hits", len(pts), "frame ms", round(el * 1e3, 4), {k: round(v, 4) for k, v in tr.timings().items() if v})
    tr.close()
