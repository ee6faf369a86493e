"""Diagnostic: trace-only time vs number of rays (azimuth columns) on the bench scene."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lidarshooter_amd import capi
import bench
sensor, meshes = bench.build_workload("syn128x1m")
tr = capi.Tracer(sensor["vertical"], sensor["h_begin"], sensor["h_end"], sensor["h_count"], sensor["Rinv"], sensor["t"])
for name, v, t in meshes:
    tr.addGeometry(name, v.shape[0], t.shape[0]); tr.updateGeometry(name, capi.IDENTITY_AFFINE, v, t)
tr.commitScene()
for naz in (64, 256, 512, 1024, 2048, 4096):
    tr.setShard(0, naz)
    for _ in range(5): tr.traceSceneAsync(0)
    tr.synchronize()
    tr.setOption(capi.LS_OPT_TIMING, 2); tr.timings()
    for i in range(30): tr.traceSceneAsync(i)
    tm = tr.timings(); tr.setOption(capi.LS_OPT_TIMING, 0)
    tr.setOption(capi.LS_OPT_COUNT_VISITS, 1); tr.traceSceneAsync(0); st = tr.visitStats(); tr.setOption(capi.LS_OPT_COUNT_VISITS, 0)
    print(f"naz {naz:5d} rays {128*naz:7d} trace_ms {tm['trace']:.4f}  Mrays/s {128*naz/tm['trace']/1e3:9.1f} node/ray {st[0]/(128*naz):.1f} maxtrips {st[3]}")
