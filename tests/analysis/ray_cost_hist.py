"""Diagnostic: per-ray node-fetch distribution of the bench workload on the BVH the GPU built."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa
from lidarshooter_amd import capi, hostapi, synth
from oracle import oracle as O
import bench
sensor, meshes = bench.build_workload("syn128x1m")
tr = capi.Tracer(sensor["vertical"], sensor["h_begin"], sensor["h_end"], sensor["h_count"], sensor["Rinv"], sensor["t"])
for name, v, t in meshes:
    tr.addGeometry(name, v.shape[0], t.shape[0]); tr.updateGeometry(name, capi.IDENTITY_AFFINE, v, t)
tr.commitScene(); tr.traceScene(0)
nodes, tri, g = tr.downloadBvh()
s = O.Sensor(uid="b", vertical=sensor["vertical"], h_begin=sensor["h_begin"], h_end=sensor["h_end"], h_count=sensor["h_count"], R=sensor["Rinv"], Rinv=sensor["Rinv"], t=sensor["t"])
dirs = O.ray_dirs(s)
per = np.zeros(dirs.shape[0], np.uint32)
t, gid, st = O.fat_traverse_stats(nodes, tri, g, dirs, per)
tg, gg = tr.denseHits()
print("match", np.array_equal(gg, gid), np.array_equal(tg, t), "stats", st)
print("per-ray nodes: mean %.2f max %d p50 %d p90 %d p99 %d p99.9 %d" % (per.mean(), per.max(), *np.percentile(per, [50, 90, 99, 99.9])))
pc = per.reshape(len(sensor["vertical"]), -1)
print("per-channel mean:", np.round(pc.mean(axis=1), 1).tolist())
print("per-channel max:", pc.max(axis=1).tolist())
