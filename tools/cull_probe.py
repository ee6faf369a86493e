"""How many wave-chunks survive k_cull on the headline frame, and what the triangles of the survivors look like."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from lidarshooter_amd import capi, hostapi, synth
DATA = os.path.join(ROOT, "tests", "golden", "data")
dev = hostapi.LidarDevice(os.path.join(DATA, "config", "hesai-pandar-XT-32-lidar_0000.json"))
d = dict(dev.desc())
v, t = synth.syn_1m()
for world in (1, 8):
    tr = capi.Tracer(synth.syn_vertical(128), 0.0, 360.0, 4096, d["Rinv"], d["t"])
    if world > 1:
        tr.setShard(0, 4096 // world)
    tr.addGeometry("g", v.shape[0], t.shape[0])
    tr.updateGeometry("g", capi.IDENTITY_AFFINE, v, t)
    tr.setOption(capi.LS_OPT_COUNT_VISITS, 1)
    tr.commitScene()
    tr.traceScene(0)
    n_node, n_tri, live, _ = tr.visitStats()
    print(f"world {world}: candidate tests {n_tri}, live groups {live} of {(t.shape[0] + 3) // 4} -> {live * 4 / 64:.0f} dense waves")
    tr.close()
