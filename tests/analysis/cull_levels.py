"""Offline (numpy, no GPU): what fraction of Morton-ordered triangle blocks of 4 / 16 / 64 / 256 / 1024 triangles can be
dropped because no ring of the SYN-128 raster passes between the block's lowest and highest vertex elevation -- the
optimum any per-block bound can reach.  usage: cull_levels.py [1m|10m]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import oracle as O
from lidarshooter_amd import synth
DATA = os.path.join(ROOT, "tests", "golden", "data")
s = O.load_sensor(os.path.join(DATA, "config", "hesai-pandar-XT-32-lidar_0000.json"))
which = sys.argv[1] if len(sys.argv) > 1 else "1m"
v, t = synth.syn_10m() if which == "10m" else synth.syn_1m()
vs = O.transform_vertices(v, O.IDENTITY_AFFINE, s).astype(np.float64)
tan_e = vs[:, 2] / np.hypot(vs[:, 0], vs[:, 1])
# Morton order of the centroids (mesh space, isotropic 10-bit quantisation: k_mesh_morton)
c = v[t].mean(axis=1)
lo = v.min(axis=0); ext = (v.max(axis=0) - lo).max()
q = np.clip(((c - lo) * (1024.0 / ext)).astype(np.int64), 0, 1023)
def spread(x):
    x = (x | (x << 16)) & 0x030000FF; x = (x | (x << 8)) & 0x0300F00F
    x = (x | (x << 4)) & 0x030C30C3; x = (x | (x << 2)) & 0x09249249
    return x
key = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
order = np.argsort(key, kind="stable")
ts = t[order]
tlo = tan_e[ts].min(axis=1); thi = tan_e[ts].max(axis=1)
chan = np.sort(np.tan(np.deg2rad(synth.syn_vertical(128).astype(np.float64))))
m = np.tan(np.deg2rad(0.005))
def survives(lo, hi):
    i = np.searchsorted(chan, lo - m)          # first channel >= lo - margin
    return (i < len(chan)) & (chan[np.minimum(i, len(chan) - 1)] <= hi + m)
n = len(ts)
print(f"{which}: {n} triangles; a single triangle survives: {survives(tlo, thi).mean():.3f}")
prev = None
for g in (4, 16, 64, 256, 1024, 4096):
    nb = (n + g - 1) // g
    pad = nb * g - n
    L = np.concatenate([tlo, np.full(pad, np.inf)]).reshape(nb, g).min(axis=1)
    H = np.concatenate([thi, np.full(pad, -np.inf)]).reshape(nb, g).max(axis=1)
    sv = survives(L, H)
    print(f"  blocks of {g:5d}: {sv.mean():.3f} survive ({sv.sum()} blocks, {sv.sum() * g} triangles)")
