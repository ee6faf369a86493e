"""Offline (numpy, no GPU): how the raster cells of an azimuth shard's triangles fall onto waves of 64 triangles.
For SYN-1M / SYN-10M under SYN-128, rank r of 8: the in-sector triangles that meet a ring (what group culling leaves,
in Morton order), their footprint cells (channels in the elevation band x columns in the azimuth arc, inside the
sector), and for waves of 64 consecutive survivors the distribution of 64-cell trips -- plain, with footprints above a
threshold sent to the gather queue, and with the survivors dealt to the waves at a stride.
usage: shard_balance.py [1m|10m] [rank]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import oracle as O
from lidarshooter_amd import synth
DATA = os.path.join(ROOT, "tests", "golden", "data")
s = O.load_sensor(os.path.join(DATA, "config", "hesai-pandar-XT-32-lidar_0000.json"))
which = sys.argv[1] if len(sys.argv) > 1 else "1m"
rank = int(sys.argv[2]) if len(sys.argv) > 2 else 4
v, t = synth.syn_10m() if which == "10m" else synth.syn_1m()
vs = O.transform_vertices(v, O.IDENTITY_AFFINE, s).astype(np.float64)
H, V = 4096, 128
step = 360.0 / (H - 1)
az = np.rad2deg(np.arctan2(vs[:, 1], vs[:, 0])) % 360.0
el = np.rad2deg(np.arctan2(vs[:, 2], np.hypot(vs[:, 0], vs[:, 1])))
rho = np.hypot(vs[:, 0], vs[:, 1])
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
# azimuth arc of a triangle (small triangles: no wrap handling beyond the 0/360 seam)
a = az[ts]
amin, amax = a.min(axis=1), a.max(axis=1)
wrap = (amax - amin) > 180.0
a2 = np.where(a < 180.0, a + 360.0, a)
amin = np.where(wrap, a2.min(axis=1), amin); amax = np.where(wrap, a2.max(axis=1), amax)
emin, emax = el[ts].min(axis=1), el[ts].max(axis=1)
chan = np.sort(synth.syn_vertical(V).astype(np.float64))
m = 0.005
i0 = np.searchsorted(chan, emin - m, side="left"); i1 = np.searchsorted(chan, emax + m, side="right")
nch = np.maximum(i1 - i0, 0)
naz = H // 8
c0, c1 = rank * naz, rank * naz + naz - 1
def cols(lo_deg, hi_deg):
    l = np.ceil((lo_deg - m) / step - 0.0625); h = np.floor((hi_deg + m) / step + 0.0625)
    l = np.maximum(l, c0); h = np.minimum(h, c1)
    return np.maximum(h - l + 1, 0)
ncol = cols(amin, amax) + np.where(wrap | (amax >= 360.0), cols(amin - 360.0, amax - 360.0), 0)
cells = (nch * ncol).astype(np.int64)
in_sector = ncol > 0
print(f"{which}, rank {rank} of 8: {in_sector.sum()} of {len(ts)} triangles reach the sector, {((cells > 0)).sum()} have cells, {cells.sum()} cells in all "
      f"({cells.sum() / 64:.0f} dense trips); cells per triangle with cells: median {np.median(cells[cells > 0]):.0f}, 99 % {np.percentile(cells[cells > 0], 99):.0f}, max {cells.max()}")
# group culling: groups of 4 sorted triangles that reach the sector and meet a ring (the optimum)
g = len(ts) // 4
keep_g = (cells[:g * 4].reshape(g, 4) > 0).any(axis=1)
surv = np.repeat(keep_g, 4)
sc = cells[:g * 4][surv]
print(f"  groups of 4 with a cell: {keep_g.sum()} ({keep_g.sum() * 4} triangles, {np.ceil(keep_g.sum() / 16):.0f} waves of 16 groups)")
def report(name, c, big):
    queued = c > big
    cc = np.where(queued, 0, c)
    pad = (-len(cc)) % 64
    w = np.concatenate([cc, np.zeros(pad, np.int64)]).reshape(-1, 64).sum(axis=1)
    trips = np.ceil(w / 64)
    print(f"  {name:34s} big > {big:5d}: queue {queued.sum():6d}; trips per wave mean {trips.mean():5.2f}, 90 % {np.percentile(trips, 90):4.0f}, 99 % {np.percentile(trips, 99):4.0f}, max {trips.max():4.0f}; waves with > 4 trips: {(trips > 4).sum()}")
for big in (128, 64, 32, 16, 8):
    report("consecutive survivors", sc, big)
n_w = int(np.ceil(len(sc) / 64))
# dealt at a stride: wave r takes groups r, r + W, r + 2 W ...
gsc = np.concatenate([sc, np.zeros((-len(sc)) % (64 * 1), np.int64)])
ng = len(gsc) // 4
W = int(np.ceil(ng / 16))
padg = W * 16 - ng
gs = np.concatenate([gsc.reshape(ng, 4), np.zeros((padg, 4), np.int64)]).reshape(16, W, 4).transpose(1, 0, 2).reshape(-1)
for big in (128, 32):
    report("groups dealt at a stride of W", gs, big)
# cells by distance
r3 = rho[ts].min(axis=1)
for d0, d1 in ((0, 5), (5, 10), (10, 20), (20, 40), (40, 80)):
    sel = (r3 >= d0) & (r3 < d1) & in_sector
    print(f"  rho in [{d0:2d}, {d1:2d}) m: {sel.sum():7d} triangles in sector, {(cells[sel] > 0).sum():6d} with cells, {cells[sel].sum():8d} cells, mean {cells[sel].sum() / max(1, (cells[sel] > 0).sum()):6.1f} per footprint")
