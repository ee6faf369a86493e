"""Per-wave timeline of one k_project launch (build variant -DLS_EXP_TIMELINE): when waves start and end, where their
time goes.  usage: exp_timeline.py gpurun_out/exp/timeline.bin"""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8)
a = a[a[:, 0] != 0]
n = len(a)
w0, w1 = a[:, 0].astype(np.int64), a[:, 1].astype(np.int64)
c = a[:, 2:7].astype(np.int64)
total = (a[:, 7] >> np.uint64(32)).astype(np.int64)
xcc = ((a[:, 7] >> np.uint64(28)) & np.uint64(0xF)).astype(np.int64)
hw = (a[:, 7] & np.uint64(0x0FFFFFFF)).astype(np.int64)
t0 = w0.min()
start = (w0 - t0) / 100.0    # us (100 MHz)
end = (w1 - t0) / 100.0
life = end - start
cyc = c[:, 4] - c[:, 0]
ghz = np.median(cyc[life > 1] / (life[life > 1] * 1e3))
print(f"waves {n}; kernel span {end.max():.2f} us; clock ~{ghz:.2f} GHz (cycles / realtime)")
print("start time  percentiles 10/50/90/99/100: " + " ".join(f"{np.percentile(start, p):.2f}" for p in (10, 50, 90, 99, 100)))
print("end time    percentiles 10/50/90/99/100: " + " ".join(f"{np.percentile(end, p):.2f}" for p in (10, 50, 90, 99, 100)))
print("lifetime us percentiles 10/50/90/99/100: " + " ".join(f"{np.percentile(life, p):.2f}" for p in (10, 50, 90, 99, 100)) + f"  mean {life.mean():.2f}")
seg = np.diff(c, axis=1) / (ghz * 1e3)
for i, name in enumerate(("prologue+loads issued+table staging+barrier", "wait for index/vertex loads", "transform+footprint+staging", "scan+trips")):
    print(f"  {name:45s} mean {seg[:, i].mean():.2f} us  p50 {np.percentile(seg[:, i], 50):.2f}  p90 {np.percentile(seg[:, i], 90):.2f}  p99 {np.percentile(seg[:, i], 99):.2f}")
trips = (total + 63) // 64
print("cells per wave: mean %.1f p50 %d p90 %d p99 %d max %d; trips mean %.2f max %d" % (total.mean(), np.percentile(total, 50), np.percentile(total, 90), np.percentile(total, 99), total.max(), trips.mean(), trips.max()))
for k in range(0, int(trips.max()) + 1):
    m = trips == k
    if m.sum() > 20: print(f"  trips {k}: waves {m.sum():6d}  scan+trips mean {seg[m, 3].mean():.2f} us  lifetime {life[m].mean():.2f}")
# occupancy over time
ts = np.arange(0, end.max(), 0.5)
occ = [(np.sum((start <= t) & (end > t))) for t in ts]
print("resident waves over time (0.5 us steps): " + " ".join(str(o) for o in occ))
# first round vs second: waves started after 3 us
late = start > 3.0
print(f"waves started in the first 3 us: {np.sum(~late)} (mean life {life[~late].mean():.2f}); later: {np.sum(late)} (mean life {life[late].mean():.2f})")
print("per XCC waves:", np.bincount(xcc))
