"""Why k_trace_inst takes what it takes (round 6): per-ray node fetches of the hierarchy actually built (classic mode: the
sensor-frame tree, downloaded; the CPU walks it in the kernel's order: oracle.fat_traverse_stats), then the kernel's
scheduling replayed on those numbers -- 2048 resident waves, a wave takes 64 consecutive columns of one ring from its XCD's
queue when at most 8 of its lanes are still busy, a trip advances every busy lane by one node.  Prints per-ring statistics, the
makespan in trips of the replay against the perfectly balanced one, and the same with other orders / refill rules.
usage (GPU box): python tools/trace_balance.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from lidarshooter_amd import capi, synth
import bench
from oracle import oracle as O

O.build()
sensor, meshes = bench.build_workload(os.environ.get("W", "syn128x1m"))
V, H = int(sensor["vertical"].shape[0]), int(sensor["h_count"])
tr = capi.Tracer(sensor["vertical"], sensor["h_begin"], sensor["h_end"], H, sensor["Rinv"], sensor["t"])
tr.setOption(capi.LS_OPT_ENGINE, 1)
tr.setOption(capi.LS_OPT_BVH_INSTANCED, 0)
for n, v, t in meshes:
    tr.addGeometry(n, v.shape[0], t.shape[0])
    tr.updateGeometry(n, capi.IDENTITY_AFFINE, v, t)
assert tr.commitScene() == 0
nodes, tri, leaf = tr.downloadBvh()
tr.close()
base = O.load_sensor(os.path.join(ROOT, "tests", "golden", "data", "config", "hesai-pandar-XT-32-lidar_0000.json"))
s = O.Sensor(uid="syn", vertical=np.asarray(sensor["vertical"], np.float32), h_begin=np.float32(sensor["h_begin"]), h_end=np.float32(sensor["h_end"]), h_count=H,
             R=base.R, Rinv=base.Rinv, t=base.t)
dirs = O.ray_dirs(s)
per_ray = np.zeros(V * H, np.uint32)
t, gid, stats = O.fat_traverse_stats(nodes, tri, leaf, dirs, per_ray)
n = per_ray.reshape(V, H).astype(np.int64) + (gid.reshape(V, H) != O.INVALID) * 1   # + a trip for the leaf test of a hit
print("node fetches per ray: mean %.2f, max %d; rays above 40: %d, above 80: %d" % (per_ray.mean(), per_ray.max(), (per_ray > 40).sum(), (per_ray > 80).sum()))
order = np.argsort(np.asarray(sensor["vertical"]))[::-1]     # rings from the highest down (the kernel's order)
print("ring (elevation): mean / max node fetches; mean of the 64-column batches' maxima")
for v in order[::8]:
    bm = n[v].reshape(-1, 64).max(axis=1)
    print("  %3d (%+6.2f deg): %6.2f / %3d ; batch max mean %6.2f, largest %3d" % (v, sensor["vertical"][v], n[v].mean(), n[v].max(), bm.mean(), bm.max()))


np.save(os.path.join(ROOT, "gpurun_out", os.environ.get("OUT", "."), "trace_balance_nodes_per_ray.npy"), n.astype(np.uint8))
import heapq


def replay(ring_order, refill_idle=56, waves=2048, queues=8, end_factor=0.0, end_idle=16, end_chunk=64):
    """-> (makespan in trips, mean busy trips per wave).  end_factor: once fewer than end_factor * waves * 64 rays are left in
    the queues, a wave refills as soon as end_idle of its lanes are idle and takes at most end_chunk rays at a time."""
    qs = []
    for x in range(queues):
        first, width = x * H // queues, (x + 1) * H // queues - x * H // queues
        qs.append(np.concatenate([n[v, first:first + width] for v in ring_order]))
    heads = [0] * queues
    left = sum(len(q) for q in qs)
    heap = [(0, w) for w in range(waves)]
    lanes = {w: np.zeros(0, np.int64) for w in range(waves)}
    home = {w: w % queues for w in range(waves)}
    heapq.heapify(heap)
    end, total = 0, 0
    while heap:
        tnow, w = heapq.heappop(heap)
        rem = lanes[w]
        rem = rem[rem > 0]
        ending = left < end_factor * waves * 64
        idle_needed = end_idle if ending else refill_idle
        if left > 0 and 64 - len(rem) >= idle_needed:
            want = min(64 - len(rem), end_chunk if ending else 64)
            for k in range(queues):
                q = (home[w] + k) % queues
                if heads[q] < len(qs[q]):
                    got = qs[q][heads[q]:heads[q] + want]
                    heads[q] += len(got)
                    left -= len(got)
                    home[w] = q
                    rem = np.concatenate([rem, got[got > 0]])
                    if not len(rem):      # nothing but free rays: one trip to find out, then again
                        total += 1
                        lanes[w] = rem
                        heapq.heappush(heap, (tnow + 1, w))
                        rem = None
                    break
            if rem is None:
                continue
        if not len(rem):
            end = max(end, tnow)
            continue
        srt = np.sort(rem)
        ending = left < end_factor * waves * 64
        idle_needed = end_idle if ending else refill_idle
        keep = 64 - idle_needed               # busy lanes at which the wave refills
        step = int(srt[-keep - 1]) if (left > 0 and len(srt) > keep) else int(srt[-1])
        step = max(step, 1)
        lanes[w] = rem - step
        total += step
        heapq.heappush(heap, (tnow + step, w))
    return end, total / waves


ground = [v for v in order if n[v].max() > 1]
sky = [v for v in order if n[v].max() <= 1]
print("rings that meet the scene: %d of %d" % (len(ground), V))
for name, ro in (("highest ring first (shipped)", list(order)), ("scene rings first, sky last", ground + sky), ("lowest first", list(order[::-1]))):
    for kw in ({}, {"refill_idle": 64}, {"end_factor": 0.5}, {"end_factor": 1.0}, {"end_factor": 2.0}, {"end_factor": 1.0, "end_idle": 32}, {"end_factor": 1.0, "end_chunk": 16},
               {"end_factor": 2.0, "end_chunk": 16}, {"end_factor": 1.0, "end_idle": 8, "end_chunk": 16}, {"waves": 1024}, {"waves": 3072}):
        mk, mean = replay(ro, **kw)
        print("%-30s %-55s makespan %4d trips, mean per wave %6.1f, balance %.2f" % (name, str(kw), mk, mean, mean / mk), flush=True)
