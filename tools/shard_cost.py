"""Per-rank cost of an azimuth shard (what one of N GPUs does per frame, without the collective).

For every (world, rank) asked for: the streamed frame time through the C ABI in ONE C++ loop (host_capi.cpp:
lsh_stream_frames -- CPython's per-call overhead, 12-16 us per frame, is otherwise what a small shard measures) with one
frame in flight, with three (LS_OPT_PIPELINE = 2) and with three as captured frame graphs (LS_OPT_FRAME_GRAPH); and the
device-side stage times by hipEvents (trace = k_cull + k_project, trace_aux = finish, pack; every bracket carries ~3 us
of barrier packets).

usage: W=syn128x1m|syn128x10m shard_cost.py [LS_OPT_BLOCK_CULL: 0 off, 1 on, 2 auto] [worlds, e.g. 1,8] [ranks per world: all|two]
"""
import ctypes as C
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from lidarshooter_amd import capi, hostapi, shards
import bench

bench.pin_to_gpu_numa_node(0)
sensor, meshes = bench.build_workload(os.environ.get("W", "syn128x1m"))
dev = torch.device("cuda", 0)
dm = [(n, torch.from_numpy(np.ascontiguousarray(v, np.float32)).to(dev), torch.from_numpy(np.ascontiguousarray(t, np.uint32).view(np.int32)).to(dev), v.shape[0], t.shape[0]) for n, v, t in meshes]
tr = capi.Tracer(sensor["vertical"], sensor["h_begin"], sensor["h_end"], int(sensor["h_count"]), sensor["Rinv"], sensor["t"])
tr.setOption(capi.LS_OPT_ENGINE, 2)
for n, dv, dt, nv, nt in dm: tr.addGeometry(n, nv, nt)
for n, dv, dt, nv, nt in dm: tr.updateGeometryDeviceShared(n, capi.IDENTITY_AFFINE, dv.data_ptr(), 12, dt.data_ptr())
cull = int(sys.argv[1]) if len(sys.argv) > 1 else 2
tr.setOption(capi.LS_OPT_BLOCK_CULL, cull)
L, h = tr.L, tr.h
HL = hostapi.load()
f32p = C.POINTER(C.c_float)
HL.lsh_stream_frames.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(f32p), C.POINTER(C.c_uint), C.c_uint,
                                 C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_uint, C.c_uint, C.c_uint, C.c_uint]
ident = (C.c_float * 12)(*[float(x) for x in capi.IDENTITY_AFFINE])
names = (C.c_char_p * len(dm))(*[n.encode() for n, *_ in dm])
aff = (f32p * len(dm))(*[C.cast(ident, f32p) for _ in dm])
n_aff = (C.c_uint * len(dm))(*[1 for _ in dm])
none3 = (C.c_void_p * 3)()


def stream(first, n):
    if HL.lsh_stream_frames(h, names, aff, n_aff, len(dm), none3, none3, none3, 0, 0, first, n) < 0:
        raise RuntimeError(tr.last_error())


def streamed_us(frames=3000, windows=5):
    stream(0, 600)
    tr.synchronize()
    best = []
    for _ in range(windows):
        t0 = time.perf_counter()
        stream(0, frames)
        tr.synchronize()
        best.append((time.perf_counter() - t0) / frames * 1e6)
    return float(np.median(best)), float(min(best))


H = int(sensor["h_count"])
worlds = [int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else (1, 8)
which = sys.argv[3] if len(sys.argv) > 3 else "two"
modes = os.environ.get("MODES", "one,three,graph").split(",")   # (under rocprofv3: one mode per run, so that the kernel averages mean something)


HL.lsh_stream_frames_timed.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(f32p), C.POINTER(C.c_uint), C.c_uint, C.c_uint, C.c_uint, C.POINTER(C.c_double)]


def host_us(frames=24, reps=20):
    """host time per frame of the enqueue alone: short bursts into an idle queue"""
    ts = []
    for _ in range(reps):
        tr.synchronize()
        t0 = time.perf_counter()
        stream(0, frames)
        ts.append((time.perf_counter() - t0) / frames * 1e6)
        tr.synchronize()
    if os.environ.get("HOST_SPLIT"):   # the same bursts with a clock around every call: pose updates + commit + trace
        ns = (C.c_double * 3)()
        acc = np.zeros(3)
        for _ in range(reps):
            tr.synchronize()
            HL.lsh_stream_frames_timed(h, names, aff, n_aff, len(dm), 0, frames, ns)
            acc += np.array(list(ns)) / frames / 1e3
        tr.synchronize()
        print("    host split per frame: updates %.2f + commit %.2f + trace %.2f us" % tuple(acc / reps), flush=True)
    return float(np.median(ts))

print("workload %s, LS_OPT_BLOCK_CULL %d" % (os.environ.get("W", "syn128x1m"), cull), flush=True)
for world in worlds:
    ranks = range(world) if which == "all" else sorted({0, world // 2})
    if os.environ.get("RANKS"): ranks = [int(x) for x in os.environ["RANKS"].split(",")]
    for rank in ranks:
        first, n = shards.shard_columns(H, world, rank)
        tr.setOption(capi.LS_OPT_FRAME_GRAPH, 0)
        tr.setOption(capi.LS_OPT_PIPELINE, 0)
        tr.setShard(first, n)
        out = ["world %d rank %d: streamed per frame:" % (world, rank)]
        if "one" in modes:
            one = streamed_us(1500, 3)
            out.append("one in flight %.2f us (host %.2f);" % (one[0], host_us()))
            tr.setOption(capi.LS_OPT_TIMING, 1); tr.timings()
            stream(0, 100)
            tm = tr.timings(); tr.setOption(capi.LS_OPT_TIMING, 0)
            out.append("device by events: trace %.2f + finish %.2f + pack %.2f us;" % (tm["trace"] * 1e3, tm["trace_aux"] * 1e3, tm["pack"] * 1e3))
        tr.setOption(capi.LS_OPT_PIPELINE, 2)
        if "three" in modes:
            three = streamed_us()
            out.append("three in flight %.2f us (best window %.2f, host %.2f);" % (three[0], three[1], host_us()))
        if "graph" in modes:
            tr.setOption(capi.LS_OPT_FRAME_GRAPH, 1)
            c0 = [tr.info(w) for w in (capi.LS_INFO_FRAME_GRAPH_CAPTURES, capi.LS_INFO_FRAME_GRAPH_REPLAYS, capi.LS_INFO_FRAME_GRAPH_PATCHES)]
            graph = streamed_us()
            c1 = [tr.info(w) for w in (capi.LS_INFO_FRAME_GRAPH_CAPTURES, capi.LS_INFO_FRAME_GRAPH_REPLAYS, capi.LS_INFO_FRAME_GRAPH_PATCHES)]
            out.append("three as frame graphs %.2f us (best %.2f, host %.2f; %d captures, %d replays, %d patches)" %
                       (graph[0], graph[1], host_us(), c1[0] - c0[0], c1[1] - c0[1], c1[2] - c0[2]))
            tr.setOption(capi.LS_OPT_FRAME_GRAPH, 0)
        tr.setOption(capi.LS_OPT_PIPELINE, 0)
        tr.synchronize()
        print(" ".join(out), flush=True)
