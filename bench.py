#!/usr/bin/env python3
"""bench.py -- headline benchmark of the LiDAR tracer hot path on MI355X.

Metric (BASELINE.json): Mrays/s (and LiDAR frames/s) for a 128-channel x 4096-azimuth sensor over
a 1M-triangle mesh (BASELINE.json configs[3] = BASELINE.md section 4 config 4, "SYN-128 x SYN-1M").
One "step" = one LiDAR frame = the reference's per-frame sequence (MeshProjector.cpp:446-464):
updateGeometry(every mesh) + commitScene (vertex transform + full BVH rebuild) + traceScene
(ray generation + closest hit + point packing), with the mesh already resident in HBM.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  N > 1: one rank per GPU over RCCL, either launched by the driver through torch.distributed.run (RANK / WORLD_SIZE in
  the environment) or -- when WORLD_SIZE is not set -- by this script itself, which then starts
  `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child process BEFORE anything here touches a
  GPU and passes the child's one JSON line through.  It never measures fewer GPUs than --gpus asks for.  Rays are
  sharded by azimuth sector, every rank keeps a scene replica, and one all-gather of the fixed-capacity hit-record
  slots per frame (count word in the slot header) collects the cloud.

Rank 0 prints ONE JSON line (see README/DESIGN.md for the extra keys).
"""
from __future__ import annotations

import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import ctypes as C_  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from lidarshooter_amd import capi, hostapi, shards, synth  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PROFILE_TAG = "r06"     # profiles/<tag>_<engine>_hbm.json: the rocprofv3 PMC summary this round's kernels were profiled into
NODE_BYTES, TRI_BYTES, RAY_OUT_BYTES = 64, 48, 8  # DESIGN.md "algorithmic bytes" (BVH engine)
DATA = os.path.join(ROOT, "tests", "golden", "data")
HOST_NUMA = None
SHIM = False
LIBRARY = None   # ensure_fresh_library(): which sources the loaded library was built from


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--min-ms", type=float, default=50.0,
                    help="keep timing windows of --steps frames until this much has been timed; the median window is reported")
    ap.add_argument("--prime-ms", type=int, default=50,
                    help="untimed frames for this long before the W warm-up steps (clocks, caches)")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="one frame in flight (LS_OPT_PIPELINE off)")
    ap.add_argument("--pipeline", type=int, default=0, choices=[0, 1, 2],
                    help="single GPU: LS_OPT_PIPELINE mode, 1 = two frames in flight on one stream (finish + pack ride in "
                         "the next frame's launch: one launch per frame), 2 = three frames in flight on three streams; "
                         "0 (default) = 2 for scenes of 200 000 triangles or more, 1 for small scenes (host-bound)")
    ap.add_argument("--workload", default="syn128x1m", choices=["syn128x1m", "syn128x10m", "xt32", "cfg5", "syn128x2m", "syn128x3m", "syn128x5m", "syn128x1500k"],
                    help="syn128x1m = the headline config; syn128x10m = BASELINE.json configs[4]'s scene size; "
                         "cfg5 = configs[4] itself: one SYN-128 sensor per GPU on a 20 m circle over SYN-10M + ben "
                         "animated by config/trajectory.json, replicas only (weak scaling, no collective)")
    ap.add_argument("--leaf", type=int, default=0, help="triangles per BVH leaf (0 = library default)")
    ap.add_argument("--engine", default="auto", choices=["auto", "bvh", "projection"],
                    help="closest-hit engine (auto = the library default: sensor-space projection)")
    ap.add_argument("--reregister", action="store_true",
                    help="hand the (unchanged) device mesh over again every frame (ls_update_geometry_device_shared) instead of "
                         "a transform-only update: the library must then assume new vertices and redo its per-block bounds")
    ap.add_argument("--no-cull", action="store_true", help="LS_OPT_BLOCK_CULL off")
    ap.add_argument("--classic-bvh", action="store_true", help="BVH engine: LS_OPT_BVH_INSTANCED off (one hierarchy in the sensor frame, refitted every frame)")
    ap.add_argument("--cull", action="store_true", help="LS_OPT_BLOCK_CULL on (default: the library's auto rule)")
    ap.add_argument("--bvh-rebuild", action="store_true",
                    help="BVH engine: the reference's literal per-frame sequence (OptixTracer.cpp:517-571 with OPERATION_BUILD, EmbreeTracer.cpp:290-295): "
                         "one hierarchy in the sensor frame, built from scratch at every commit (LS_OPT_BVH_INSTANCED 0, LS_OPT_BVH_REFIT 0)")
    ap.add_argument("--no-also", action="store_true",
                    help="single GPU, headline workload: skip the other configurations the line reports under also_measured (the BVH engine "
                         "poses-only and with a rebuild per frame, SYN-10M, configs[4] per GPU)")
    ap.add_argument("--multi", default="sharded", choices=["interleaved", "sharded"],
                    help="N > 1: which way of spreading the frame stream over the GPUs is reported as `value` (the other one is "
                         "measured too and reported under also_measured): sharded (default, BASELINE.json's split) = azimuth "
                         "sectors of every frame + one all-gather of hit-record slots per frame (strong scaling: K frames in "
                         "all); interleaved = whole frames per rank, no collective (weak scaling: K frames per rank)")
    ap.add_argument("--multi-driver", default="c", choices=["c", "torch"],
                    help="sharded mode: c (default) = include/lidarshooter_group.h -- the frame loop, the RCCL all-gather and the "
                         "rebuild of the cloud run in C, three frames in flight per rank; torch = torch.distributed issues the "
                         "all-gather (two frames in flight; the path LS_BENCH_REHEARSAL rehearses over gloo)")
    ap.add_argument("--group-flags", type=int, default=0,
                    help="sharded mode, C driver: LS_GROUP_FLAG_* of include/lidarshooter_group.h (0 = per-set communicators and streams, "
                         "every frame one captured HIP graph; 1 = round 3's arrangement: one communicator on a collective stream; 2 = per-set, plain launches)")
    ap.add_argument("--frame-graph", type=int, default=0, choices=[0, 1],
                    help="single GPU: LS_OPT_FRAME_GRAPH (the three launches of a frame as one captured HIP graph)")
    ap.add_argument("--spawn-check", action="store_true",
                    help="start the ranks, have them meet (one all-reduce), print how many did, and exit: no GPU work (tests)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dropin", action="store_true", help="skip the ITracer-adapter (host buffers, PointCloud2) legs")
    ap.add_argument("--cpu-frames", type=int, default=6)
    ap.add_argument("--breakdown", action="store_true", help="extra pass with per-stage hipEvent timings")
    return ap.parse_args()


def build_workload(name, rank=0):
    """-> (sensor dict for capi.Tracer, list of (mesh name, verts f32[n,3], tris u32[m,3]))."""
    dev = hostapi.LidarDevice(os.path.join(DATA, "config", "hesai-pandar-XT-32-lidar_0000.json"))
    d = dev.desc()
    if name == "xt32":
        g = hostapi.PolygonMesh(os.path.join(DATA, "mesh", "ground.stl"))
        b = hostapi.PolygonMesh(os.path.join(DATA, "mesh", "ben.stl"))
        return d, [("ground", g.points(), g.polygons()), ("face", b.points(), b.polygons())]
    d = dict(d)
    d["vertical"] = synth.syn_vertical(128)      # +15 .. -25 deg
    d["h_begin"], d["h_end"], d["h_count"] = np.float32(0.0), np.float32(360.0), 4096
    if os.environ.get("LS_BENCH_SYN_CHANNELS"):   # (tools/cull_crossover.sh: the same span with fewer, wider-spaced rings)
        d["vertical"] = synth.syn_vertical(int(os.environ["LS_BENCH_SYN_CHANNELS"]))
    between = {"syn128x1500k": (1224, 612), "syn128x2m": (1414, 707), "syn128x3m": (1732, 866), "syn128x5m": (2236, 1118)}   # BASELINE.md section 4's formula, 2 : 1 cells, other sizes
    v, t = synth.grid_mesh(*between[name]) if name in between else (synth.syn_10m() if name in ("syn128x10m", "cfg5") else synth.syn_1m())
    if name == "cfg5":
        ang = 2.0 * np.pi * rank / 8.0            # 8 sensor poses on a 20 m circle around lidar_0000's
        d["t"] = (d["t"] + np.array([20.0 * np.cos(ang), 20.0 * np.sin(ang), 0.0], np.float32)).astype(np.float32)
        b = hostapi.PolygonMesh(os.path.join(DATA, "mesh", "ben.stl"))
        return d, [("ground", v, t), ("face", b.points(), b.polygons())]
    return d, [("ground", v, t)]


def _affine(lin, ang):
    """Translation(lin) * Rz * Ry * Rx as row-major 3x4 (MeshTransformer.cpp:467-477), built on the host."""
    cx, sx, cy, sy, cz, sz = np.cos(ang[0]), np.sin(ang[0]), np.cos(ang[1]), np.sin(ang[1]), np.cos(ang[2]), np.sin(ang[2])
    rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]], np.float32)
    ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]], np.float32)
    rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]], np.float32)
    m = (rz @ ry @ rx).astype(np.float32)
    return np.concatenate([m, np.asarray(lin, np.float32).reshape(3, 1)], axis=1).reshape(12)


def pin_to_gpu_numa_node(dev_index: int):
    """The frame loop is a host thread that writes packets and doorbells to ONE GPU: on a two-socket host (the pool's boxes: 2 x
    EPYC 9575F, the cgroup grants 16 CPUs' worth of time anywhere on 256) the scheduler puts it on either socket, and from
    far one -- or on CPUs it shares with the box's other tenants -- a frame's three launches have been seen to cost 13.7 us of
    host time instead of 8.3 - 10.6, enough for the three streams to run dry between frames: 18.3 - 18.7 us per frame instead
    of 16 - 17 (the process-to-process spread of rounds 2 - 4; lsbench and tools/variance_probe.py, which
    keep the GPU's queues thousands of frames deep, never show it).  So the process is confined to the CPUs of the GPU's own NUMA
    node, as numactl would -- all of its threads, the runtime's helper threads that exist by then included: 8 of 8 processes at
    16.0 - 16.3 us (16.2 - 16.9 with the calling thread alone) against 3 of 6 above 18.3 without.  Returns what it did."""
    try:
        # the device's PCI address from the HIP runtime (does not initialise more than torch already has), then sysfs
        import ctypes
        buf = ctypes.create_string_buffer(64)
        node = None
        try:
            hip = ctypes.CDLL("libamdhip64.so.7")   # (by soname: the runtime torch has loaded already, never a second copy)
            if hip.hipDeviceGetPCIBusId(buf, 64, dev_index) == 0:
                v = int(open(f"/sys/bus/pci/devices/{buf.value.decode().lower()}/numa_node").read().strip())
                node = v if v >= 0 else None
        except Exception:
            node = None
        if node is None:
            return {"pinned": False, "why": "no numa_node for the GPU in sysfs"}
        cpus = set()
        for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        cpus &= set(os.sched_getaffinity(0))
        if not cpus:
            return {"pinned": False, "why": "the GPU's node has no CPU this process may run on", "node": node}
        # every thread the process has by now (sched_setaffinity(0, ...) is the calling thread alone; the runtime's helper
        # threads that exist already would stay where they are), and, by inheritance, every thread made later
        tids = 0
        for t in os.listdir("/proc/self/task"):
            try:
                os.sched_setaffinity(int(t), cpus)
                tids += 1
            except OSError:
                pass
        os.sched_setaffinity(0, cpus)
        return {"pinned": True, "node": node, "cpus": len(cpus), "threads": tids}
    except Exception as e:   # never in the way of the measurement
        return {"pinned": False, "why": repr(e)}


def kernel_source_sha() -> str:
    """sha256 over the library's sources: a committed profile only speaks for the kernels it was taken from
    (the GPU box has no .git, so the commit id is not available there)."""
    import hashlib
    hsh = hashlib.sha256()
    src = os.path.join(ROOT, "lidarshooter_amd", "csrc")
    for fn in sorted(os.listdir(src)):
        if fn.endswith((".hip", ".h", ".cpp")):
            hsh.update(fn.encode())
            hsh.update(open(os.path.join(src, fn), "rb").read())
    return hsh.hexdigest()[:16]


def library_source_sha_on_disk() -> str | None:
    """the hash the built library carries (ls_source_hash(): 'LS_SOURCE_HASH=<16 hex digits>' inside the file), read WITHOUT loading it"""
    try:
        blob = open(capi.LIB_PATH, "rb").read()
        at = blob.index(b"LS_SOURCE_HASH=")
        return blob[at + 15:at + 31].decode()
    except Exception:
        return None


def ensure_fresh_library() -> dict:
    """the GPU box runs binaries built elsewhere and `make` says "nothing to be done" by time stamps -- a stale
    .so could speak for newer sources.  Before anything is loaded: the library's own source hash against the sources here; a
    mismatch rebuilds (make, then make -B) and the line records what happened.  A library that is still stale is reported as such
    (`library_is_stale`), never silently measured as if it were these sources."""
    want = kernel_source_sha()
    had = library_source_sha_on_disk()
    rec = {"sources": want, "library": had, "rebuilt": False}
    if had != want and os.environ.get("LS_LIB_PATH") is None:
        import subprocess
        src = os.path.join(ROOT, "lidarshooter_amd", "csrc")
        for extra in ([], ["-B"]):
            subprocess.call(["make", "-C", src, "-j8"] + extra, stdout=sys.stderr, stderr=sys.stderr)
            subprocess.call(["make", "-C", os.path.join(ROOT, "lidarshooter_amd", "host")] + extra, stdout=sys.stderr, stderr=sys.stderr)
            rec["rebuilt"] = True
            rec["library"] = library_source_sha_on_disk()
            if rec["library"] == want:
                break
    rec["library_is_stale"] = rec["library"] != want
    return rec


def embree_probe(sensor, meshes, total_rays, ncpu):
    """Baseline A: if the box has Embree 3, time rtcCommitScene + rtcIntersect1M on all cores over
    the same (pre-transformed) scene and diff its hits against `gpu_hits` later.  Returns {"present": False} otherwise."""
    import ctypes as C
    try:
        E = C.CDLL("libembree3.so.3")
    except OSError:
        try:
            E = C.CDLL("libembree3.so")
        except OSError:
            return {"present": False, "note": "libembree3.so.3 not found on this box: Embree 3.13.4 is an un-vendored binary "
                                              "of the reference (.devcontainer/Dockerfile:24-27)"}
    try:
        import threading
        from oracle import oracle as O
        s = O.Sensor(uid="bench", vertical=sensor["vertical"], h_begin=sensor["h_begin"], h_end=sensor["h_end"],
                     h_count=sensor["h_count"], R=np.eye(3, dtype=np.float32).reshape(9), Rinv=sensor["Rinv"], t=sensor["t"])
        vp = C.c_void_p
        for f in ("rtcNewDevice", "rtcNewScene", "rtcNewGeometry", "rtcSetSharedGeometryBuffer"):
            getattr(E, f).restype = vp
        E.rtcNewDevice.argtypes = [C.c_char_p]
        E.rtcNewScene.argtypes = [vp]
        E.rtcNewGeometry.argtypes = [vp, C.c_int]
        E.rtcSetSharedGeometryBuffer.argtypes = [vp, C.c_int, C.c_uint, C.c_int, vp, C.c_size_t, C.c_size_t, C.c_size_t]
        E.rtcSetSharedGeometryBuffer.restype = None
        E.rtcCommitGeometry.argtypes = [vp]
        E.rtcAttachGeometry.argtypes = [vp, vp]
        E.rtcAttachGeometry.restype = C.c_uint
        E.rtcCommitScene.argtypes = [vp]
        E.rtcIntersect1M.argtypes = [vp, vp, vp, C.c_uint, C.c_size_t]
        dev = E.rtcNewDevice(None)
        scene = E.rtcNewScene(dev)
        keep = []
        for _, v, t in meshes:   # EmbreeTracer.cpp:140-176: FLOAT3 vertices (stride 12), UINT3 indices
            tv = np.ascontiguousarray(O.transform_vertices(v, O.IDENTITY_AFFINE, s), np.float32)
            tv = np.concatenate([tv.reshape(-1), np.zeros(4, np.float32)])   # Embree reads 16 bytes per vertex
            ti = np.ascontiguousarray(t, np.uint32)
            g = E.rtcNewGeometry(dev, 0)
            E.rtcSetSharedGeometryBuffer(g, 1, 0, 0x9003, tv.ctypes.data, 0, 12, v.shape[0])      # RTC_BUFFER_TYPE_VERTEX, RTC_FORMAT_FLOAT3
            E.rtcSetSharedGeometryBuffer(g, 0, 0, 0x5003, ti.ctypes.data, 0, 12, t.shape[0])      # RTC_BUFFER_TYPE_INDEX, RTC_FORMAT_UINT3
            E.rtcCommitGeometry(g)
            E.rtcAttachGeometry(scene, g)
            keep.append((tv, ti))
        t0 = time.perf_counter()
        E.rtcCommitScene(scene)
        build = time.perf_counter() - t0
        dirs = O.ray_dirs(s)
        n = dirs.shape[0]
        rh = np.zeros(n, np.dtype([("org", "<f4", 3), ("tnear", "<f4"), ("dir", "<f4", 3), ("time", "<f4"), ("tfar", "<f4"),
                                   ("mask", "<u4"), ("id", "<u4"), ("flags", "<u4"), ("Ng", "<f4", 3), ("u", "<f4"), ("v", "<f4"),
                                   ("primID", "<u4"), ("geomID", "<u4"), ("instID", "<u4")], align=False))
        assert rh.dtype.itemsize == 80
        rh["dir"] = dirs
        rh["tfar"] = np.inf
        rh["mask"] = 0xFFFFFFFF
        rh["geomID"] = 0xFFFFFFFF
        rh["instID"] = 0xFFFFFFFF
        ctxb = (C.c_ubyte * 32)()                      # RTCIntersectContext {flags @0, filter @8, instID[1] @16} (rtcInitIntersectContext)
        C.memmove(ctxb, (C.c_uint * 2)(0, 0), 8)
        C.memmove(C.addressof(ctxb) + 16, (C.c_uint * 1)(0xFFFFFFFF), 4)
        per = (n + ncpu - 1) // ncpu

        def work(i):
            lo, hi = i * per, min(n, (i + 1) * per)
            if lo < hi:
                E.rtcIntersect1M(scene, C.addressof(ctxb), rh.ctypes.data + 80 * lo, hi - lo, 80)
        t0 = time.perf_counter()
        th = [threading.Thread(target=work, args=(i,)) for i in range(ncpu)]
        [x.start() for x in th]
        [x.join() for x in th]
        trace = time.perf_counter() - t0
        return {"present": True, "threads": ncpu, "commit_ms": build * 1e3, "trace_ms": trace * 1e3,
                "frames_per_s": 1.0 / (build + trace), "value": total_rays / (build + trace) / 1e6, "unit": "Mrays/s",
                "hits": int((rh["geomID"] != 0xFFFFFFFF).sum()),
                "_t": rh["tfar"].copy(), "_prim": rh["primID"].copy(), "_geom": rh["geomID"].copy()}
    except Exception as e:   # a present but unusable library must not take the benchmark down
        return {"present": True, "error": repr(e)}


def cpu_baseline(sensor, meshes, frames, total_rays):
    """The oracle's CPU path (oracle/: binned-SAH BVH2, threaded single-ray traversal) timed on
    the host cores for the SAME frame definition: transform + full BVH build + trace + pack."""
    from oracle import oracle as O
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:  # a cgroup CPU quota (the GPU box gives one GPU's share of the host) bounds the useful thread count
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            ncpu = max(1, min(ncpu, int(round(int(quota) / int(period)))))
    except Exception:
        pass
    s = O.Sensor(uid="bench", vertical=sensor["vertical"], h_begin=sensor["h_begin"], h_end=sensor["h_end"],
                 h_count=sensor["h_count"], R=np.eye(3, dtype=np.float32).reshape(9), Rinv=sensor["Rinv"], t=sensor["t"])
    ml = [(i, v, t, O.IDENTITY_AFFINE) for i, (_, v, t) in enumerate(meshes)]
    dirs = O.ray_dirs(s)
    times, parts = [], []
    for _ in range(frames):
        t0 = time.perf_counter()
        scene = O.assemble_scene(s, ml)
        t1 = time.perf_counter()
        bvh = O.CpuBvh(scene, ncpu)
        t2 = time.perf_counter()
        tt, gid, _ = bvh.trace(dirs, ncpu)
        O.pack_points(tt, gid, dirs, s.H, scene)
        t3 = time.perf_counter()
        bvh.close()
        times.append(t3 - t0)
        parts.append((t1 - t0, t2 - t1, t3 - t2))
        if sum(times) > 30.0:
            break
    med = float(np.median(times))
    p = np.median(np.array(parts), axis=0)
    # Baseline C: what the reference itself does -- 4 worker threads (EmbreeTracer.cpp:306,
    # MeshTransformer.cpp:158) and every mesh re-sent + the scene re-committed every frame (MeshProjector.cpp:446-464)
    t4 = []
    for _ in range(2):
        t0 = time.perf_counter()
        scene = O.assemble_scene(s, ml)
        bvh = O.CpuBvh(scene, 4)
        tt, gid, _ = bvh.trace(dirs, 4)
        O.pack_points(tt, gid, dirs, s.H, scene)
        t4.append(time.perf_counter() - t0)
        bvh.close()
    t4 = float(np.median(t4))
    emb = embree_probe(sensor, meshes, total_rays, ncpu)
    if emb.get("present") and "_t" in emb:   # the first measurement of the 1e-4 claim against a real Embree
        hit_e, hit_o = emb["_geom"] != 0xFFFFFFFF, gid != O.INVALID
        both = hit_e & hit_o
        rel = np.abs(emb["_t"][both] - tt[both]) / np.abs(tt[both]) if both.any() else np.zeros(1)
        emb["vs_oracle"] = {"hit_set_mismatches": int((hit_e != hit_o).sum()), "max_rel_t": float(rel.max())}
        if len(meshes) == 1:   # global triangle id == primID: ids may differ only where two triangles share the hit distance
            emb["vs_oracle"]["prim_mismatches_at_unequal_t"] = int(((emb["_prim"][both] != gid[both]) & (rel > 0)).sum())
        for k in ("_t", "_prim", "_geom"):
            emb.pop(k)
    return {
        "value": total_rays / med / 1e6, "unit": "Mrays/s", "cores": ncpu, "kind": "port",
        "frames_per_s": 1.0 / med,
        "sample": f"{len(times)} full frames of the same workload (median): transform {p[0]*1e3:.0f} ms + "
                  f"binned-SAH BVH2 build {p[1]*1e3:.0f} ms + trace/pack {p[2]*1e3:.0f} ms, {ncpu} threads; "
                  "CPU restatement (Embree 3.13.4 is not installed)",
        "threads4_frames_per_s": 1.0 / t4, "threads4_value": total_rays / t4 / 1e6,
        "threads4_sample": "2 full frames, 4 threads, rebuild every frame: the reference's own threading (EmbreeTracer.cpp:306)",
        "embree": emb,
    }


def front_loaded(out: dict) -> dict:
    """The same line with its short numeric fields first and last (a record that keeps only the head or the tail of the line still
    shows them): the contract's scalars, the drop-in frame, the roofline's fractions, the CPU baseline's value -- the long
    descriptive strings and objects sit in the middle."""
    rf, cb = out.get("roofline") or {}, out.get("cpu_baseline") or {}
    short = {k: out.get(k) for k in ("value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                                     "ms_per_step_one_frame_in_flight", "ms_per_step_windows_of_1000", "dropin_ms_per_step", "dropin_static_ms_per_step",
                                     "frames_per_s") if k in out}
    short["roofline_frac"] = rf.get("frac")
    short["roofline_kernel_ms"] = rf.get("kernel_ms")
    short["roofline_issue_frac"] = (rf.get("secondary") or {}).get("issue_frac")
    short["cpu_baseline_value"] = cb.get("value")
    for a in out.get("also_measured") or []:   # (single GPU: the other configurations' step times and roofline fractions, by label)
        if isinstance(a, dict) and a.get("label") and "ms_per_step" in a:
            short["also_%s_ms" % a["label"]] = a["ms_per_step"]
            short["also_%s_frac" % a["label"]] = (a.get("roofline") or {}).get("frac")
    middle = {k: v for k, v in out.items() if k not in short}
    return {**short, **middle, "summary": dict(short)}


class Watchdog:
    """N > 1 only.  The multi-rank RCCL path -- a communicator per buffer set, three collectives in flight, a collective
    inside a captured HIP graph -- has never run on hardware (the build pool gives one GPU; DESIGN.md section 8 "untested because
    it cannot be here").  A wait on a stuck collective cannot be cancelled, so a phase that does not finish in time ends the
    job instead of holding the node until somebody else's limit: every rank says which phase it was, rank 0 still prints
    ONE line -- the measurements that did finish (the frame-interleaved split is measured first for that reason: nothing
    is exchanged on its frame path), `value` taken from them, and an "error" field that says what did not -- and the ranks
    leave with os._exit: 4 when there is such a line (a job whose watchdog fired did NOT succeed -- ADVICE round 5 -- but its
    line is there to be read), 3 when there is nothing to report."""
    EXIT_WITH_LINE, EXIT_WITHOUT = 4, 3

    def __init__(self, result_fd: int, rank: int, store=None):
        import threading
        # store: torch.distributed's key-value store.  A rank other than 0 that ends the job says so THERE first, and rank 0's
        # watchdog thread, which polls it, writes the line and leaves within a quarter of a second -- before the launcher, seeing
        # a rank leave with a non-zero code, terminates the others (it would otherwise end rank 0 before its line is out).
        self.result_fd, self.rank, self.store = result_fd, rank, store
        self.lock = threading.Lock()
        self.deadline, self.what, self.seconds, self.fallback = None, None, 0.0, None
        t = threading.Thread(target=self._run, name="bench-watchdog", daemon=True)
        t.start()

    def arm(self, what: str, seconds: float, fallback=None):
        with self.lock:
            self.what, self.seconds, self.fallback = what, seconds, fallback
            self.deadline = time.monotonic() + seconds

    def disarm(self):
        with self.lock:
            self.deadline = None

    def _run(self):
        while True:
            time.sleep(0.25)
            with self.lock:
                late = self.deadline is not None and time.monotonic() > self.deadline
                what, seconds, fallback = self.what, self.seconds, self.fallback
            if late:
                self._expire(what, seconds, fallback)
            if self.rank == 0 and self.store is not None and what is not None:
                try:
                    if self.store.check(["ls_bench_watchdog"]):
                        self._end("a peer ended the job: " + self.store.get("ls_bench_watchdog").decode(), fallback)
                except Exception:
                    pass

    def _expire(self, what, seconds, fallback):
        self._end(f"'{what}' did not finish within {seconds:.0f} s on rank {self.rank}: taken to be hung (a collective that never "
                  "completes cannot be waited out); the job ends here", fallback)

    def failed(self, error: BaseException):
        """The armed phase raised on this rank (a communicator that could not be made, a refused capture): its peers are, or
        soon will be, inside collectives this rank will never join -- same ending, without waiting for the deadline."""
        import traceback
        traceback.print_exc()
        with self.lock:
            self.deadline = None
            what, fallback = self.what, self.fallback
        self._end(f"'{what}' raised on rank {self.rank}: {error!r}; the job ends here", fallback)

    def _end(self, msg, fallback):
        print("bench.py watchdog: " + msg, file=sys.stderr, flush=True)
        if self.rank == 0:
            if fallback is not None:
                line = dict(fallback)
                line["error"] = msg + f"; `value` is the measurement that did finish ({fallback['config']['parallelism']})"
                os.write(self.result_fd, (json.dumps(front_loaded(line)) + "\n").encode())
        else:
            try:
                if self.store is not None:
                    self.store.set("ls_bench_watchdog", msg)
            except Exception:
                pass
            time.sleep(3.0)   # rank 0 writes its line before a launcher that sees a rank leave ends the others
        os._exit(self.EXIT_WITH_LINE if fallback is not None else self.EXIT_WITHOUT)


def spawn_ranks(args) -> int:
    """--gpus N without a launcher: start the N ranks as a fresh child process tree (torch.distributed.run) -- nothing in
    THIS process has touched a GPU yet (importing torch does not; torch.cuda.device_count() does not initialise one on
    this image either) -- and hand the child's stdout (rank 0's one JSON line) through.  Returns the child's exit code."""
    import socket
    import subprocess
    rehearsal = os.environ.get("LS_BENCH_REHEARSAL") == "1" or os.environ.get("LS_BENCH_SHIM") == "1"
    if not rehearsal and not args.spawn_check:
        n = torch.cuda.device_count()
        if n < args.gpus:
            print(f"bench.py: --gpus {args.gpus} but this node shows {n} GPU(s): refusing to measure fewer than asked for", file=sys.stderr)
            return 2
    with socket.socket() as so:   # a free rendezvous port on the loopback interface
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL's peer mappings need it on this driver
    env["LS_BENCH_SPAWNED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher around us: be the launcher (before any GPU call), never a silent one-GPU run
        raise SystemExit(spawn_ranks(args))
    # stdout carries ONE JSON line: whatever native libraries print there (RCCL's version banner at communicator creation)
    # goes to stderr instead -- file descriptor 1 is pointed at stderr, the real stdout is kept for the result line
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the line would not say what was measured")
    global LIBRARY
    if not args.spawn_check:
        if rank == 0 or world == 1:
            LIBRARY = ensure_fresh_library()
        if world > 1 and not LIBRARY:
            LIBRARY = {"sources": kernel_source_sha(), "library": library_source_sha_on_disk(), "rebuilt": False}
            LIBRARY["library_is_stale"] = LIBRARY["library"] != LIBRARY["sources"]
    if args.spawn_check:
        # the ranks meet and count themselves; rank 0 reports (gloo under LS_BENCH_REHEARSAL=1 or without a GPU, else RCCL)
        use_gloo = os.environ.get("LS_BENCH_REHEARSAL") == "1" or not torch.cuda.is_available()
        if world > 1:
            if not use_gloo:
                torch.cuda.set_device(local_rank)
            dist.init_process_group("gloo" if use_gloo else "nccl")
            one = torch.ones(1, dtype=torch.int64, device="cpu" if use_gloo else f"cuda:{local_rank}")
            dist.all_reduce(one)
            seen = int(one.item())
            dist.destroy_process_group()
        else:
            seen = 1
        if rank == 0:
            os.write(result_fd, (json.dumps({"spawn_check": True, "n_gpus": world, "ranks_met": seen, "backend": "gloo" if use_gloo else "nccl",
                                             "spawned_by_bench": os.environ.get("LS_BENCH_SPAWNED") == "1"}) + "\n").encode())
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    # LS_BENCH_REHEARSAL=1: several ranks share GPU 0 over gloo -- only to rehearse the N > 1 code path on
    # a one-GPU box (RCCL refuses two ranks on one device); never used for a reported number
    rehearsal = os.environ.get("LS_BENCH_REHEARSAL") == "1"
    # LS_BENCH_SHIM=1 (tests/test_gpu_group_shim.py): the ranks share GPU 0 too, but the frames go through the C group driver
    # like on a real node -- its collectives on tests/shim/librccl_shim.so (LS_GROUP_RCCL_LIBRARY), torch.distributed's control
    # plane on gloo.  Every phase of the N > 1 flow meets a real peer; the line is labelled and is never a number to quote.
    global SHIM
    SHIM = os.environ.get("LS_BENCH_SHIM") == "1" and world > 1
    if SHIM:
        if not os.environ.get("LS_GROUP_RCCL_LIBRARY"):
            raise SystemExit("LS_BENCH_SHIM=1 needs LS_GROUP_RCCL_LIBRARY=<tests/shim/librccl_shim.so>")
        if args.group_flags == 0:
            args.group_flags = 2   # (the shim refuses a collective on a capturing stream: per-set communicators, plain launches)
    dev_index = 0 if (rehearsal or SHIM) else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    global HOST_NUMA
    HOST_NUMA = pin_to_gpu_numa_node(dev_index) if os.environ.get("LS_BENCH_NO_PIN") != "1" else {"pinned": False, "why": "LS_BENCH_NO_PIN=1"}
    if world > 1:
        if rehearsal or SHIM:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)

    # LS_BENCH_FORCE_GROUP=1 on one GPU: the N > 1 path below, every phase of it, through include/lidarshooter_group.h with a
    # one-rank RCCL communicator (every line of the C group driver runs; never a reported number)
    force_group = world == 1 and os.environ.get("LS_BENCH_FORCE_GROUP") == "1"
    if (world == 1 and not force_group) or args.workload == "cfg5":
        t_start = time.monotonic()
        out = measure(args, rank, world, device, dev_index, rehearsal, None)
        # the other configurations under the driver's own clock, next to the headline -- the BVH engine
        # `north_star` names (poses only, and with the reference's build-per-frame), SYN-10M, configs[4] as one GPU runs it.
        # Lean passes of the same measure(): the timed windows and the dominant kernel's own duration, nothing else; skipped when
        # the headline run already took long (the default run stays within minutes).
        if world == 1 and args.workload == "syn128x1m" and args.engine in ("auto", "projection") and not args.no_also and not args.reregister:
            also, budget_s = [], float(os.environ.get("LS_BENCH_ALSO_BUDGET_S", "240"))
            for label, changes in (("bvh_poses_only", {"engine": "bvh"}),
                                   ("bvh_rebuild_every_frame", {"engine": "bvh", "bvh_rebuild": True}),
                                   ("syn128x10m", {"workload": "syn128x10m"}),
                                   ("cfg5_per_gpu", {"workload": "cfg5"})):
                if time.monotonic() - t_start > budget_s:
                    also.append({"label": label, "skipped": f"the run had taken {time.monotonic() - t_start:.0f} s already (LS_BENCH_ALSO_BUDGET_S)"})
                    continue
                extra = argparse.Namespace(**vars(args))
                extra.lean, extra.no_cpu_baseline, extra.no_dropin, extra.breakdown = True, True, True, False
                extra.steps = max(args.steps, 20)
                for k, v in changes.items():
                    setattr(extra, k, v)
                try:
                    m = measure(extra, rank, world, device, dev_index, rehearsal, None)
                    rf = m.get("roofline") or {}
                    also.append({"label": label, "workload": m["config"]["workload"], "engine": m["config"]["engine"], "frame": m["config"]["frame"],
                                 "ms_per_step": m["ms_per_step"], "value": m["value"], "unit": m["unit"], "frames_per_s": m["frames_per_s"],
                                 "ms_per_step_windows_of_1000": m.get("ms_per_step_windows_of_1000"),
                                 "ms_per_step_one_frame_in_flight": m.get("ms_per_step_one_frame_in_flight"),
                                 "steps": m["steps"], "timing": m["timing"], "points_sha256": m.get("points_sha256"), "hits_per_frame": m.get("hits_per_frame_rank0"),
                                 "roofline": {k: rf.get(k) for k in ("kernel", "frac", "achieved", "kernel_ms", "algorithmic_bytes_per_launch", "unit", "peak",
                                                                      "nodes_per_ray", "tris_per_ray", "cull")}})
                except Exception as e:   # never in the way of the headline
                    import traceback
                    traceback.print_exc()
                    also.append({"label": label, "error": repr(e)})
            out["also_measured"] = also
    else:
        # N > 1, one scene: two ways to spread a stream of frames over the GPUs, both measured, args.multi is `value`;
        # the sharded split is measured on SYN-10M too (where a shard is work-bound), reported next to it
        # The interleaved split first (no collective on its frame path), then the sharded ones under the watchdog: if a
        # multi-rank collective hangs, the line still carries what was measured and says what was not.
        keys = ("value", "unit", "ms_per_step", "frames_per_s", "scaling", "gathered_points_rank0", "host_enqueue_ms_per_step", "timing")

        def brief(m):
            b = {k: m.get(k) for k in keys}
            b["workload"] = m["config"]["workload"]
            b["parallelism"] = m["config"]["parallelism"]
            return b
        store = None
        if world > 1:
            try:
                store = dist.distributed_c10d._get_default_store()
            except Exception:
                store = None
        dog = Watchdog(result_fd, rank, store)
        limit = float(os.environ.get("LS_BENCH_WATCHDOG_S", "240"))
        dog.arm("frames interleaved over the ranks", limit)
        inter = measure(args, rank, world, device, dev_index, rehearsal, "interleaved", force_group=force_group)
        # The sharded split in two arrangements, the plain one first ("keep ONE_COMMUNICATOR as the default
        # until a run with two or more ranks exists" -- it is not the default, but it is what the line falls back on): one
        # communicator, the collective on a stream of its own behind an event, plain launches -- how RCCL is used
        # everywhere; then the arrangement --group-flags asks for (default: a communicator and a stream per buffer set,
        # the frame one captured graph), whose multi-rank collectives inside a captured graph nobody has run yet.
        plain = None
        if args.multi_driver == "c" and not rehearsal and args.group_flags != 1:
            safe = argparse.Namespace(**vars(args))
            safe.group_flags = 1
            dog.arm("azimuth shards + all-gather on one communicator, " + args.workload, limit, inter)
            try:
                plain = measure(safe, rank, world, device, dev_index, rehearsal, "sharded", force_group=force_group)
            except (Exception, SystemExit) as e:
                dog.failed(e)
        fallback = inter
        if plain is not None and args.multi == "sharded":
            fallback = dict(plain)
            fallback["also_measured"] = [brief(inter)]
        dog.arm("azimuth shards + all-gather" + (", per-set communicators and frame graphs, " if plain is not None else ", ") + args.workload, limit, fallback)
        try:
            shard = measure(args, rank, world, device, dev_index, rehearsal, "sharded", force_group=force_group)
        except (Exception, SystemExit) as e:
            dog.failed(e)
        out, also = (shard, [brief(inter)]) if args.multi == "sharded" else (inter, [brief(shard)])
        if plain is not None:
            also.insert(0, brief(plain))
        if args.workload == "syn128x1m" and not rehearsal:
            big = argparse.Namespace(**vars(args))
            big.workload = "syn128x10m"
            partial = dict(out)
            partial["also_measured"] = list(also)
            dog.arm("azimuth shards + all-gather, syn128x10m", limit, partial)
            try:
                also.append(brief(measure(big, rank, world, device, dev_index, rehearsal, "sharded", force_group=force_group)))
            except (Exception, SystemExit) as e:
                dog.failed(e)
        dog.disarm()
        if rank == 0:
            out["also_measured"] = also
    if rank == 0:
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(front_loaded(out)) + "\n").encode())
    if world > 1:
        dist.destroy_process_group()


def measure(args, rank, world, device, dev_index, rehearsal, multi, force_group=False):
    """One measurement.  multi: None (single GPU, or cfg5's replicas), "sharded" (azimuth sectors + one all-gather of
    hit-record slots per frame: a frame's latency is what is split) or "interleaved" (rank g traces the whole raster of
    frames f with f mod N == g, nothing is exchanged on the frame path: the stream's throughput is what is multiplied)."""
    if multi and os.environ.get("LS_BENCH_DEBUG_STALL") == multi:   # (tests: what the watchdog does about a phase that hangs)
        time.sleep(1e6)
    if multi and os.environ.get("LS_BENCH_DEBUG_RAISE") == f"{multi}:{rank}":   # (tests: ... or that raises on one rank)
        raise RuntimeError("LS_BENCH_DEBUG_RAISE")
    sensor, meshes = build_workload(args.workload, rank)
    pipeline_arg = args.pipeline
    if args.pipeline == 0:
        args.pipeline = 2 if sum(t.shape[0] for _, _, t in meshes) >= 200000 else 1
    replicas = args.workload == "cfg5"               # every rank traces its own full sensor (own pose): no shards, no collective
    independent = replicas or multi == "interleaved" # every rank traces whole frames of its own: nothing is exchanged
    if replicas:  # the AffineMesh pose rule played over config/trajectory.json, scaled to keep ben inside the scene
        poses = hostapi.trajectory_play(os.path.join(DATA, "config", "trajectory.json"), 0.1)
        affines = [_affine(p[:3] * np.float32(0.05), p[3:]) for p in poses]
    V, H = int(sensor["vertical"].shape[0]), int(sensor["h_count"])
    first_az, n_az = (0, H) if independent else shards.shard_columns(H, world, rank)
    cap = V * H if independent else shards.slot_capacity(V, H, world)  # records per slot

    tr = capi.Tracer(sensor["vertical"], sensor["h_begin"], sensor["h_end"], H, sensor["Rinv"], sensor["t"],
                     device=dev_index)
    if args.leaf:
        tr.setOption(capi.LS_OPT_LEAF_SIZE, args.leaf)
    tr.setOption(capi.LS_OPT_ENGINE, {"auto": 0, "bvh": 1, "projection": 2}[args.engine])
    if args.classic_bvh or args.bvh_rebuild:
        tr.setOption(capi.LS_OPT_BVH_INSTANCED, 0)
    if args.bvh_rebuild:
        tr.setOption(capi.LS_OPT_BVH_REFIT, 0)
    if os.environ.get("LS_BENCH_BVH_WIDE") is not None:   # (tools/bvh_wide_ab.sh: the binary walk for comparison)
        tr.setOption(capi.LS_OPT_BVH_WIDE, int(os.environ["LS_BENCH_BVH_WIDE"]))
    if args.no_cull or args.cull:
        tr.setOption(capi.LS_OPT_BLOCK_CULL, 1 if args.cull else 0)
    engine = "bvh" if args.engine == "bvh" else "projection"
    tr.setShard(first_az, n_az)
    # a dedicated (non-default) stream shared by the tracer's kernels and, through torch, by RCCL's
    # stream dependencies: the all-gather of frame i is ordered after frame i's pack kernel
    stream = torch.cuda.Stream(device)
    rccl_info = None
    # sharded frames through include/lidarshooter_group.h: frame loop, RCCL all-gather and cloud rebuild in C (RCCL does not
    # take two ranks on one device: the gloo rehearsal on a one-GPU box goes through the torch driver)
    cgroup = (world > 1 or force_group) and not independent and args.multi_driver == "c" and not rehearsal
    if world > 1 and not independent and not cgroup:   # no collective to order otherwise: the tracer keeps its own stream
        tr.setStream(stream.cuda_stream)
    torch.cuda.set_stream(stream)

    # inputs resident in HBM before the timed region
    d_meshes = []
    for name, v, t in meshes:
        dv = torch.from_numpy(np.ascontiguousarray(v, np.float32)).to(device)
        dt = torch.from_numpy(np.ascontiguousarray(t, np.uint32).view(np.int32)).to(device)
        assert tr.addGeometry(name, v.shape[0], t.shape[0]) >= 0
        d_meshes.append((name, dv, dt))
    ident = capi.IDENTITY_AFFINE
    single = (world == 1 and not force_group) or independent
    pipeline = single_gpu_pipeline = single and not args.no_pipeline and engine == "projection"
    if single:
        # single GPU: [n_points u32 | pad to 64 B | points 32*cap | hits 16*cap] in caller-owned buffers, two of
        # them: with two frames in flight (LS_OPT_PIPELINE) consecutive frames write alternate buffers
        out_bufs = [torch.zeros(64 + 48 * cap, dtype=torch.uint8, device=device) for _ in range(3)]

        def set_out(i):
            base = out_bufs[i % 3].data_ptr()
            tr.setOutputBuffers(base + 64, base + 64 + 32 * cap, base, cap)
        set_out(0)
        count_words = out_bufs
        if pipeline:
            torch.cuda.synchronize(device)   # (the handle times candidate streams against each other here: nothing of torch's may be running)
            tr.setOption(capi.LS_OPT_PIPELINE, args.pipeline)
            if args.frame_graph:
                tr.setOption(capi.LS_OPT_FRAME_GRAPH, 1)
    elif cgroup:
        from lidarshooter_amd import groupapi
        uid = torch.zeros(groupapi.ID_BYTES, dtype=torch.uint8, device=device)
        if rank == 0:
            GL = groupapi.load()
            idbuf = (C_.c_uint8 * groupapi.ID_BYTES)()
            if GL.ls_group_unique_id(idbuf) != 0:
                raise SystemExit("ls_group_unique_id failed (no RCCL?)")
            uid = torch.tensor(list(bytes(idbuf)), dtype=torch.uint8, device=device)
        if world > 1:
            uid = uid.cpu() if dist.get_backend() == "gloo" else uid
            dist.broadcast(uid, src=0)
        grp = groupapi.Group(tr, world, rank, groupapi.SHARDED, bytes(uid.cpu().numpy().tobytes()), flags=args.group_flags)
        if bool(grp.info(groupapi.INFO_COLLECTIVES_ARE_A_SHIM)) != SHIM:
            raise SystemExit("the group's collective library and LS_BENCH_SHIM disagree: a shim run must be labelled, a real one must not use it")
        rccl_info = grp.rccl()
        count_words = None
    else:
        # N > 1: the travelling slot holds the count word and the 16-byte hit records only (shards.py);
        # two slots / two gather buffers alternate so that the all-gather of frame i overlaps frame i+1
        sb = shards.slot_bytes(cap)
        slots = [torch.zeros(sb, dtype=torch.uint8, device=device) for _ in range(2)]
        gathered = [torch.zeros(world * sb, dtype=torch.uint8, device=device) for _ in range(2)]
        local_points = torch.zeros(32 * cap, dtype=torch.uint8, device=device)
        cloud_points = torch.zeros(32 * cap * world, dtype=torch.uint8, device=device)
        cloud_hits = torch.zeros(16 * cap * world, dtype=torch.uint8, device=device)
        cloud_n = torch.zeros(16, dtype=torch.int32, device=device)
        works = [None, None]
        count_words = slots

    # what the collective library itself says about the job ("no field says how many ranks RCCL saw")
    rccl_out = None
    if not single:
        if rccl_info is None:   # torch driver: torch.distributed's communicator
            rccl_info = {"version": None if rehearsal else int("%d%02d%02d" % tuple(torch.cuda.nccl.version()[:3])),
                         "comm_ranks": dist.get_world_size() if world > 1 else 1, "device": dev_index, "communicators": 1,
                         "per_set_streams": False, "frame_graph_state": 0, "through": "torch.distributed (" + dist.get_backend() + ")" if world > 1 else "none"}
        else:
            rccl_info["through"] = "include/lidarshooter_group.h (ncclCommCount / ncclCommCuDevice / ncclGetVersion of the group's own communicator)"
        mine = {"rank": rank, "hip_device": dev_index, "comm_device": rccl_info.get("device"), "comm_ranks": rccl_info.get("comm_ranks")}
        every = [mine]
        if world > 1:
            every = [None] * world
            dist.all_gather_object(every, mine)
        rccl_out = dict(rccl_info)
        rccl_out.pop("device", None)
        rccl_out["devices"] = [e["comm_device"] for e in every]
        rccl_out["comm_ranks_on_every_rank"] = [e["comm_ranks"] for e in every]
        rccl_out["launched_by"] = "bench.py itself (torch.distributed.run child)" if os.environ.get("LS_BENCH_SPAWNED") == "1" else "the caller's launcher"

    # The same four C-ABI calls per frame as below, bound once with their constant arguments: the loop is the
    # host side of the product path, and a per-call numpy conversion / string encode is Python's cost, not its
    import ctypes as C
    L, h = tr.L, tr.h
    f32p = C.POINTER(C.c_float)
    ident_c = (C.c_float * 12)(*[float(x) for x in ident])
    fast_meshes = [(name.encode(), C.c_void_p(dv.data_ptr()), C.c_void_p(dt.data_ptr()), name == "face") for name, dv, dt in d_meshes]
    fast_aff = [(C.c_float * 12)(*[float(x) for x in a]) for a in affines] if replicas else None
    fast_frame_struct = capi.Frame()
    fast_out = [(C.c_void_p(b.data_ptr() + 64), C.c_void_p(b.data_ptr() + 64 + 32 * cap), C.c_void_p(b.data_ptr())) for b in out_bufs] if single else None

    registered = set()

    def fast_update_and_trace(i):
        for nm, pv, pt, moving in fast_meshes:
            a = fast_aff[i % len(fast_aff)] if (moving and replicas) else ident_c
            if args.reregister or nm not in registered:
                # the mesh is handed over (in place, in HBM): the library may find any vertices there
                rc = L.ls_update_geometry_device_shared(h, nm, C.cast(a, f32p), pv, 12, pt)
                registered.add(nm)
            else:
                # steady state: the mesh has not changed, only its pose is (re)stated -- what the ITracer adapter does
                # for an unchanged pcl::PolygonMesh and what AffineMesh's pose integration produces (AffineMesh.cpp:108-128)
                rc = L.ls_update_geometry_transform(h, nm, C.cast(a, f32p))
            if rc < 0:
                raise RuntimeError(tr.last_error())
        if L.ls_commit_scene(h) < -1:
            raise RuntimeError(tr.last_error())
        o = fast_out[i % 3]
        L.ls_tracer_set_output_buffers(h, o[0], o[1], o[2], cap)
        if L.ls_trace_scene_async(h, i, C.byref(fast_frame_struct)) < -1:
            raise RuntimeError(tr.last_error())

    # The same sequence for a run of frames in one C++ call (lidarshooter_amd/host/host_capi.cpp: lsh_stream_frames): the
    # timed windows go through it so that what is timed is the C ABI, not CPython's per-call overhead (4 ctypes calls per
    # frame cost 12-16 us of host time on the GPU box's cores against 17 us of GPU time per frame)
    stream_frames = None
    if single and not args.reregister:
        HL = hostapi.load()
        HL.lsh_stream_frames.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(f32p), C.POINTER(C.c_uint), C.c_uint,
                                         C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_uint, C.c_uint,
                                         C.c_uint, C.c_uint]
        sf_names = (C.c_char_p * len(fast_meshes))(*[m[0] for m in fast_meshes])
        sf_lists, sf_counts = [], []
        for nm, pv, pt, moving in fast_meshes:
            if moving and replicas:
                flat = (C.c_float * (12 * len(affines)))(*[float(x) for a in affines for x in a])
                sf_lists.append(flat)
                sf_counts.append(len(affines))
            else:
                sf_lists.append(ident_c)
                sf_counts.append(1)
        sf_aff = (f32p * len(fast_meshes))(*[C.cast(x, f32p) for x in sf_lists])
        sf_n = (C.c_uint * len(fast_meshes))(*sf_counts)
        sf_p = (C.c_void_p * 3)(*[o[0] for o in fast_out])
        sf_h = (C.c_void_p * 3)(*[o[1] for o in fast_out])
        sf_c = (C.c_void_p * 3)(*[o[2] for o in fast_out])

        def stream_frames(first, n):
            if len(registered) < len(fast_meshes):      # the hand-over of the meshes happens once, through the per-call path
                fast_update_and_trace(first)
                first, n = first + 1, n - 1
            if n > 0 and HL.lsh_stream_frames(h, sf_names, sf_aff, sf_n, len(fast_meshes), sf_p, sf_h, sf_c, 3, cap, first, n) < 0:
                raise RuntimeError(tr.last_error())

    group_last_frame = [0]
    if cgroup:
        HL = hostapi.load()
        HL.lsh_group_stream_frames.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(f32p), C.POINTER(C.c_uint), C.c_uint,
                                               C.c_uint, C.c_uint]
        gsf_names = (C.c_char_p * len(fast_meshes))(*[m[0] for m in fast_meshes])
        gsf_aff = (f32p * len(fast_meshes))(*[C.cast(ident_c, f32p) for _ in fast_meshes])
        gsf_n = (C.c_uint * len(fast_meshes))(*[1 for _ in fast_meshes])

        def stream_frames(first, n):
            if len(registered) < len(fast_meshes):      # the hand-over of the meshes happens once
                for nm, pv, pt, _ in fast_meshes:
                    if L.ls_update_geometry_device_shared(h, nm, C.cast(ident_c, f32p), pv, 12, pt) < 0:
                        raise RuntimeError(tr.last_error())
                    registered.add(nm)
            if n > 0 and HL.lsh_group_stream_frames(grp.g, h, gsf_names, gsf_aff, gsf_n, len(fast_meshes), first, n) < 0:
                raise RuntimeError(tr.last_error() + " / " + grp.L.ls_group_last_error(grp.g).decode())
            if n > 0:
                group_last_frame[0] = first + n - 1      # (the group keeps three frames: the one downloaded at the end is this one)

    def update_and_trace(i, copy):
        if single and not copy:
            return fast_update_and_trace(i)
        for name, dv, dt in d_meshes:                      # MeshProjector.cpp:448-461: every mesh, every frame
            if replicas and name == "face":                # the animated instance: a new pose every frame
                tr.updateGeometryDeviceShared(name, affines[i % len(affines)], dv.data_ptr(), 12, dt.data_ptr())
                continue
            if copy:   # defensive D2D copy of the mesh into library-owned buffers (18 MB per frame at 1M triangles)
                tr.updateGeometryDevice(name, ident, dv.data_ptr(), 12, dt.data_ptr())
            else:      # the mesh is read in place from the caller's HBM buffers (vertex transform fused into the trace)
                tr.updateGeometryDeviceShared(name, ident, dv.data_ptr(), 12, dt.data_ptr())
        tr.commitScene()
        if single:
            set_out(i)
        tr.traceSceneAsync(i)

    def collect(b):
        """finish the gather that was started from slot b: wait, then compact the world's slots into one cloud"""
        if works[b] is not None:
            works[b].wait()                                # the tracer's stream waits for the collective
            works[b] = None
            tr.expandGatheredHits(gathered[b].data_ptr(), world, cap, cloud_points.data_ptr(), cloud_hits.data_ptr(),
                                  cloud_n.data_ptr())

    # N > 1: with two frames in flight (LS_OPT_PIPELINE = 1, one stream: the finish + pack of frame i ride in the
    # launch of frame i+1) the slot of frame i is complete on the stream after call i+1, so its all-gather starts
    # one call later and has two frames of work to hide behind.  G(f) reads slot[f & 1], which the riders in the
    # launch of frame f+3 write again: G(f) is waited for (and its cloud rebuilt) at the start of frame f+3.
    lagged = (not single) and not cgroup and engine == "projection" and not args.no_pipeline
    if lagged:
        tr.setOption(capi.LS_OPT_PIPELINE, 1)
    state = {"prev": None}

    def start_gather(f):
        b = f & 1
        works[b] = shards.all_gather_slots(slots[b], gathered[b], async_op=True)

    def frame(i, copy=False):
        if single:
            update_and_trace(i, copy)
            return
        if cgroup:
            stream_frames(i, 1)
            return
        b = i & 1
        if lagged:
            collect(b ^ 1)                                 # G(i-3): its slot is written again by the riders in this launch
            tr.setOutputBuffers(local_points.data_ptr(), slots[b].data_ptr() + shards.HEADER, slots[b].data_ptr(), cap)
            update_and_trace(i, copy)
            if state["prev"] is not None:
                start_gather(state["prev"])                # frame i-1 is complete on the stream now
            state["prev"] = i
            return
        collect(b)                                         # slot b is free again (its gather is two frames old)
        tr.setOutputBuffers(local_points.data_ptr(), slots[b].data_ptr() + shards.HEADER, slots[b].data_ptr(), cap)
        update_and_trace(i, copy)                          # runs while the previous frame's gather is in flight
        collect(b ^ 1)                                     # previous frame: gather done -> its cloud
        start_gather(i)

    def flush():
        if cgroup:
            grp.synchronize()
        elif not single:
            if lagged:
                tr.flush()                                 # the last frame's riders
                if state["prev"] is not None:
                    collect(state["prev"] & 1)             # an older gather from the same slot pair first
                    start_gather(state["prev"])
                    state["prev"] = None
            collect(0)
            collect(1)
        elif pipeline:
            tr.flush()   # no-op while LS_OPT_PIPELINE is off

    def sync():
        flush()
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(device)

    # ---- node / triangle visits per ray on this BVH (algorithmic bytes), outside the timed region
    if engine == "bvh":   # (a big hierarchy's four-wide twins are made by the second frame that finds it unchanged: count on the walk that is timed)
        for i in range(3):
            frame(i)
        sync()
    tr.setOption(capi.LS_OPT_COUNT_VISITS, 1)
    frame(0)
    sync()
    n_node, n_tri, wave_trips, max_trips = tr.visitStats()
    tr.setOption(capi.LS_OPT_COUNT_VISITS, 0)
    shard_rays = tr.getTotalRays()
    points_sha = None
    if count_words is not None:
        n_hits = int(count_words[0][:4].view(torch.int32).item())
        if single:   # the frame's cloud (32-byte points, ray-index order): lsbench prints the same hash for the same scene
            import hashlib
            points_sha = hashlib.sha256(out_bufs[0][64:64 + 32 * n_hits].cpu().numpy().tobytes()).hexdigest()
    else:   # C group: the slot's count word is the group's; the shard's dense result says how many of its rays hit
        n_hits = int((tr.denseHits()[1] != 0xFFFFFFFF).sum())
    info = tr.sceneSize()
    n_tris_total = info["n_tris"]

    # CPython's cyclic garbage collector runs a full (generation 2) collection some 700 frames into the run --
    # every frame allocates a few ctypes objects -- and with torch imported that pause is 30-40 ms, a thousand
    # frames' worth (tools/long_run.py).  The measurement loop allocates nothing that needs it: freeze what
    # exists and switch the collector off.  The priming frames that follow are for clocks and caches.
    gc.collect()
    gc.freeze()
    gc.disable()
    prime = max(0, args.prime_ms) * 1e-3
    t_prime = time.perf_counter()
    i = 0
    # N > 1: every rank must issue the same collectives, so the count is fixed there (1000 frames)
    batches = None if world == 1 else (20 if prime > 0 else 0)
    while (time.perf_counter() - t_prime < prime) if batches is None else (batches > 0):
        for _ in range(50):
            frame(i)
            i += 1
        sync()
        if batches is not None:
            batches -= 1
    for i in range(args.warmup):
        frame(i)
    sync()
    # ---- the timed region: windows of exactly K frames, each bracketed by barrier + synchronize, no instrumentation
    #      inside (hipEvent records would put barrier packets between the kernels).  Whatever --steps says, windows
    #      are repeated until at least --min-ms have been timed; the reported step time is the MEDIAN window's.
    def window_sync():
        # single GPU: the device-wide wait the contract asks for covers the three streams the frames rotate over by itself; the
        # handle's flush (three event records + three waits, ~15 us of host calls that order its own stream behind them) is not
        # part of a frame and is left to the first call that needs the handle's stream ordered (sync() does it outside the windows)
        if single:
            torch.cuda.synchronize(device)
        else:
            sync()

    def window():
        sync()
        t0 = time.perf_counter()
        if stream_frames is not None:
            stream_frames(0, args.steps)         # K frames, one C++ loop over the C ABI
        else:
            for i in range(args.steps):
                frame(i)
        enq = time.perf_counter() - t0           # host time to enqueue K frames (diagnostic: host- or GPU-bound?)
        window_sync()
        el = time.perf_counter() - t0
        if world > 1:
            e = torch.tensor([el], dtype=torch.float64, device=device)
            dist.all_reduce(e, op=dist.ReduceOp.MAX)
            el = float(e.item())
        return el, enq
    first = window()
    n_windows = max(1, min(2000, int(np.ceil(args.min_ms * 1e-3 / max(first[0], 1e-9)))))   # the same count on every rank (MAX-reduced time)
    wins = [first] + [window() for _ in range(n_windows - 1)]
    order = np.argsort([w[0] for w in wins])
    elapsed, enqueue_s = wins[int(order[len(order) // 2])]
    window_ms = sorted(w[0] * 1e3 for w in wins)
    # ---- the dominant kernel's duration: the same K frames again, with hipEvents on the tracer's
    #      stream bracketing that kernel only (recorded without synchronising, read after the loop)
    tr.setOption(capi.LS_OPT_TIMING, 2)
    tr.timings()
    if stream_frames is not None:
        stream_frames(0, args.steps)             # back to back, like the timed windows (the GPU stays busy)
    else:
        for i in range(args.steps):
            frame(i)
    sync()
    tm = tr.timings()
    tr.setOption(capi.LS_OPT_TIMING, 0)

    def windows_of(fn, min_s=0.02):
        """median seconds per frame over windows of K frames, at least min_s timed (single rank only)"""
        res = []
        total = 0.0
        while total < min_s or len(res) < 3:
            sync()
            t1 = time.perf_counter()
            if fn is frame and stream_frames is not None:
                stream_frames(0, args.steps)
            else:
                for i in range(args.steps):
                    fn(i)
            sync()
            res.append(time.perf_counter() - t1)
            total += res[-1]
        return float(np.median(res)) / args.steps

    # ---- the same frames in long windows (single GPU, C++ loop): what the step costs once the window's edges -- the first
    #      launch's latency, the last frame's whole latency instead of a step, the runtime's bookkeeping inside the device-wide
    #      wait: ~70 us per window, 3.5 us per frame of a 20-frame window -- are amortised (never `value`)
    steady_s = None
    if stream_frames is not None and world == 1:
        long_k = 1000
        res = []
        for _ in range(3):
            sync()
            t1 = time.perf_counter()
            stream_frames(0, long_k)
            sync()
            res.append((time.perf_counter() - t1) / long_k)
        steady_s = float(np.median(res))

    latency_frame_s = None
    if pipeline:
        # the same K frames with one frame in flight: what a consumer that needs every frame before the next sees
        tr.setOption(capi.LS_OPT_PIPELINE, 0)
        latency_frame_s = windows_of(frame) if world == 1 else None
        if world > 1:
            sync()
            t1 = time.perf_counter()
            for i in range(args.steps):
                frame(i)
            sync()
            latency_frame_s = (time.perf_counter() - t1) / args.steps
        tr.setOption(capi.LS_OPT_PIPELINE, args.pipeline)

    breakdown = None
    if (args.breakdown or world == 1) and not replicas and not getattr(args, "lean", False):
        tr.setOption(capi.LS_OPT_TIMING, 1)
        tr.timings()
        for i in range(min(args.steps, 50)):
            frame(i)
        sync()
        breakdown = tr.timings()
        tr.setOption(capi.LS_OPT_TIMING, 0)
        # trace-only frames (static scene): the trace kernels + pack, nothing else
        trace_only_s = windows_of(lambda i: tr.traceSceneAsync(i))
        # the same frame with the mesh copied into library-owned buffers on every update
        copy_frame_s = windows_of(lambda i: frame(i, copy=True))
        frame(0)
        sync()

    # ---- the drop-in path (N = 1): the reference's per-frame sequence through the ROS-typed adapter
    #      integration/HipTracer.hpp (lidarshooter::HipTracer : ITracer over stand-in ROS/PCL types), in C++:
    #      for every mesh updateGeometry(name, translation, rotation, pcl::PolygonMesh::Ptr&) + commitScene() +
    #      traceScene(frame) filling a sensor_msgs::PointCloud2 in host memory (MeshProjector.cpp:446-464).
    dropin = None
    if world == 1 and not replicas and engine == "projection" and not args.no_dropin and not getattr(args, "lean", False):
        import tempfile
        from lidarshooter_amd import adapterapi
        cfg = synth.write_sensor_json(os.path.join(DATA, "config", "hesai-pandar-XT-32-lidar_0000.json"),
                                      os.path.join(tempfile.mkdtemp(), "sensor.json"), sensor["vertical"], float(sensor["h_begin"]),
                                      float(sensor["h_end"]), H)
        at = adapterapi.AdapterTracer(cfg, dev_index)
        for name, v, t in meshes:
            at.meshFromArrays(name, v, t, point_step=16)      # pcl::PointXYZ records, as pcl::io::loadPolygonFileSTL leaves them
            at.addGeometry(name)
        at.frameLoop(3)

        def adapter_ms():
            res, total = [], 0.0
            while total < 0.05 or len(res) < 3:
                sec = at.frameLoop(max(10, min(args.steps, 50)))
                res.append(sec)
                total += sec * max(10, min(args.steps, 50))
            return float(np.median(res)) * 1e3
        up_ms = adapter_ms()                                   # default policy: vertices re-sent every frame
        at.setSkipUnchanged(True)
        at.frameLoop(3)
        st_ms = adapter_ms()                                   # unchanged cloud: pose-only update, no vertex traffic
        ac = at.cloud()
        vbytes = sum(v.shape[0] * 16 for _, v, _ in meshes)
        # ---- the PCIe floor of the drop-in frame, from rates measured here: the vertex upload as the adapter does it
        #      (updateGeometry alone: one copy straight from pageable memory, the call returns when it has been read) and the
        #      cloud's way back (8 bytes per point -- ray number and t -- into pinned host memory) at this link's D2H rate
        at.setSkipUnchanged(False)
        ups = []
        for _ in range(12):
            t1 = time.perf_counter()
            for name, _, _ in meshes:
                at.updateGeometry(name)
            ups.append(time.perf_counter() - t1)
        upload_ms = float(np.median(ups[2:])) * 1e3
        down_bytes = int(ac["width"]) * 8
        src = torch.empty(max(down_bytes, 1 << 16), dtype=torch.uint8, device=device)
        dst = torch.empty(src.numel(), dtype=torch.uint8).pin_memory()
        downs = []
        for _ in range(12):
            torch.cuda.synchronize(device)
            t1 = time.perf_counter()
            dst.copy_(src, non_blocking=True)
            torch.cuda.synchronize(device)
            downs.append(time.perf_counter() - t1)
        download_ms = float(np.median(downs[2:])) * 1e3
        del src, dst
        dropin = {"dropin_ms_per_step": up_ms, "dropin_static_ms_per_step": st_ms, "points": int(ac["width"]),
                  "cloud_equals_headline_hits": int(ac["width"]) == n_hits,
                  "vertex_bytes_uploaded_per_frame": vbytes, "point_bytes_over_pcie_per_frame": down_bytes,
                  "floor_ms": upload_ms + download_ms, "gap_ms": up_ms - (upload_ms + download_ms),
                  "floor": {"upload_ms": upload_ms, "upload_gb_per_s": vbytes / (upload_ms * 1e-3) / 1e9,
                            "download_ms": download_ms, "download_gb_per_s": down_bytes / (download_ms * 1e-3) / 1e9,
                            "what": "floor_ms = the two PCIe transfers of a frame back to back at the rates measured in this run: the vertex "
                                    "upload (updateGeometry calls alone) + a device-to-pinned-host copy of the cloud's 8-byte records; gap_ms = "
                                    "dropin_ms_per_step - floor_ms = kernels + launches + the host-side rebuild that do not hide under a transfer"},
                  "what": "MeshProjector::traceAffineMesh through lidarshooter::HipTracer (integration/HipTracer.hpp, stand-in ROS/PCL "
                          "types): updateGeometry(translation, rotation, mesh) per mesh + commitScene + traceScene into PointCloud2::data; "
                          "dropin = vertices re-sent every frame (default), dropin_static = unchanged cloud detected, pose-only update "
                          "(setMeshPolicy(SkipUnchanged)); polygons are sent once in both",
                  "pcie_floor_note": "the cloud has to reach host memory: 8 B per point ((ray, t): the 32-byte record is rebuilt on the "
                                     "host from the factor tables) = %.2f MB per frame; the vertex upload moves %.1f MB"
                                     % (down_bytes / 1e6, vbytes / 1e6)}
        at.close()

    total_rays = V * H * (world if independent else 1)
    ms_per_step = elapsed / args.steps * 1e3
    value = total_rays * args.steps / elapsed / 1e6
    trace_ms = tm["trace"]
    if engine == "bvh":
        # k_trace: 64 B per node fetch + 48 B per triangle test + 8 B per ray written
        kernel = "k_trace_inst" if tr.info(capi.LS_INFO_BVH_INSTANCED) else "k_trace"
        wide_walk = bool(tr.info(capi.LS_INFO_BVH_WIDE))
        node_bytes = 2 * NODE_BYTES if wide_walk else NODE_BYTES   # (a four-wide node is 128 bytes: four boxes + four references)
        b_launch = RAY_OUT_BYTES * shard_rays + node_bytes * n_node + TRI_BYTES * n_tri
        b_frame = None
        units = {"rays_per_launch": shard_rays, "bytes_per_ray": b_launch / shard_rays,
                 "nodes_per_ray": n_node / shard_rays, "tris_per_ray": n_tri / shard_rays, "node_bytes": node_bytes,
                 "walk": "four-wide nodes (k_widen: a node's slots are its grandchildren)" if wide_walk else "binary nodes",
                 "wave_trips_mean": wave_trips / max(1, (shard_rays + 63) // 64), "wave_trips_max": max_trips}
    else:
        # k_project: every triangle is streamed once (12 B indices + 36 B vertex gather) and every hit folds 8 B into
        # the per-ray key; the angle tables (V+H entries) stay on chip.  With group culling in effect
        # (LS_OPT_BLOCK_CULL; the timed stage is then k_cull + k_project) the triangles of culled groups are never read:
        # the bytes are the bounds k_cull reads (32 B per 256-triangle block, 32 B per group of 4 inside the blocks that
        # survive), the survivor list (4 B written + 4 B read per surviving group) and 48 B per SURVIVING triangle
        kernel = "k_project"
        groups_alive, group_bounds_read = int(wave_trips), int(max_trips)
        culled = groups_alive > 0
        if culled:
            n_blocks = sum(((t.shape[0] + 3) // 4 + 63) // 64 for _, _, t in meshes if t.shape[0] >= 524288)
            tris_read = min(4 * groups_alive, n_tris_total) + sum(t.shape[0] for _, _, t in meshes if t.shape[0] < 524288)
            b_launch = 32 * n_blocks + 32 * group_bounds_read + 8 * groups_alive + 48 * tris_read + 8 * n_hits
            kernel = "k_cull + k_project"
        else:
            tris_read = n_tris_total
            b_launch = 48 * n_tris_total + 8 * n_hits
        # the whole frame: + k_project_finish (8 B key read per ray) + k_pack (8 B key read + 8 B re-arm per ray,
        # 48 B point + hit record per hit)
        b_frame = b_launch + 24 * shard_rays + 48 * n_hits
        units = {"triangles_per_launch": n_tris_total, "triangles_read_per_launch": tris_read,
                 "bytes_per_triangle": b_launch / max(1, n_tris_total),
                 "candidate_tests_per_launch": n_tri, "tests_per_triangle": n_tri / max(1, n_tris_total),
                 "rays_per_launch": shard_rays,
                 "nominal_bytes_per_launch": 48 * n_tris_total + 8 * n_hits}
        if culled:
            units["cull"] = {"blocks_of_256": n_blocks, "group_bounds_read": group_bounds_read, "groups_surviving": groups_alive,
                             "what": "bytes are priced on what the culled stage has to read: 32 B per block bound, 32 B per group bound "
                                     "inside surviving blocks, 8 B per surviving group (list), 48 B per surviving triangle, 8 B per hit"}
    achieved = b_launch / (trace_ms * 1e-3) / 1e9 if trace_ms > 0 else 0.0
    # HBM bytes of that kernel from the committed rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE, KiB).  Counters cannot
    # be collected from inside this process, so the figure is the profiled one -- and only if the profile was taken
    # from these very kernel sources (its recorded source hash equals the current one)
    traffic, traffic_src = None, None
    sha = kernel_source_sha()
    prof = os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_{engine}_hbm.json")
    if args.workload == "syn128x1m" and world == 1 and os.path.exists(prof):  # only the profiled configuration
        try:
            pj = json.load(open(prof))
            if pj.get("kernel_source_sha") != sha:
                traffic_src = f"{os.path.relpath(prof, ROOT)} is stale (taken from kernel sources {pj.get('kernel_source_sha')}, these are {sha}): not reported"
            else:
                k = pj["kernels"]
                # the timed variant: template argument COUNT (the first one) is false
                key = next((n for n in k if n.split("<")[0] == kernel and not n.split("<")[-1].startswith("true")), None)
                if key:
                    traffic, traffic_src = k[key]["hbm_bytes_per_launch"], os.path.relpath(prof, ROOT)
        except Exception:
            pass
    # What binds the kernel besides bytes (SURVEY.md 8d "secondary ceilings"): the committed SQ / TCC counter summary of the same
    # kernel sources (tools/sq_profile.sh -> profiles/<tag>_<engine>_sq.json), same staleness rule as `traffic`
    secondary = None
    sq = os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_{engine}_sq.json")
    if args.workload == "syn128x1m" and world == 1 and os.path.exists(sq):
        try:
            sj = json.load(open(sq))
            if sj.get("kernel_source_sha") != sha:
                secondary = {"source": f"{os.path.relpath(sq, ROOT)} is stale (taken from kernel sources {sj.get('kernel_source_sha')}, these are {sha}): not reported"}
            else:
                key = next((n for n in sj["kernels"] if n.split("<")[0] == kernel and not n.split("<")[-1].startswith("true")), None)
                if key:
                    d = sj["kernels"][key]["derived"]
                    secondary = {"issue_frac": d.get("valu_issue_frac"), "lanes_active_per_valu": d.get("lanes_active_per_valu"), "l2_hit": d.get("l2_hit"),
                                 "wave_cycles_waiting": d.get("wait"), "wave_cycles_issue_stalled": d.get("stall"), "valu_per_wave": d.get("valu_per_wave"),
                                 "wave_life_us": d.get("wave_life_us"), "source": os.path.relpath(sq, ROOT),
                                 "what": "issue_frac = VALU wave-instructions x 2 cycles / (1024 SIMDs x 2.3 GHz) / kernel time: the share of the kernel during which "
                                         "the vector ALUs could have been issuing at all; lanes_active_per_valu of 64; l2_hit = TCC_HIT / (TCC_HIT + TCC_MISS); the two "
                                         "wave_cycles shares are SQ_WAIT_ANY (parked in s_waitcnt / barrier) and SQ_WAIT_INST_ANY (issue-stalled) over SQ_WAVE_CYCLES"}
        except Exception:
            pass
    one_ms = latency_frame_s * 1e3 if latency_frame_s is not None else ms_per_step

    if cgroup:   # the whole frame's cloud as this rank holds it after the gather + rebuild (the group keeps three frames)
        import hashlib
        gathered_cloud, gathered_hits = grp.download(group_last_frame[0])
        # (sector-major as gathered: rank 0's records, then rank 1's ...; hashed in ray order, the order of the one-GPU cloud)
        points_sha = hashlib.sha256(np.ascontiguousarray(gathered_cloud[np.argsort(gathered_hits[:, 0], kind="stable")]).tobytes()).hexdigest()
    out = {
        "metric": {"syn128x1m": "Mrays/s (LiDAR frame = updateGeometry + commitScene + traceScene incl. point packing, 128ch x 4096az over 1M tris; "
                                "geometry resident in HBM, every mesh's pose restated per frame; cloud left in HBM -- the frame with vertices "
                                "re-sent from host memory and the cloud delivered to PointCloud2::data is dropin_ms_per_step)",
                   "syn128x10m": "Mrays/s (LiDAR frame as above; 128ch x 4096az over 10M tris; geometry resident in HBM, cloud left in HBM)",
                   "syn128x1500k": "Mrays/s (LiDAR frame as above; 1.5M tris)", "syn128x2m": "Mrays/s (LiDAR frame as above; 2M tris)", "syn128x3m": "Mrays/s (LiDAR frame as above; 3M tris)", "syn128x5m": "Mrays/s (LiDAR frame as above; 5M tris)",
                   "cfg5": "Mrays/s (one 128ch x 4096az sensor per GPU over a shared 10M-tri scene + animated instance)",
                   "xt32": "Mrays/s (XT-32 over ground+ben)"}[args.workload],
        "value": value, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak" if independent else "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic" + (" (LS_BENCH_SHIM: %d ranks on ONE device, collectives through tests/shim -- a test of semantics, not a measurement)" % world if SHIM else ""),
        "config": {"workload": {"syn128x1m": "SYN-128 (128ch x 4096az, pose lidar_0000) x SYN-1M (1,000,000 tris)",
                                "syn128x10m": "SYN-128 x SYN-10M (9,998,244 tris)",
                                "syn128x1500k": "SYN-128 x 1224 x 612-cell grid (1,498,176 tris)",
                                "syn128x2m": "SYN-128 x 1414 x 707-cell grid (1,999,396 tris)", "syn128x3m": "SYN-128 x 1732 x 866-cell grid (2,999,824 tris)",
                                "syn128x5m": "SYN-128 x 2236 x 1118-cell grid (4,999,696 tris)",
                                "cfg5": "8-pose SYN-128 ring x (SYN-10M + ben.stl animated by trajectory.json), replicas only",
                                "xt32": "XT-32 lidar_0000 x ground.stl+ben.stl"}[args.workload],
                   "rays_per_frame": total_rays, "triangles": info["n_tris"], "engine": engine,
                   "frame": ("updateGeometry(device mesh handed over in place, every frame) + commitScene + traceScene" if args.reregister else
                             "updateGeometry(mesh resident in HBM and unchanged: its pose is restated, ls_update_geometry_transform) "
                             "+ commitScene + traceScene")
                            + ((" (BVH engine, instanced: per-geometry hierarchies in mesh space, nothing is built when only poses change)"
                                if tr.info(capi.LS_INFO_BVH_INSTANCED) else
                                (" (BVH engine, classic: the hierarchy is BUILT FROM SCRATCH every frame -- Morton keys, radix sort, leaves, range tree, radix tree -- "
                                 "the reference's own per-frame sequence)" if args.bvh_rebuild else " (BVH engine, classic: the hierarchy is refitted every frame)"))
                               if engine == "bvh" else "")
                            + (("; two frames in flight (finish + pack of frame i ride in the launch of frame i+1)"
                                if args.pipeline == 1 else "; three frames in flight (whole frames rotate over three streams)")
                               if pipeline else ""),
                   "parallelism": ((f"azimuth-sector shards x{world}, scene replica per GPU, one ncclAllGather of hit-record slots per frame, "
                                    f"cloud rebuilt on every rank; three frames in flight per rank, each buffer set with a communicator and a "
                                    f"stream of its own: trace + gather + rebuild of a frame are ONE captured HIP graph launch (poses patched into "
                                    f"the k_project node); frame loop in C (include/lidarshooter_group.h)"
                                    if rccl_out and rccl_out.get("per_set_streams") and rccl_out.get("frame_graph_state") == 1 else
                                    f"azimuth-sector shards x{world}, scene replica per GPU, one ncclAllGather of hit-record slots per "
                                    f"frame on a collective stream that waits for that frame alone, cloud rebuilt on every rank; three "
                                    f"frames in flight per rank; frame loop, collective and rebuild in C (include/lidarshooter_group.h)")
                                   if cgroup else
                                   f"azimuth-sector shards x{world}, scene replica per GPU, one async all-gather of "
                                   f"hit-record slots per frame through torch.distributed (overlapped with the next "
                                   f"{'two frames; two frames in flight per rank' if lagged else 'frame'}), cloud rebuilt on every rank")
                   if not single else ("single GPU" if world == 1 else
                                       (f"{world} independent replicas (one sensor pose per GPU), no collective" if replicas else
                                        f"frames interleaved over {world} GPUs: every rank traces the whole raster of its own frames "
                                        f"(K per rank in the timed region, {world}K in all), scene replica per GPU, nothing exchanged "
                                        f"on the frame path (a 20 us frame is bound by launch and memory latency: splitting it costs "
                                        f"more in the collective than it saves)"))},
        "frames_per_s": args.steps / elapsed,
        "timing": {"windows": len(wins), "frames_per_window": args.steps, "window_ms_min_median_max":
                   [window_ms[0], window_ms[len(window_ms) // 2], window_ms[-1]], "reported": "median window",
                   "timed_ms_total": float(sum(window_ms))},
        "host_enqueue_ms_per_step": enqueue_s / args.steps * 1e3,
        "hits_per_frame_rank0": n_hits,
        "points_sha256": points_sha,
        # N > 1: points of the whole frame as rebuilt from the gathered slots on rank 0 (= the 1-GPU hit count)
        "gathered_points_rank0": None if single else (int(gathered_cloud.shape[0]) if cgroup else int(cloud_n[0].item())),
        "rehearsal_gloo_shared_gpu": True if rehearsal else None,
        "rccl": rccl_out,
        "host_numa": HOST_NUMA,
        "library": dict(LIBRARY or {}, loaded=capi.load().ls_source_hash().decode()),
        "frame_graph": {"state": tr.info(capi.LS_INFO_FRAME_GRAPH_STATE), "captures": tr.info(capi.LS_INFO_FRAME_GRAPH_CAPTURES),
                        "replays": tr.info(capi.LS_INFO_FRAME_GRAPH_REPLAYS), "patches": tr.info(capi.LS_INFO_FRAME_GRAPH_PATCHES)},
        "roofline": dict({
            "bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src, "secondary": secondary,
            "kernel_ms": trace_ms, "kernel_launches_timed": tm["frames"],
            "kernel_timing": "ISOLATED kernel: its own begin / end timestamps (hipEvents attached to the dispatch with hipExtLaunchKernel, what rocprofv3 reports per dispatch; events recorded around a launch add ~3 us of barrier packets) with ONE frame in flight (frames do not overlap while it is timed), in a second pass of the same K frames; compare with ms_per_step_one_frame_in_flight, not with ms_per_step (frames overlap there), and with profiles/*_kernel_isolated.json (the same figure from the rocprofv3 trace of this command)",
            "ms_per_step_one_frame_in_flight": one_ms,
            "algorithmic_bytes_per_launch": b_launch, "kernel_source_sha": sha,
            # the same roofline on the frame as timed: all kernels' algorithmic bytes over the reported step time
            "step_basis": None if b_frame is None else {
                "algorithmic_bytes_per_frame": b_frame, "ms_per_step": ms_per_step,
                "achieved": b_frame / (ms_per_step * 1e-3) / 1e9, "frac": b_frame / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "what": "k_project + k_project_finish + k_pack bytes of one frame / the reported (overlapped) step time"}},
            **units),
    }
    if latency_frame_s is not None:
        out["ms_per_step_one_frame_in_flight"] = latency_frame_s * 1e3
    if steady_s is not None:
        out["ms_per_step_windows_of_1000"] = steady_s * 1e3
        out["window_note"] = ("ms_per_step is the median window of exactly --steps frames between two device-wide waits; a window carries ~70 us of "
                              "edges (first launch, the last frame's whole latency, the runtime retiring the window's commands inside the wait), i.e. "
                              "3.5 us per frame at --steps 20 and 0.07 us at 1000: ms_per_step_windows_of_1000 is the same loop in windows of 1000 frames")
    if dropin is not None:
        out["dropin_ms_per_step"] = dropin["dropin_ms_per_step"]
        out["dropin_static_ms_per_step"] = dropin["dropin_static_ms_per_step"]
        out["dropin"] = dropin
    if breakdown is not None:
        out["stage_ms"] = {k: round(v, 5) for k, v in breakdown.items() if k != "frames"}
        out["copy_update_ms_per_step"] = copy_frame_s * 1e3
        out["trace_only_ms"] = trace_only_s * 1e3
        out["trace_only_mrays_per_s"] = shard_rays / trace_only_s / 1e6
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not replicas and not getattr(args, "lean", False):
        out["cpu_baseline"] = cpu_baseline(sensor, meshes, args.cpu_frames, total_rays)
    if cgroup:
        grp.close()
    tr.close()
    args.pipeline = pipeline_arg
    torch.cuda.set_stream(torch.cuda.default_stream(device))
    gc.enable()
    return out


if __name__ == "__main__":
    main()
