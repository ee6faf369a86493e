"""BASELINE configs[4] on the BVH engine at its own size: SYN-10M + ben.stl, ben moved by config/trajectory.json
("per-GPU BVH replica, trajectory.json animated refit"; the OptiX counterpart is buildAccelStructure's BUILD / UPDATE,
OptixTracer.cpp:517-571).  Host-clock times per phase; run under tools/rocprof_kernels.sh with PHASE=<one phase> for that
phase's kernel list.  Phases:
  build     the first commit of the scene: both instanced hierarchies from nothing (10 M + 5 489 triangles)
  poses     per frame: ben's pose from the trajectory, the ground's restated -- instanced: nothing is built or refitted
  refit_ben per frame: ben's VERTICES uploaded again (a deforming mesh) -- its hierarchy alone is refitted
  classic   per frame, LS_OPT_BVH_INSTANCED = 0: the sensor-frame hierarchy over all 10 M triangles is refitted (the pose changed)
usage: [PHASE=build|poses|refit_ben|classic|all] bvh_cfg5_cost.py [frames]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from lidarshooter_amd import capi, hostapi
import bench
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 60
phase = os.environ.get("PHASE", "all")
bench.pin_to_gpu_numa_node(0)
sensor, meshes = bench.build_workload("cfg5", 0)
poses = hostapi.trajectory_play(os.path.join(bench.DATA, "config", "trajectory.json"), 0.1)
affines = [bench._affine(p[:3] * np.float32(0.05), p[3:]) for p in poses]
dev = torch.device("cuda", 0)
out = {"workload": "cfg5 rank 0: SYN-128 x (SYN-10M %d triangles + ben.stl %d), %d trajectory poses" % (meshes[0][2].shape[0], meshes[1][2].shape[0], len(affines)), "frames": frames}


def make(instanced):
    tr = capi.Tracer(sensor["vertical"], sensor["h_begin"], sensor["h_end"], int(sensor["h_count"]), sensor["Rinv"], sensor["t"])
    tr.setOption(capi.LS_OPT_ENGINE, 1)
    tr.setOption(capi.LS_OPT_BVH_INSTANCED, instanced)
    keep = []
    for n, v, t in meshes:
        dv = torch.from_numpy(np.ascontiguousarray(v, np.float32)).to(dev)
        dt = torch.from_numpy(np.ascontiguousarray(t, np.uint32).view(np.int32)).to(dev)
        keep.append((n, dv, dt, v))
        tr.addGeometry(n, v.shape[0], t.shape[0])
        tr.updateGeometryDeviceShared(n, capi.IDENTITY_AFFINE, dv.data_ptr(), 12, dt.data_ptr())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    assert tr.commitScene() == 0
    tr.synchronize()
    return tr, keep, (time.perf_counter() - t0) * 1e3


def loop(tr, keep, n, reupload_ben=False):
    for f in range(n):
        for name, dv, dt, v in keep:
            if name == "face" and reupload_ben:
                tr.updateGeometryDevice(name, affines[f % len(affines)], dv.data_ptr(), 12, None)   # vertices again, same topology
            else:
                tr.updateGeometryTransform(name, affines[f % len(affines)] if name == "face" else capi.IDENTITY_AFFINE)
        assert tr.commitScene() == 0
        tr.traceSceneAsync(f)
    tr.synchronize()


def timed(tr, keep, **kw):
    loop(tr, keep, 10, **kw)
    t0 = time.perf_counter()
    loop(tr, keep, frames, **kw)
    return (time.perf_counter() - t0) / frames * 1e3


if phase in ("all", "build", "poses", "refit_ben"):
    tr, keep, build_ms = make(1)
    out["build_ms_first_commit_instanced"] = build_ms
    if phase in ("all", "poses"):
        out["poses_ms_per_frame"] = timed(tr, keep)
        out["poses_last_commit_built_anything"] = tr.info(capi.LS_INFO_BVH_INSTANCED) != 1
        rc, pts, _ = tr.traceScene(9999)
        out["points_last_frame"] = int(pts.shape[0])
    if phase in ("all", "refit_ben"):
        out["refit_ben_ms_per_frame"] = timed(tr, keep, reupload_ben=True)
    tr.close()
    del keep
    torch.cuda.empty_cache()
if phase in ("all", "classic"):
    tr, keep, build_ms = make(0)
    out["build_ms_first_commit_classic"] = build_ms
    out["classic_refit_ms_per_frame"] = timed(tr, keep)
    out["classic_last_commit_was_a_refit"] = bool(tr.info(capi.LS_INFO_LAST_COMMIT_REFIT))
    tr.close()
print(json.dumps(out), flush=True)
