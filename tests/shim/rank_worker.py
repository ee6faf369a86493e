#!/usr/bin/env python3
"""One rank of a world > 1 group ON ONE DEVICE -- a child process of tests/test_gpu_group_shim.py (test infrastructure).

The group's collectives go through tests/shim/librccl_shim.so (LS_GROUP_RCCL_LIBRARY, set by the parent): the product code
that runs here is include/lidarshooter_group.h / lidarshooter_hip.h exactly as on a node with one GPU per rank --
ls_group_create_opts (communicators, duplicates, the agreement collective), ls_group_trace / lsh_group_stream_frames,
ls_group_download_cloud -- against a real peer.  Every cloud is compared here, on the rank, with the CPU oracle; the
verdict goes to the parent as one JSON file.  No torch in this process (LS_HIP_STANDALONE=1): ctypes + numpy only.
"""
from __future__ import annotations

import argparse
import ctypes as C
import hashlib
import json
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("LS_HIP_STANDALONE", "1")

import numpy as np  # noqa: E402

from lidarshooter_amd import capi, groupapi, hostapi, synth  # noqa: E402
from oracle import oracle as O  # noqa: E402  (the checker)

DATA = os.path.join(ROOT, "tests", "golden", "data")


def exchange_id(args) -> bytes:
    """rank 0 makes the id (ncclGetUniqueId through ls_group_unique_id) and leaves it in a file; the others wait for the file"""
    if args.rank == 0:
        buf = (C.c_uint8 * groupapi.ID_BYTES)()
        assert groupapi.load().ls_group_unique_id(buf) == 0
        tmp = args.id_file + ".tmp"
        with open(tmp, "wb") as f:
            f.write(bytes(buf))
        os.rename(tmp, args.id_file)
        return bytes(buf)
    t0 = time.time()
    while not os.path.exists(args.id_file):
        if time.time() - t0 > 120:
            raise RuntimeError("rank 0 never wrote the id")
        time.sleep(0.01)
    return open(args.id_file, "rb").read()


def with_raster(base, vertical, h_begin, h_end, h_count):
    """the shipped sensor's pose with another raster (BASELINE.md section 4's SYN sensors)"""
    return O.Sensor(uid="syn", vertical=np.asarray(vertical, np.float32), h_begin=np.float32(h_begin), h_end=np.float32(h_end), h_count=int(h_count),
                    R=base.R, Rinv=base.Rinv, t=base.t)


def tracer_for(s):
    same_cloud.H = s.H
    tr = capi.Tracer(s.vertical, s.h_begin, s.h_end, s.h_count, s.Rinv, s.t, device=0)
    tr.setOption(capi.LS_OPT_ENGINE, capi.ENGINE_PROJECTION)
    return tr


def group_facts(g, tr):
    return {"comm_ranks": g.info(groupapi.INFO_COMM_RANKS), "communicators": g.info(groupapi.INFO_COMMUNICATORS),
            "per_set": g.info(groupapi.INFO_PER_SET), "mine": g.info(groupapi.INFO_ARRANGEMENT_MINE),
            "common": g.info(groupapi.INFO_ARRANGEMENT_COMMON), "shim": g.info(groupapi.INFO_COLLECTIVES_ARE_A_SHIM),
            "version": g.info(groupapi.INFO_RCCL_VERSION), "device": g.info(groupapi.INFO_COMM_DEVICE),
            "frame_graph_state": tr.info(capi.LS_INFO_FRAME_GRAPH_STATE), "pipeline_mode": tr.info(capi.LS_INFO_PIPELINE_MODE)}


CHECKS = []   # every comparison of a downloaded cloud with the oracle's, for the parent's failure message


def ray_order(got, H, world):
    """The gathered cloud is SECTOR-major -- rank 0's records (every channel, its columns), then rank 1's, ... (lidarshooter_group.h);
    the oracle's is in ray-index order (the reference's own order is a thread interleaving, SURVEY.md appendix: compare per ray).
    -> the same cloud in ray order, after checking that it really is sector-major with ascending rays inside every sector."""
    pts, hits = got
    if hits.shape[0] == 0:
        return pts, hits
    col = (hits[:, 0] % np.uint32(H)).astype(np.int64)
    sector = np.zeros_like(col)
    for r in range(world):
        first, n = groupapi.shard_columns(H, world, r)
        sector[(col >= first) & (col < first + n)] = r
    assert np.all(np.diff(sector) >= 0), "the gathered cloud is not sector-major"
    ray = hits[:, 0].astype(np.int64)
    assert np.all((np.diff(ray) > 0) | (np.diff(sector) > 0)), "rays are not ascending inside a sector"
    order = np.argsort(hits[:, 0], kind="stable")
    return pts[order], hits[order]


def same_cloud(got, ref, what=""):
    pts, hits = ray_order(got, same_cloud.H, same_cloud.world)
    ok = bool(np.array_equal(pts, ref["points"]) and np.array_equal(hits, ref["hits"]))
    rec = {"what": what, "points": int(pts.shape[0]), "oracle": int(ref["points"].shape[0]), "equal": ok}
    if not ok and pts.shape[0] and ref["hits"].shape[0]:
        got_rays, ref_rays = set(hits[:, 0].tolist()), set(ref["hits"][:, 0].tolist())
        rec["rays_missing"] = sorted(ref_rays - got_rays)[:8]
        rec["rays_extra"] = sorted(got_rays - ref_rays)[:8]
        rec["n_missing_extra"] = [len(ref_rays - got_rays), len(got_rays - ref_rays)]
    CHECKS.append(rec)
    return ok


def xt32(args, out):
    """XT-32 lidar_0000 over ground + ben with ben's pose changing every frame: every frame's whole cloud on THIS rank is the
    oracle's (1781 points at the identity pose -- OptixTracer_test.cpp:122-169)."""
    s = O.load_sensor(os.path.join(DATA, "config", "hesai-pandar-XT-32-lidar_0000.json"))
    ground = O.load_stl(os.path.join(DATA, "mesh", "ground.stl"))
    ben = O.load_stl(os.path.join(DATA, "mesh", "ben.stl"))
    tr = tracer_for(s)
    tr.addGeometry("ground", ground[0].shape[0], ground[1].shape[0])
    tr.addGeometry("face", ben[0].shape[0], ben[1].shape[0])
    tr.updateGeometry("ground", O.IDENTITY_AFFINE, *ground)
    tr.updateGeometry("face", O.IDENTITY_AFFINE, *ben)
    g = groupapi.Group(tr, args.world, args.rank, groupapi.SHARDED, exchange_id(args), flags=args.flags)
    out["group"] = group_facts(g, tr)
    poses = [O.affine_from_components(np.array((0.4 * k, -0.3 * k, 0.02 * k), np.float32), np.array((0.0, 0.0, 0.15 * k), np.float32)) for k in range(7)]
    refs = [O.trace_frame(s, [(0, *ground, O.IDENTITY_AFFINE), (1, *ben, A)]) for A in poses]
    out["points"] = [int(r["points"].shape[0]) for r in refs]
    ok = True
    for f, A in enumerate(poses):
        tr.updateGeometryTransform("face", A)
        assert tr.commitScene() == 0
        assert g.trace(f) == 0
        if f >= 2 and f % 2 == 0:
            for k in (f - 2, f - 1, f):
                ok = same_cloud(g.download(k), refs[k], f"trace {k} at {f}") and ok
    # ... and through the C++ frame loop (host_capi.cpp: lsh_group_stream_frames), poses restated from a list
    HL = hostapi.load()
    f32p = C.POINTER(C.c_float)
    HL.lsh_group_stream_frames.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(f32p), C.POINTER(C.c_uint), C.c_uint, C.c_uint, C.c_uint]
    names = (C.c_char_p * 2)(b"ground", b"face")
    ident = (C.c_float * 12)(*[float(x) for x in O.IDENTITY_AFFINE])
    flat = (C.c_float * (12 * len(poses)))(*[float(x) for A in poses for x in np.asarray(A, np.float32).reshape(-1)])
    aff = (f32p * 2)(C.cast(ident, f32p), C.cast(flat, f32p))
    n_aff = (C.c_uint * 2)(1, len(poses))
    first = 14   # (a multiple of len(poses): frame f takes pose f mod 7)
    assert HL.lsh_group_stream_frames(g.g, tr.h, names, aff, n_aff, 2, first, 9) == 0, g.L.ls_group_last_error(g.g)
    for k in (first + 6, first + 7, first + 8):
        ok = same_cloud(g.download(k), refs[k % len(poses)], f"stream {k}") and ok
    out["identity_points"] = int(g.download(first + 7)[0].shape[0])   # frame 21: pose 0
    out["clouds_equal_oracle"] = ok
    g.close()
    rc, pts, _ = tr.traceScene(99)   # the tracer is the caller's again: the whole raster, on its own, in ray order
    out["after_close_equal_oracle"] = bool(rc == 0 and np.array_equal(pts, refs[(first + 8) % len(poses)]["points"]))
    tr.close()


def syn(args, out):
    """SYN-128 x SYN-1M (BASELINE.json configs[3], the headline): the sharded frame's whole cloud, hashed."""
    s = O.load_sensor(os.path.join(DATA, "config", "hesai-pandar-XT-32-lidar_0000.json"))
    s = with_raster(s, synth.syn_vertical(128), 0.0, 360.0, 4096)
    v, t = synth.syn_1m()
    tr = tracer_for(s)
    tr.addGeometry("ground", v.shape[0], t.shape[0])
    tr.updateGeometry("ground", O.IDENTITY_AFFINE, v, t)
    g = groupapi.Group(tr, args.world, args.rank, groupapi.SHARDED, exchange_id(args), flags=args.flags)
    out["group"] = group_facts(g, tr)
    for f in range(5):
        assert tr.commitScene() == 0 and g.trace(f) == 0
    pts, hits = ray_order(g.download(4), s.H, args.world)   # (checks the sector-major layout on the way)
    out["points"] = int(pts.shape[0])
    out["points_sha256"] = hashlib.sha256(np.ascontiguousarray(pts).tobytes()).hexdigest()
    out["rays_ascending"] = bool(np.all(np.diff(hits[:, 0].astype(np.int64)) > 0))
    g.close()
    tr.close()


def small_scene():
    """a 64 x 1024 raster at lidar_0000's pose over a 200 x 100-cell ground: 65 536 rays, 40 000 triangles (the oracle traces it in seconds)"""
    s = O.load_sensor(os.path.join(DATA, "config", "hesai-pandar-XT-32-lidar_0000.json"))
    s = with_raster(s, synth.syn_vertical(64), 0.0, 360.0, 1024)
    v, t = synth.grid_mesh(200, 100)
    return s, v, t


def wall_in_second_half(s):
    """two huge triangles in the plane y = -5 of the SENSOR frame, given in world coordinates: every ray of the azimuth columns
    with sin(phi) < 0 -- the second half of the turn, rank 1's sector of two -- hits it, no ray of the first half does"""
    R = np.linalg.inv(np.asarray(s.Rinv, np.float64).reshape(3, 3))
    t = np.asarray(s.t, np.float64)
    c = np.array([(-1e4, -5.0, -1e4), (1e4, -5.0, -1e4), (1e4, -5.0, 1e4), (-1e4, -5.0, 1e4)], np.float64)
    world = (c @ R.T + t).astype(np.float32)
    return world, np.array([(0, 1, 2), (0, 2, 3)], np.uint32)


def sized(args, out):
    """LS_GROUP_FLAG_SIZED_GATHER with a truncation caused by ONE rank: the gather shrinks to what the ground needs; then a wall
    that only rank 1's sector sees makes rank 1's hits outgrow it.  Both ranks must report that frame truncated -- alike, the
    number comes out of the gathered headers -- never deliver it short; three frames later the gather has grown and the cloud
    with the wall is the oracle's on both ranks."""
    s, v, t = small_scene()
    wv, wt = wall_in_second_half(s)
    tr = tracer_for(s)
    tr.addGeometry("ground", v.shape[0], t.shape[0])
    tr.updateGeometry("ground", O.IDENTITY_AFFINE, v, t)
    g = groupapi.Group(tr, args.world, args.rank, groupapi.SHARDED, exchange_id(args), flags=args.flags | groupapi.FLAG_SIZED_GATHER)
    out["group"] = group_facts(g, tr)
    cap_full = g.info(groupapi.INFO_GATHER_CAPACITY)
    ref0 = O.trace_frame(s, [(0, v, t, O.IDENTITY_AFFINE)])
    f = 0
    for _ in range(9):
        assert tr.commitScene() == 0 and g.trace(f) == 0
        f += 1
    ok = same_cloud(g.download(f - 1), ref0)
    out["capacity_full_then_sized"] = [cap_full, g.info(groupapi.INFO_GATHER_CAPACITY)]
    out["truncated_before"] = g.info(groupapi.INFO_TRUNCATED_FRAMES)
    tr.addGeometry("wall", wv.shape[0], wt.shape[0])
    tr.updateGeometry("wall", O.IDENTITY_AFFINE, wv, wt)
    ref1 = O.trace_frame(s, [(0, v, t, O.IDENTITY_AFFINE), (1, wv, wt, O.IDENTITY_AFFINE)])
    out["points_without_and_with_wall"] = [int(ref0["points"].shape[0]), int(ref1["points"].shape[0])]
    assert tr.commitScene() == 0 and g.trace(f) == 0
    g.synchronize()
    out["status_of_the_outgrown_frame"] = g.frameStatus(f)
    try:
        g.download(f)
        out["outgrown_frame_was_delivered"] = True
    except capi.LidarShooterHipError as e:
        out["outgrown_frame_was_delivered"] = False
        out["download_error"] = str(e)
    f += 1
    for _ in range(5):
        assert tr.commitScene() == 0 and g.trace(f) == 0
        f += 1
    ok = ok and same_cloud(g.download(f - 1), ref1)
    out["truncated_after"] = g.info(groupapi.INFO_TRUNCATED_FRAMES)
    out["capacity_at_the_end"] = g.info(groupapi.INFO_GATHER_CAPACITY)
    out["clouds_equal_oracle"] = ok
    g.close()
    tr.close()


def empty(args, out):
    """a frame over an EMPTY scene in the middle of a stream (OptixTracer.cpp:263-288: -1 and a cleared cloud): an empty slot
    travels from every rank, the cloud has no points, the frames before and after are the oracle's"""
    s, v, t = small_scene()
    tr = tracer_for(s)
    tr.addGeometry("ground", v.shape[0], t.shape[0])
    tr.updateGeometry("ground", O.IDENTITY_AFFINE, v, t)
    g = groupapi.Group(tr, args.world, args.rank, groupapi.SHARDED, exchange_id(args), flags=args.flags)
    out["group"] = group_facts(g, tr)
    ref = O.trace_frame(s, [(0, v, t, O.IDENTITY_AFFINE)])
    rcs = []
    for f in range(4):
        assert tr.commitScene() == 0
        rcs.append(g.trace(f))
    ok = same_cloud(g.download(3), ref)
    assert tr.removeGeometry("ground") == 0
    rcs.append(tr.commitScene())          # -1: nothing to commit
    rcs.append(g.trace(4))                # -1: an empty slot travels
    pts, hits = g.download(4)
    out["points_of_the_empty_frame"] = int(pts.shape[0])
    ok = ok and same_cloud(g.download(3), ref)    # (its neighbours in the other sets are untouched)
    tr.addGeometry("ground", v.shape[0], t.shape[0])
    tr.updateGeometry("ground", O.IDENTITY_AFFINE, v, t)
    for f in range(5, 9):
        assert tr.commitScene() == 0
        rcs.append(g.trace(f))
    ok = ok and same_cloud(g.download(8), ref) and same_cloud(g.download(6), ref)
    out["return_codes"] = rcs
    out["clouds_equal_oracle"] = ok
    g.close()
    tr.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rank", type=int, required=True)
    ap.add_argument("--world", type=int, required=True)
    ap.add_argument("--scenario", required=True, choices=["xt32", "syn", "sized", "empty"])
    ap.add_argument("--flags", type=int, default=2)
    ap.add_argument("--id-file", required=True)
    ap.add_argument("--out", required=True)
    args = ap.parse_args()
    out = {"rank": args.rank, "scenario": args.scenario, "flags": args.flags, "ok": False}
    same_cloud.world = args.world
    try:
        O.build()
        {"xt32": xt32, "syn": syn, "sized": sized, "empty": empty}[args.scenario](args, out)
        out["ok"] = True
    except BaseException as e:   # the parent reads the file whatever happened
        out["error"] = repr(e)
        out["traceback"] = traceback.format_exc()
    out["checks"] = CHECKS
    with open(args.out + ".tmp", "w") as f:
        json.dump(out, f)
    os.rename(args.out + ".tmp", args.out)
    sys.exit(0 if out["ok"] else 1)


if __name__ == "__main__":
    main()
