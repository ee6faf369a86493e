"""CPU ORACLE front-end (numpy + ctypes over oracle/liblsoracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under lidarshooter_amd/ may import this module.

Restates the reference's host-side steps that feed the tracer (paths relative to
/root/reference/ros_ws/src/lidarshooter/src/):
  * sensor JSON parsing ............ LidarDevice.cpp:482-633, :758-822 (jsoncpp, comments allowed)
  * STL ingest ..................... test/EmbreeTracer_test.cpp:42-45 (pcl::io::loadPolygonFileSTL:
                                     binary STL, exactly-equal vertices merged in first-seen order,
                                     triangle order kept; pinned by 98 vertices / 162 triangles)
  * per-frame sequence ............. MeshProjector.cpp:446-464 (update every mesh, commit, trace)
and drives the C restatement in ls_oracle.c for everything numeric.
"""
from __future__ import annotations

import ctypes as C
import json
import os
import struct
import subprocess
from dataclasses import dataclass, field

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liblsoracle.so")
INVALID = 0xFFFFFFFF


def build(force: bool = False) -> str:
    """Compile liblsoracle.so with the committed Makefile (gcc only)."""
    src = os.path.join(_HERE, "ls_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "liblsoracle.so"])
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        f32p, u32p, u8p, u64p = (C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.POINTER(C.c_uint8),
                                 C.POINTER(C.c_uint64))
        L.lso_pose_from_quat.argtypes = [C.c_float] * 4 + [f32p, f32p]
        L.lso_azimuth_step.argtypes = [C.c_float, C.c_float, C.c_uint32]
        L.lso_azimuth_step.restype = C.c_float
        L.lso_ray_tables.argtypes = [f32p, C.c_uint32, C.c_float, C.c_float, C.c_uint32, f32p, f32p, f32p, f32p]
        L.lso_ray_dirs.argtypes = [f32p, C.c_uint32, C.c_float, C.c_float, C.c_uint32, f32p]
        L.lso_affine_from_components.argtypes = [f32p, f32p, f32p]
        L.lso_transform_vertices.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, f32p, f32p, f32p, f32p]
        L.lso_cloud_to_world.argtypes = [u8p, C.c_uint32, f32p, f32p, f32p, u8p]
        L.lso_tri_intersect.argtypes = [f32p] * 5 + [f32p]
        L.lso_tri_intersect.restype = C.c_int
        L.lso_set_edge_rule.argtypes = [C.c_int]
        L.lso_get_edge_rule.restype = C.c_int
        L.lso_trace_bruteforce.argtypes = [f32p, C.c_uint32, f32p, u32p, C.c_uint32, f32p, u32p, C.c_int]
        L.lso_bvh_build.argtypes = [f32p, u32p, C.c_uint32, C.c_int]
        L.lso_bvh_build.restype = C.c_void_p
        L.lso_bvh_free.argtypes = [C.c_void_p]
        L.lso_bvh_node_count.argtypes = [C.c_void_p]
        L.lso_bvh_node_count.restype = C.c_uint32
        L.lso_bvh_trace.argtypes = [C.c_void_p, f32p, C.c_uint32, f32p, u32p, C.c_int, u64p]
        L.lso_pack_points.argtypes = [f32p, u32p, f32p, C.c_uint32, C.c_uint32, u32p, u32p, C.c_uint32, u8p, u32p]
        L.lso_pack_points.restype = C.c_uint32
        L.lso_fat_traverse_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, f32p,
                                             C.c_uint32, f32p, u32p, u64p, u32p]
        _lib = L
    return _lib


class edge_rule:
    """TEST-ONLY: `with edge_rule(1):` runs the closest-hit test with the packet form of Embree's third edge test
    (W = absDen - U - V >= 0) instead of the single-ray form (U + V <= absDen); ls_oracle.c:tri_test.  The default (0)
    is what every parity test, smoke() and bench.py's cpu_baseline use."""

    def __init__(self, rule: int):
        self.rule = int(rule)

    def __enter__(self):
        self.prev = lib().lso_get_edge_rule()
        lib().lso_set_edge_rule(self.rule)
        return self

    def __exit__(self, *exc):
        lib().lso_set_edge_rule(self.prev)
        return False


def _p(a: np.ndarray, ty):
    return a.ctypes.data_as(C.POINTER(ty))


# ---------------------------------------------------------------------------------------------
# a-1  sensor JSON  (LidarDevice.cpp:482-633).  The shipped files carry `//` comment lines AND a
# "http://..." string, so comments are stripped only outside string literals.
# ---------------------------------------------------------------------------------------------
def strip_json_comments(text: str) -> str:
    out, i, n, in_str = [], 0, len(text), False
    while i < n:
        ch = text[i]
        if in_str:
            out.append(ch)
            if ch == "\\" and i + 1 < n:
                out.append(text[i + 1])
                i += 1
            elif ch == '"':
                in_str = False
        elif ch == '"':
            in_str = True
            out.append(ch)
        elif ch == "/" and i + 1 < n and text[i + 1] == "/":
            while i < n and text[i] != "\n":
                i += 1
            continue
        elif ch == "/" and i + 1 < n and text[i + 1] == "*":
            j = text.find("*/", i + 2)
            i = n if j < 0 else j + 2
            continue
        else:
            out.append(ch)
        i += 1
    return "".join(out)


@dataclass
class Sensor:
    uid: str
    vertical: np.ndarray          # float32[V], degrees above horizon
    h_begin: np.float32
    h_end: np.float32
    h_count: int
    R: np.ndarray                 # float32[9] row-major
    Rinv: np.ndarray              # float32[9] row-major
    t: np.ndarray                 # float32[3] = (base_to_origin.tx, .ty, sensor_to_base.tz)
    frame_id: str = "PandarXT-32"
    point_step: int = 32
    is_bigendian: bool = False
    is_dense: bool = True
    fields: list = field(default_factory=list)

    @property
    def V(self) -> int:
        return int(self.vertical.shape[0])

    @property
    def H(self) -> int:
        return int(self.h_count)

    @property
    def total_rays(self) -> int:   # LidarDevice.cpp:617
        return self.V * self.H

    def step(self) -> np.float32:
        return np.float32(lib().lso_azimuth_step(float(self.h_begin), float(self.h_end), self.h_count))


def pose_from_quat(qw, qx, qy, qz):
    R = np.zeros(9, np.float32)
    Rinv = np.zeros(9, np.float32)
    lib().lso_pose_from_quat(float(np.float32(qw)), float(np.float32(qx)), float(np.float32(qy)),
                             float(np.float32(qz)), _p(R, C.c_float), _p(Rinv, C.c_float))
    return R, Rinv


def load_sensor(path: str) -> Sensor:
    with open(path, "r") as fh:
        j = json.loads(strip_json_comments(fh.read()))
    dev = j["device"]
    if "sensorConfig" in dev:                       # inline pose has the highest precedence (:505)
        tr = dev["sensorConfig"]
    elif "sensorConfigFile" in dev:                 # (:511-531)
        with open(dev["sensorConfigFile"], "r") as fh:
            tr = json.loads(strip_json_comments(fh.read()))
    else:
        raise ValueError("sensor pose over HTTP (LidarDevice.cpp:533-554) is out of scope")
    b2o, s2b = tr.get("base_to_origin", {}), tr.get("sensor_to_base", {})
    f32 = np.float32                                # Json::Value::asFloat: double -> float
    R, Rinv = pose_from_quat(f32(s2b.get("qw", 0.0)), f32(s2b.get("qx", 0.0)),
                             f32(s2b.get("qy", 0.0)), f32(s2b.get("qz", 0.0)))
    t = np.array([f32(b2o.get("tx", 0.0)), f32(b2o.get("ty", 0.0)), f32(s2b.get("tz", 0.0))], np.float32)
    ch = j["channels"]
    msg = j.get("message", {})
    return Sensor(
        uid=dev.get("sensorUid", tr.get("uid", "")),
        vertical=np.array([f32(a) for a in ch["vertical"]], np.float32),
        h_begin=f32(ch["horizontal"]["range"].get("begin", 0.0)),
        h_end=f32(ch["horizontal"]["range"].get("end", 360.0)),
        h_count=int(ch["horizontal"].get("count", 128)),
        R=R, Rinv=Rinv, t=t,
        frame_id=msg.get("frameId", "PandarXT-32"),
        point_step=int(msg.get("pointStep", 32)),
        is_bigendian=bool(msg.get("isBigendian", False)),
        is_dense=bool(msg.get("isDense", True)),
        fields=[(f["name"], int(f["offset"]), int(f["datatype"]), int(f["count"]))
                for f in msg.get("pointFields", [])],
    )


def init_message(sensor: Sensor, frame_index: int) -> dict:
    """LidarDevice::initMessage (LidarDevice.cpp:94-115) minus the ROS time stamp."""
    return dict(frame_id=sensor.frame_id, seq=int(frame_index), height=1, width=0,
                point_step=sensor.point_step, row_step=0, is_bigendian=sensor.is_bigendian,
                is_dense=sensor.is_dense, fields=list(sensor.fields))


# ---------------------------------------------------------------------------------------------
# STL ingest (pcl::io::loadPolygonFileSTL -> vtkSTLReader with point merging)
# ---------------------------------------------------------------------------------------------
def load_stl(path: str):
    """-> (verts float32[Nv,3], tris uint32[Nt,3]); exact-equality weld, first-seen order."""
    with open(path, "rb") as fh:
        raw = fh.read()
    (nt,) = struct.unpack_from("<I", raw, 80)
    assert len(raw) == 84 + 50 * nt, "not a binary STL"
    rec = np.frombuffer(raw, dtype=np.uint8, count=50 * nt, offset=84).reshape(nt, 50)
    corners = rec[:, 12:48].copy().view(np.float32).reshape(nt * 3, 3)
    keys = corners.view(np.uint32).reshape(-1, 3).copy()
    keys[keys == 0x80000000] = 0                    # -0.0 == +0.0 under float equality
    seen: dict = {}
    idx = np.empty(nt * 3, np.uint32)
    order = []
    for i, k in enumerate(map(tuple, keys.tolist())):
        j = seen.get(k)
        if j is None:
            j = len(order)
            seen[k] = j
            order.append(i)
        idx[i] = j
    return corners[np.array(order, np.int64)].copy(), idx.reshape(nt, 3).copy()


# ---------------------------------------------------------------------------------------------
# numeric wrappers
# ---------------------------------------------------------------------------------------------
def ray_tables(sensor: Sensor):
    V, H = sensor.V, sensor.H
    st, ct = np.zeros(V, np.float32), np.zeros(V, np.float32)
    sp, cp = np.zeros(H, np.float32), np.zeros(H, np.float32)
    lib().lso_ray_tables(_p(sensor.vertical, C.c_float), V, float(sensor.h_begin), float(sensor.h_end), H,
                         _p(st, C.c_float), _p(ct, C.c_float), _p(sp, C.c_float), _p(cp, C.c_float))
    return st, ct, sp, cp


def ray_dirs(sensor: Sensor) -> np.ndarray:
    d = np.zeros((sensor.total_rays, 3), np.float32)
    lib().lso_ray_dirs(_p(sensor.vertical, C.c_float), sensor.V, float(sensor.h_begin), float(sensor.h_end),
                       sensor.H, _p(d, C.c_float))
    return d


RAY_DTYPE = np.dtype([("origin", "<f4", 3), ("tmin", "<f4"), ("direction", "<f4", 3), ("tmax", "<f4")])   # Ray.hpp:16-35
HIT_DTYPE = np.dtype([("t", "<f4"), ("normal", "<f4", 3), ("intensity", "<f4"), ("ring", "<i4")])       # Hit.hpp:16-29


def all_rays_aos(sensor: Sensor):
    """LidarDevice::allRaysGPU's two buffers (LidarDeviceKernels.cu:38-51), ray index v*H + h: Ray{origin = 0,
    direction = the table direction} and Hit{t = 1e16, intensity = 64.0, ring = v}.  The reference's kernel leaves
    tmin / tmax / normal as it finds them; they are restated as its OptiX programs use them (tmin 0, tmax 1e16,
    OptixTracerModules.cu:45-46) and as zeros.  Directions are the CPU path's (libm tables, LidarDevice.cpp:306-316): the
    parity target is the Embree backend, not CUDA's sinf/cosf."""
    n = sensor.total_rays
    rays = np.zeros(n, RAY_DTYPE)
    rays["direction"] = ray_dirs(sensor)
    rays["tmax"] = np.float32(1e16)
    hits = np.zeros(n, HIT_DTYPE)
    hits["t"] = np.float32(1e16)
    hits["intensity"] = np.float32(64.0)
    hits["ring"] = np.repeat(np.arange(sensor.V, dtype=np.int32), sensor.H)
    return rays, hits


def affine_from_components(lin, ang) -> np.ndarray:
    lin = np.ascontiguousarray(lin, np.float32)
    ang = np.ascontiguousarray(ang, np.float32)
    A = np.zeros(12, np.float32)
    lib().lso_affine_from_components(_p(lin, C.c_float), _p(ang, C.c_float), _p(A, C.c_float))
    return A


IDENTITY_AFFINE = np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], np.float32)


def play_trajectory(path: str, period: float = 0.1) -> np.ndarray:
    """config/trajectory.json through AffineMesh::joystickCallback (AffineMesh.cpp:107-128, :275-284):
    lin += (Rz*Ry*Rx)(ang) * twist.linear; ang += twist.angular, one message per `period` seconds,
    round(dt/period) messages per segment.  -> float32[n,6] poses (linear, angular)."""
    with open(path, "r") as fh:
        j = json.loads(strip_json_comments(fh.read()))
    lin = np.zeros(3, np.float32)
    ang = np.zeros(3, np.float32)
    out = []
    for seg in j["trajectory"]:
        tl = np.array([np.float32(x) for x in seg["linear"]], np.float32)
        ta = np.array([np.float32(x) for x in seg["angular"]], np.float32)
        for _ in range(int(round(float(np.float32(seg["dt"])) / period))):
            R = affine_from_components(np.zeros(3, np.float32), ang).reshape(3, 4)[:, :3]
            g = np.float32(np.float32(np.float32(R[:, 0] * tl[0]) + np.float32(R[:, 1] * tl[1])) + np.float32(R[:, 2] * tl[2]))
            lin = (lin + g).astype(np.float32)
            ang = (ang + ta).astype(np.float32)
            out.append(np.concatenate([lin, ang]))
    return np.array(out, np.float32).reshape(-1, 6)


def transform_vertices(verts: np.ndarray, affine: np.ndarray, sensor: Sensor, stride: int | None = None):
    verts = np.ascontiguousarray(verts)
    if stride is None:
        verts = verts.astype(np.float32, copy=False)
        stride, n = 12, verts.shape[0]
    else:
        n = verts.nbytes // stride
    out = np.zeros((n, 3), np.float32)
    A = np.ascontiguousarray(affine, np.float32)
    lib().lso_transform_vertices(verts.ctypes.data, stride, n, _p(A, C.c_float), _p(sensor.Rinv, C.c_float),
                                 _p(sensor.t, C.c_float), _p(out, C.c_float))
    return out


def cloud_to_world(points: np.ndarray, sensor: Sensor, affine: np.ndarray | None = None) -> np.ndarray:
    """f-4: CloudTransformer::applyInverseTransform (CloudTransformer.cpp:283-318) of a cloud of 32-byte
    XYZIR records: p_world = R * (T * p) + (tx, ty, tz) (LidarDevice.cpp:393-401).  -> uint8[n, 32]."""
    pts = np.ascontiguousarray(points, np.uint8).reshape(-1, 32)
    out = np.empty_like(pts)
    A = np.ascontiguousarray(IDENTITY_AFFINE if affine is None else affine, np.float32)
    lib().lso_cloud_to_world(_p(pts, C.c_uint8), pts.shape[0], _p(A, C.c_float), _p(sensor.R, C.c_float),
                             _p(sensor.t, C.c_float), _p(out, C.c_uint8))
    return out


@dataclass
class Scene:
    """Geometries in geomID order, concatenated (global triangle id orders by (geomID, primID))."""
    verts: np.ndarray          # float32[Nv,3], sensor frame
    tris: np.ndarray           # uint32[Nt,3], indices into verts
    geom_first: np.ndarray     # uint32[G] first global triangle id per geometry slot
    geom_ids: np.ndarray       # uint32[G] geomID of each slot (ascending)
    geom_quad: np.ndarray = None   # bool[G]: RTC_GEOMETRY_TYPE_QUAD slot (two triangles per element, primID = triangle // 2)


def quads_to_triangles(quads: np.ndarray) -> np.ndarray:
    """Embree 3's quad = the triangle pair (v0,v1,v3), (v2,v3,v1) (its API documentation of RTC_GEOMETRY_TYPE_QUAD; the
    reference hands quads to Embree as they are, EmbreeTracer.cpp:179-198): uint32[n,4] -> uint32[2n,3]."""
    q = np.asarray(quads, np.uint32)
    t = np.empty((2 * q.shape[0], 3), np.uint32)
    t[0::2] = q[:, [0, 1, 3]]
    t[1::2] = q[:, [2, 3, 1]]
    return t


def assemble_scene(sensor: Sensor, meshes) -> Scene:
    """meshes: iterable of (geomID, verts[Nv,3], elements, affine[12]); elements uint32[Nt,3] (triangles) or [Nq,4]
    (quads); sorted by geomID here."""
    meshes = sorted(meshes, key=lambda m: m[0])
    vs, ts, first, ids, quad, vo, to = [], [], [], [], [], 0, 0
    for gid, v, t, A in meshes:
        tv = transform_vertices(v, A, sensor)
        vs.append(tv)
        t = np.asarray(t, np.uint32)
        is_quad = t.ndim == 2 and t.shape[1] == 4
        if is_quad:
            t = quads_to_triangles(t)
        ts.append(t + np.uint32(vo))
        first.append(to)
        ids.append(gid)
        quad.append(is_quad)
        vo += tv.shape[0]
        to += len(t)
    if not vs:
        return Scene(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.uint32), np.zeros(0, np.uint32),
                     np.zeros(0, np.uint32), np.zeros(0, bool))
    return Scene(np.concatenate(vs), np.concatenate(ts).astype(np.uint32), np.array(first, np.uint32),
                 np.array(ids, np.uint32), np.array(quad, bool))


def trace_bruteforce(dirs: np.ndarray, scene: Scene, nthreads: int = 8):
    n = dirs.shape[0]
    t = np.full(n, -1.0, np.float32)
    gid = np.full(n, INVALID, np.uint32)
    dirs = np.ascontiguousarray(dirs, np.float32)
    v = np.ascontiguousarray(scene.verts, np.float32)
    tr = np.ascontiguousarray(scene.tris, np.uint32)
    lib().lso_trace_bruteforce(_p(dirs, C.c_float), n, _p(v, C.c_float), _p(tr, C.c_uint32), tr.shape[0],
                               _p(t, C.c_float), _p(gid, C.c_uint32), nthreads)
    return t, gid


class CpuBvh:
    """Binned-SAH BVH2 + threaded single-ray tracer (cpu_baseline and full-size checker)."""

    def __init__(self, scene: Scene, nthreads: int = 8):
        self._v = np.ascontiguousarray(scene.verts, np.float32)
        self._t = np.ascontiguousarray(scene.tris, np.uint32)
        self.h = lib().lso_bvh_build(_p(self._v, C.c_float), _p(self._t, C.c_uint32), self._t.shape[0], nthreads)

    def node_count(self) -> int:
        return int(lib().lso_bvh_node_count(self.h))

    def trace(self, dirs: np.ndarray, nthreads: int = 8):
        n = dirs.shape[0]
        t = np.full(n, -1.0, np.float32)
        gid = np.full(n, INVALID, np.uint32)
        stats = np.zeros(2, np.uint64)
        dirs = np.ascontiguousarray(dirs, np.float32)
        lib().lso_bvh_trace(self.h, _p(dirs, C.c_float), n, _p(t, C.c_float), _p(gid, C.c_uint32), nthreads,
                            _p(stats, C.c_uint64))
        return t, gid, stats

    def close(self):
        if self.h:
            lib().lso_bvh_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def pack_points(t, gid, dirs, H, scene: Scene):
    """-> (points uint8[n,32], hits uint32[n,4] = {ray, geomID, primID, t bits})."""
    n = t.shape[0]
    pts = np.zeros((n, 32), np.uint8)
    hits = np.zeros((n, 4), np.uint32)
    dirs = np.ascontiguousarray(dirs, np.float32)
    gf = np.ascontiguousarray(scene.geom_first, np.uint32)
    gi = np.ascontiguousarray(scene.geom_ids, np.uint32)
    k = lib().lso_pack_points(_p(t, C.c_float), _p(gid, C.c_uint32), _p(dirs, C.c_float), n, H,
                              _p(gf, C.c_uint32), _p(gi, C.c_uint32), gf.shape[0], _p(pts, C.c_uint8),
                              _p(hits, C.c_uint32))
    pts, hits = pts[:k].copy(), hits[:k].copy()
    if scene.geom_quad is not None and scene.geom_quad.any() and k:   # primID of a quad geometry = triangle // 2
        slot_of = {int(g): i for i, g in enumerate(scene.geom_ids)}
        is_quad = np.array([scene.geom_quad[slot_of[int(g)]] for g in hits[:, 1]], bool)
        hits[is_quad, 2] //= 2
    return pts, hits


def fat_traverse_stats(nodes: np.ndarray, tris: np.ndarray, leaf_size: int, dirs: np.ndarray, per_ray=None):
    """Walk a BVH downloaded from the HIP library (ls_debug_download_bvh layout, see
    include/lidarshooter_hip.h) with the kernel's per-ray order and count node fetches / triangle
    tests.  -> (t, gid, stats[3] = node fetches, triangle tests, deepest stack)."""
    n = dirs.shape[0]
    t = np.full(n, -1.0, np.float32)
    gid = np.full(n, INVALID, np.uint32)
    stats = np.zeros(3, np.uint64)
    dirs = np.ascontiguousarray(dirs, np.float32)
    nodes = np.ascontiguousarray(nodes)
    tris = np.ascontiguousarray(tris)
    ntris = tris.nbytes // 48
    nleaves = (ntris + leaf_size - 1) // leaf_size
    lib().lso_fat_traverse_stats(nodes.ctypes.data, tris.ctypes.data, nleaves, leaf_size, ntris,
                                 _p(dirs, C.c_float), n, _p(t, C.c_float), _p(gid, C.c_uint32),
                                 _p(stats, C.c_uint64), _p(per_ray, C.c_uint32) if per_ray is not None else None)
    return t, gid, stats


def reference_packet_walk(total_rays: int):
    """The ray indices EmbreeTracer::traceScene actually traces, packet by packet (EmbreeTracer.cpp:304-307, :315-318):
    iterations = ceil(rays / 16), chunks = ceil(iterations / 4), and FOUR threads run `chunks` iterations each -- 4 * chunks
    calls of nextRay16 on the one shared iterator, which stops a packet at the last ray and starts again at ray 0 on the next
    call (LidarDevice.cpp:294-342, advanceRayIndex :824-845).  When ceil(rays / 16) is not a multiple of 4 the surplus
    calls re-trace rays from index 0: -> list of (first ray, count) for every executed packet."""
    N = int(total_rays)
    iterations = N // 16 + (1 if N % 16 else 0)
    chunks = iterations // 4 + (1 if iterations % 4 else 0)
    out, pos = [], 0
    for _ in range(4 * chunks):
        n = min(16, N - pos)
        out.append((pos, n))
        pos = (pos + n) % N
    return out


def reference_width(sensor: Sensor, hit_mask) -> int:
    """`cloud.width` the reference's Embree backend reports for a frame whose per-ray hit mask (ray-index order) is
    `hit_mask`: every hit of every executed packet is appended (EmbreeTracer.cpp:338-352), duplicates included, and
    width = the number appended (:364).  Equal to the number of hits whenever ceil(rays / 16) is a multiple of 4 -- every
    shipped config (4800 rays: 300 iterations) and SYN-128 (32 768) -- and larger by the hits among the re-traced leading
    rays otherwise.  The HIP backend (and the OptiX one, OptixTracer.cpp:895-942) reports every hit once."""
    hit = np.asarray(hit_mask, bool).reshape(-1)
    assert hit.shape[0] == sensor.total_rays
    return int(sum(int(hit[a:a + n].sum()) for a, n in reference_packet_walk(sensor.total_rays)))


def trace_frame(sensor: Sensor, meshes, use_bvh: bool = False, nthreads: int = 8):
    """The reference's per-frame sequence on the CPU: update every mesh, commit, trace, pack.
    -> dict(t, gid, points, hits, scene, dirs)."""
    scene = assemble_scene(sensor, meshes)
    dirs = ray_dirs(sensor)
    if scene.tris.shape[0] == 0:
        t = np.full(dirs.shape[0], -1.0, np.float32)
        gid = np.full(dirs.shape[0], INVALID, np.uint32)
    elif use_bvh:
        b = CpuBvh(scene, nthreads)
        t, gid, _ = b.trace(dirs, nthreads)
        b.close()
    else:
        t, gid = trace_bruteforce(dirs, scene, nthreads)
    pts, hits = pack_points(t, gid, dirs, sensor.H, scene)
    return dict(t=t, gid=gid, points=pts, hits=hits, scene=scene, dirs=dirs)
