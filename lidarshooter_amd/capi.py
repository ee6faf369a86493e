"""ctypes binding of the C ABI in include/lidarshooter_hip.h (liblidarshooter_hip.so).

Used by tests/, bench.py and __graft_entry__.py to drive the HIP library exactly the way the
reference's C++ adapter would (one call per ITracer virtual, ITracer.hpp:50-94).  This module
holds no algorithm: if the shared library is missing it raises, it never computes on the CPU.
"""
from __future__ import annotations

import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# LS_LIB_PATH: kernel-variant experiments (tools/) load another build of the same library
LIB_PATH = os.environ.get("LS_LIB_PATH") or os.path.join(_HERE, "liblidarshooter_hip.so")
INVALID = 0xFFFFFFFF
ABI_VERSION = 4   # include/lidarshooter_hip.h: LS_ABI_VERSION (checked at load)

LS_OPT_LEAF_SIZE, LS_OPT_TIMING, LS_OPT_COUNT_VISITS, LS_OPT_ENGINE, LS_OPT_PIPELINE = 1, 2, 3, 5, 6
LS_OPT_HOST_OUTPUT, LS_OPT_READBACK_HITS, LS_OPT_DEBUG_FAULT, LS_OPT_BLOCK_CULL, LS_OPT_BVH_REFIT = 7, 8, 9, 10, 11
LS_INFO_LAST_COMMIT_REFIT = 6
LS_OPT_BVH_INSTANCED = 12
LS_OPT_UPLOAD_MODE = 13
LS_OPT_FRAME_GRAPH = 14
LS_OPT_EMIT_POINTS = 15
LS_OPT_BVH_WIDE = 16
LS_INFO_BVH_WIDE = 16
LS_INFO_FRAME_GRAPH_LAST_PATCHED = 13
LS_INFO_EMIT_POINTS, LS_INFO_FRAME_GRAPH_PATCH_WAITS = 14, 15
LS_INFO_NEXT_SLOT, LS_INFO_FRAME_GRAPH_STATE, LS_INFO_FRAME_GRAPH_CAPTURES, LS_INFO_FRAME_GRAPH_REPLAYS, LS_INFO_FRAME_GRAPH_PATCHES = 8, 9, 10, 11, 12
LS_GEOMETRY_TYPE_TRIANGLE, LS_GEOMETRY_TYPE_QUAD = 0, 1
RAY_DTYPE = np.dtype([("origin", "<f4", 3), ("tmin", "<f4"), ("direction", "<f4", 3), ("tmax", "<f4")])   # Ray.hpp:16-35
REFHIT_DTYPE = np.dtype([("t", "<f4"), ("normal", "<f4", 3), ("intensity", "<f4"), ("ring", "<i4")])       # Hit.hpp:16-29
LS_INFO_BVH_INSTANCED = 7
LS_INFO_CONCURRENT_STREAMS, LS_INFO_PIPELINE_MODE, LS_INFO_DEVICE_STATUS, LS_INFO_HOST_THREADS, LS_INFO_AZIMUTH_COUNT = 1, 2, 3, 4, 5
ENGINE_AUTO, ENGINE_BVH, ENGINE_PROJECTION = 0, 1, 2
STAGES = ("transform", "morton", "sort", "leaves", "range_tree", "hierarchy", "trace", "trace_aux", "pack")

# every symbol include/lidarshooter_hip.h declares (tests/test_abi.py checks the .so exports them all)
SYMBOLS = (
    "ls_abi_version", "ls_source_hash", "ls_tracer_create", "ls_tracer_create_tables", "ls_tracer_destroy", "ls_parallel_copy", "ls_expand_points", "ls_get_info", "ls_affine_from_components", "ls_add_geometry", "ls_remove_geometry",
    "ls_update_geometry", "ls_update_geometry_components", "ls_update_geometry_device",
    "ls_update_geometry_device_shared", "ls_update_geometry_transform", "ls_commit_scene", "ls_trace_scene", "ls_trace_scene_async",
    "ls_geometry_count", "ls_geometry_id", "ls_vertex_count", "ls_element_count", "ls_total_rays",
    "ls_total_channels", "ls_last_error", "ls_tracer_set_shard", "ls_tracer_set_stream",
    "ls_tracer_synchronize", "ls_tracer_flush", "ls_tracer_set_output_buffers", "ls_tracer_set_hit_buffers", "ls_expand_gathered_hits", "ls_expand_gathered_hits_on", "ls_cloud_to_world", "ls_tracer_set_option", "ls_get_timings",
    "ls_get_visit_counts", "ls_generate_rays", "ls_generate_rays_aos", "ls_geometry_type", "ls_tracer_order_after_last_frame", "ls_tracer_wait_event", "ls_tracer_next_frame_waits", "ls_trace_scene_begin", "ls_trace_scene_expand",
    "ls_frame_graph_begin", "ls_frame_graph_stream", "ls_frame_graph_end", "ls_frame_graph_reset",
    "ls_tracer_set_sensor", "ls_tracer_set_sensor_tables", "ls_expand_gathered_hits_sized",
)
# include/lidarshooter_hip_debug.h: test / measurement hooks (not part of the drop-in surface)
DEBUG_SYMBOLS = ("ls_debug_dense_hits", "ls_debug_trace_bruteforce", "ls_debug_scene_size", "ls_debug_download_scene",
                 "ls_debug_download_bvh", "ls_debug_sort_pairs", "ls_debug_expand_hits")


class SensorDesc(C.Structure):
    _fields_ = [("vertical_deg", C.POINTER(C.c_float)), ("n_vertical", C.c_uint32), ("h_begin", C.c_float),
                ("h_end", C.c_float), ("h_count", C.c_uint32), ("Rinv", C.c_float * 9), ("t", C.c_float * 3)]


class SensorTables(C.Structure):
    _fields_ = [("sin_theta", C.POINTER(C.c_float)), ("cos_theta", C.POINTER(C.c_float)),
                ("elevation_deg", C.POINTER(C.c_float)), ("n_vertical", C.c_uint32),
                ("sin_phi", C.POINTER(C.c_float)), ("cos_phi", C.POINTER(C.c_float)), ("h_count", C.c_uint32),
                ("h_begin_deg", C.c_float), ("h_step_deg", C.c_float), ("Rinv", C.c_float * 9), ("t", C.c_float * 3)]


class Frame(C.Structure):
    _fields_ = [("points32", C.POINTER(C.c_uint8)), ("hits", C.c_void_p), ("n_points", C.c_uint32),
                ("n_rays", C.c_uint32), ("frame", C.c_uint32), ("d_points32", C.c_void_p),
                ("d_hits", C.c_void_p), ("d_n_points", C.c_void_p), ("compact16", C.c_void_p)]


HIT_DTYPE = np.dtype([("ray", "<u4"), ("geom", "<u4"), ("prim", "<u4"), ("t", "<f4")])
NODE_DTYPE = np.dtype([("llo", "<f4", 3), ("left", "<u4"), ("lhi", "<f4", 3), ("right", "<u4"),
                       ("rlo", "<f4", 3), ("pad0", "<u4"), ("rhi", "<f4", 3), ("pad1", "<u4")])
LEAF_BIT = 0x80000000
TRI_DTYPE = np.dtype([("v0", "<f4", 3), ("gid", "<u4"), ("e1", "<f4", 3), ("NgC", "<f4"), ("e2", "<f4", 3),
                      ("pad", "<u4")])

_lib = None


class LidarShooterHipError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load the HIP library; fails loudly when it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64 (soname
    # libamdhip64.so.7, the one this library needs).  Importing torch first makes the loader bind
    # this library to that copy; loading /opt/rocm's copy first and torch's afterwards would put
    # two HIP runtimes in the process and the second one finds no GPU.
    if "torch" not in sys.modules and os.environ.get("LS_HIP_STANDALONE") != "1":
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    if not os.path.exists(LIB_PATH):
        raise LidarShooterHipError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(make -C lidarshooter_amd/csrc). There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp, u32, i32, f32p, u32p = C.c_void_p, C.c_uint32, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_uint32)
    L.ls_abi_version.restype = i32
    if L.ls_abi_version() != ABI_VERSION:   # (a stale .so next to a newer binding, or the other way round)
        raise LidarShooterHipError(f"{LIB_PATH} speaks ABI {L.ls_abi_version()}, this binding {ABI_VERSION}: rebuild (make -C lidarshooter_amd/csrc)")
    L.ls_source_hash.argtypes = []
    L.ls_source_hash.restype = C.c_char_p
    L.ls_tracer_create.argtypes = [C.POINTER(SensorDesc), i32, C.POINTER(vp)]
    L.ls_tracer_create_tables.argtypes = [C.POINTER(SensorTables), i32, C.POINTER(vp)]
    L.ls_tracer_set_sensor.argtypes = [vp, C.POINTER(SensorDesc)]
    L.ls_tracer_set_sensor_tables.argtypes = [vp, C.POINTER(SensorTables)]
    L.ls_affine_from_components.argtypes = [f32p, f32p, f32p]
    L.ls_affine_from_components.restype = None
    L.ls_expand_points.argtypes = [vp, vp, u32]
    L.ls_parallel_copy.argtypes = [vp, vp, C.c_uint64]
    L.ls_get_info.argtypes = [vp, i32]
    L.ls_get_info.restype = C.c_long
    L.ls_tracer_destroy.argtypes = [vp]
    L.ls_tracer_destroy.restype = None
    L.ls_add_geometry.argtypes = [vp, C.c_char_p, i32, i32, i32]
    L.ls_remove_geometry.argtypes = [vp, C.c_char_p]
    L.ls_update_geometry.argtypes = [vp, C.c_char_p, f32p, vp, u32, vp]
    L.ls_update_geometry_components.argtypes = [vp, C.c_char_p, f32p, f32p, vp, u32, vp]
    L.ls_update_geometry_device.argtypes = [vp, C.c_char_p, f32p, vp, u32, vp]
    L.ls_update_geometry_device_shared.argtypes = [vp, C.c_char_p, f32p, vp, u32, vp]
    L.ls_update_geometry_transform.argtypes = [vp, C.c_char_p, f32p]
    L.ls_commit_scene.argtypes = [vp]
    L.ls_trace_scene.argtypes = [vp, u32, C.POINTER(Frame)]
    L.ls_trace_scene_async.argtypes = [vp, u32, C.POINTER(Frame)]
    L.ls_trace_scene_begin.argtypes = [vp, u32, u32p]
    L.ls_trace_scene_expand.argtypes = [vp, vp]
    L.ls_geometry_count.argtypes = [vp]
    L.ls_geometry_count.restype = C.c_long
    L.ls_geometry_id.argtypes = [vp, C.c_char_p]
    L.ls_vertex_count.argtypes = [vp, C.c_char_p]
    L.ls_vertex_count.restype = C.c_long
    L.ls_element_count.argtypes = [vp, C.c_char_p]
    L.ls_element_count.restype = C.c_long
    L.ls_total_rays.argtypes = [vp]
    L.ls_total_rays.restype = u32
    L.ls_total_channels.argtypes = [vp]
    L.ls_total_channels.restype = u32
    L.ls_last_error.argtypes = [vp]
    L.ls_last_error.restype = C.c_char_p
    L.ls_tracer_set_shard.argtypes = [vp, u32, u32]
    L.ls_tracer_set_stream.argtypes = [vp, vp]
    L.ls_tracer_synchronize.argtypes = [vp]
    L.ls_tracer_set_output_buffers.argtypes = [vp, vp, vp, vp, u32]
    L.ls_tracer_set_hit_buffers.argtypes = [vp, vp, vp, u32]
    L.ls_expand_gathered_hits.argtypes = [vp, vp, u32, u32, vp, vp, vp]
    L.ls_expand_gathered_hits_on.argtypes = [vp, vp, vp, u32, u32, vp, vp, vp]
    L.ls_cloud_to_world.argtypes = [vp, f32p, f32p, vp, vp, vp, vp, vp, u32]
    L.ls_tracer_flush.argtypes = [vp]
    L.ls_tracer_order_after_last_frame.argtypes = [vp, vp]
    L.ls_tracer_wait_event.argtypes = [vp, vp]
    L.ls_tracer_next_frame_waits.argtypes = [vp, vp]
    L.ls_frame_graph_begin.argtypes = [vp, C.c_uint64]
    L.ls_frame_graph_stream.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_uint32), C.POINTER(C.c_int)]
    L.ls_frame_graph_end.argtypes = [vp]
    L.ls_frame_graph_reset.argtypes = [vp]
    L.ls_tracer_set_option.argtypes = [vp, i32, i32]
    L.ls_get_timings.argtypes = [vp, f32p]
    L.ls_get_visit_counts.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.ls_generate_rays.argtypes = [vp, vp, vp, vp]
    L.ls_generate_rays_aos.argtypes = [vp, vp, vp]
    L.ls_geometry_type.argtypes = [vp, C.c_char_p]
    L.ls_debug_dense_hits.argtypes = [vp, f32p, u32p]
    L.ls_debug_trace_bruteforce.argtypes = [vp, f32p, u32p]
    L.ls_debug_scene_size.argtypes = [vp, u32p, u32p, u32p, u32p]
    L.ls_debug_download_scene.argtypes = [vp, vp, vp]
    L.ls_debug_download_bvh.argtypes = [vp, vp, vp]
    L.ls_debug_sort_pairs.argtypes = [vp, vp, vp, u32]
    L.ls_debug_expand_hits.argtypes = [vp, vp, u32, vp, vp, vp, u32, u32]
    _lib = L
    return L


def _f32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


IDENTITY_AFFINE = np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], np.float32)


class Tracer:
    """Thin object wrapper over an ls_tracer handle; method names follow ITracer."""

    def __init__(self, vertical_deg, h_begin, h_end, h_count, Rinv, t, device: int = 0):
        self.L = load()
        self._vert = np.ascontiguousarray(vertical_deg, np.float32)
        sd = SensorDesc()
        sd.vertical_deg = _f32p(self._vert)
        sd.n_vertical = self._vert.shape[0]
        sd.h_begin, sd.h_end, sd.h_count = float(h_begin), float(h_end), int(h_count)
        sd.Rinv = (C.c_float * 9)(*[float(x) for x in np.asarray(Rinv, np.float32).reshape(9)])
        sd.t = (C.c_float * 3)(*[float(x) for x in np.asarray(t, np.float32).reshape(3)])
        self.V, self.H = int(sd.n_vertical), int(h_count)
        self.az0, self.naz = 0, self.H
        h = C.c_void_p()
        rc = self.L.ls_tracer_create(C.byref(sd), device, C.byref(h))
        if rc != 0:
            raise LidarShooterHipError(f"ls_tracer_create failed with status {rc} (no HIP device / bad sensor)")
        self.h = h

    @classmethod
    def fromTables(cls, sin_theta, cos_theta, elevation_deg, sin_phi, cos_phi, h_begin_deg, h_step_deg, Rinv, t, device: int = 0):
        """ls_tracer_create_tables: the sensor as ray-direction factor tables (what integration/HipTracer.hpp probes out of a live
        LidarDevice); raises when the library refuses them (elevation_deg / h_begin_deg / h_step_deg off their own tables)"""
        self = cls.__new__(cls)
        self.L = load()
        self._tabs = [np.ascontiguousarray(a, np.float32) for a in (sin_theta, cos_theta, elevation_deg, sin_phi, cos_phi)]
        st = SensorTables()
        st.sin_theta, st.cos_theta, st.elevation_deg, st.sin_phi, st.cos_phi = [_f32p(a) for a in self._tabs]
        st.n_vertical, st.h_count = self._tabs[0].shape[0], self._tabs[3].shape[0]
        st.h_begin_deg, st.h_step_deg = float(h_begin_deg), float(h_step_deg)
        st.Rinv = (C.c_float * 9)(*[float(x) for x in np.asarray(Rinv, np.float32).reshape(9)])
        st.t = (C.c_float * 3)(*[float(x) for x in np.asarray(t, np.float32).reshape(3)])
        self.V, self.H = int(st.n_vertical), int(st.h_count)
        self.az0, self.naz = 0, self.H
        h = C.c_void_p()
        rc = self.L.ls_tracer_create_tables(C.byref(st), device, C.byref(h))
        if rc != 0:
            raise LidarShooterHipError(f"ls_tracer_create_tables failed with status {rc} (no HIP device / tables refused)")
        self.h = h
        return self

    def setSensorTables(self, sin_theta, cos_theta, elevation_deg, sin_phi, cos_phi, h_begin_deg, h_step_deg, Rinv, t):
        """ls_tracer_set_sensor_tables: another sensor, given as factor tables, for this handle (its geometries stay)"""
        tabs = [np.ascontiguousarray(a, np.float32) for a in (sin_theta, cos_theta, elevation_deg, sin_phi, cos_phi)]
        st = SensorTables()
        st.sin_theta, st.cos_theta, st.elevation_deg, st.sin_phi, st.cos_phi = [_f32p(a) for a in tabs]
        st.n_vertical, st.h_count = tabs[0].shape[0], tabs[3].shape[0]
        st.h_begin_deg, st.h_step_deg = float(h_begin_deg), float(h_step_deg)
        st.Rinv = (C.c_float * 9)(*[float(x) for x in np.asarray(Rinv, np.float32).reshape(9)])
        st.t = (C.c_float * 3)(*[float(x) for x in np.asarray(t, np.float32).reshape(3)])
        self._check(self.L.ls_tracer_set_sensor_tables(self.h, C.byref(st)), "ls_tracer_set_sensor_tables")
        self._tabs = tabs
        self.V, self.H = int(st.n_vertical), int(st.h_count)
        self.az0, self.naz = 0, self.H

    def setSensor(self, vertical_deg, h_begin, h_end, h_count, Rinv, t):
        """ITracer::setSensorConfig: another sensor for this handle, its geometries stay (ls_tracer_set_sensor)"""
        self._vert = np.ascontiguousarray(vertical_deg, np.float32)
        sd = SensorDesc()
        sd.vertical_deg = _f32p(self._vert)
        sd.n_vertical = self._vert.shape[0]
        sd.h_begin, sd.h_end, sd.h_count = float(h_begin), float(h_end), int(h_count)
        sd.Rinv = (C.c_float * 9)(*[float(x) for x in np.asarray(Rinv, np.float32).reshape(9)])
        sd.t = (C.c_float * 3)(*[float(x) for x in np.asarray(t, np.float32).reshape(3)])
        self._check(self.L.ls_tracer_set_sensor(self.h, C.byref(sd)), "ls_tracer_set_sensor")
        self.V, self.H = int(sd.n_vertical), int(h_count)
        self.az0, self.naz = 0, self.H

    # ---- lifetime
    def close(self):
        if getattr(self, "h", None):
            self.L.ls_tracer_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc < 0 and rc != -1:
            raise LidarShooterHipError(f"{what}: status {rc}: {self.last_error()}")
        return rc

    def last_error(self) -> str:
        return (self.L.ls_last_error(self.h) or b"").decode()

    # ---- ITracer surface
    def addGeometry(self, name: str, n_vertices: int, n_elements: int, geometry_type: int = 0) -> int:
        return self.L.ls_add_geometry(self.h, name.encode(), geometry_type, int(n_vertices), int(n_elements))

    def removeGeometry(self, name: str) -> int:
        return self.L.ls_remove_geometry(self.h, name.encode())

    def updateGeometry(self, name: str, affine, verts: np.ndarray, tris: np.ndarray | None, stride: int | None = None):
        A = np.ascontiguousarray(affine, np.float32).reshape(12)
        verts = np.ascontiguousarray(verts)
        if stride is None:
            verts = np.ascontiguousarray(verts, np.float32)
            stride = 12
        tp = None
        if tris is not None:   # uint32[n,3] triangles, or [n,4] quads for a geometry added with geometry_type=1
            tris = np.ascontiguousarray(tris, np.uint32)
            tp = tris.ctypes.data
        return self._check(self.L.ls_update_geometry(self.h, name.encode(), _f32p(A), verts.ctypes.data, stride, tp),
                           "ls_update_geometry")

    def updateGeometryComponents(self, name: str, lin, ang, verts: np.ndarray, tris: np.ndarray | None):
        lin = np.ascontiguousarray(lin, np.float32)
        ang = np.ascontiguousarray(ang, np.float32)
        verts = np.ascontiguousarray(verts, np.float32)
        tp = None
        if tris is not None:
            tris = np.ascontiguousarray(tris, np.uint32)
            tp = tris.ctypes.data
        return self._check(self.L.ls_update_geometry_components(self.h, name.encode(), _f32p(lin), _f32p(ang),
                                                                verts.ctypes.data, 12, tp),
                           "ls_update_geometry_components")

    def updateGeometryDevice(self, name: str, affine, d_verts: int, stride: int, d_tris: int | None):
        A = np.ascontiguousarray(affine, np.float32).reshape(12)
        return self._check(self.L.ls_update_geometry_device(self.h, name.encode(), _f32p(A), d_verts, stride, d_tris),
                           "ls_update_geometry_device")

    def updateGeometryDeviceShared(self, name: str, affine, d_verts: int, stride: int, d_tris: int | None):
        A = np.ascontiguousarray(affine, np.float32).reshape(12)
        return self._check(self.L.ls_update_geometry_device_shared(self.h, name.encode(), _f32p(A), d_verts, stride, d_tris),
                           "ls_update_geometry_device_shared")

    def updateGeometryTransform(self, name: str, affine):
        A = np.ascontiguousarray(affine, np.float32).reshape(12)
        return self._check(self.L.ls_update_geometry_transform(self.h, name.encode(), _f32p(A)),
                           "ls_update_geometry_transform")

    def commitScene(self) -> int:
        return self._check(self.L.ls_commit_scene(self.h), "ls_commit_scene")

    def traceScene(self, frame_index: int = 0):
        """-> (rc, points uint8[n,32], hits HIT_DTYPE[n]) copied out of the handle's pinned buffers."""
        fr = Frame()
        rc = self._check(self.L.ls_trace_scene(self.h, frame_index, C.byref(fr)), "ls_trace_scene")
        n = int(fr.n_points)
        if n and fr.compact16:   # LS_OPT_HOST_OUTPUT = 2
            pts = np.zeros((n, 32), np.uint8)
            self._check(self.L.ls_expand_points(pts.ctypes.data, fr.compact16, n), "ls_expand_points")
            hits = (np.frombuffer(C.string_at(fr.hits, n * 16), dtype=HIT_DTYPE).copy() if fr.hits else np.zeros(0, HIT_DTYPE))
        elif n:
            pts = np.ctypeslib.as_array(fr.points32, shape=(n * 32,)).reshape(n, 32).copy()
            hits = (np.frombuffer(C.string_at(fr.hits, n * 16), dtype=HIT_DTYPE).copy() if fr.hits
                    else np.zeros(0, HIT_DTYPE))   # LS_OPT_READBACK_HITS = 0: the records stay on the device
        else:
            pts = np.zeros((0, 32), np.uint8)
            hits = np.zeros(0, HIT_DTYPE)
        self.last_frame = fr
        return rc, pts, hits

    def traceSceneTwoStep(self, frame_index: int = 0):
        """ls_trace_scene_begin + ls_trace_scene_expand into a numpy buffer: -> (rc, points uint8[n,32])"""
        n = C.c_uint32()
        rc = self.L.ls_trace_scene_begin(self.h, frame_index, C.byref(n))
        if rc < -1:
            self._check(rc, "ls_trace_scene_begin")
        pts = np.empty((int(n.value), 32), np.uint8)
        if rc >= 0:
            self._check(self.L.ls_trace_scene_expand(self.h, pts.ctypes.data), "ls_trace_scene_expand")
        return rc, pts

    def traceSceneAsync(self, frame_index: int = 0) -> Frame:
        fr = Frame()
        self._check(self.L.ls_trace_scene_async(self.h, frame_index, C.byref(fr)), "ls_trace_scene_async")
        return fr

    def getGeometryCount(self) -> int:
        return int(self.L.ls_geometry_count(self.h))

    def getGeometryId(self, name: str) -> int:
        return self.L.ls_geometry_id(self.h, name.encode())

    def getGeometryType(self, name: str) -> int:
        """EmbreeTracer::getGeometryType (EmbreeTracer.cpp:103-113): LS_GEOMETRY_TYPE_* or a negative status."""
        return self.L.ls_geometry_type(self.h, name.encode())

    def getVertexCount(self, name: str) -> int:
        return int(self.L.ls_vertex_count(self.h, name.encode()))

    def getElementCount(self, name: str) -> int:
        return int(self.L.ls_element_count(self.h, name.encode()))

    def getTotalRays(self) -> int:
        return int(self.L.ls_total_rays(self.h))

    # ---- extensions
    def setShard(self, first_az: int, n_az: int):
        rc = self._check(self.L.ls_tracer_set_shard(self.h, first_az, n_az), "ls_tracer_set_shard")
        self.az0, self.naz = first_az, n_az
        return rc

    def setStream(self, stream_ptr: int | None):
        return self._check(self.L.ls_tracer_set_stream(self.h, stream_ptr), "ls_tracer_set_stream")

    def synchronize(self):
        return self._check(self.L.ls_tracer_synchronize(self.h), "ls_tracer_synchronize")

    def setOutputBuffers(self, d_points: int | None, d_hits: int | None, d_n: int | None, capacity: int):
        return self._check(self.L.ls_tracer_set_output_buffers(self.h, d_points, d_hits, d_n, capacity),
                           "ls_tracer_set_output_buffers")

    def setHitBuffers(self, d_hits: int | None, d_n: int | None, capacity: int):
        return self._check(self.L.ls_tracer_set_hit_buffers(self.h, d_hits, d_n, capacity), "ls_tracer_set_hit_buffers")

    def expandGatheredHits(self, d_gathered: int, world: int, capacity: int, d_points: int, d_hits: int, d_n: int):
        return self._check(self.L.ls_expand_gathered_hits(self.h, d_gathered, world, capacity, d_points, d_hits, d_n),
                           "ls_expand_gathered_hits")

    def expandGatheredHitsOn(self, stream_ptr: int, d_gathered: int, world: int, capacity: int, d_points: int, d_hits: int, d_n: int):
        self._check(self.L.ls_expand_gathered_hits_on(self.h, stream_ptr, d_gathered, world, capacity, d_points, d_hits, d_n),
                    "ls_expand_gathered_hits_on")

    def cloudToWorld(self, R, d_points_in: int, d_n_points: int, d_points_out: int, out_capacity: int, affine=None,
                     d_out_base: int | None = None, d_out_total: int | None = None):
        """ls_cloud_to_world: sensor-frame points (device) -> world frame, appended at *d_out_base."""
        R = np.ascontiguousarray(R, np.float32).reshape(9)
        A = None if affine is None else np.ascontiguousarray(affine, np.float32).reshape(12)
        return self._check(self.L.ls_cloud_to_world(self.h, None if A is None else _f32p(A), _f32p(R), d_points_in, d_n_points,
                                                    d_points_out, d_out_base, d_out_total, out_capacity),
                           "ls_cloud_to_world")

    def info(self, what: int) -> int:
        return int(self._check(self.L.ls_get_info(self.h, what), "ls_get_info"))

    def orderAfterLastFrame(self, stream_ptr: int):
        self._check(self.L.ls_tracer_order_after_last_frame(self.h, stream_ptr), "ls_tracer_order_after_last_frame")

    def waitEvent(self, event_ptr: int):
        self._check(self.L.ls_tracer_wait_event(self.h, event_ptr), "ls_tracer_wait_event")

    def nextFrameWaits(self, event_ptr: int):
        """only the frame issued next starts after the hipEvent_t (one wait on that frame's stream)"""
        self._check(self.L.ls_tracer_next_frame_waits(self.h, event_ptr), "ls_tracer_next_frame_waits")

    def flush(self):
        return self._check(self.L.ls_tracer_flush(self.h), "ls_tracer_flush")

    def setOption(self, option: int, value: int):
        return self._check(self.L.ls_tracer_set_option(self.h, option, value), "ls_tracer_set_option")

    def timings(self) -> dict:
        """Mean stage durations in ms over the frames recorded since the last call (+ 'frames')."""
        ms = np.zeros(len(STAGES), np.float32)
        n = self._check(self.L.ls_get_timings(self.h, _f32p(ms)), "ls_get_timings")
        d = dict(zip(STAGES, [float(x) for x in ms]))
        d["frames"] = int(n)
        return d

    def visitCounts(self):
        """-> (node fetches, triangle tests) of the last counted trace."""
        return self.visitStats()[:2]

    def visitStats(self):
        """-> (node fetches, triangle tests, sum of per-wave loop trips, max loop trips)."""
        c = (C.c_uint64 * 4)()
        self._check(self.L.ls_get_visit_counts(self.h, c), "ls_get_visit_counts")
        return int(c[0]), int(c[1]), int(c[2]), int(c[3])

    def generateRays(self, d_dx: int, d_dy: int, d_dz: int):
        return self._check(self.L.ls_generate_rays(self.h, d_dx, d_dy, d_dz), "ls_generate_rays")

    # ---- test hooks
    def generateRaysAos(self, d_rays: int | None, d_hits: int | None):
        """LidarDevice::allRaysGPU's two buffers (Ray 32 B, Hit 24 B per ray) in device memory of the caller."""
        self._check(self.L.ls_generate_rays_aos(self.h, d_rays, d_hits), "ls_generate_rays_aos")

    def denseHits(self):
        n = self.getTotalRays()
        t = np.zeros(n, np.float32)
        gid = np.zeros(n, np.uint32)
        self._check(self.L.ls_debug_dense_hits(self.h, _f32p(t), gid.ctypes.data_as(C.POINTER(C.c_uint32))),
                    "ls_debug_dense_hits")
        return t, gid

    def bruteForce(self):
        n = self.getTotalRays()
        t = np.zeros(n, np.float32)
        gid = np.zeros(n, np.uint32)
        self._check(self.L.ls_debug_trace_bruteforce(self.h, _f32p(t), gid.ctypes.data_as(C.POINTER(C.c_uint32))),
                    "ls_debug_trace_bruteforce")
        return t, gid

    def sortPairs(self, keys: np.ndarray, vals: np.ndarray):
        """The build path's radix sort on its own: -> (sorted keys, values); stable, 30-bit keys."""
        k, v = np.ascontiguousarray(keys, np.uint32).copy(), np.ascontiguousarray(vals, np.uint32).copy()
        self._check(self.L.ls_debug_sort_pairs(self.h, k.ctypes.data, v.ctypes.data, k.shape[0]), "ls_debug_sort_pairs")
        return k, v

    def sceneSize(self):
        v = [C.c_uint32() for _ in range(4)]
        self._check(self.L.ls_debug_scene_size(self.h, *[C.byref(x) for x in v]), "ls_debug_scene_size")
        return dict(n_verts=v[0].value, n_tris=v[1].value, n_slots=v[2].value, leaf_size=v[3].value)

    def downloadScene(self):
        s = self.sceneSize()
        verts = np.zeros((s["n_verts"], 3), np.float32)
        tris = np.zeros((s["n_tris"], 3), np.uint32)
        self._check(self.L.ls_debug_download_scene(self.h, verts.ctypes.data, tris.ctypes.data), "ls_debug_download_scene")
        return verts, tris

    def downloadBvh(self):
        s = self.sceneSize()
        nodes = np.zeros(s["n_slots"], NODE_DTYPE)
        tri = np.zeros(s["n_tris"], TRI_DTYPE)
        self._check(self.L.ls_debug_download_bvh(self.h, nodes.ctypes.data, tri.ctypes.data), "ls_debug_download_bvh")
        return nodes, tri, s["leaf_size"]
