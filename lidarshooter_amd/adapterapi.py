"""ctypes binding of libls_adapter_test.so: the ROS-typed adapter integration/HipTracer.hpp compiled against
integration/stubs, driven the way lidarshooter's MeshProjector drives an ITracer (integration/test/adapter_capi.cpp).
Used by tests/test_adapter.py and bench.py's drop-in legs only; holds no algorithm."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import capi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libls_adapter_test.so")
_lib = None


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    capi.load()                      # torch's HIP runtime first (see capi.load)
    if not os.path.exists(LIB_PATH):
        raise capi.LidarShooterHipError(f"{LIB_PATH} is missing: make -C integration")
    L = C.CDLL(LIB_PATH)
    vp, f32p, u32p = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_uint)
    L.lsa_last_error.restype = C.c_char_p
    L.lsa_create.argtypes = [C.c_char_p, C.c_int]
    L.lsa_create.restype = vp
    L.lsa_destroy.argtypes = [vp]
    L.lsa_destroy.restype = None
    L.lsa_probe_sensor.argtypes = [C.c_char_p, u32p, f32p, C.c_uint, f32p, f32p, f32p, f32p]
    L.lsa_mesh_from_arrays.argtypes = [vp, C.c_char_p, f32p, C.c_uint, u32p, C.c_uint, C.c_uint]
    L.lsa_mesh_from_arrays_ex.argtypes = [vp, C.c_char_p, f32p, C.c_uint, u32p, C.c_uint, C.c_uint, C.c_uint]
    L.lsa_mesh_from_stl.argtypes = [vp, C.c_char_p, C.c_char_p]
    L.lsa_mesh_vertices.argtypes = [vp, C.c_char_p]
    L.lsa_mesh_vertices.restype = C.c_uint
    L.lsa_mesh_polygons.argtypes = [vp, C.c_char_p]
    L.lsa_mesh_polygons.restype = C.c_uint
    L.lsa_mesh_set_vertices.argtypes = [vp, C.c_char_p, f32p, C.c_uint]
    L.lsa_mesh_set_displacement.argtypes = [vp, C.c_char_p, f32p, f32p]
    L.lsa_mesh_set_displacement.restype = None
    L.lsa_add_geometry.argtypes = [vp, C.c_char_p, C.c_int]
    L.lsa_remove_geometry.argtypes = [vp, C.c_char_p]
    L.lsa_geometry_count.argtypes = [vp]
    L.lsa_geometry_count.restype = C.c_long
    L.lsa_update_components.argtypes = [vp, C.c_char_p]
    L.lsa_update_affine.argtypes = [vp, C.c_char_p, f32p]
    L.lsa_commit.argtypes = [vp]
    L.lsa_trace.argtypes = [vp, C.c_uint]
    L.lsa_frame_loop.argtypes = [vp, C.c_uint]
    L.lsa_frame_loop.restype = C.c_double
    L.lsa_cloud.argtypes = [vp, u32p, C.POINTER(C.c_ulonglong)]
    L.lsa_cloud.restype = vp
    L.lsa_set_skip_unchanged.argtypes = [vp, C.c_int]
    L.lsa_set_skip_unchanged.restype = None
    L.lsa_upload_counts.argtypes = [vp, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]
    L.lsa_upload_counts.restype = None
    L.lsa_handle.argtypes = [vp]
    L.lsa_handle.restype = vp
    L.lsa_getter.argtypes = [vp, C.c_int, C.c_char_p]
    L.lsa_getter.restype = C.c_long
    L.lsa_invalidate_mesh.argtypes = [vp, C.c_char_p]
    L.lsa_invalidate_mesh.restype = None
    L.lsa_set_sensor_config.argtypes = [vp, C.c_char_p]
    L.lsa_reinitialize_sensor.argtypes = [vp, C.c_char_p]
    L.lsa_invalidate_sensor.argtypes = [vp]
    L.lsa_invalidate_sensor.restype = None
    L.lsa_sensor_probe_count.argtypes = [vp]
    L.lsa_sensor_probe_count.restype = C.c_ulonglong
    _lib = L
    return L


def _f32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def probe_sensor(config_path: str) -> dict:
    """HipTracer::SensorProbe on a stub LidarDevice made from `config_path` (pure host code, no GPU)."""
    L = load()
    vh = (C.c_uint * 2)()
    cap = 1 << 20
    tab = np.zeros(cap, np.float32)
    el = np.zeros(1 << 16, np.float32)
    misc, rinv, t = np.zeros(2, np.float32), np.zeros(9, np.float32), np.zeros(3, np.float32)
    if L.lsa_probe_sensor(config_path.encode(), vh, _f32p(tab), cap, _f32p(el), _f32p(misc), _f32p(rinv), _f32p(t)) != 0:
        raise capi.LidarShooterHipError(L.lsa_last_error().decode())
    V, H = int(vh[0]), int(vh[1])
    return dict(V=V, H=H, sin_theta=tab[:V].copy(), cos_theta=tab[V:2 * V].copy(), sin_phi=tab[2 * V:2 * V + H].copy(),
                cos_phi=tab[2 * V + H:2 * V + 2 * H].copy(), elevation=el[:V].copy(), begin=misc[0], step=misc[1], Rinv=rinv, t=t)


class AdapterTracer:
    """One sensor's (stub LidarDevice, lidarshooter::HipTracer) pair plus the meshes MeshProjector would track."""

    def __init__(self, config_path: str, device: int = 0):
        self.L = load()
        self.c = self.L.lsa_create(config_path.encode(), device)
        if not self.c:
            raise capi.LidarShooterHipError(self.L.lsa_last_error().decode())

    def close(self):
        if getattr(self, "c", None):
            self.L.lsa_destroy(self.c)
            self.c = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc == -100:
            raise capi.LidarShooterHipError(f"{what}: {self.L.lsa_last_error().decode()}")
        return rc

    def meshFromSTL(self, name: str, path: str):
        return self._check(self.L.lsa_mesh_from_stl(self.c, name.encode(), path.encode()), "lsa_mesh_from_stl")

    def meshFromArrays(self, name: str, verts, tris, point_step: int = 16):
        v = np.ascontiguousarray(verts, np.float32)
        t = np.ascontiguousarray(tris, np.uint32)   # [n,3] triangles or [n,4] quads
        return self._check(self.L.lsa_mesh_from_arrays_ex(self.c, name.encode(), _f32p(v), v.shape[0],
                                                          t.ctypes.data_as(C.POINTER(C.c_uint)), t.shape[0], point_step, t.shape[1]),
                           "lsa_mesh_from_arrays")

    def setVertices(self, name: str, verts, seq: int = 0):
        v = np.ascontiguousarray(verts, np.float32)
        return self.L.lsa_mesh_set_vertices(self.c, name.encode(), _f32p(v), seq)

    def setDisplacement(self, name: str, linear, angular):
        a = np.ascontiguousarray(linear, np.float32)
        b = np.ascontiguousarray(angular, np.float32)
        self.L.lsa_mesh_set_displacement(self.c, name.encode(), _f32p(a), _f32p(b))

    def addGeometry(self, name: str, geometry_type: int = 0) -> int:
        return self._check(self.L.lsa_add_geometry(self.c, name.encode(), geometry_type), "addGeometry")

    def removeGeometry(self, name: str) -> int:
        return self._check(self.L.lsa_remove_geometry(self.c, name.encode()), "removeGeometry")

    def getGeometryCount(self) -> int:
        return int(self.L.lsa_geometry_count(self.c))

    def updateGeometry(self, name: str, affine=None) -> int:
        """affine None: the (translation, rotation) overload with the mesh's displacement; else the Affine3f overload."""
        if affine is None:
            return self._check(self.L.lsa_update_components(self.c, name.encode()), "updateGeometry")
        A = np.ascontiguousarray(affine, np.float32).reshape(12)
        return self._check(self.L.lsa_update_affine(self.c, name.encode(), _f32p(A)), "updateGeometry")

    def commitScene(self) -> int:
        return self._check(self.L.lsa_commit(self.c), "commitScene")

    def traceScene(self, frame: int) -> int:
        return self._check(self.L.lsa_trace(self.c, frame), "traceScene")

    def frameLoop(self, frames: int) -> float:
        """MeshProjector::traceAffineMesh `frames` times in C++; seconds per frame."""
        s = self.L.lsa_frame_loop(self.c, frames)
        if s < 0:
            raise capi.LidarShooterHipError(self.L.lsa_last_error().decode())
        return float(s)

    def cloud(self) -> dict:
        out = (C.c_uint * 8)()
        nbytes = C.c_ulonglong()
        p = self.L.lsa_cloud(self.c, out, C.byref(nbytes))
        n = int(nbytes.value)
        data = np.frombuffer(C.string_at(p, n), np.uint8).copy() if n else np.zeros(0, np.uint8)
        return dict(width=out[0], height=out[1], point_step=out[2], row_step=out[3], seq=out[4], n_fields=out[5],
                    is_bigendian=bool(out[6]), is_dense=bool(out[7]), data=data)

    def setSkipUnchanged(self, on: bool):
        self.L.lsa_set_skip_unchanged(self.c, int(bool(on)))

    def uploadCounts(self):
        a, b = C.c_ulonglong(), C.c_ulonglong()
        self.L.lsa_upload_counts(self.c, C.byref(a), C.byref(b))
        return int(a.value), int(b.value)

    def handle(self) -> int:
        return self.L.lsa_handle(self.c)

    def getter(self, what: str, name: str) -> int:
        """EmbreeTracer's per-name getters through the adapter: 'id', 'type', 'vertices', 'elements'.  A TraceException
        comes back as -1000 - its error code (EmbreeTracer.cpp:369-415: codes 1, 4, 8)."""
        return int(self.L.lsa_getter(self.c, {"id": 0, "type": 1, "vertices": 2, "elements": 3}[what], name.encode()))

    def invalidateMesh(self, name: str):
        self.L.lsa_invalidate_mesh(self.c, name.encode())

    def setSensorConfig(self, sensor_config: str):
        """ITracer::setSensorConfig with a new LidarDevice made from that file (through the base-class pointer)"""
        if self.L.lsa_set_sensor_config(self.c, sensor_config.encode()) != 0:
            raise RuntimeError(self.L.lsa_last_error().decode())

    def reinitializeSensor(self, sensor_config: str):
        """LidarDevice::initialize on the device the tracer already holds"""
        if self.L.lsa_reinitialize_sensor(self.c, sensor_config.encode()) != 0:
            raise RuntimeError(self.L.lsa_last_error().decode())

    def invalidateSensor(self):
        self.L.lsa_invalidate_sensor(self.c)

    def sensorProbeCount(self) -> int:
        return int(self.L.lsa_sensor_probe_count(self.c))
