"""ctypes binding of the C++ host mirror (liblidarshooter_host.so: LidarDevice, PolygonMesh/STL
ingest, HipTracer).  Lets tests and bench.py use the host classes the way a C++ user of the
reference would use LidarDevice / EmbreeTracer.  No algorithm lives here."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import capi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liblidarshooter_host.so")
_lib = None


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    capi.load()  # the HIP library first (and torch's HIP runtime before it, see capi.load)
    if not os.path.exists(LIB_PATH):
        raise capi.LidarShooterHipError(f"{LIB_PATH} is missing: run __graft_entry__.build()")
    L = C.CDLL(LIB_PATH)
    vp, u32p, f32p = C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_float)
    L.lsh_last_error.restype = C.c_char_p
    L.lsh_device_create.argtypes = [C.c_char_p, C.c_char_p]
    L.lsh_device_create.restype = vp
    L.lsh_device_destroy.argtypes = [vp]
    for f in ("lsh_device_total_rays", "lsh_device_total_channels", "lsh_device_scan_ray_count"):
        getattr(L, f).argtypes = [vp]
        getattr(L, f).restype = C.c_uint32
    L.lsh_device_uid.argtypes = [vp]
    L.lsh_device_uid.restype = C.c_char_p
    L.lsh_device_desc.argtypes = [vp, C.POINTER(capi.SensorDesc)]
    L.lsh_device_rotation.argtypes = [vp, f32p, f32p]
    L.lsh_device_step.argtypes = [vp]
    L.lsh_device_step.restype = C.c_float
    L.lsh_device_ray_direction.argtypes = [vp, C.c_uint32, C.c_uint32, f32p]
    L.lsh_device_origin_to_sensor.argtypes = [vp, f32p, C.c_int]
    L.lsh_device_init_message.argtypes = [vp, C.c_int, u32p, u32p, C.c_char_p, C.c_uint32]
    L.lsh_mesh_load_stl.argtypes = [C.c_char_p]
    L.lsh_mesh_load_stl.restype = vp
    L.lsh_mesh_destroy.argtypes = [vp]
    for f in ("lsh_mesh_num_points", "lsh_mesh_num_polygons", "lsh_mesh_point_step"):
        getattr(L, f).argtypes = [vp]
        getattr(L, f).restype = C.c_uint32
    L.lsh_mesh_point_data.argtypes = [vp]
    L.lsh_mesh_point_data.restype = vp
    L.lsh_mesh_copy_polygons.argtypes = [vp, u32p]
    L.lsh_tracer_create.argtypes = [vp, C.c_int]
    L.lsh_tracer_create.restype = vp
    L.lsh_tracer_destroy.argtypes = [vp]
    L.lsh_tracer_add_geometry.argtypes = [vp, C.c_char_p, C.c_int, C.c_int, C.c_int]
    L.lsh_tracer_remove_geometry.argtypes = [vp, C.c_char_p]
    L.lsh_tracer_update_geometry.argtypes = [vp, C.c_char_p, f32p, vp]
    L.lsh_tracer_update_geometry_components.argtypes = [vp, C.c_char_p, f32p, f32p, vp]
    L.lsh_tracer_commit_scene.argtypes = [vp]
    L.lsh_tracer_trace_scene.argtypes = [vp, C.c_uint32]
    L.lsh_tracer_geometry_count.argtypes = [vp]
    L.lsh_tracer_geometry_count.restype = C.c_long
    L.lsh_tracer_vertex_count.argtypes = [vp, C.c_char_p]
    L.lsh_tracer_vertex_count.restype = C.c_long
    L.lsh_tracer_element_count.argtypes = [vp, C.c_char_p]
    L.lsh_tracer_element_count.restype = C.c_long
    L.lsh_tracer_geometry_id.argtypes = [vp, C.c_char_p]
    L.lsh_tracer_geometry_type.argtypes = [vp, C.c_char_p]
    L.lsh_tracer_cloud.argtypes = [vp, u32p]
    L.lsh_tracer_cloud.restype = vp
    L.lsh_tracer_hits.argtypes = [vp, u32p]
    L.lsh_tracer_hits.restype = vp
    L.lsh_tracer_handle.argtypes = [vp]
    L.lsh_tracer_handle.restype = vp
    L.lsh_trajectory_play.argtypes = [C.c_char_p, C.c_float, f32p, C.c_int]
    _lib = L
    return L


def _f32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class LidarDevice:
    """lidarshooter::LidarDevice (host/LidarDevice.hpp)."""

    def __init__(self, config_path: str, sensor_uid: str = ""):
        self.L = load()
        self.h = self.L.lsh_device_create(config_path.encode(), sensor_uid.encode())
        if not self.h:
            raise capi.LidarShooterHipError(self.L.lsh_last_error().decode())

    def close(self):
        if getattr(self, "h", None):
            self.L.lsh_device_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def getTotalRays(self):
        return int(self.L.lsh_device_total_rays(self.h))

    def getTotalChannels(self):
        return int(self.L.lsh_device_total_channels(self.h))

    def getScanRayCount(self):
        return int(self.L.lsh_device_scan_ray_count(self.h))

    def getSensorUid(self):
        return self.L.lsh_device_uid(self.h).decode()

    def step(self):
        return np.float32(self.L.lsh_device_step(self.h))

    def rotation(self):
        R, Ri = np.zeros(9, np.float32), np.zeros(9, np.float32)
        self.L.lsh_device_rotation(self.h, _f32p(R), _f32p(Ri))
        return R, Ri

    def desc(self):
        """-> dict(vertical, h_begin, h_end, h_count, Rinv, t) as held by ls_sensor_desc."""
        d = capi.SensorDesc()
        self.L.lsh_device_desc(self.h, C.byref(d))
        vert = np.ctypeslib.as_array(d.vertical_deg, shape=(d.n_vertical,)).copy()
        return dict(vertical=vert, h_begin=np.float32(d.h_begin), h_end=np.float32(d.h_end), h_count=int(d.h_count),
                    Rinv=np.array(list(d.Rinv), np.float32), t=np.array(list(d.t), np.float32))

    def rayDirection(self, v: int, h: int):
        d = np.zeros(3, np.float32)
        self.L.lsh_device_ray_direction(self.h, v, h, _f32p(d))
        return d

    def originToSensor(self, p, inverse=False):
        q = np.array(p, np.float32)
        self.L.lsh_device_origin_to_sensor(self.h, _f32p(q), 1 if inverse else 0)
        return q

    def initMessage(self, frame: int):
        out = (C.c_uint32 * 6)()
        flags = C.c_uint32()
        fid = C.create_string_buffer(64)
        self.L.lsh_device_init_message(self.h, frame, out, C.byref(flags), fid, 64)
        return dict(seq=out[0], height=out[1], width=out[2], point_step=out[3], row_step=out[4], n_fields=out[5],
                    is_bigendian=bool(flags.value & 1), is_dense=bool(flags.value & 2), frame_id=fid.value.decode())


class PolygonMesh:
    """pcl::PolygonMesh stand-in loaded by loadPolygonFileSTL (host/HostTypes.hpp)."""

    def __init__(self, stl_path: str):
        self.L = load()
        self.h = self.L.lsh_mesh_load_stl(stl_path.encode())
        if not self.h:
            raise capi.LidarShooterHipError(self.L.lsh_last_error().decode())

    def close(self):
        if getattr(self, "h", None):
            self.L.lsh_mesh_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def numPoints(self):
        return int(self.L.lsh_mesh_num_points(self.h))

    def numPolygons(self):
        return int(self.L.lsh_mesh_num_polygons(self.h))

    def pointStep(self):
        return int(self.L.lsh_mesh_point_step(self.h))

    def points(self) -> np.ndarray:
        n, st = self.numPoints(), self.pointStep()
        raw = np.frombuffer(C.string_at(self.L.lsh_mesh_point_data(self.h), n * st), np.uint8).reshape(n, st)
        return raw[:, :12].copy().view(np.float32).reshape(n, 3)

    def polygons(self) -> np.ndarray:
        out = np.zeros((self.numPolygons(), 3), np.uint32)
        self.L.lsh_mesh_copy_polygons(self.h, out.ctypes.data_as(C.POINTER(C.c_uint32)))
        return out


class HipTracer:
    """lidarshooter::HipTracer (host/HipTracer.hpp): ITracer's surface."""

    def __init__(self, device: LidarDevice, hip_device: int = 0):
        self.L = load()
        self.dev = device
        self.h = self.L.lsh_tracer_create(device.h, hip_device)
        if not self.h:
            raise capi.LidarShooterHipError(self.L.lsh_last_error().decode())

    def close(self):
        if getattr(self, "h", None):
            self.L.lsh_tracer_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def addGeometry(self, name, n_vertices, n_elements, geometry_type=0):
        return self.L.lsh_tracer_add_geometry(self.h, name.encode(), geometry_type, n_vertices, n_elements)

    def removeGeometry(self, name):
        return self.L.lsh_tracer_remove_geometry(self.h, name.encode())

    def updateGeometry(self, name, affine, mesh: PolygonMesh):
        A = np.ascontiguousarray(affine, np.float32).reshape(12)
        return self.L.lsh_tracer_update_geometry(self.h, name.encode(), _f32p(A), mesh.h)

    def updateGeometryComponents(self, name, lin, ang, mesh: PolygonMesh):
        lin = np.ascontiguousarray(lin, np.float32)
        ang = np.ascontiguousarray(ang, np.float32)
        return self.L.lsh_tracer_update_geometry_components(self.h, name.encode(), _f32p(lin), _f32p(ang), mesh.h)

    def commitScene(self):
        return self.L.lsh_tracer_commit_scene(self.h)

    def traceScene(self, frame: int):
        return self.L.lsh_tracer_trace_scene(self.h, frame)

    def getGeometryCount(self):
        return int(self.L.lsh_tracer_geometry_count(self.h))

    def getVertexCount(self, name):
        return int(self.L.lsh_tracer_vertex_count(self.h, name.encode()))

    def getElementCount(self, name):
        return int(self.L.lsh_tracer_element_count(self.h, name.encode()))

    def getGeometryId(self, name):
        return self.L.lsh_tracer_geometry_id(self.h, name.encode())

    def getGeometryType(self, name):
        """RTCGeometryType value, or -1000 - the TraceException's code for an unknown name (8)"""
        return self.L.lsh_tracer_geometry_type(self.h, name.encode())

    def getTraceCloud(self):
        out = (C.c_uint32 * 4)()
        p = self.L.lsh_tracer_cloud(self.h, out)
        w, hgt, ps, seq = out[0], out[1], out[2], out[3]
        n = w * hgt
        data = np.frombuffer(C.string_at(p, n * 32), np.uint8).reshape(n, 32).copy() if n else np.zeros((0, 32), np.uint8)
        return dict(width=w, height=hgt, point_step=ps, seq=seq, data=data)

    def getHits(self):
        n = C.c_uint32()
        p = self.L.lsh_tracer_hits(self.h, C.byref(n))
        if not n.value:
            return np.zeros(0, capi.HIT_DTYPE)
        return np.frombuffer(C.string_at(p, n.value * 16), capi.HIT_DTYPE).copy()


def trajectory_play(path: str, period: float = 0.1) -> np.ndarray:
    """lidarshooter::Trajectory::play (host/Trajectory.hpp): -> float32[n, 6] = (linear xyz, angular xyz)
    after each twist message, per the reference's AffineMesh::joystickCallback rule."""
    L = load()
    n = L.lsh_trajectory_play(path.encode(), period, None, 0)
    if n < 0:
        raise capi.LidarShooterHipError(L.lsh_last_error().decode())
    out = np.zeros((n, 6), np.float32)
    L.lsh_trajectory_play(path.encode(), period, _f32p(out), n)
    return out
