"""ctypes binding of include/lidarshooter_group.h (liblidarshooter_group.so): the C multi-GPU path.  The slot
arithmetic works without a GPU; ls_group_create needs one (and RCCL).  Tests and tools only."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import capi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liblidarshooter_group.so")
SHARDED, INTERLEAVED = 0, 1
ID_BYTES, SLOT_HEADER = 128, 64
SYMBOLS = ("ls_group_shard_columns", "ls_group_slot_capacity", "ls_group_slot_bytes", "ls_group_write_slot",
           "ls_group_decode_gathered", "ls_group_unique_id", "ls_group_create", "ls_group_destroy", "ls_group_trace",
           "ls_group_owns_frame", "ls_group_cloud", "ls_group_download_cloud", "ls_group_synchronize", "ls_group_last_error",
           "ls_group_create_opts", "ls_group_info", "ls_group_frame_status")
FLAG_ONE_COMMUNICATOR, FLAG_NO_GRAPH, FLAG_SIZED_GATHER, FLAG_DEBUG_PEER_REFUSES = 1, 2, 4, 0x100
INFO_RCCL_VERSION, INFO_COMM_RANKS, INFO_COMM_DEVICE, INFO_COMMUNICATORS, INFO_PER_SET, INFO_FRAME_GRAPH = 1, 2, 3, 4, 5, 6
INFO_GATHER_CAPACITY, INFO_TRUNCATED_FRAMES, INFO_ARRANGEMENT_MINE, INFO_ARRANGEMENT_COMMON, INFO_COLLECTIVES_ARE_A_SHIM = 7, 8, 9, 10, 11
_lib = None


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    capi.load()
    if not os.path.exists(LIB_PATH):
        raise capi.LidarShooterHipError(f"{LIB_PATH} is missing: make -C lidarshooter_amd/csrc")
    L = C.CDLL(LIB_PATH)
    vp, u32, u32p = C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)
    L.ls_group_shard_columns.argtypes = [u32, u32, u32, u32p, u32p]
    L.ls_group_shard_columns.restype = None
    L.ls_group_slot_capacity.argtypes = [u32, u32, u32]
    L.ls_group_slot_capacity.restype = u32
    L.ls_group_slot_bytes.argtypes = [u32]
    L.ls_group_slot_bytes.restype = C.c_uint64
    L.ls_group_write_slot.argtypes = [vp, u32, vp, u32]
    L.ls_group_write_slot.restype = None
    L.ls_group_decode_gathered.argtypes = [vp, u32, u32, vp]
    L.ls_group_decode_gathered.restype = u32
    L.ls_group_unique_id.argtypes = [vp]
    L.ls_group_create.argtypes = [vp, u32, u32, C.c_int, vp, C.POINTER(vp)]
    L.ls_group_create_opts.argtypes = [vp, u32, u32, C.c_int, u32, vp, C.POINTER(vp)]
    L.ls_group_info.argtypes = [vp, C.c_int]
    L.ls_group_info.restype = C.c_long
    L.ls_group_destroy.argtypes = [vp]
    L.ls_group_destroy.restype = None
    L.ls_group_trace.argtypes = [vp, u32]
    L.ls_group_owns_frame.argtypes = [vp, u32]
    L.ls_group_cloud.argtypes = [vp, u32, C.POINTER(capi.Frame)]
    L.ls_group_download_cloud.argtypes = [vp, u32, vp, vp, u32]
    L.ls_group_frame_status.argtypes = [vp, u32]
    L.ls_group_download_cloud.restype = C.c_long
    L.ls_group_synchronize.argtypes = [vp]
    L.ls_group_last_error.argtypes = [vp]
    L.ls_group_last_error.restype = C.c_char_p
    _lib = L
    return L


def shard_columns(H: int, world: int, rank: int):
    a, b = C.c_uint32(), C.c_uint32()
    load().ls_group_shard_columns(H, world, rank, C.byref(a), C.byref(b))
    return int(a.value), int(b.value)


def slot_capacity(V: int, H: int, world: int) -> int:
    return int(load().ls_group_slot_capacity(V, H, world))


def slot_bytes(cap: int) -> int:
    return int(load().ls_group_slot_bytes(cap))


def write_slot(slot: np.ndarray, cap: int, hits: np.ndarray):
    h = np.ascontiguousarray(hits, np.uint32)
    load().ls_group_write_slot(slot.ctypes.data, cap, h.ctypes.data, h.shape[0])


def decode_gathered(gathered: np.ndarray, world: int, cap: int) -> np.ndarray:
    out = np.zeros((world * cap, 4), np.uint32)
    n = load().ls_group_decode_gathered(gathered.ctypes.data, world, cap, out.ctypes.data)
    return out[:n].copy()


class Group:
    def __init__(self, tracer: "capi.Tracer", world: int, rank: int, mode: int, uid: bytes | None = None, flags: int = 0):
        self.L = load()
        if uid is None and mode == SHARDED:
            buf = (C.c_uint8 * ID_BYTES)()
            if self.L.ls_group_unique_id(buf) != 0:
                raise capi.LidarShooterHipError("ls_group_unique_id failed (no RCCL?)")
            uid = bytes(buf)
        self.uid = uid
        g = C.c_void_p()
        idbuf = (C.c_uint8 * ID_BYTES).from_buffer_copy(uid) if uid else None
        rc = self.L.ls_group_create_opts(idbuf, world, rank, mode, flags, tracer.h, C.byref(g))
        if rc != 0:
            raise capi.LidarShooterHipError(f"ls_group_create_opts: status {rc}")
        self.g, self.tracer = g, tracer

    def trace(self, frame: int) -> int:
        rc = self.L.ls_group_trace(self.g, frame)
        if rc < -1:
            raise capi.LidarShooterHipError(self.L.ls_group_last_error(self.g).decode())
        return rc

    def info(self, what: int) -> int:
        return int(self.L.ls_group_info(self.g, what))

    def rccl(self) -> dict:
        """what RCCL itself says about this rank's communicator (bench.py's "rccl" object)"""
        return {"version": self.info(INFO_RCCL_VERSION), "comm_ranks": self.info(INFO_COMM_RANKS), "device": self.info(INFO_COMM_DEVICE),
                "communicators": self.info(INFO_COMMUNICATORS), "per_set_streams": bool(self.info(INFO_PER_SET)),
                "frame_graph_state": self.info(INFO_FRAME_GRAPH), "shim": bool(self.info(INFO_COLLECTIVES_ARE_A_SHIM))}

    def owns(self, frame: int) -> bool:
        return bool(self.L.ls_group_owns_frame(self.g, frame))

    def cloud(self, frame: int) -> "capi.Frame":
        f = capi.Frame()
        if self.L.ls_group_cloud(self.g, frame, C.byref(f)) != 0:
            raise capi.LidarShooterHipError(self.L.ls_group_last_error(self.g).decode())
        return f

    def frameStatus(self, frame: int) -> int:
        """ls_group_frame_status: 0 complete, LS_ERR_OUT_OF_RANGE (-5) truncated / reused, LS_ERR_NOT_COMMITTED while it is still in flight"""
        return int(self.L.ls_group_frame_status(self.g, frame))

    def download(self, frame: int):
        """-> (points uint8[n,32], hits uint32[n,4]) of that frame's whole cloud"""
        cap = self.tracer.V * self.tracer.H
        pts = np.zeros((cap, 32), np.uint8)
        hits = np.zeros((cap, 4), np.uint32)
        n = self.L.ls_group_download_cloud(self.g, frame, pts.ctypes.data, hits.ctypes.data, cap)
        if n < 0:
            raise capi.LidarShooterHipError(self.L.ls_group_last_error(self.g).decode())
        return pts[:n].copy(), hits[:n].copy()

    def synchronize(self):
        if self.L.ls_group_synchronize(self.g) != 0:
            raise capi.LidarShooterHipError(self.L.ls_group_last_error(self.g).decode())

    def close(self):
        if getattr(self, "g", None):
            self.L.ls_group_destroy(self.g)
            self.g = None
