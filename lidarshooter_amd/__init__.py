"""lidarshooter_amd -- MI355X (gfx950) LiDAR ray-casting backend behind lidarshooter's ITracer surface.

The product is the C-ABI shared library `liblidarshooter_hip.so` (sources in csrc/, interface in
include/lidarshooter_hip.h) plus the C++ host mirror in host/.  The Python modules here are only
the ctypes binding (`capi`) and synthetic workload generators (`synth`) used by tests and bench.
"""
from . import capi, hostapi, shards, synth  # noqa: F401

__all__ = ["capi", "hostapi", "shards", "synth"]
