#!/usr/bin/env python3
"""Write one of lidarshooter_amd/synth.py's meshes as a raw dump for `lsbench --mesh-raw name=path`, so that the C++ harness
traces BASELINE.md's SYN-1M / SYN-10M bit for bit (its cloud then hashes to bench.py's):
    "LSMESH1\\0" | uint32 n_verts | uint32 n_tris | float32 xyz[n_verts] | uint32 idx[3 n_tris]
usage: dump_mesh.py syn1m|syn10m|grid:<cx>x<cy> out.lsmesh"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from lidarshooter_amd import synth  # noqa: E402


def write(path, v, t):
    v = np.ascontiguousarray(v, np.float32)
    t = np.ascontiguousarray(t, np.uint32)
    with open(path, "wb") as f:
        f.write(b"LSMESH1\0")
        f.write(np.array([v.shape[0], t.shape[0]], np.uint32).tobytes())
        f.write(v.tobytes())
        f.write(t.tobytes())


if __name__ == "__main__":
    which, out = sys.argv[1], sys.argv[2]
    if which == "syn1m":
        v, t = synth.syn_1m()
    elif which == "syn10m":
        v, t = synth.syn_10m()
    else:
        cx, cy = which.split(":")[1].split("x")
        v, t = synth.grid_mesh(int(cx), int(cy))
    write(out, v, t)
    print(f"{out}: {v.shape[0]} vertices, {t.shape[0]} triangles")
