"""Regenerates tests/golden/xt32_golden.npz from the CPU oracle (oracle/).

The reference itself cannot be built or run in this image (needs ROS1, PCL/VTK, Eigen, Embree
3.13.4, ... see DESIGN.md), so these vectors are NOT reference outputs: they are the oracle's
per-ray results for the shipped XT-32 configs, pinned to the reference by the hit counts its own
gtests hold (1668 / 1781; EmbreeTracer_test.cpp:122-135, OptixTracer_test.cpp:122-169).
Run:  python tests/golden/make_golden.py
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as O  # noqa: E402

DATA = os.path.join(HERE, "data")


def main():
    g = O.load_stl(os.path.join(DATA, "mesh", "ground.stl"))
    b = O.load_stl(os.path.join(DATA, "mesh", "ben.stl"))
    out = {}
    for uid in ("0000", "0001"):
        s = O.load_sensor(os.path.join(DATA, "config", f"hesai-pandar-XT-32-lidar_{uid}.json"))
        for scene, meshes in (("ground", [(0, g[0], g[1], O.IDENTITY_AFFINE)]),
                              ("ground_ben", [(0, g[0], g[1], O.IDENTITY_AFFINE), (1, b[0], b[1], O.IDENTITY_AFFINE)])):
            r = O.trace_frame(s, meshes)
            k = f"lidar_{uid}_{scene}"
            out[k + "_t"] = r["t"]
            out[k + "_gid"] = r["gid"]
            out[k + "_points_sha256"] = np.frombuffer(hashlib.sha256(r["points"].tobytes()).digest(), np.uint8)
            out[k + "_hits_sha256"] = np.frombuffer(hashlib.sha256(r["hits"].tobytes()).digest(), np.uint8)
            print(k, len(r["points"]))
        out[f"lidar_{uid}_dirs"] = O.ray_dirs(s)
        out[f"lidar_{uid}_Rinv"] = s.Rinv
    # a moved + rotated ben (updateGeometry(name, translation, rotation, mesh), EmbreeTracer.cpp:276-288)
    s = O.load_sensor(os.path.join(DATA, "config", "hesai-pandar-XT-32-lidar_0000.json"))
    A = O.affine_from_components([1.5, -2.0, 0.25], [0.1, -0.2, 0.7])
    r = O.trace_frame(s, [(0, g[0], g[1], O.IDENTITY_AFFINE), (1, b[0], b[1], A)])
    out["lidar_0000_ground_benmoved_t"] = r["t"]
    out["lidar_0000_ground_benmoved_gid"] = r["gid"]
    out["benmoved_affine"] = A
    print("moved ben", len(r["points"]))
    np.savez_compressed(os.path.join(HERE, "xt32_golden.npz"), **out)


if __name__ == "__main__":
    main()
