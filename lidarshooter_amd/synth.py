"""Synthetic sensor / mesh workloads of BASELINE.md section 4 (inputs only, no algorithm).

SYN-128 sensor: 128 channels from +15 deg to -25 deg (linear), 4096 azimuth columns over
[0, 360] inclusive, pose of lidar_0000 (config/hesai-pandar-XT-32-lidar_0000.json:21-31).
SYN-1M mesh: 1000 x 500-cell grid on [-50,50]^2, 2 triangles per cell = 1 000 000 triangles,
z = 0.25 sin(0.35 x) cos(0.27 y) + U(-0.01, 0.01), numpy default_rng(20240), float32.
"""
from __future__ import annotations

import numpy as np


def syn_vertical(n_channels: int = 128, top: float = 15.0, span: float = 40.0) -> np.ndarray:
    i = np.arange(n_channels, dtype=np.float64)
    return (top - i * (span / (n_channels - 1))).astype(np.float32)


def grid_mesh(cells_x: int = 1000, cells_y: int = 500, half: float = 50.0, seed: int = 20240,
              relief: float = 0.25, noise: float = 0.01):
    """-> (verts float32[(cx+1)*(cy+1), 3], tris uint32[2*cx*cy, 3]); row-major cells, CCW."""
    xs = np.linspace(-half, half, cells_x + 1)
    ys = np.linspace(-half, half, cells_y + 1)
    X, Y = np.meshgrid(xs, ys, indexing="xy")            # shape (cy+1, cx+1): rows along y
    rng = np.random.default_rng(seed)
    Z = relief * np.sin(0.35 * X) * np.cos(0.27 * Y) + rng.uniform(-noise, noise, size=X.shape)
    verts = np.stack([X, Y, Z], axis=-1).reshape(-1, 3).astype(np.float32)
    j, i = np.meshgrid(np.arange(cells_y), np.arange(cells_x), indexing="ij")
    v00 = (j * (cells_x + 1) + i).reshape(-1)
    v10 = v00 + 1
    v01 = v00 + (cells_x + 1)
    v11 = v01 + 1
    tris = np.empty((2 * cells_x * cells_y, 3), np.uint32)
    tris[0::2] = np.stack([v00, v10, v11], axis=-1)
    tris[1::2] = np.stack([v00, v11, v01], axis=-1)
    return verts, tris


def syn_1m():
    return grid_mesh(1000, 500)


def syn_10m():
    return grid_mesh(3162, 1581)


def shard_columns(H: int, world: int, rank: int):
    """Contiguous azimuth sector of `rank` out of `world`: -> (first_az, n_az)."""
    from .shards import shard_columns as _sc
    return _sc(H, world, rank)


def write_sensor_json(template_path: str, out_path: str, vertical, h_begin: float, h_end: float, h_count: int) -> str:
    """A sensor config like `template_path` (a shipped XT-32 JSON) with other channel tables: only the
    `channels` block is rewritten, textually, so pose, message fields and comments stay as shipped."""
    import re
    txt = open(template_path).read()
    vert = ", ".join(repr(float(np.float32(x))) for x in np.asarray(vertical, np.float32))   # float32 -> shortest double repr: round-trips
    txt, n1 = re.subn(r'("vertical"\s*:\s*\[)[^\]]*(\])', lambda m: m.group(1) + vert + m.group(2), txt, count=1)
    txt, n2 = re.subn(r'("begin"\s*:\s*)[-0-9.eE+]+', lambda m: m.group(1) + repr(float(np.float32(h_begin))), txt, count=1)
    txt, n3 = re.subn(r'("end"\s*:\s*)[-0-9.eE+]+', lambda m: m.group(1) + repr(float(np.float32(h_end))), txt, count=1)
    at = txt.index('"horizontal"', txt.index('"channels"'))   # "count" also names a field of every message pointField
    tail, n4 = re.subn(r'("count"\s*:\s*)[0-9]+', lambda m: m.group(1) + str(int(h_count)), txt[at:], count=1)
    txt = txt[:at] + tail
    assert (n1, n2, n3, n4) == (1, 1, 1, 1), "template does not look like a lidarshooter sensor config"
    with open(out_path, "w") as f:
        f.write(txt)
    return out_path
