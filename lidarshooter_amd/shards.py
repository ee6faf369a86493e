"""Multi-GPU plumbing of the tracer path: azimuth-sector shards and the one
collective per frame.  No algorithm here -- partitioning arithmetic and torch.distributed calls.

Every rank traces the azimuth columns [first, first+n) of every channel over its own replica of the
scene and leaves its result in a fixed-capacity *slot* in device memory:

    [ n_points u32 | pad to 64 B | 32-byte points x cap | 16-byte ls_hit records x cap ]

One `all_gather_into_tensor` of the slots per frame is the all-gatherv of hit records: the count
word travels in the slot header, so no second collective is needed.  On the fully connected xGMI
node each rank's slot goes straight to its 7 peers.
"""
from __future__ import annotations

import numpy as np

HEADER = 64


def shard_columns(H: int, world: int, rank: int):
    """Contiguous azimuth sector of `rank` out of `world`: -> (first_az, n_az)."""
    base, rem = divmod(H, world)
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


def slot_capacity(V: int, H: int, world: int) -> int:
    """Records per slot: the largest shard's ray count."""
    return V * max(shard_columns(H, world, r)[1] for r in range(world))


def slot_bytes(cap: int) -> int:
    return HEADER + 48 * cap


def slot_offsets(cap: int):
    """-> (offset of n_points, offset of points, offset of hits) inside a slot."""
    return 0, HEADER, HEADER + 32 * cap


def write_slot(slot: np.ndarray, cap: int, points: np.ndarray, hits: np.ndarray):
    """Fill a host-side slot (uint8[slot_bytes]) -- what the GPU writes through ls_tracer_set_output_buffers."""
    n = points.shape[0]
    assert n <= cap and hits.shape[0] == n
    _, po, ho = slot_offsets(cap)
    slot[:4] = np.frombuffer(np.uint32(n).tobytes(), np.uint8)
    slot[po:po + 32 * n] = points.reshape(-1).view(np.uint8)
    slot[ho:ho + 16 * n] = hits.reshape(-1).view(np.uint8)


def decode_gathered(gathered: np.ndarray, world: int, cap: int):
    """gathered: uint8[world * slot_bytes] -> (points uint8[n,32], hits uint8[n,16]) in rank order
    (= ascending azimuth sector; ray indices inside the hit records are global)."""
    sb = slot_bytes(cap)
    _, po, ho = slot_offsets(cap)
    pts, hts = [], []
    for r in range(world):
        s = gathered[r * sb:(r + 1) * sb]
        n = int(np.frombuffer(s[:4].tobytes(), np.uint32)[0])
        pts.append(s[po:po + 32 * n].reshape(n, 32))
        hts.append(s[ho:ho + 16 * n].reshape(n, 16))
    return np.concatenate(pts), np.concatenate(hts)


def all_gather_slots(slot, gathered):
    """The frame's one collective (RCCL on GPUs, gloo in the CPU tests)."""
    import torch.distributed as dist
    dist.all_gather_into_tensor(gathered, slot)
