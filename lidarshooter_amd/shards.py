"""Multi-GPU plumbing of the tracer path: azimuth-sector shards and the one
collective per frame.  No algorithm here -- partitioning arithmetic and torch.distributed calls.

Every rank traces the azimuth columns [first, first+n) of every channel over its own replica of the
scene and leaves its result in a fixed-capacity *slot* in device memory:

    [ n_points u32 | pad to 64 B | 16-byte ls_hit records x cap ]          (the slot that travels)
    [ 32-byte points x cap ]                                                (stays local)

One `all_gather_into_tensor` of the slots per frame is the all-gatherv of hit records: the count
word travels in the slot header, so no second collective is needed; the 32-byte points are a
function of (ray, t) and are rebuilt on the receiving side (`ls_expand_gathered_hits`), which cuts the
payload to a third.  On the fully connected xGMI node each rank's slot goes straight to its 7 peers.
Slots are double-buffered and the collective is asynchronous, so the gather of frame i overlaps the
tracing of frame i+1.
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
    """Bytes of the travelling slot: header + hit records."""
    return HEADER + 16 * cap


def write_slot(slot: np.ndarray, cap: int, hits: np.ndarray):
    """Fill a host-side slot (uint8[slot_bytes]) -- what the GPU writes through ls_tracer_set_output_buffers."""
    n = hits.shape[0]
    assert n <= cap
    slot[:4] = np.frombuffer(np.uint32(n).tobytes(), np.uint8)
    slot[HEADER:HEADER + 16 * n] = hits.reshape(-1).view(np.uint8)


def decode_gathered(gathered: np.ndarray, world: int, cap: int) -> np.ndarray:
    """gathered: uint8[world * slot_bytes] -> hit records uint8[n,16] in rank order (= ascending azimuth
    sector; ray indices inside the records are global).  CPU twin of ls_expand_gathered_hits."""
    sb = slot_bytes(cap)
    hts = []
    for r in range(world):
        s = gathered[r * sb:(r + 1) * sb]
        n = int(np.frombuffer(s[:4].tobytes(), np.uint32)[0])
        hts.append(s[HEADER:HEADER + 16 * n].reshape(n, 16))
    return np.concatenate(hts)


def all_gather_slots(slot, gathered, async_op: bool = False):
    """The frame's one collective (RCCL on GPUs, gloo in the CPU tests)."""
    import torch.distributed as dist
    return dist.all_gather_into_tensor(gathered, slot, async_op=async_op)
