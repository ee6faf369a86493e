"""The N > 1 path on CPU: world_size-2 (and 3) gloo process groups run the same shard arithmetic,
slot layout and single all-gather that bench.py runs over RCCL; the per-rank results come from the
CPU oracle restricted to the rank's azimuth sector (no GPU here).  The gathered, decoded cloud
must equal the oracle's full-frame cloud.  Both implementations of the slot protocol are driven: the Python one
(lidarshooter_amd/shards.py, what bench.py uses) and the C one (include/lidarshooter_group.h, what lsbench --ranks
and any C++ caller use) -- and a slot written by one must decode with the other."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import DATA, ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir, impl="python"):
    import sys
    sys.path.insert(0, ROOT)
    from lidarshooter_amd import shards
    if impl == "c":                       # the C slot protocol (liblidarshooter_group.so), same call shapes
        from lidarshooter_amd import groupapi
        proto_w = groupapi
        proto_r = shards if rank % 2 else groupapi   # odd ranks decode with the other implementation
    else:
        proto_w = proto_r = shards
    from oracle import oracle as O
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    s = O.load_sensor(os.path.join(DATA, "config", "hesai-pandar-XT-32-lidar_0000.json"))
    ground = O.load_stl(os.path.join(DATA, "mesh", "ground.stl"))
    ben = O.load_stl(os.path.join(DATA, "mesh", "ben.stl"))
    ml = [(0, *ground, O.IDENTITY_AFFINE), (1, *ben, O.IDENTITY_AFFINE)]
    full = O.trace_frame(s, ml)
    # this rank's sector: keep the hits whose azimuth column is in [first, first+n)
    first, n = proto_w.shard_columns(s.H, world, rank)
    assert (first, n) == shards.shard_columns(s.H, world, rank)
    col = full["hits"][:, 0] % s.H
    mine = (col >= first) & (col < first + n)
    cap = proto_w.slot_capacity(s.V, s.H, world)
    assert cap == shards.slot_capacity(s.V, s.H, world) and proto_w.slot_bytes(cap) == shards.slot_bytes(cap)
    slot = np.full(proto_w.slot_bytes(cap), 0xAB, np.uint8)
    slot[:shards.HEADER] = 0
    proto_w.write_slot(slot, cap, full["hits"][mine])
    t_slot = torch.from_numpy(slot)
    gathered = torch.zeros(world * slot.shape[0], dtype=torch.uint8)
    work = shards.all_gather_slots(t_slot, gathered, async_op=True)     # the asynchronous form bench.py uses
    work.wait()
    hts = proto_r.decode_gathered(gathered.numpy(), world, cap)
    hits = np.ascontiguousarray(hts).view(np.uint32).reshape(-1, 4)
    order = np.argsort(hits[:, 0], kind="stable")
    # points are rebuilt on the receiving side from (ray, t): xyz = t * dir(ray)
    t = hits[order][:, 3].view(np.float32)
    xyz = (t[:, None] * full["dirs"][hits[order][:, 0]]).astype(np.float32)
    ok = np.array_equal(hits[order], full["hits"]) and np.array_equal(
        xyz, full["points"].view(np.float32).reshape(-1, 8)[:, :3])
    # every rank must see the same gathered bytes
    h = torch.tensor([int(np.frombuffer(gathered.numpy().tobytes()[:8], np.uint64)[0] % (1 << 62)), int(ok)])
    hs = [torch.zeros_like(h) for _ in range(world)]
    dist.all_gather(hs, h)
    same = all(bool((x == hs[0]).all()) for x in hs)
    with open(os.path.join(out_dir, f"rank{rank}.txt"), "w") as fh:
        fh.write(f"{int(ok)} {int(same)} {len(hits)}\n")
    dist.destroy_process_group()


@pytest.mark.parametrize("world,impl", [(2, "python"), (3, "python"), (2, "c"), (3, "c")])
def test_sharded_gather_equals_full_frame(tmp_path, world, impl):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path), impl), nprocs=world, join=True)
    for r in range(world):
        ok, same, n = open(tmp_path / f"rank{r}.txt").read().split()
        assert ok == "1" and same == "1" and n == "1781"


def test_shard_columns_partition():
    from lidarshooter_amd import shards
    for H in (150, 4096, 7):
        for world in (1, 2, 3, 8):
            cols = []
            for r in range(world):
                f, n = shards.shard_columns(H, world, r)
                cols += list(range(f, f + n))
            assert cols == list(range(H))
    assert shards.slot_capacity(128, 4096, 8) == 128 * 512
    assert shards.slot_bytes(10) == 64 + 160


def test_c_slot_arithmetic_equals_python():
    # include/lidarshooter_group.h against lidarshooter_amd/shards.py, without a GPU
    from lidarshooter_amd import groupapi, shards
    for H in (150, 4096, 7, 2):
        for world in (1, 2, 3, 8):
            for r in range(world):
                assert groupapi.shard_columns(H, world, r) == shards.shard_columns(H, world, r)
            assert groupapi.slot_capacity(32, H, world) == shards.slot_capacity(32, H, world)
    assert groupapi.slot_bytes(10) == shards.slot_bytes(10) and groupapi.SLOT_HEADER == shards.HEADER
    rng = np.random.default_rng(1)
    cap, world = 40, 3
    recs = [rng.integers(0, 2**32, size=(n, 4), dtype=np.uint32) for n in (0, 40, 17)]
    g_c = np.zeros(world * shards.slot_bytes(cap), np.uint8)
    g_p = g_c.copy()
    for r, h in enumerate(recs):
        groupapi.write_slot(g_c[r * shards.slot_bytes(cap):(r + 1) * shards.slot_bytes(cap)], cap, h)
        shards.write_slot(g_p[r * shards.slot_bytes(cap):(r + 1) * shards.slot_bytes(cap)], cap, h)
    assert np.array_equal(g_c, g_p)
    want = np.concatenate(recs)
    assert np.array_equal(groupapi.decode_gathered(g_p, world, cap), want)
    assert np.array_equal(shards.decode_gathered(g_c, world, cap).view(np.uint32).reshape(-1, 4), want)


# ---------------------------------------------------------------------------------------------------------------------
# bench.py --gpus N without a launcher (missing 1a): the script starts the N ranks itself -- as a fresh
# torch.distributed.run child, before anything touches a GPU -- or exits non-zero; it never prints a one-GPU line for --gpus N.
# ---------------------------------------------------------------------------------------------------------------------
def _bench(args, env_extra=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK")):
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=300, env=env)
    lines = [ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")]
    return out.returncode, [json.loads(ln) for ln in lines], out.stderr


def test_bench_gpus_2_without_launcher_spawns_two_ranks():
    rc, lines, err = _bench(["--gpus", "2", "--spawn-check"], {"LS_BENCH_REHEARSAL": "1"})
    assert rc == 0, err[-2000:]
    assert len(lines) == 1, lines                       # ONE JSON line, rank 0's
    rec = lines[0]
    assert rec["n_gpus"] == 2 and rec["ranks_met"] == 2 and rec["backend"] == "gloo" and rec["spawned_by_bench"] is True


def test_bench_never_reports_fewer_gpus_than_asked_for():
    # this container has no GPU: --gpus 8 must fail, not fall back to what is there (round 3 printed "n_gpus": 1)
    rc, lines, err = _bench(["--gpus", "8", "--steps", "2"])
    assert rc != 0 and lines == [] and "refusing" in err
    # a launcher whose world disagrees with --gpus is refused as well
    rc, lines, err = _bench(["--gpus", "4", "--spawn-check"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"}, drop=())
    assert rc != 0 and lines == []


def test_bench_watchdog_ends_a_hung_phase_and_still_reports():
    """bench.py's N > 1 watchdog (the multi-rank RCCL path has never run on hardware): a phase that does not finish in time
    ends the process from a second thread; rank 0 prints the measurements that did finish as ONE line with an "error"
    field and exit code 4 -- a job whose watchdog fired has NOT succeeded, its line is there to be read -- or
    nothing and exit code 3 when there is nothing to report; another rank prints nothing (and leaves with 4 as well)."""
    import json
    import subprocess
    import sys
    prog = ("import sys, time; sys.path.insert(0, %r); import bench; d = bench.Watchdog(1, int(sys.argv[1])); "
            "fb = {'value': 5.0, 'unit': 'Mrays/s', 'n_gpus': 2, 'config': {'workload': 'w', 'parallelism': 'frames interleaved'}} if sys.argv[2] == '1' else None; "
            "d.arm('phase one', 30.0); d.arm('azimuth shards + all-gather', 0.5, fb); time.sleep(60)") % ROOT
    def run(rank, with_fallback):
        return subprocess.run([sys.executable, "-c", prog, str(rank), "1" if with_fallback else "0"], capture_output=True, text=True, timeout=120)
    out = run(0, True)
    assert out.returncode == 4, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["value"] == 5.0 and "azimuth shards + all-gather" in rec["error"] and "frames interleaved" in rec["error"]
    assert rec["summary"]["value"] == 5.0                      # (the line keeps bench.py's front-loaded shape)
    assert "watchdog" in out.stderr
    out = run(0, False)
    assert out.returncode == 3 and out.stdout.strip() == "" and "did not finish within" in out.stderr
    out = run(1, True)
    assert out.returncode == 4 and out.stdout.strip() == ""   # only rank 0 reports
    # a phase that raises on this rank ends the same way, at once
    prog3 = ("import sys, time; sys.path.insert(0, %r); import bench; d = bench.Watchdog(1, 0); "
             "d.arm('azimuth shards + all-gather', 600.0, {'value': 7.0, 'config': {'workload': 'w', 'parallelism': 'frames interleaved'}})\n"
             "try:\n    raise RuntimeError('no communicator')\nexcept Exception as e:\n    d.failed(e)\n") % ROOT
    out = subprocess.run([sys.executable, "-c", prog3], capture_output=True, text=True, timeout=120)
    assert out.returncode == 4, out.stderr[-2000:]
    rec = json.loads([ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")][0])
    assert rec["value"] == 7.0 and "raised on rank 0" in rec["error"] and "no communicator" in rec["error"]
    # a phase that finishes: disarm, nothing happens
    prog2 = ("import sys, time; sys.path.insert(0, %r); import bench; d = bench.Watchdog(1, 0); d.arm('quick', 0.5); d.disarm(); time.sleep(1.5); print('alive')") % ROOT
    out = subprocess.run([sys.executable, "-c", prog2], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip() == "alive"
