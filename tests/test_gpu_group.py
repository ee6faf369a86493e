"""include/lidarshooter_group.h on one GPU: a one-rank RCCL communicator (ncclAllGather of the rank's own slot, then
ls_expand_gathered_hits on the collective's stream) and the frame-interleaved mode, frames rotating over the group's three
buffer sets while the pose of a mesh changes; every frame's whole cloud equals the oracle's.  (Two ranks on one device
are refused by RCCL, so N > 1 of the C path is covered on the CPU: tests/test_multirank_gloo.py drives its slot
protocol over gloo.)"""
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import DATA, ROOT, make_tracer

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode_name,flags", [("sharded", 0), ("sharded", 1), ("sharded", 2), ("interleaved", 0), ("interleaved", 2)])
@pytest.mark.parametrize("pipeline", [0, 1, 2])
def test_group_one_rank(oracle, capi, sensors, meshes, mode_name, flags, pipeline):
    """flags: 0 = per-set mode (a communicator and a stream per buffer set, every frame one captured graph: three launches
    + ncclAllGather + rebuild), 1 = LS_GROUP_FLAG_ONE_COMMUNICATOR (round 3's arrangement), 2 = LS_GROUP_FLAG_NO_GRAPH."""
    from lidarshooter_amd import groupapi
    s = sensors["0001"]
    tr = make_tracer(capi, s, "projection")
    tr.addGeometry("ground", *[a.shape[0] for a in meshes["ground"]])
    tr.addGeometry("face", *[a.shape[0] for a in meshes["ben"]])
    tr.updateGeometry("ground", oracle.IDENTITY_AFFINE, *meshes["ground"])
    tr.updateGeometry("face", oracle.IDENTITY_AFFINE, *meshes["ben"])
    tr.setOption(capi.LS_OPT_PIPELINE, pipeline)
    g = groupapi.Group(tr, 1, 0, groupapi.SHARDED if mode_name == "sharded" else groupapi.INTERLEAVED, flags=flags)
    three = tr.info(capi.LS_INFO_PIPELINE_MODE) == 2
    if mode_name == "sharded":
        info = g.rccl()
        assert info["comm_ranks"] == 1 and info["version"] > 20000 and info["device"] == 0
        assert info["per_set_streams"] == (three and flags != 1) and info["communicators"] == (3 if info["per_set_streams"] else 1)
    assert tr.info(capi.LS_INFO_FRAME_GRAPH_STATE) == (1 if three and flags == 0 else 0)
    poses = [oracle.affine_from_components(np.array((0.4 * k, -0.3 * k, 0.02 * k), np.float32), np.array((0.0, 0.0, 0.15 * k), np.float32))
             for k in range(7)]
    refs = [oracle.trace_frame(s, [(0, *meshes["ground"], oracle.IDENTITY_AFFINE), (1, *meshes["ben"], A)]) for A in poses]
    assert len({r["points"].shape[0] for r in refs}) > 2
    for f, A in enumerate(poses):
        tr.updateGeometryTransform("face", A)
        assert tr.commitScene() == 0
        assert g.owns(f) and g.trace(f) == 0
        if f >= 2 and f % 2 == 0:                  # frames f-2 .. f are the three sets' current tenants
            for k in (f - 2, f - 1, f):
                pts, hits = g.download(k)
                assert np.array_equal(pts, refs[k]["points"]) and np.array_equal(hits, refs[k]["hits"])
    assert tr.info(capi.LS_INFO_PIPELINE_MODE) == 2    # the group keeps three frames in flight per rank
    if three and flags == 0:   # every frame went out as one graph launch: three captures, the moving mesh's pose patched in
        assert tr.info(capi.LS_INFO_FRAME_GRAPH_STATE) == 1 and tr.info(capi.LS_INFO_FRAME_GRAPH_CAPTURES) == 3
        assert tr.info(capi.LS_INFO_FRAME_GRAPH_REPLAYS) == len(poses)
    g.close()
    assert tr.info(capi.LS_INFO_FRAME_GRAPH_STATE) == 0
    assert tr.info(capi.LS_INFO_PIPELINE_MODE) == pipeline
    rc, pts, hits = tr.traceScene(99)              # the tracer is the caller's again
    assert rc == 0 and np.array_equal(pts, refs[-1]["points"])
    tr.close()


def test_group_arrangement_is_agreed_on_and_hit_buffers_refuse_points(oracle, capi, sensors, meshes):
    """ADVICE round 4.  (1) The group's arrangement is the AND of what the ranks can do, gathered over the first
    communicator: with LS_GROUP_FLAG_DEBUG_PEER_REFUSES the gathered answers read as if a peer could do neither, and this
    rank -- which could -- runs the one-communicator path like that peer (its duplicates destroyed, no frame graphs), frames
    still the oracle's.  (2) The group installs hit buffers alone: LS_OPT_EMIT_POINTS = 1 is refused while it is attached,
    and the caller's own value of the option comes back at destroy."""
    from lidarshooter_amd import groupapi
    s = sensors["0000"]
    tr = make_tracer(capi, s, "projection")
    tr.addGeometry("ground", *[a.shape[0] for a in meshes["ground"]])
    tr.updateGeometry("ground", oracle.IDENTITY_AFFINE, *meshes["ground"])
    ref = oracle.trace_frame(s, [(0, *meshes["ground"], oracle.IDENTITY_AFFINE)])
    g = groupapi.Group(tr, 1, 0, groupapi.SHARDED)
    mine = g.info(groupapi.INFO_ARRANGEMENT_MINE)
    assert g.info(groupapi.INFO_ARRANGEMENT_COMMON) == mine and bool(g.info(groupapi.INFO_PER_SET)) == (mine == 3)
    assert tr.info(capi.LS_INFO_EMIT_POINTS) == 0
    with pytest.raises(capi.LidarShooterHipError, match="hit buffers alone"):
        tr.setOption(capi.LS_OPT_EMIT_POINTS, 1)
    g.close()
    assert tr.info(capi.LS_INFO_EMIT_POINTS) == 1
    tr.setOption(capi.LS_OPT_EMIT_POINTS, 0)            # a caller that had it off gets it back off
    g = groupapi.Group(tr, 1, 0, groupapi.SHARDED, flags=groupapi.FLAG_DEBUG_PEER_REFUSES)
    assert g.info(groupapi.INFO_ARRANGEMENT_MINE) == mine and g.info(groupapi.INFO_ARRANGEMENT_COMMON) == 0
    assert g.info(groupapi.INFO_PER_SET) == 0 and g.info(groupapi.INFO_COMMUNICATORS) == 1
    assert tr.info(capi.LS_INFO_FRAME_GRAPH_STATE) == 0
    for f in range(5):
        assert tr.commitScene() == 0 and g.trace(f) == 0
    for f in (2, 3, 4):
        pts, hits = g.download(f)
        assert np.array_equal(pts, ref["points"]) and np.array_equal(hits, ref["hits"])
    g.close()
    assert tr.info(capi.LS_INFO_EMIT_POINTS) == 0
    tr.setOption(capi.LS_OPT_EMIT_POINTS, 1)
    # hit buffers alone, by hand: a trace with points switched on is refused, not written through the marker pointer
    import torch
    dev = torch.device("cuda", 0)
    hits_d = torch.zeros(16 * s.V * s.H, dtype=torch.uint8, device=dev)
    n_d = torch.zeros(16, dtype=torch.int32, device=dev)
    with pytest.raises(capi.LidarShooterHipError, match="EMIT_POINTS"):
        tr.setHitBuffers(hits_d.data_ptr(), n_d.data_ptr(), s.V * s.H)
    tr.setOption(capi.LS_OPT_EMIT_POINTS, 0)
    tr.setHitBuffers(hits_d.data_ptr(), n_d.data_ptr(), s.V * s.H)
    assert tr.commitScene() == 0
    tr.traceSceneAsync(0)
    tr.synchronize()
    n = int(n_d[0].item())
    assert n == ref["hits"].shape[0]
    assert np.array_equal(hits_d.cpu().numpy()[:16 * n].view(np.uint32).reshape(n, 4), ref["hits"])
    tr.setHitBuffers(None, None, 0)
    tr.setOption(capi.LS_OPT_EMIT_POINTS, 1)
    rc, pts, _ = tr.traceScene(1)
    assert rc == 0 and np.array_equal(pts, ref["points"])
    tr.close()


@pytest.mark.parametrize("group,flags", [("sharded", 0), ("sharded", 1), ("interleaved", 0)])
def test_lsbench_ranks_one(oracle, sensors, meshes, group, flags):
    """lsbench --ranks 1: the C++ harness through lidarshooter_group.h (fork per rank, id through a file, RCCL)."""
    import hashlib
    exe = os.path.join(ROOT, "lidarshooter_amd", "lsbench")
    out = subprocess.run([exe, "--config", os.path.join(DATA, "config", "hesai-pandar-XT-32-lidar_0000.json"),
                          "--mesh", "ground=" + os.path.join(DATA, "mesh", "ground.stl"),
                          "--mesh", "face=" + os.path.join(DATA, "mesh", "ben.stl"),
                          "--frames", "120", "--warmup", "10", "--ranks", "1", "--group", group, "--group-flags", str(flags)],
                         capture_output=True, text=True, timeout=180)
    assert out.returncode == 0, out.stderr
    rec = json.loads(out.stdout.strip().splitlines()[-1])
    assert rec["ranks"] == 1 and rec["group"] == group and rec["points_last_frame"] == 1781
    if group == "sharded":
        assert rec["rccl"]["comm_ranks"] == 1 and rec["rccl"]["communicators"] == (3 if flags == 0 else 1)
        assert rec["frame_graph_state"] == (1 if flags == 0 else 0) and rec["frame_graphs_captured"] == (3 if flags == 0 else 0)
    ref = oracle.trace_frame(sensors["0000"], [(0, *meshes["ground"], oracle.IDENTITY_AFFINE), (1, *meshes["ben"], oracle.IDENTITY_AFFINE)])
    assert rec["points_sha256"] == hashlib.sha256(ref["points"].tobytes()).hexdigest()


def test_sharded_path_two_logical_ranks_on_one_device(oracle, capi, sensors, meshes):
    """Every kernel and every ordering primitive of the N > 1 sharded path on one device, against the oracle: world = 2 as
    two handles (RCCL refuses two ranks on one GPU, so the device-side concatenation of the two slots stands in for
    ncclAllGather).  Per frame and rank: ls_tracer_wait_event guards the slot that is written again (three buffer sets),
    ls_tracer_set_shard'ed handle with the sector cull and group culling on, three frames in flight (LS_OPT_PIPELINE 2),
    k_pack writes count word + hit records straight into the rank's slot of the gathered buffer,
    ls_tracer_order_after_last_frame orders the collective stream after THAT frame only, ls_expand_gathered_hits_on
    rebuilds the cloud there -- what ls_group_trace does (lidarshooter_amd/csrc/ls_group.cpp) -- while a mesh moves."""
    import torch
    from lidarshooter_amd import shards, synth
    base = sensors["0000"]
    s = oracle.Sensor(uid="syn", vertical=synth.syn_vertical(64), h_begin=np.float32(0.0), h_end=np.float32(360.0), h_count=1024,
                      R=base.R, Rinv=base.Rinv, t=base.t)
    gv, gt = synth.grid_mesh(600, 450, half=50.0, seed=7)        # 540 000 triangles: group culling applies (>= 524 288)
    bv, bt = meshes["ben"]
    world, sets, frames = 2, 3, 9
    cap = shards.slot_capacity(s.V, s.H, world)
    sb = shards.slot_bytes(cap)
    dev = "cuda:0"
    trs = []
    for rank in range(world):
        tr = make_tracer(capi, s, "projection")
        tr.setShard(*shards.shard_columns(s.H, world, rank))
        tr.setOption(capi.LS_OPT_BLOCK_CULL, 1)
        tr.addGeometry("ground", gv.shape[0], gt.shape[0])
        tr.addGeometry("face", bv.shape[0], bt.shape[0])
        tr.updateGeometry("ground", oracle.IDENTITY_AFFINE, gv, gt)
        tr.updateGeometry("face", oracle.IDENTITY_AFFINE, bv, bt)
        assert tr.commitScene() == 0
        tr.setOption(capi.LS_OPT_PIPELINE, 2)
        assert tr.info(capi.LS_INFO_PIPELINE_MODE) in (1, 2)
        trs.append(tr)
    collect = torch.cuda.Stream(dev)
    gathered = [torch.zeros(world * sb, dtype=torch.uint8, device=dev) for _ in range(sets)]
    local_pts = [[torch.zeros(32 * cap, dtype=torch.uint8, device=dev) for _ in range(sets)] for _ in range(world)]
    cloud = [(torch.zeros(32 * cap * world, dtype=torch.uint8, device=dev), torch.zeros(16 * cap * world, dtype=torch.uint8, device=dev),
              torch.zeros(4, dtype=torch.int32, device=dev)) for _ in range(sets)]
    collected = [torch.cuda.Event() for _ in range(sets)]
    used = [False] * sets
    poses = [oracle.affine_from_components(np.array((6.0 + 0.5 * k, -4.0 + 1.1 * k, 0.1 * k), np.float32), np.array((0.0, 0.05 * k, 0.4 * k), np.float32))
             for k in range(frames)]
    refs = [oracle.trace_frame(s, [(0, gv, gt, oracle.IDENTITY_AFFINE), (1, bv, bt, A)], use_bvh=True) for A in poses]
    assert len({r["points"].shape[0] for r in refs}) > 2
    for f, A in enumerate(poses):
        b = f % sets
        for rank, tr in enumerate(trs):
            if used[b]:                                          # the slot's previous tenant has been rebuilt into its cloud
                (tr.waitEvent if rank == 0 else tr.nextFrameWaits)(collected[b].cuda_event)   # (both forms of the guard)
            slot = gathered[b].data_ptr() + rank * sb
            tr.setOutputBuffers(local_pts[rank][b].data_ptr(), slot + shards.HEADER, slot, cap)
            tr.updateGeometryTransform("face", A)
            assert tr.commitScene() == 0
            tr.traceSceneAsync(f)
            tr.orderAfterLastFrame(collect.cuda_stream)           # this frame only: the two before it are still in flight
        pts, hts, n = cloud[b]
        trs[0].expandGatheredHitsOn(collect.cuda_stream, gathered[b].data_ptr(), world, cap, pts.data_ptr(), hts.data_ptr(), n.data_ptr())
        collected[b].record(collect)
        used[b] = True
        if f % sets == sets - 1:                                   # frames f-2 .. f are the sets' tenants
            collect.synchronize()
            for k in range(f - sets + 1, f + 1):
                pts, hts, n = cloud[k % sets]
                cnt = int(n[0].item())
                assert cnt == refs[k]["points"].shape[0], (k, cnt)
                got_h = hts.cpu().numpy()[:16 * cnt].view(np.uint32).reshape(cnt, 4)
                got_p = pts.cpu().numpy()[:32 * cnt].reshape(cnt, 32)
                order = np.argsort(got_h[:, 0], kind="stable")     # the gathered cloud is sector-major: rank 0's records, then rank 1's
                assert np.array_equal(got_h[order], refs[k]["hits"]) and np.array_equal(got_p[order], refs[k]["points"])
                half = int((got_h[:, 0] % s.H < s.H // 2).sum())
                assert np.all(got_h[:half, 0] % s.H < s.H // 2) and np.all(got_h[half:, 0] % s.H >= s.H // 2)
    for tr in trs:
        tr.synchronize()
        tr.close()


def test_bench_two_ranks_rehearsal_without_a_launcher():
    """`bench.py --gpus 2` with WORLD_SIZE unset: the script starts its two ranks itself (torch.distributed.run as a child,
    before anything touches the GPU) and rank 0 prints ONE line that says n_gpus 2.  LS_BENCH_REHEARSAL=1: the two ranks
    share this box's one GPU over gloo (RCCL refuses two ranks on a device), which drives the sharded N > 1 code of bench.py
    -- shards, slots, gather, rebuild -- end to end; the gathered cloud is the one-GPU cloud."""
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["LS_BENCH_REHEARSAL"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "2", "--workload", "xt32",
                          "--no-cpu-baseline", "--min-ms", "5", "--prime-ms", "0"], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["value"] > 0 and rec["scaling"] == "strong"
    assert rec["gathered_points_rank0"] == 1781                      # OptixTracer_test.cpp:122-169 through two shards
    assert rec["rccl"]["comm_ranks"] == 2 and rec["rccl"]["comm_ranks_on_every_rank"] == [2, 2]
    assert rec["rccl"]["launched_by"].startswith("bench.py itself")
    assert rec["rehearsal_gloo_shared_gpu"] is True


def test_bench_every_phase_of_the_multi_rank_flow_through_a_one_rank_communicator():
    """LS_BENCH_FORCE_GROUP=1: bench.py's N > 1 flow on one GPU, phase by phase -- frames interleaved, the sharded split on one
    communicator (the arrangement the line falls back on), the sharded split with per-set communicators and frame graphs
    (`value`), under the watchdog -- through include/lidarshooter_group.h with a one-rank RCCL communicator.  Every cloud the
    reference's 1781 points."""
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["LS_BENCH_FORCE_GROUP"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "10", "--warmup", "2", "--workload", "xt32",
                          "--no-cpu-baseline", "--no-dropin", "--min-ms", "5", "--prime-ms", "0"], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert "error" not in rec and rec["n_gpus"] == 1 and rec["value"] > 0
    assert rec["gathered_points_rank0"] == 1781                                  # OptixTracer_test.cpp:122-169
    assert rec["rccl"]["communicators"] == 3 and "ONE captured HIP graph launch" in rec["config"]["parallelism"]
    also = rec["also_measured"]
    assert len(also) == 2
    assert "collective stream" in also[0]["parallelism"] and also[0]["gathered_points_rank0"] == 1781 and also[0]["value"] > 0
    assert also[1]["parallelism"] == "single GPU" and also[1]["value"] > 0      # (frames "interleaved" over one GPU)


def test_bench_two_ranks_watchdog_reports_what_finished():
    """The sharded phase of `bench.py --gpus 2` made to hang (LS_BENCH_DEBUG_STALL): the watchdog ends both ranks, rank 0's ONE
    line carries the frame-interleaved measurement that did finish -- labelled as such -- and an "error" that names the phase
    (the multi-rank RCCL path has never run on hardware: a hang there must not hold an 8-GPU node until somebody's limit)."""
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update({"LS_BENCH_REHEARSAL": "1", "LS_BENCH_DEBUG_STALL": "sharded", "LS_BENCH_WATCHDOG_S": "45"})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "2", "--workload", "xt32",
                          "--no-cpu-baseline", "--min-ms", "5", "--prime-ms", "0"], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode != 0, out.stderr[-3000:]   # (a job whose watchdog fired did not succeed: the ranks leave with 4, the launcher says so; the line is there)
    lines = [ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["value"] > 0 and rec["scaling"] == "weak"
    assert "azimuth shards + all-gather" in rec["error"] and "interleaved" in rec["config"]["parallelism"]
    assert "watchdog" in out.stderr
    # ... and the phase raising on ONE rank (a communicator that could not be made there): that rank leaves, its peer's
    # collective fails or times out, rank 0 reports the same way
    env.pop("LS_BENCH_DEBUG_STALL")
    env["LS_BENCH_DEBUG_RAISE"] = "sharded:1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "2", "--workload", "xt32",
                          "--no-cpu-baseline", "--min-ms", "5", "--prime-ms", "0"], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode != 0, out.stderr[-3000:]   # (a job whose watchdog fired did not succeed: the ranks leave with 4, the launcher says so; the line is there)
    lines = [ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["scaling"] == "weak" and "azimuth shards + all-gather" in rec["error"]
    assert "LS_BENCH_DEBUG_RAISE" in out.stderr


@pytest.mark.parametrize("flags", [0, 1])
def test_group_empty_scene_and_back(oracle, capi, sensors, meshes, flags):
    """OptixTracer.cpp:263-288 through the group: a frame of an empty scene traces nothing (-1) and still travels -- an empty
    slot is gathered, the cloud holds zero points -- in the per-set arrangement (where that frame is plain work on the set's
    stream, outside the cached graph) and in the one-communicator one; geometry added afterwards, frames are clouds again."""
    from lidarshooter_amd import groupapi
    s = sensors["0000"]
    tr = make_tracer(capi, s, "projection")
    g = groupapi.Group(tr, 1, 0, groupapi.SHARDED, flags=flags)
    for f in range(4):
        assert tr.commitScene() == -1
        assert g.trace(f) == -1
    pts, hits = g.download(3)
    assert pts.shape[0] == 0 and hits.shape[0] == 0
    tr.addGeometry("ground", *[a.shape[0] for a in meshes["ground"]])
    tr.updateGeometry("ground", oracle.IDENTITY_AFFINE, *meshes["ground"])
    ref = oracle.trace_frame(s, [(0, *meshes["ground"], oracle.IDENTITY_AFFINE)])
    for f in range(4, 9):
        assert tr.commitScene() == 0 and g.trace(f) == 0
    for f in (6, 7, 8):
        pts, hits = g.download(f)
        assert pts.shape[0] == 1668 and np.array_equal(pts, ref["points"]) and np.array_equal(hits, ref["hits"])   # EmbreeTracer_test.cpp:122-135
    assert tr.removeGeometry("ground") == 0
    assert g.trace(9) == -1
    pts, _ = g.download(9)
    assert pts.shape[0] == 0
    g.close()
    tr.close()


def test_group_sized_gather(oracle, capi, sensors, meshes):
    """LS_GROUP_FLAG_SIZED_GATHER on a one-rank communicator: after three frames the gather moves the front of the slot only
    (what the set's previous frame needed + a quarter, in steps of capacity / 16), the clouds stay the oracle's; when the scene
    suddenly has more hits than travel, those frames are REPORTED as truncated -- never delivered short -- and the frames that
    follow are sized up again."""
    from lidarshooter_amd import groupapi
    s = sensors["0000"]
    tr = make_tracer(capi, s, "projection")
    tr.addGeometry("face", *[a.shape[0] for a in meshes["ben"]])
    tr.updateGeometry("face", oracle.IDENTITY_AFFINE, *meshes["ben"])
    g = groupapi.Group(tr, 1, 0, groupapi.SHARDED, flags=groupapi.FLAG_SIZED_GATHER)
    if not g.info(groupapi.INFO_PER_SET):
        g.close(); tr.close()
        pytest.skip("fewer than three concurrent streams on this device")
    cap = s.V * s.H
    step = (cap + 15) // 16
    assert g.info(groupapi.INFO_GATHER_CAPACITY) == cap
    few = oracle.trace_frame(s, [(0, *meshes["ben"], oracle.IDENTITY_AFFINE)])
    n_few = few["points"].shape[0]
    assert 0 < n_few < 400
    for f in range(8):
        assert tr.commitScene() == 0 and g.trace(f) == 0
    small = g.info(groupapi.INFO_GATHER_CAPACITY)
    assert small == -(-(n_few + n_few // 4 + 1024) // step) * step < cap     # sized from the frames before
    for f in (5, 6, 7):
        pts, hits = g.download(f)
        assert np.array_equal(pts, few["points"]) and np.array_equal(hits, few["hits"])
    assert tr.info(capi.LS_INFO_FRAME_GRAPH_CAPTURES) >= 6                   # the change of size captured the three graphs anew
    # ---- the ground appears: 1 668 + hits, more than the 1 200-odd records that travel
    tr.addGeometry("ground", *[a.shape[0] for a in meshes["ground"]])
    tr.updateGeometry("ground", oracle.IDENTITY_AFFINE, *meshes["ground"])
    both = oracle.trace_frame(s, [(0, *meshes["ben"], oracle.IDENTITY_AFFINE), (1, *meshes["ground"], oracle.IDENTITY_AFFINE)])
    assert both["points"].shape[0] > small
    assert tr.commitScene() == 0 and g.trace(8) == 0
    with pytest.raises(capi.LidarShooterHipError, match="outgrew"):
        g.download(8)
    # a caller that reads ls_group_cloud's device pointers itself asks ls_group_frame_status: truncated,
    # and counted at once -- not only when the set is reused three frames later
    g.cloud(8)
    assert g.frameStatus(8) < 0 and g.frameStatus(7) == 0 and g.frameStatus(2) < 0   # (7: complete; 2: its buffers are long reused)
    assert g.info(groupapi.INFO_TRUNCATED_FRAMES) == 1
    for f in range(9, 16):
        assert tr.commitScene() == 0 and g.trace(f) == 0
    assert g.info(groupapi.INFO_TRUNCATED_FRAMES) >= 3                        # frames 8 .. 10 were sized from frames 5 .. 7
    assert g.info(groupapi.INFO_GATHER_CAPACITY) >= both["points"].shape[0]
    for f in (13, 14, 15):
        pts, hits = g.download(f)
        assert np.array_equal(pts, both["points"]) and np.array_equal(hits, both["hits"])
    g.close()
    tr.close()
