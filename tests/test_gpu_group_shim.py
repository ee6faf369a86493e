"""include/lidarshooter_group.h with a REAL PEER on the one GPU the pool grants.

RCCL refuses two ranks on one device, so until round 6 the world > 1 control flow of csrc/ls_group.cpp -- communicator
duplicates, the agreement collective, per-set collectives, the sized gather -- had run against nobody.  Here two fresh child
processes (tests/shim/rank_worker.py, or lsbench / bench.py themselves) form a group on device 0 with tests/shim/librccl_shim.so
standing in for RCCL (LS_GROUP_RCCL_LIBRARY; shared-memory rendezvous, host-blocking collectives: semantics, not speed, and no
claim about xGMI).  Every rank compares its whole clouds with the CPU oracle.  What the shim cannot do -- a collective inside a
captured HIP graph -- stays untested (DESIGN.md section 8 says so): the groups here run with LS_GROUP_FLAG_NO_GRAPH or
LS_GROUP_FLAG_ONE_COMMUNICATOR.  No scaling number comes out of this file.
"""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import DATA, ROOT

pytestmark = pytest.mark.gpu

SHIM = os.path.join(ROOT, "tests", "shim", "librccl_shim.so")
WORKER = os.path.join(ROOT, "tests", "shim", "rank_worker.py")
# the headline cloud (SYN-128 x SYN-1M, 256 482 points): the same bytes since round 1 on one GPU, in every pipeline mode
HEADLINE_SHA = "5eeb1f648fe65ccb8cd13398a416c6e48d755fcc2fc353f6b31ff3f39c770de8"


def shim_env(**extra):
    assert os.path.exists(SHIM), "tests/shim/librccl_shim.so is missing: make -C tests/shim (build() does it)"
    env = dict(os.environ)
    env["LS_GROUP_RCCL_LIBRARY"] = SHIM
    env["LS_SHIM_TIMEOUT_S"] = "90"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update({k: str(v) for k, v in extra.items()})
    return env


def run_ranks(tmp_path, scenario, flags, world=2, timeout=420, **extra_env):
    """start `world` fresh children on device 0, wait for all of them, -> their verdicts (rank order)"""
    idf = str(tmp_path / "id.bin")
    procs, outs = [], []
    for r in range(world):
        out = str(tmp_path / f"rank{r}.json")
        outs.append(out)
        log = open(str(tmp_path / f"rank{r}.log"), "w")
        procs.append((subprocess.Popen([sys.executable, WORKER, "--rank", str(r), "--world", str(world), "--scenario", scenario, "--flags", str(flags),
                                        "--id-file", idf, "--out", out], env=shim_env(**extra_env), stdout=log, stderr=subprocess.STDOUT), log))
    failed = []
    for r, (p, log) in enumerate(procs):
        try:
            p.wait(timeout=timeout)
        except subprocess.TimeoutExpired:
            failed.append(r)
    for p, log in procs:          # exactly the processes started here
        if p.poll() is None:
            p.kill()
            p.wait()
        log.close()
    logs = "\n".join(f"--- rank {r}\n" + open(str(tmp_path / f"rank{r}.log")).read()[-3000:] for r in range(world))
    assert not failed, f"ranks {failed} did not finish\n{logs}"
    res = []
    for r, o in enumerate(outs):
        assert os.path.exists(o), f"rank {r} left no verdict\n{logs}"
        res.append(json.load(open(o)))
    for v in res:
        assert v["ok"], f"rank {v['rank']}: {v.get('error')}\n{v.get('traceback')}\n{logs}"
        bad = [c for c in v.get("checks", []) if not c["equal"]]
        assert not bad, f"rank {v['rank']}: clouds that differ from the oracle's: {bad}"
    return res


def assert_two_ranks_on_the_shim(res, world=2):
    assert len(res) == world
    for v in res:
        g = v["group"]
        assert g["comm_ranks"] == world and g["shim"] == 1 and g["version"] == 1 and g["device"] == 0
        assert g["pipeline_mode"] == 2           # the group keeps three frames in flight per rank


@pytest.mark.parametrize("flags", [1, 2])
def test_two_ranks_xt32_clouds_equal_the_oracle(tmp_path, flags):
    """BASELINE.json configs[1] through a two-rank sharded group: 1 = LS_GROUP_FLAG_ONE_COMMUNICATOR, 2 = per-set communicators
    (ncclCommSplit twice, the agreement collective) with plain launches.  Seven poses of ben through ls_group_trace, nine
    frames through the C++ loop lsh_group_stream_frames; every downloaded cloud equals the oracle's on BOTH ranks, 1781 points at
    the identity pose (OptixTracer_test.cpp:122-169); afterwards the tracer is a single-GPU tracer again."""
    res = run_ranks(tmp_path, "xt32", flags)
    assert_two_ranks_on_the_shim(res)
    for v in res:
        assert v["clouds_equal_oracle"] and v["after_close_equal_oracle"], v
        assert v["identity_points"] == 1781 == v["points"][0] and len(set(v["points"])) > 2
        g = v["group"]
        assert g["frame_graph_state"] == 0
        if flags == 1:
            assert g["communicators"] == 1 and g["per_set"] == 0
        else:   # per-set iff BOTH ranks found three concurrent streams (the calibration is a timing measurement: either is legal, alike)
            assert g["per_set"] == (1 if g["common"] == 3 else 0) and g["communicators"] == (3 if g["per_set"] else 1)
    assert res[0]["group"]["common"] == res[1]["group"]["common"]
    assert res[0]["group"]["per_set"] == res[1]["group"]["per_set"]


@pytest.mark.parametrize("world,flags", [(3, 2), (4, 1)])
def test_three_and_four_ranks_ragged_shards(tmp_path, world, flags):
    """more than one peer, and shards of unequal width: XT-32's 150 azimuth columns over 4 ranks are 38 + 38 + 37 + 37 (the slot
    capacity is the widest shard's); three ranks in the per-set arrangement, four on one communicator (parent + 4 children: the
    box allows six processes on the card)"""
    res = run_ranks(tmp_path, "xt32", flags, world=world)
    assert_two_ranks_on_the_shim(res, world)
    for v in res:
        assert v["clouds_equal_oracle"] and v["after_close_equal_oracle"] and v["identity_points"] == 1781
    assert len({(v["group"]["common"], v["group"]["per_set"], v["group"]["communicators"]) for v in res}) == 1


@pytest.mark.parametrize("flags", [1, 2])
def test_two_ranks_headline_cloud(tmp_path, flags):
    """BASELINE.json configs[3] -- SYN-128 x SYN-1M, azimuth halves, one all-gather of 4 MiB slots per frame: both ranks end up with the
    headline cloud, byte for byte (SHA-256 of 256 482 points), rays ascending."""
    res = run_ranks(tmp_path, "syn", flags)
    assert_two_ranks_on_the_shim(res)
    for v in res:
        assert v["points"] == 256482 and v["rays_ascending"]
        assert v["points_sha256"] == HEADLINE_SHA


def test_a_rank_whose_comm_split_fails(tmp_path):
    """ncclCommSplit fails on rank 1 ONLY (after the collective part, so its peer's calls succeed).  Before round 6
    rank 1 went on alone into a broadcast its peer was not in and the group hung in ls_group_create.  Now every rank attempts every
    split, the ranks find out together that not everybody holds everything, BOTH drop what they hold and make the duplicates from
    broadcast ids instead, together: the group still has its communicator per buffer set, clouds the oracle's."""
    res = run_ranks(tmp_path, "xt32", 2, LS_SHIM_FAIL_SPLIT_RANK=1)
    assert_two_ranks_on_the_shim(res)
    for v in res:
        g = v["group"]
        assert g["mine"] & 1 == 1 and g["common"] & 1 == 1
        assert g["per_set"] == (1 if g["common"] == 3 else 0) and g["communicators"] == (3 if g["per_set"] else 1)
        assert v["clouds_equal_oracle"] and v["identity_points"] == 1781
    assert res[0]["group"]["per_set"] == res[1]["group"]["per_set"]


def test_a_rank_that_cannot_hold_any_duplicate(tmp_path):
    """... and when the broadcast ids fail on rank 1 as well (its ncclCommInitRank, after the rendezvous): rank 0 holds both of its
    duplicates, rank 1 none; the agreement collective (the AND over the ranks) makes BOTH run the one-communicator arrangement."""
    res = run_ranks(tmp_path, "xt32", 2, LS_SHIM_FAIL_SPLIT_RANK=1, LS_SHIM_FAIL_INIT_RANK=1)
    assert_two_ranks_on_the_shim(res)
    assert res[0]["group"]["mine"] & 1 == 1 and res[1]["group"]["mine"] & 1 == 0
    for v in res:
        g = v["group"]
        assert g["common"] & 1 == 0 and g["per_set"] == 0 and g["communicators"] == 1
        assert v["clouds_equal_oracle"] and v["identity_points"] == 1781


def test_duplicates_by_a_broadcast_id_when_the_library_cannot_split(tmp_path):
    """a collective library whose ncclCommSplit does not work (on every rank alike): the duplicates come from fresh ids that rank 0
    makes and the first communicator broadcasts (ncclBroadcast + ncclCommInitRank)"""
    res = run_ranks(tmp_path, "xt32", 2, LS_SHIM_NO_SPLIT=1)
    assert_two_ranks_on_the_shim(res)
    for v in res:
        g = v["group"]
        assert g["mine"] & 1 == 1 and g["common"] & 1 == 1
        assert g["communicators"] == (3 if g["per_set"] else 1)
        assert v["clouds_equal_oracle"] and v["identity_points"] == 1781


def test_sized_gather_truncated_by_one_rank_is_reported_by_both(tmp_path):
    """LS_GROUP_FLAG_SIZED_GATHER: the gather shrinks to what the scene needs; then a wall that only rank 1's half of the turn sees
    makes rank 1's hits outgrow it.  BOTH ranks report that frame as truncated (LS_ERR_OUT_OF_RANGE = -9 from ls_group_frame_status
    and from the download) and neither delivers it; a few frames later the gather has grown back and both hold the oracle's cloud
    with the wall.  (Per-set mode is a precondition of the flag: where the two ranks did not both find three concurrent streams the
    fixed-capacity gather runs and nothing is ever truncated -- asserted as that.)"""
    res = run_ranks(tmp_path, "sized", 2)
    assert_two_ranks_on_the_shim(res)
    assert res[0]["group"]["per_set"] == res[1]["group"]["per_set"]
    for v in res:
        assert v["clouds_equal_oracle"], v
        assert v["points_without_and_with_wall"][1] > v["points_without_and_with_wall"][0] + 10000
        full, small = v["capacity_full_then_sized"]
        assert full == 64 * 512
        if v["group"]["per_set"]:
            assert small < full and v["truncated_before"] == 0
            assert v["status_of_the_outgrown_frame"] < 0 and not v["outgrown_frame_was_delivered"], v
            assert "outgrew" in v["download_error"]
            # three frames are truncated, not one: a set's gather is sized from the set's PREVIOUS tenant, three frames back, so
            # the two frames after the wall's first still travel at the old size (lidarshooter_group.h: "within three frames")
            assert v["truncated_after"] == 3 and v["capacity_at_the_end"] == full
        else:
            assert small == full and v["outgrown_frame_was_delivered"] and v["truncated_after"] == 0
    assert res[0]["status_of_the_outgrown_frame"] == res[1]["status_of_the_outgrown_frame"]


@pytest.mark.parametrize("flags", [1, 2])
def test_an_empty_scene_in_the_middle_of_a_stream(tmp_path, flags):
    """OptixTracer.cpp:263-288 through a two-rank group: commitScene and the frame return -1 on both ranks, an empty slot travels,
    the frame's cloud has no points; the frames before and after it are the oracle's."""
    res = run_ranks(tmp_path, "empty", flags)
    assert_two_ranks_on_the_shim(res)
    for v in res:
        assert v["return_codes"] == [0, 0, 0, 0, -1, -1, 0, 0, 0, 0]
        assert v["points_of_the_empty_frame"] == 0 and v["clouds_equal_oracle"]


@pytest.mark.parametrize("group,flags", [("sharded", 1), ("sharded", 2), ("interleaved", 2)])
def test_lsbench_two_ranks_on_one_device(oracle, sensors, meshes, group, flags):
    """the C++ harness, no Python in the ranks: lsbench --ranks 2 forks two processes that share device 0 (lsbench.cpp picks device 0 for
    every rank when the node has fewer devices than ranks) and meet through the shim"""
    exe = os.path.join(ROOT, "lidarshooter_amd", "lsbench")
    out = subprocess.run([exe, "--config", os.path.join(DATA, "config", "hesai-pandar-XT-32-lidar_0000.json"),
                          "--mesh", "ground=" + os.path.join(DATA, "mesh", "ground.stl"), "--mesh", "face=" + os.path.join(DATA, "mesh", "ben.stl"),
                          "--frames", "60", "--warmup", "6", "--ranks", "2", "--group", group, "--group-flags", str(flags)],
                         capture_output=True, text=True, timeout=300, env=shim_env())
    assert out.returncode == 0, out.stderr[-3000:]
    rec = json.loads(out.stdout.strip().splitlines()[-1])
    assert rec["ranks"] == 2 and rec["group"] == group and rec["points_last_frame"] == 1781
    if group == "sharded":
        assert rec["rccl"]["comm_ranks"] == 2 and rec["rccl"]["version"] == 1
    ref = oracle.trace_frame(sensors["0000"], [(0, *meshes["ground"], oracle.IDENTITY_AFFINE), (1, *meshes["ben"], oracle.IDENTITY_AFFINE)])
    assert rec["points_sha256"] == hashlib.sha256(ref["points"].tobytes()).hexdigest()


def test_bench_two_gpus_end_to_end_on_the_shim():
    """`python bench.py --gpus 2` as the driver would start it, on one device: LS_BENCH_SHIM=1 makes the ranks share device 0, puts
    torch.distributed's control plane (barriers, the broadcast of the id, the MAX over ranks) on gloo and the group's collectives on
    the shim.  Every phase of the N > 1 flow runs against a real peer -- frames interleaved, shards + all-gather on one communicator,
    shards + per-set communicators (plain launches), and SYN-10M -- and the line says what it is: "shim": true, never a number to quote."""
    env = shim_env(LS_BENCH_SHIM=1, LS_BENCH_WATCHDOG_S=200)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "12", "--warmup", "3", "--min-ms", "5",
                          "--prime-ms", "5", "--no-cpu-baseline", "--no-dropin"], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-4000:]
    rec = json.loads(out.stdout.strip().splitlines()[-1])
    assert "error" not in rec, rec.get("error")
    assert rec["n_gpus"] == 2 and rec["scaling"] == "strong" and rec["config"]["parallelism"].startswith("azimuth")
    assert rec["rccl"]["comm_ranks_on_every_rank"] == [2, 2] and rec["rccl"]["shim"] is True and rec["rccl"]["devices"] == [0, 0]
    assert rec["gathered_points_rank0"] == 256482
    assert rec["points_sha256"] == HEADLINE_SHA
    also = {(a["parallelism"].split(",")[0], a["workload"][:18]): a for a in rec["also_measured"]}
    assert len(rec["also_measured"]) >= 3, rec["also_measured"]
    assert any(a["scaling"] == "weak" for a in rec["also_measured"])
    assert "shim" in rec["data"]
