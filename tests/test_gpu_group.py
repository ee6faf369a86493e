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


@pytest.mark.parametrize("mode_name", ["sharded", "interleaved"])
@pytest.mark.parametrize("pipeline", [0, 1])
def test_group_one_rank(oracle, capi, sensors, meshes, mode_name, pipeline):
    from lidarshooter_amd import groupapi
    s = sensors["0001"]
    tr = make_tracer(capi, s, "projection")
    tr.addGeometry("ground", *[a.shape[0] for a in meshes["ground"]])
    tr.addGeometry("face", *[a.shape[0] for a in meshes["ben"]])
    tr.updateGeometry("ground", oracle.IDENTITY_AFFINE, *meshes["ground"])
    tr.updateGeometry("face", oracle.IDENTITY_AFFINE, *meshes["ben"])
    tr.setOption(capi.LS_OPT_PIPELINE, pipeline)
    g = groupapi.Group(tr, 1, 0, groupapi.SHARDED if mode_name == "sharded" else groupapi.INTERLEAVED)
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
    g.close()
    rc, pts, hits = tr.traceScene(99)              # the tracer is the caller's again
    assert rc == 0 and np.array_equal(pts, refs[-1]["points"])
    tr.close()


@pytest.mark.parametrize("group", ["sharded", "interleaved"])
def test_lsbench_ranks_one(oracle, sensors, meshes, group):
    """lsbench --ranks 1: the C++ harness through lidarshooter_group.h (fork per rank, id through a file, RCCL)."""
    import hashlib
    exe = os.path.join(ROOT, "lidarshooter_amd", "lsbench")
    out = subprocess.run([exe, "--config", os.path.join(DATA, "config", "hesai-pandar-XT-32-lidar_0000.json"),
                          "--mesh", "ground=" + os.path.join(DATA, "mesh", "ground.stl"),
                          "--mesh", "face=" + os.path.join(DATA, "mesh", "ben.stl"),
                          "--frames", "120", "--warmup", "10", "--ranks", "1", "--group", group],
                         capture_output=True, text=True, timeout=180)
    assert out.returncode == 0, out.stderr
    rec = json.loads(out.stdout.strip().splitlines()[-1])
    assert rec["ranks"] == 1 and rec["group"] == group and rec["points_last_frame"] == 1781
    ref = oracle.trace_frame(sensors["0000"], [(0, *meshes["ground"], oracle.IDENTITY_AFFINE), (1, *meshes["ben"], oracle.IDENTITY_AFFINE)])
    assert rec["points_sha256"] == hashlib.sha256(ref["points"].tobytes()).hexdigest()
