"""Where the libraries of a profiled bench process sit relative to each other (resolving the frames of
gpurun_out/r5e_graph_fuse512/out.txt, whose stack below frame_graph_close is '(unknown)').  Same imports in the same order as
tools/shard_cost.py, a few frames through captured frame graphs so that everything lazily loaded is loaded, then the
executable mappings of this process, one line each: start end offset path.  Run it under the profiler the record was taken
under (rocprofv3 --kernel-trace --stats -- python3 tools/maps_probe.py OUT); tools/resolve_stack.py does the arithmetic.
"""
import ctypes as C
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from lidarshooter_amd import capi, hostapi, shards
import bench

bench.pin_to_gpu_numa_node(0)
sensor, meshes = bench.build_workload("xt32")
dev = torch.device("cuda", 0)
tr = capi.Tracer(sensor["vertical"], sensor["h_begin"], sensor["h_end"], int(sensor["h_count"]), sensor["Rinv"], sensor["t"])
tr.setOption(capi.LS_OPT_ENGINE, 2)
for n, v, t in meshes:
    tr.addGeometry(n, v.shape[0], t.shape[0])
    tr.updateGeometry(n, capi.IDENTITY_AFFINE, v, t)
tr.commitScene()
tr.setOption(capi.LS_OPT_PIPELINE, 2)
tr.setOption(capi.LS_OPT_FRAME_GRAPH, 1)
for i in range(30):
    tr.updateGeometryTransform("face", capi.IDENTITY_AFFINE)
    tr.commitScene()
    tr.traceSceneAsync(i)
tr.synchronize()
out = sys.argv[1] if len(sys.argv) > 1 else "/dev/stdout"
with open(out, "w") as f:
    for ln in open("/proc/self/maps"):
        p = ln.split()
        if len(p) >= 6 and "x" in p[1] and p[5].startswith("/"):
            a, b = p[0].split("-")
            f.write(f"{a} {b} {p[2]} {p[5]}\n")
    f.write("# frame_graph_state %d replays %d\n" % (tr.info(capi.LS_INFO_FRAME_GRAPH_STATE), tr.info(capi.LS_INFO_FRAME_GRAPH_REPLAYS)))
tr.close()
