"""ls_update_geometry (8 MB of pcl::PointXYZ vertices) under the library's upload knobs; one process per setting.
LS_UPLOAD_MODE here selects LS_OPT_UPLOAD_MODE (an option of the handle); the chunk / run / thread knobs are read only
by a library built with -DLS_EXPERIMENTAL (tools/exp_build.sh, then LS_LIB_PATH=build/exp/<name>/liblidarshooter_hip.so)."""
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np
    from lidarshooter_amd import capi, synth
    v, t = synth.syn_1m()
    padded = np.zeros((v.shape[0], 4), np.float32)
    padded[:, :3] = v
    tr = capi.Tracer(synth.syn_vertical(128), 0.0, 360.0, 4096, np.eye(3, dtype=np.float32).reshape(9), np.zeros(3, np.float32))
    tr.addGeometry("g", v.shape[0], t.shape[0])
    tr.setOption(capi.LS_OPT_UPLOAD_MODE, int(os.environ.get("LS_UPLOAD_MODE", "1")))
    tr.updateGeometry("g", capi.IDENTITY_AFFINE, padded, t, stride=16)
    tr.synchronize()
    L, h = tr.L, tr.h
    A = capi.IDENTITY_AFFINE
    f32p = C.POINTER(C.c_float)
    res = {}
    for _ in range(10):
        L.ls_update_geometry(h, b"g", A.ctypes.data_as(f32p), padded.ctypes.data, 16, None)
    L.ls_tracer_synchronize(h)
    n = 100
    t0 = time.perf_counter()
    for _ in range(n):
        L.ls_update_geometry(h, b"g", A.ctypes.data_as(f32p), padded.ctypes.data, 16, None)
        L.ls_tracer_synchronize(h)
    res["update_sync_ms"] = (time.perf_counter() - t0) / n * 1e3
    t0 = time.perf_counter()
    for _ in range(n):
        L.ls_update_geometry(h, b"g", A.ctypes.data_as(f32p), padded.ctypes.data, 16, None)
    res["update_enqueue_ms"] = (time.perf_counter() - t0) / n * 1e3
    L.ls_tracer_synchronize(h)
    print(json.dumps(res))
    sys.exit(0)

for env in ({"LS_UPLOAD_MODE": "1"}, {"LS_UPLOAD_MODE": "2"}, {}, {"LS_DMA_RUN": "1"}, {"LS_DMA_RUN": "2"}, {"LS_DMA_RUN": "16"},
            {"LS_COPY_CHUNK_KB": "256"}, {"LS_COPY_CHUNK_KB": "1024", "LS_DMA_RUN": "2"}, {"LS_COPY_CHUNK_KB": "128", "LS_DMA_RUN": "8"},
            {"LS_HOST_THREADS": "2"}, {"LS_HOST_THREADS": "4"}, {"LS_HOST_THREADS": "12"}, {"LS_HOST_THREADS": "16"},
            {"LS_HOST_THREADS": "4", "LS_COPY_CHUNK_KB": "256", "LS_DMA_RUN": "2"}):
    e = dict(os.environ)
    e.update(env)
    out = subprocess.run([sys.executable, __file__, "child"], env=e, capture_output=True, text=True)
    print(env, out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:], flush=True)
