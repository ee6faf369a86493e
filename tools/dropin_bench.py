"""Drop-in (ITracer adapter) frame cost on SYN-128 x SYN-1M: MeshProjector::traceAffineMesh through
integration/HipTracer.hpp (stub-typed), mesh re-uploaded every frame vs pose-only change."""
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from lidarshooter_amd import adapterapi, capi, synth  # noqa: E402

DATA = os.path.join(ROOT, "tests", "golden", "data")


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    cfg = synth.write_sensor_json(os.path.join(DATA, "config", "hesai-pandar-XT-32-lidar_0000.json"),
                                  os.path.join(tempfile.mkdtemp(), "syn128.json"), synth.syn_vertical(128), 0.0, 360.0, 4096)
    v, t = synth.syn_1m()
    t0 = time.perf_counter()
    tr = adapterapi.AdapterTracer(cfg)
    t_create = time.perf_counter() - t0
    tr.meshFromArrays("ground", v, t, point_step=16)
    tr.addGeometry("ground")
    t0 = time.perf_counter()
    tr.frameLoop(1)
    first = time.perf_counter() - t0
    out = {"create_s": t_create, "first_frame_ms": first * 1e3, "host_threads": tr.L and capi.load().ls_get_info(tr.handle(), capi.LS_INFO_HOST_THREADS)}
    for label, skip in (("upload_always", False), ("skip_unchanged", True)):
        tr.setSkipUnchanged(skip)
        tr.frameLoop(5)
        runs = [tr.frameLoop(frames) * 1e3 for _ in range(5)]
        out[label + "_ms"] = float(np.median(runs))
        out[label + "_runs"] = [round(r, 4) for r in runs]
    c = tr.cloud()
    out["points"] = int(c["width"])
    # piece by piece through the C ABI
    L = capi.load()
    h = tr.handle()
    import ctypes as C
    fr = capi.Frame()
    for ho in (2, 1, 0):
        L.ls_tracer_set_option(h, capi.LS_OPT_HOST_OUTPUT, ho)
        for _ in range(5):
            L.ls_trace_scene(h, 0, C.byref(fr))
        t0 = time.perf_counter()
        for _ in range(frames):
            L.ls_trace_scene(h, 0, C.byref(fr))
        out[f"trace_scene_sync_ms_host_output{ho}"] = (time.perf_counter() - t0) / frames * 1e3
    L.ls_tracer_set_option(h, capi.LS_OPT_HOST_OUTPUT, 1)
    L.ls_trace_scene(h, 0, C.byref(fr))
    dst = np.zeros(int(fr.n_points) * 32, np.uint8)
    t0 = time.perf_counter()
    for _ in range(frames):
        L.ls_expand_points(dst.ctypes.data, C.cast(fr.points32, C.c_void_p), fr.n_points // 2)
    out["expand_half_ms"] = (time.perf_counter() - t0) / frames * 1e3
    t0 = time.perf_counter()
    for _ in range(frames):
        L.ls_parallel_copy(dst.ctypes.data, C.cast(fr.points32, C.c_void_p), dst.size)
    out["parallel_copy_ms"] = (time.perf_counter() - t0) / frames * 1e3
    t0 = time.perf_counter()
    for _ in range(10):
        C.memmove(dst.ctypes.data, C.cast(fr.points32, C.c_void_p), dst.size)
    out["memmove_ms"] = (time.perf_counter() - t0) / 10 * 1e3
    padded = np.zeros((v.shape[0], 4), np.float32)
    padded[:, :3] = v
    A = capi.IDENTITY_AFFINE
    f32p = C.POINTER(C.c_float)
    t0 = time.perf_counter()
    for _ in range(frames):
        L.ls_update_geometry(h, b"ground", A.ctypes.data_as(f32p), padded.ctypes.data, 16, None)
    t_enq = (time.perf_counter() - t0) / frames * 1e3
    L.ls_tracer_synchronize(h)
    out["update_vertices_enqueue_ms"] = t_enq
    t0 = time.perf_counter()
    for _ in range(frames):
        L.ls_update_geometry(h, b"ground", A.ctypes.data_as(f32p), padded.ctypes.data, 16, None)
        L.ls_tracer_synchronize(h)
    out["update_vertices_sync_ms"] = (time.perf_counter() - t0) / frames * 1e3
    L.ls_tracer_set_option(h, capi.LS_OPT_HOST_OUTPUT, 2)
    print(json.dumps(out))
    tr.close()


if __name__ == "__main__":
    main()
