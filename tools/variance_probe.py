"""Where the process-to-process spread of the three-stream frame time comes from: the headline frame loop (C++ loop of
host_capi.cpp) on a tracer that is created, measured and destroyed several times inside ONE process; run the script several
times for the spread across processes.  usage: variance_probe.py [handles per process] [mode: 2 | 1 | 0]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get("PROBE_PRELOAD") == "1":   # the library (and with it /opt/rocm's HIP runtime) before torch brings its own copy
    from lidarshooter_amd import capi as _c
    _c.load()
import numpy as np, torch
from lidarshooter_amd import capi, hostapi
import bench
if os.environ.get("PROBE_NO_PIN") != "1": bench.pin_to_gpu_numa_node(0)   # (as bench.py does: the host's enqueue cost decides the three-stream frame time)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 2
sensor, meshes = bench.build_workload("syn128x1m")
dev = torch.device("cuda", 0)
if os.environ.get("PROBE_EXTRA_STREAM") == "1":   # what bench.py did until round 4: a torch stream of its own, made current, before the tracer's streams exist
    _extra = torch.cuda.Stream(dev)
    torch.cuda.set_stream(_extra)
HL = hostapi.load()
f32p = C.POINTER(C.c_float)
HL.lsh_stream_frames.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(f32p), C.POINTER(C.c_uint), C.c_uint,
                                 C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_uint, C.c_uint, C.c_uint, C.c_uint]
ident = (C.c_float * 12)(*[float(x) for x in capi.IDENTITY_AFFINE])
V, H = len(sensor["vertical"]), int(sensor["h_count"])
cap = V * H
res = []
for rep in range(reps):
    F = os.environ.get("PROBE_FLAGS", "")
    tr = capi.Tracer(sensor["vertical"], sensor["h_begin"], sensor["h_end"], H, sensor["Rinv"], sensor["t"])
    tr.setOption(capi.LS_OPT_ENGINE, 2)
    if os.environ.get("PROBE_CULL"): tr.setOption(capi.LS_OPT_BLOCK_CULL, int(os.environ["PROBE_CULL"]))
    if "f1" in F: tr.setShard(0, H)
    if "f6" in F: outs_early = [torch.zeros(64 + 48 * cap, dtype=torch.uint8, device=dev) for _ in range(3)]
    if "f2" in F: tr.setOption(capi.LS_OPT_PIPELINE, mode)
    keep = []
    for n, v, t in meshes:
        dv = torch.from_numpy(np.ascontiguousarray(v, np.float32)).to(dev)
        dt = torch.from_numpy(np.ascontiguousarray(t, np.uint32).view(np.int32)).to(dev)
        keep.append((dv, dt))
        tr.addGeometry(n, v.shape[0], t.shape[0])
        tr.updateGeometryDeviceShared(n, capi.IDENTITY_AFFINE, dv.data_ptr(), 12, dt.data_ptr())
    tr.commitScene()
    tr.setOption(capi.LS_OPT_PIPELINE, mode)
    outs = outs_early if "f6" in F else [torch.zeros(64 + 48 * cap, dtype=torch.uint8, device=dev) for _ in range(3)]
    names = (C.c_char_p * len(meshes))(*[m[0].encode() for m in meshes])
    aff = (f32p * len(meshes))(*[C.cast(ident, f32p) for _ in meshes])
    na = (C.c_uint * len(meshes))(*[1 for _ in meshes])
    P = (C.c_void_p * 3)(*[b.data_ptr() + 64 for b in outs])
    Hh = (C.c_void_p * 3)(*[b.data_ptr() + 64 + 32 * cap for b in outs])
    Cn = (C.c_void_p * 3)(*[b.data_ptr() for b in outs])
    def run(first, n):
        assert HL.lsh_stream_frames(tr.h, names, aff, na, len(meshes), P, Hh, Cn, 3, cap, first, n) == 0
    if "f3" in F:
        tr.setOption(capi.LS_OPT_COUNT_VISITS, 1); tr.setOutputBuffers(outs[0].data_ptr() + 64, outs[0].data_ptr() + 64 + 32 * cap, outs[0].data_ptr(), cap)
        tr.traceSceneAsync(0); tr.flush(); torch.cuda.synchronize(dev); tr.visitStats(); tr.setOption(capi.LS_OPT_COUNT_VISITS, 0)
    if "f4" in F:
        import ctypes
        fr = capi.Frame()
        for b in range(20):
            for i in range(50):
                tr.L.ls_update_geometry_transform(tr.h, names[0], C.cast(ident, f32p)); tr.L.ls_commit_scene(tr.h)
                tr.L.ls_tracer_set_output_buffers(tr.h, C.c_void_p(P[i % 3]), C.c_void_p(Hh[i % 3]), C.c_void_p(Cn[i % 3]), cap)
                tr.L.ls_trace_scene_async(tr.h, i, C.byref(fr))
            tr.flush(); torch.cuda.synchronize(dev)
    run(0, 600); tr.synchronize()
    if "f5" in F or "f7" in F or "f8" in F:
        def wsync():
            if "f5" in F: tr.flush(); torch.cuda.synchronize(dev)            # bench.py until round 4: the device-wide wait does the waiting
            elif "f7" in F: tr.synchronize(); torch.cuda.synchronize(dev)    # the handle's streams first: the device-wide wait finds an idle device
            else: tr.synchronize()
        ws5 = []
        gap = float(os.environ.get("PROBE_GAP_MS", "0")) * 1e-3
        for w in range(12):
            wsync()
            if gap: time.sleep(gap)
            t0 = time.perf_counter(); run(0, 200); wsync(); ws5.append((time.perf_counter() - t0) / 200 * 1e6)
        print("windows of 200: median %.2f" % float(np.median(ws5)))
    if os.environ.get("PROBE_KSWEEP"):   # what a window of K frames costs beyond K steady steps: K = 1 is a frame's whole latency + the wait
        # the three slot streams (the stream of the frame issued last, three frames running), for a wait that polls them
        hip = C.CDLL("libamdhip64.so.7")
        hip.hipStreamQuery.argtypes = [C.c_void_p]
        slot_streams = []
        for i in range(3):
            run(i, 1)
            sp = C.c_void_p()
            assert tr.L.ls_frame_graph_stream(tr.h, C.byref(sp), None, None) == 0
            slot_streams.append(sp.value)
        torch.cuda.synchronize(dev)
        print("slot streams:", ["%x" % (x or 0) for x in slot_streams])
        for poll in (0, 1, 0, 1):
            for K in [int(x) for x in os.environ["PROBE_KSWEEP"].split(",")]:
                el, en = [], []
                for w in range(40):
                    torch.cuda.synchronize(dev)
                    t0 = time.perf_counter(); run(0, K); t1 = time.perf_counter()
                    if poll:
                        for sp in slot_streams:
                            while hip.hipStreamQuery(sp) != 0: pass
                    torch.cuda.synchronize(dev); t2 = time.perf_counter()
                    el.append((t2 - t0) * 1e6); en.append((t1 - t0) * 1e6)
                print("%s window of %3d frames: median %.1f us (min %.1f), enqueue %.1f us, per frame %.2f" % ("polled " if poll else "blocked", K, float(np.median(el)), min(el), float(np.median(en)), float(np.median(el)) / K), flush=True)
    if os.environ.get("PROBE_PER_FRAME"):   # the host's enqueue time of every frame of a window that starts on an idle device
        K = int(os.environ["PROBE_PER_FRAME"])
        rows = []
        for w in range(30):
            torch.cuda.synchronize(dev)
            ts = [time.perf_counter()]
            for i in range(K):
                run(i, 1); ts.append(time.perf_counter())
            torch.cuda.synchronize(dev); t_end = time.perf_counter()
            rows.append([(ts[i + 1] - ts[i]) * 1e6 for i in range(K)] + [(t_end - ts[-1]) * 1e6, (t_end - ts[0]) * 1e6])
        if os.environ.get("PROBE_PER_FRAME_SPLIT"):   # the same windows through the loop that clocks its three calls
            HL.lsh_stream_frames_timed.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(f32p), C.POINTER(C.c_uint), C.c_uint, C.c_uint, C.c_uint, C.POINTER(C.c_double)]
            ns3 = (C.c_double * 3)()
            split = []
            for w in range(30):
                torch.cuda.synchronize(dev)
                one = []
                for i in range(K):
                    assert HL.lsh_stream_frames_timed(tr.h, names, aff, na, len(meshes), i, 1, ns3) == 0
                    one.append([ns3[0] * 1e-3, ns3[1] * 1e-3, ns3[2] * 1e-3])
                torch.cuda.synchronize(dev)
                split.append(one)
            m3 = np.median(np.array(split), axis=0)
            print("pose / commit / trace us per frame of the window: " + " ".join("%.1f/%.1f/%.1f" % tuple(x) for x in m3[:8]), flush=True)
        med = np.median(np.array(rows), axis=0)
        print("per-frame enqueue us (median of 30 windows of %d): %s | final wait %.1f | window %.1f" % (K, " ".join("%.1f" % x for x in med[:K]), med[K], med[K + 1]), flush=True)
    ws, enq = [], []
    K = int(os.environ.get("PROBE_WINDOW", "1000"))   # frames per timed window
    for w in range(int(os.environ.get("PROBE_WINDOWS", "5"))):
        t0 = time.perf_counter(); run(K * w, K); t1 = time.perf_counter(); tr.synchronize(); ws.append((time.perf_counter() - t0) / K * 1e6); enq.append((t1 - t0) / K * 1e6)
    res.append(ws)
    print("handle %d: host enqueue us per frame: %s" % (rep, " ".join("%.2f" % x for x in enq)))
    print("handle %d: us per frame per window: %s   streams %s" % (rep, " ".join("%.2f" % x for x in ws), tr.info(capi.LS_INFO_CONCURRENT_STREAMS)), flush=True)
    tr.close()
    del outs, keep
    torch.cuda.empty_cache()
hip = sorted({l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l})
print("PID %d medians: %s   HIP runtime: %s" % (os.getpid(), " ".join("%.2f" % float(np.median(w)) for w in res), hip))
