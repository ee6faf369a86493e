"""The C-ABI library loads and exports every symbol include/lidarshooter_hip.h declares.
No compute calls (there is no GPU where `-m "not gpu"` runs)."""
import ctypes
import os
import re

from conftest import ROOT


def _declared_symbols(header="lidarshooter_hip.h"):
    hdr = open(os.path.join(ROOT, "include", header)).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(ls_[a-z0-9_]+)\s*\(", hdr)))


def test_header_symbols_are_exported(capi):
    syms = _declared_symbols()
    assert len(syms) >= 30
    assert sorted(capi.SYMBOLS) == syms, "capi.SYMBOLS must list exactly the header's entry points"
    lib = ctypes.CDLL(capi.LIB_PATH)
    for s in syms:
        assert hasattr(lib, s), f"{s} is declared in the header but not exported"


def test_debug_header_symbols_are_exported(capi):
    # include/lidarshooter_hip_debug.h: the test / measurement hooks live apart from the drop-in surface
    syms = _declared_symbols("lidarshooter_hip_debug.h")
    assert sorted(capi.DEBUG_SYMBOLS) == syms
    assert not [s for s in _declared_symbols() if s.startswith("ls_debug_")]
    lib = ctypes.CDLL(capi.LIB_PATH)
    for s in syms:
        assert hasattr(lib, s), s


def test_abi_version(capi):
    assert capi.load().ls_abi_version() == 4 == capi.ABI_VERSION


def test_shipped_library_reads_no_tuning_knobs_from_the_environment():
    # experiment knobs go through lsi::tune_int (ls_tuning.h), which only reads the environment in an EXPERIMENTAL build
    src = os.path.join(ROOT, "lidarshooter_amd", "csrc")
    for fn in os.listdir(src):
        if fn.endswith((".cpp", ".hip", ".h")) and fn != "ls_tuning.h":
            txt = open(os.path.join(src, fn)).read()
            if fn == "ls_group.cpp":   # WHICH collective library to load is deployment, not tuning: the one variable the header documents
                assert re.findall(r'getenv\("([A-Z_]+)"\)', txt) == ["LS_GROUP_RCCL_LIBRARY"] and txt.count("getenv") == 1
                continue
            assert "getenv" not in txt, fn
    assert "LS_EXPERIMENTAL" in open(os.path.join(src, "ls_tuning.h")).read()


def test_library_was_built_from_these_sources(capi):
    """ls_source_hash(): the binary says which sources it was made of (a stale .so must not speak for newer
    sources -- the library is built in-tree and travels to the GPU box as a binary).  After build() the two agree; the marker
    can be read out of the file without loading it (bench.py does that before it loads anything)."""
    import hashlib
    src = os.path.join(ROOT, "lidarshooter_amd", "csrc")
    h = hashlib.sha256()
    for fn in sorted(os.listdir(src)):
        if fn.endswith((".hip", ".h", ".cpp")):
            h.update(fn.encode())
            h.update(open(os.path.join(src, fn), "rb").read())
    want = h.hexdigest()[:16]
    assert capi.load().ls_source_hash().decode() == want
    blob = open(capi.LIB_PATH, "rb").read()
    at = blob.index(b"LS_SOURCE_HASH=")
    assert blob[at + 15:at + 31].decode() == want


def test_struct_sizes(capi):
    assert capi.HIT_DTYPE.itemsize == 16
    assert capi.NODE_DTYPE.itemsize == 64
    assert capi.TRI_DTYPE.itemsize == 48


def test_product_does_not_import_oracle():
    # the oracle is test infrastructure: nothing under lidarshooter_amd/ may reference it
    pkg = os.path.join(ROOT, "lidarshooter_amd")
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".cpp", ".hip", ".h", ".hpp")):
                txt = open(os.path.join(dp, fn), errors="ignore").read()
                assert "liblsoracle" not in txt and "from oracle" not in txt and "import oracle" not in txt, fn


def test_group_header_symbols_are_exported():
    # include/lidarshooter_group.h: the C multi-GPU surface (loads without a GPU and without RCCL: dlopen at create)
    from lidarshooter_amd import groupapi
    hdr = open(os.path.join(ROOT, "include", "lidarshooter_group.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    syms = sorted(set(re.findall(r"\b(ls_group_[a-z0-9_]+)\s*\(", hdr)))
    assert syms == sorted(groupapi.SYMBOLS)
    lib = ctypes.CDLL(groupapi.LIB_PATH)
    for s in syms:
        assert hasattr(lib, s), s


def test_host_side_helpers_without_a_gpu(capi, oracle):
    """The C ABI's host-only helpers (no HIP call inside): ls_affine_from_components == the oracle's restatement of
    MeshTransformer.cpp:467-477, ls_expand_points rebuilds the 32-byte records of XYZIRBytes.cpp:24-40 from the compact
    ones, ls_parallel_copy copies."""
    import numpy as np
    L = capi.load()
    f32p = ctypes.POINTER(ctypes.c_float)
    rng = np.random.default_rng(7)
    for _ in range(20):
        lin = rng.uniform(-50, 50, 3).astype(np.float32)
        ang = rng.uniform(-3.2, 3.2, 3).astype(np.float32)
        out = np.zeros(12, np.float32)
        L.ls_affine_from_components(lin.ctypes.data_as(f32p), ang.ctypes.data_as(f32p), out.ctypes.data_as(f32p))
        assert np.array_equal(out.view(np.uint32), oracle.affine_from_components(lin, ang).view(np.uint32))
    n = 100003                                     # several work items of the copy pool, a ragged tail
    compact = np.zeros((n, 4), np.float32)
    compact[:, :3] = rng.normal(size=(n, 3)).astype(np.float32)
    ring = rng.integers(0, 128, n).astype(np.int32)
    compact[:, 3] = ring.view(np.float32)
    out = np.full((n, 32), 0xCD, np.uint8)
    assert L.ls_expand_points(out.ctypes.data, compact.ctypes.data, n) == 0
    want = np.zeros((n, 8), np.float32)
    want[:, :3] = compact[:, :3]
    want[:, 4] = 64.0                              # EmbreeTracer.cpp:343
    want[:, 5] = ring.view(np.float32)
    assert np.array_equal(out.view(np.uint32).reshape(n, 8), want.view(np.uint32))
    src = rng.integers(0, 256, 5_000_001, dtype=np.uint8)
    dst = np.zeros_like(src)
    assert L.ls_parallel_copy(dst.ctypes.data, src.ctypes.data, src.size) == 0 and np.array_equal(src, dst)
    assert L.ls_expand_points(None, None, 0) == 0 and L.ls_parallel_copy(None, None, 0) == 0


def test_host_rebuild_of_points_from_ray_and_t(capi, oracle, sensors):
    """The host half of ls_trace_scene_expand without a device: 8-byte (ray, t) records -> 32-byte points must be what the
    oracle's packing makes of the same hits (xyz = t * direction: EmbreeTracer.cpp:341-345 with LidarDevice.cpp:310-316's
    factors; XYZIRBytes.cpp:24-40's layout) -- aligned and unaligned destinations, rows without hits, a single record."""
    import numpy as np
    L = capi.load()
    s = sensors["0001"]
    st, ct, sp, cp = oracle.ray_tables(s)
    V, H = s.V, s.H
    cs = np.ascontiguousarray(np.stack([cp, sp], axis=1).astype(np.float32))
    rng = np.random.default_rng(11)
    for n in (1, 7, min(40001, V * H // 2)):
        rays = np.sort(rng.choice(V * H, size=n, replace=False)).astype(np.uint32)
        if n > 100:
            rays = rays[(rays // H) % 5 != 2]          # whole channels without a hit
        t = rng.uniform(0.5, 120.0, rays.size).astype(np.float32)
        rec = np.empty((rays.size, 2), np.uint32)
        rec[:, 0] = rays
        rec[:, 1] = t.view(np.uint32)
        v, h = rays // H, rays % H
        want = np.zeros((rays.size, 8), np.float32)
        want[:, 0] = t * (st[v] * cs[h, 0])
        want[:, 1] = t * (st[v] * cs[h, 1])
        want[:, 2] = t * ct[v]
        want[:, 4] = 64.0
        want[:, 5] = v.astype(np.int32).view(np.float32)
        for shift in (0, 4):                            # 16-byte aligned (streaming stores) and not
            buf = np.full(rays.size * 32 + 64, 0xCD, np.uint8)
            off = (-buf.ctypes.data) % 16 + shift
            assert L.ls_debug_expand_hits(buf.ctypes.data + off, rec.ctypes.data, rays.size, st.ctypes.data, ct.ctypes.data,
                                          cs.ctypes.data, V, H) == 0
            got = buf[off:off + rays.size * 32].view(np.uint32).reshape(-1, 8)
            assert np.array_equal(got, want.view(np.uint32)), (n, shift)
    assert L.ls_debug_expand_hits(None, None, 0, None, None, None, V, H) == 0


def test_given_sensor_tables_are_checked_before_anything_touches_a_device(capi, oracle, sensors):
    """ls_tracer_create_tables without a GPU: tables whose own description is grossly off them are refused (LS_ERR_INVALID_ARGUMENT,
    -2) BEFORE the device is looked for; consistent ones get as far as "no device" (-7) here.  ADVICE round 5: elevation_deg used to
    feed the footprint bounds unchecked."""
    import numpy as np
    L = capi.load()
    s = sensors["0001"]
    st, ct, sp, cp = oracle.ray_tables(s)

    def create(elev, begin, step, sin_theta=st):
        tabs = [np.ascontiguousarray(a, np.float32) for a in (sin_theta, ct, elev, sp, cp)]
        t = capi.SensorTables()
        t.sin_theta, t.cos_theta, t.elevation_deg, t.sin_phi, t.cos_phi = [a.ctypes.data_as(ctypes.POINTER(ctypes.c_float)) for a in tabs]
        t.n_vertical, t.h_count = s.V, s.H
        t.h_begin_deg, t.h_step_deg = float(begin), float(step)
        t.Rinv = (ctypes.c_float * 9)(*[float(x) for x in s.Rinv])
        t.t = (ctypes.c_float * 3)(*[float(x) for x in s.t])
        h = ctypes.c_void_p()
        rc = L.ls_tracer_create_tables(ctypes.byref(t), 0, ctypes.byref(h))
        if rc == 0:
            L.ls_tracer_destroy(h)
        return rc
    ok = create(s.vertical, s.h_begin, s.step())
    assert ok in (0, -7)                                                          # accepted (0 on a GPU box, "no device" here)
    assert create(s.vertical + np.float32(0.004), s.h_begin, s.step()) == ok      # a loose description is fine: the library derives its own
    assert create(s.vertical + np.float32(0.02), s.h_begin, s.step()) == -2
    assert create(s.vertical, s.h_begin, s.step() * np.float32(1.01)) == -2
    assert create(s.vertical, s.h_begin + np.float32(0.01), s.step()) == -2
    assert create(s.vertical, s.h_begin, s.step(), sin_theta=st * np.float32(1.01)) == -2   # not a sine and a cosine of one angle
    assert create(s.vertical, s.h_begin + np.float32(360.0), s.step()) == ok      # azimuths compare modulo a turn


import pytest  # noqa: E402


@pytest.mark.gpu
def test_the_library_the_gpu_box_loads_was_built_from_these_sources(capi):
    """the same check where the parity tests run: the GPU box executes binaries built elsewhere (they travel with the snapshot), and
    every `-m gpu` verdict speaks for the sources only if the loaded library carries their hash"""
    test_library_was_built_from_these_sources(capi)
