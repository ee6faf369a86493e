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
    assert capi.load().ls_abi_version() == 3


def test_shipped_library_reads_no_tuning_knobs_from_the_environment():
    # experiment knobs go through lsi::tune_int (ls_tuning.h), which only reads the environment in an EXPERIMENTAL build
    src = os.path.join(ROOT, "lidarshooter_amd", "csrc")
    for fn in os.listdir(src):
        if fn.endswith((".cpp", ".hip", ".h")) and fn != "ls_tuning.h":
            assert "getenv" not in open(os.path.join(src, fn)).read(), fn
    assert "LS_EXPERIMENTAL" in open(os.path.join(src, "ls_tuning.h")).read()


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
