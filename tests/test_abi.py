"""The C-ABI library loads and exports every symbol include/lidarshooter_hip.h declares.
No compute calls (there is no GPU where `-m "not gpu"` runs)."""
import ctypes
import os
import re

from conftest import ROOT


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "lidarshooter_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(ls_[a-z0-9_]+)\s*\(", hdr)))


def test_header_symbols_are_exported(capi):
    syms = _declared_symbols()
    assert len(syms) >= 30
    assert sorted(capi.SYMBOLS) == syms, "capi.SYMBOLS must list exactly the header's entry points"
    lib = ctypes.CDLL(capi.LIB_PATH)
    for s in syms:
        assert hasattr(lib, s), f"{s} is declared in the header but not exported"


def test_abi_version(capi):
    assert capi.load().ls_abi_version() == 2


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
