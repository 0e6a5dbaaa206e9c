"""CPU-only: the C-ABI library builds for gfx950, loads, and exports exactly the entry
points include/glsdet_hip.h declares (no compute call is made here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "glsdet_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(glsdet_[a-z0-9_]+)\s*\(", src)))


def test_library_builds_and_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    from glsdet_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    declared = _declared()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), "missing export " + name
    assert set(declared) == set(_lib.EXPORTS), set(declared) ^ set(_lib.EXPORTS)
    assert _lib.load().glsdet_abi_version() == _lib.ABI_VERSION


def test_header_cites_reference_call_sites():
    src = open(os.path.join(ROOT, "include", "glsdet_hip.h")).read()
    for cite in ("baseConv.py:15-16", "darknet.py:15-21", "utils_bbox.py:254-306", "Identity_Conv.py:152-173",
                 "utils_bbox.py:375-419"):
        assert cite in src


def test_struct_layout_matches_header():
    from glsdet_amd import _lib
    assert ctypes.sizeof(_lib.View) == 8 + 24 + 16 + 8 + 16      # base, strides, nhwc, dtype+pad, alloc
    assert ctypes.sizeof(_lib.ConvDesc) == 3 * ctypes.sizeof(_lib.View) + 24 + 24


def test_host_size_helpers():
    from glsdet_amd import _lib
    lib = _lib.load()
    assert lib.glsdet_conv_kpad(3, 3, 16, _lib.F16) == 192        # 144 -> 128-byte multiple
    assert lib.glsdet_conv_kpad(1, 1, 64, _lib.F32) == 64
    assert lib.glsdet_conv_cout_pad(15) == 32
    assert lib.glsdet_nms_workspace_bytes(8, 8400, 8400) > 8 * 8400 * 132 * 8
