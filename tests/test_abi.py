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


# ---- host-side validation of the entry points: every malformed call is refused with a negative
# code and a message BEFORE anything is launched (the pointers below are never dereferenced, so
# this runs without a GPU)
def _view(n, h, w, c, dtype=0, base=0x10000, extra=0):
    from glsdet_amd import _lib
    es = 2 if dtype == 0 else 4
    v = _lib.View()
    v.base, v.n, v.h, v.w, v.c, v.dtype = base, n, h, w, c, dtype
    v.sw, v.sh, v.sn = c, w * c, h * w * c
    v.alloc_lo, v.alloc_hi = base, base + n * h * w * c * es + extra
    return v


def _err(lib):
    return lib.glsdet_last_error().decode()


def test_new_entry_points_validate_their_operands_on_the_host():
    import ctypes as C
    from glsdet_amd import _lib
    lib = _lib.load()
    x = _view(2, 8, 10, 16)
    P = lambda v: C.byref(v)
    # pool2d: wrong output extent
    assert lib.glsdet_pool2d(P(x), P(_view(2, 8, 10, 16)), 3, 2, 1, None) < 0 and "pool2d" in _err(lib)
    # upsample_add: channel mismatch
    assert lib.glsdet_upsample_add(P(_view(2, 4, 5, 8)), P(x), None) < 0 and "upsample_add" in _err(lib)
    # channel_maxmean: output must have 8 channels
    assert lib.glsdet_channel_maxmean(P(x), P(_view(2, 8, 10, 16)), None) < 0 and "channel_maxmean" in _err(lib)
    # nchw_pack: extent mismatch
    assert lib.glsdet_nchw_pack(0x2000, 2, 3, 9, 10, P(_view(2, 8, 10, 8)), None) < 0 and "nchw_pack" in _err(lib)
    # groupnorm: C not divisible by groups / unsupported act / misaligned workspace
    g = (C.c_float * 16)()
    assert lib.glsdet_groupnorm(P(x), P(x), 3, g, g, 1e-5, 2, 0x4000, None) < 0 and "groupnorm" in _err(lib)
    assert lib.glsdet_groupnorm(P(x), P(x), 2, g, g, 1e-5, 1, 0x4000, None) < 0 and "act" in _err(lib)
    assert lib.glsdet_groupnorm(P(x), P(x), 2, g, g, 1e-5, 2, 0x4001, None) < 0 and "aligned" in _err(lib)
    assert lib.glsdet_groupnorm_workspace_bytes(8, 32) == 8 * 64 * 32 * 2 * 8 + 8 * 32 * 2 * 8
    # proxy_scores: class with no proxies, dots not fp32
    counts = (C.c_int32 * 2)(3, 0)
    d32, o32 = _view(2, 8, 10, 8, dtype=1, base=0x90000), _view(2, 8, 10, 8, dtype=1, base=0xA0000)
    assert lib.glsdet_proxy_scores(P(x), P(d32), counts, 2, 10.0, P(o32), None) < 0 and "proxies" in _err(lib)
    counts = (C.c_int32 * 2)(3, 2)
    assert lib.glsdet_proxy_scores(P(x), P(_view(2, 8, 10, 8, base=0x90000)), counts, 2, 10.0, P(o32), None) < 0
    # gfl_detect: level with too few channels, workspace not aligned, too many levels
    cls, reg = (_lib.View * 1)(_view(1, 4, 4, 8, dtype=1)), (_lib.View * 1)(_view(1, 4, 4, 64, dtype=1, base=0x50000))
    st = (C.c_int32 * 1)(8)
    args = lambda ws: (cls, reg, 1, st, 10, 16, 32, 32, None, None, 0.05, 100, 0.6, 160, 100, 0x6000, 0x7000, 0x8000, ws, 1 << 30, None)
    assert lib.glsdet_gfl_detect(*args(0x100000)) < 0 and "level 0" in _err(lib)
    assert lib.glsdet_gfl_workspace_bytes(1, 9, 100, 100) == 0 and lib.glsdet_gfl_workspace_bytes(2, 5, 4096, 1000) > 0
    # conv2d_multi: different shape classes, too many descriptors
    d = (_lib.ConvDesc * 2)()
    for i, cout in enumerate((16, 32)):
        d[i].x, d[i].y = _view(1, 8, 8, 16, base=0x200000), _view(1, 8, 8, cout, base=0x300000)
        d[i].w, d[i].scale, d[i].bias = 0x400000, 0x500000, 0x600000
        d[i].R = d[i].S = 1
        d[i].stride, d[i].pad, d[i].act, d[i].tile_hint = 1, 0, 0, 0
    assert lib.glsdet_conv2d_multi(d, 2, None) < 0 and "shape class" in _err(lib)
    assert lib.glsdet_conv2d_multi(d, 5, None) < 0
    # pil_resize_normalize: picture does not fit the canvas
    m = (C.c_double * 3)(0.5, 0.5, 0.5)
    assert lib.glsdet_pil_resize_normalize(0x1000, 10, 10, 0x2000, 0x3000, 5, 64, 0x2000, 0x3000, 5, 64, 0x4000, 0x5000,
                                           32, 32, 0, 0, m, m, None) < 0 and "does not fit" in _err(lib)
