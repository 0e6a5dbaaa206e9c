"""ctypes binding of libglsdet_hip.so (the C ABI declared in include/glsdet_hip.h).

The library is the product: there is no CPU or eager-PyTorch fallback.  If the shared
object is missing or was built for another ABI version, importing anything that needs
it raises ``GlsdetLibraryError``.
"""
import ctypes as C
import threading
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libglsdet_hip.so")
ABI_VERSION = 13
CAPTURE_LOCK = threading.RLock()        # hipGraph captures are serialised across host threads

F16, F32 = 0, 1
ACT = {"none": 0, "silu": 1, "relu": 2, "lrelu": 3, "gelu": 4, "sigmoid": 5}


class GlsdetLibraryError(RuntimeError):
    pass


class GlsdetError(RuntimeError):
    """A C-ABI call returned a negative code; nothing was launched."""


class View(C.Structure):
    _fields_ = [("base", C.c_void_p), ("sn", C.c_int64), ("sh", C.c_int64), ("sw", C.c_int64),
                ("n", C.c_int32), ("h", C.c_int32), ("w", C.c_int32), ("c", C.c_int32),
                ("dtype", C.c_int32), ("_pad", C.c_int32),
                ("alloc_lo", C.c_void_p), ("alloc_hi", C.c_void_p)]


class ConvDesc(C.Structure):
    _fields_ = [("x", View), ("y", View), ("res", View),
                ("w", C.c_void_p), ("scale", C.c_void_p), ("bias", C.c_void_p),
                ("R", C.c_int32), ("S", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32),
                ("act", C.c_int32), ("tile_hint", C.c_int32)]


class ConvChain(C.Structure):
    _fields_ = [("y2", View), ("w2", C.c_void_p), ("scale2", C.c_void_p), ("bias2", C.c_void_p),
                ("act2", C.c_int32), ("c0", C.c_int32), ("cin2", C.c_int32), ("flags", C.c_int32)]


_SIGS = {
    "glsdet_conv2d_chain": (C.c_int, [C.POINTER(ConvDesc), C.POINTER(ConvChain), C.c_void_p]),
    "glsdet_conv2d_chain_tune": (C.c_int, [C.POINTER(ConvDesc), C.POINTER(ConvChain), C.c_void_p, C.POINTER(C.c_int32),
                                           C.POINTER(C.c_float)]),
    "glsdet_conv2d_gnstats_bytes": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "glsdet_conv2d_gnstats": (C.c_int, [C.POINTER(ConvDesc), C.c_int32, C.c_void_p, C.c_void_p]),
    "glsdet_conv2d_gnstats_tune": (C.c_int, [C.POINTER(ConvDesc), C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32),
                                             C.POINTER(C.c_float)]),
    "glsdet_bottleneck": (C.c_int, [C.POINTER(ConvDesc), C.POINTER(ConvDesc), C.c_int32, C.c_void_p]),
    "glsdet_bottleneck_tune": (C.c_int, [C.POINTER(ConvDesc), C.POINTER(ConvDesc), C.c_void_p, C.POINTER(C.c_int32),
                                         C.POINTER(C.c_float)]),
    "glsdet_conv2d": (C.c_int, [C.POINTER(ConvDesc), C.c_void_p]),
    "glsdet_conv2d_multi": (C.c_int, [C.POINTER(ConvDesc), C.c_int32, C.c_void_p]),
    "glsdet_conv2d_multi_tune": (C.c_int, [C.POINTER(ConvDesc), C.c_int32, C.c_void_p, C.POINTER(C.c_int32),
                                           C.POINTER(C.c_float)]),
    "glsdet_conv2d_tune": (C.c_int, [C.POINTER(ConvDesc), C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_float)]),
    "glsdet_dwconv2d": (C.c_int, [C.POINTER(ConvDesc), C.c_void_p]),
    "glsdet_dwconv2d_dilated": (C.c_int, [C.POINTER(ConvDesc), C.c_int32, C.c_void_p]),
    "glsdet_gate": (C.c_int, [C.POINTER(View), C.POINTER(View), C.POINTER(View), C.POINTER(View), C.c_int32, C.c_void_p]),
    "glsdet_conv_weight_elems": (C.c_int64, [C.c_int32] * 5),
    "glsdet_conv_kpad": (C.c_int32, [C.c_int32] * 4),
    "glsdet_conv_cout_pad": (C.c_int32, [C.c_int32]),
    "glsdet_focus_pack": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                    C.POINTER(View), C.c_void_p]),
    "glsdet_focus_conv": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_int32, C.POINTER(View), C.c_void_p]),
    "glsdet_focus_conv_down": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(View),
                                         C.c_void_p]),
    "glsdet_resnet_stem_weight_elems": (C.c_int64, []),
    "glsdet_resnet_stem_pool": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.POINTER(View), C.c_void_p]),
    "glsdet_resnet_stem": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_int32, C.POINTER(View), C.c_void_p]),
    "glsdet_channel_maxmean": (C.c_int, [C.POINTER(View), C.POINTER(View), C.c_void_p]),
    "glsdet_spp_pools": (C.c_int, [C.POINTER(View), C.POINTER(View), C.POINTER(View), C.POINTER(View), C.c_void_p]),
    "glsdet_maxpool2d": (C.c_int, [C.POINTER(View), C.POINTER(View), C.c_int32, C.c_void_p]),
    "glsdet_resample_copy": (C.c_int, [C.POINTER(View), C.POINTER(View), C.c_int32, C.c_void_p]),
    "glsdet_copy_many": (C.c_int, [C.POINTER(View), C.POINTER(View), C.c_int32, C.c_void_p]),
    "glsdet_transpose_many": (C.c_int, [C.POINTER(View), C.POINTER(View), C.c_int32, C.c_void_p]),
    "glsdet_nonlocal": (C.c_int, [C.POINTER(View), C.POINTER(View), C.c_int32, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.POINTER(View), C.c_void_p]),
    "glsdet_nonlocal_multi": (C.c_int, [C.POINTER(View), C.POINTER(View), C.c_int32, C.c_int32, C.POINTER(C.c_void_p),
                                        C.POINTER(C.c_void_p), C.c_void_p, C.POINTER(View), C.c_void_p]),
    "glsdet_attn_split": (C.c_int, [C.POINTER(View), C.c_void_p, C.c_void_p]),
    "glsdet_nonlocal_split": (C.c_int, [C.POINTER(View), C.POINTER(View), C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                        C.c_void_p, C.POINTER(View), C.c_void_p, C.c_int32, C.c_void_p]),
    "glsdet_rowsplit": (C.c_int, [C.POINTER(View), C.POINTER(View), C.POINTER(View), C.c_void_p, C.c_int32, C.c_void_p]),
    "glsdet_scale_by_map": (C.c_int, [C.POINTER(View), C.POINTER(View), C.POINTER(View), C.c_void_p]),
    "glsdet_yolox_decode": (C.c_int, [C.POINTER(View), C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                      C.POINTER(C.c_int32), C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "glsdet_nms_workspace_bytes": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32]),
    "glsdet_nms": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float,
                             C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                             C.c_int64, C.c_void_p]),
    "glsdet_pack_detections": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "glsdet_nchw_pack": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(View), C.c_void_p]),
    "glsdet_pool2d": (C.c_int, [C.POINTER(View), C.POINTER(View), C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "glsdet_upsample_add": (C.c_int, [C.POINTER(View), C.POINTER(View), C.c_void_p]),
    "glsdet_groupnorm_workspace_bytes": (C.c_int64, [C.c_int32, C.c_int32]),
    "glsdet_groupnorm": (C.c_int, [C.POINTER(View), C.POINTER(View), C.c_int32, C.c_void_p, C.c_void_p, C.c_float,
                                   C.c_int32, C.c_void_p, C.c_void_p]),
    "glsdet_groupnorm_multi": (C.c_int, [C.POINTER(View), C.POINTER(View), C.c_int32, C.c_int32, C.POINTER(C.c_void_p),
                                         C.POINTER(C.c_void_p), C.c_float, C.c_int32, C.c_void_p, C.c_void_p]),
    "glsdet_groupnorm_multi_pre": (C.c_int, [C.POINTER(View), C.POINTER(View), C.c_int32, C.c_int32, C.POINTER(C.c_void_p),
                                             C.POINTER(C.c_void_p), C.c_float, C.c_int32, C.c_void_p, C.POINTER(C.c_void_p),
                                             C.c_void_p]),
    "glsdet_proxy_scores": (C.c_int, [C.POINTER(View), C.POINTER(View), C.POINTER(C.c_int32), C.c_int32, C.c_float,
                                      C.POINTER(View), C.c_void_p]),
    "glsdet_gfl_workspace_bytes": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "glsdet_gfl_detect": (C.c_int, [C.POINTER(View), C.POINTER(View), C.c_int32, C.POINTER(C.c_int32), C.c_int32,
                                    C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_float, C.c_int32,
                                    C.c_float, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_int64, C.c_void_p]),
    "glsdet_pil_resize_normalize": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                              C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                              C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_double),
                                              C.POINTER(C.c_double), C.c_void_p]),
    "glsdet_ufp_mosaic": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32,
                                    C.c_void_p]),
    "glsdet_resize_normalize_pad": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32,
                                              C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p]),
    "glsdet_resize_normalize_pad_u8": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32,
                                                 C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p]),
    "glsdet_ufp_merge_workspace_bytes": (C.c_int64, [C.c_int32]),
    "glsdet_ufp_backmap_merge": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_float, C.c_float,
                                           C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                           C.c_void_p]),
    "glsdet_coco_match": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p]),
    "glsdet_plan_create": (C.c_void_p, []),
    "glsdet_plan_destroy": (None, [C.c_void_p]),
    "glsdet_plan_begin": (C.c_int, [C.c_void_p]),
    "glsdet_plan_end": (C.c_int, [C.c_void_p]),
    "glsdet_plan_set_branch": (C.c_int, [C.c_int32]),
    "glsdet_plan_num_ops": (C.c_int32, [C.c_void_p]),
    "glsdet_plan_op_info": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_double),
                                      C.POINTER(C.c_double), C.c_char_p, C.c_int32]),
    "glsdet_plan_run": (C.c_int, [C.c_void_p, C.c_void_p]),
    "glsdet_plan_capture": (C.c_int, [C.c_void_p, C.c_void_p]),
    "glsdet_plan_launch": (C.c_int, [C.c_void_p, C.c_void_p]),
    "glsdet_plan_run_timed": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]),
    "glsdet_last_error": (C.c_char_p, []),
    "glsdet_abi_version": (C.c_int32, []),
}
EXPORTS = tuple(sorted(_SIGS))

_lib = None


def load():
    """Load (once) and return the ctypes handle; raises GlsdetLibraryError when absent."""
    global _lib
    if _lib is not None:
        return _lib
    # A/B measurements only: an alternative build of the same ABI (tools/ab_lib.sh)
    path = os.environ.get("GLSDET_LIB_PATH", LIB_PATH)
    if not os.path.exists(path):
        raise GlsdetLibraryError(
            "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  glsdet_amd has no fallback path." % LIB_PATH)
    try:
        lib = C.CDLL(path)
    except OSError as e:  # pragma: no cover
        raise GlsdetLibraryError("cannot load %s: %s" % (path, e))
    for name, (res, args) in _SIGS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise GlsdetLibraryError("%s does not export %s" % (LIB_PATH, name))
        fn.restype, fn.argtypes = res, args
    if lib.glsdet_abi_version() != ABI_VERSION:
        raise GlsdetLibraryError("ABI version mismatch: library %d, binding %d"
                                 % (lib.glsdet_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        raise GlsdetError("%s failed (%d): %s" % (what or "glsdet call", rc,
                                                   load().glsdet_last_error().decode(errors="replace")))
