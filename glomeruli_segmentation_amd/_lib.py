"""ctypes binding of libglomseg.so (include/glomseg.h).  No fallback: if the HIP library is
missing or fails to load this raises, so a GPU box can never pass on a silent CPU path."""
import ctypes
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libglomseg.so")
# Experiments only (tools/ab.sh): GLOMSEG_LIB names an alternative build of the same sources.  It is honoured only together
# with the explicit opt-in GLOMSEG_EXPERIMENT=1, and a library that reports a diagnostic build (gs_build_flags() &
# GS_BUILD_DIAG: timing variants that return wrong results by construction) is refused unless GLOMSEG_ALLOW_DIAG=1 as well,
# so that no stray environment variable can swap the product library under a test or a benchmark.
if os.environ.get("GLOMSEG_LIB") and os.environ.get("GLOMSEG_EXPERIMENT") == "1":
    LIB_PATH = os.environ["GLOMSEG_LIB"]

GS_OK = 0
GS_IN_U8_BGR_NHWC = 0
GS_IN_F32_NCHW = 1
ABI_VERSION = 5
GS_BUILD_DIAG = 1
MAX_CROPS_PER_CALL = 64

STATUS_NAMES = {0: "GS_OK", 1: "GS_ERR_INVALID", 2: "GS_ERR_HIP", 3: "GS_ERR_NOMEM", 4: "GS_ERR_UNSUPPORTED",
                5: "GS_ERR_NODEVICE", 6: "GS_ERR_DEVICE_FAULT"}


class GlomsegError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("%s: %s" % (STATUS_NAMES.get(status, status), message))
        self.status = status


class LayerDesc(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char * 96), ("offset", ctypes.c_int64), ("ndim", ctypes.c_int32),
                ("shape", ctypes.c_int32 * 4)]


class CropDesc(ctypes.Structure):
    _fields_ = [("in_off", ctypes.c_int64), ("out_off", ctypes.c_int64), ("h", ctypes.c_int32), ("w", ctypes.c_int32),
                ("x1", ctypes.c_int32), ("y1", ctypes.c_int32)]


class PasteTarget(ctypes.Structure):
    _fields_ = [("slide_map", ctypes.c_void_p), ("map_h", ctypes.c_int32), ("map_w", ctypes.c_int32), ("ds", ctypes.c_int32),
                ("sx_lut", ctypes.c_void_p), ("sy_lut", ctypes.c_void_p)]


class CropOverlay(ctypes.Structure):
    _fields_ = [("palette_rgb", ctypes.c_void_p), ("n_colours", ctypes.c_int32), ("wa", ctypes.c_float), ("wb", ctypes.c_float),
                ("out_bgr", ctypes.POINTER(ctypes.c_void_p))]


class KernelTime(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char * 64), ("total_ms", ctypes.c_double), ("launches", ctypes.c_int64),
                ("flops_per_tile", ctypes.c_double)]


_P = ctypes.c_void_p
_FP = ctypes.POINTER(ctypes.c_float)
_I = ctypes.c_int

# every symbol include/glomseg.h declares: (restype, argtypes)
PROTOTYPES = {
    "gs_last_error": (ctypes.c_char_p, []),
    "gs_abi_version": (_I, []),
    "gs_build_flags": (_I, []),
    "gs_device_fault_check": (_I, []),
    "gs_espnet_create": (_I, [_P, ctypes.POINTER(LayerDesc), _I, _I, _I, _I, _I, ctypes.POINTER(_P)]),
    "gs_espnet_destroy": (None, [_P]),
    "gs_espnet_reserve": (_I, [_P, _I, _I, _I]),
    "gs_espnet_forward": (_I, [_P, _P, _I, _I, _I, _I, _FP, _FP, _P, _P, _P, _P]),
    "gs_espnet_set_lanes": (_I, [_P, _I]),
    "gs_espnet_lanes": (_I, [_P]),
    "gs_espnet_forward_lane": (_I, [_P, _I, _P, _I, _I, _I, _I, _FP, _FP, _P, _P, _P, _P]),
    "gs_espnet_segment_host": (_I, [_P, _P, _I, _I, _I, _FP, _FP, _I, _P, _P]),
    "gs_espnet_segment_crops": (_I, [_P, _I, _P, ctypes.POINTER(CropDesc), _I, _FP, _FP, _I, _I, _P, _P, _P,
                                ctypes.POINTER(PasteTarget), _P]),
    "gs_espnet_ensemble_segment_crops": (_I, [ctypes.POINTER(_P), _I, _P, ctypes.POINTER(CropDesc), _I, _FP, _FP, _I, _I, _P, _P, _P,
                                         ctypes.POINTER(PasteTarget), _P]),
    "gs_espnet_segment_crops_host": (_I, [ctypes.POINTER(_P), _I, ctypes.POINTER(_P), ctypes.POINTER(_I), ctypes.POINTER(_I), _I, _FP, _FP,
                                     _I, _I, _I, ctypes.POINTER(_P), _P, _P, ctypes.POINTER(PasteTarget), ctypes.POINTER(_I),
                                     ctypes.POINTER(_I), ctypes.POINTER(CropOverlay)]),
    "gs_plan_crop_batches": (_I, [ctypes.POINTER(_I), ctypes.POINTER(_I), _I, _I, ctypes.POINTER(_I), _I, ctypes.POINTER(_I)]),
    "gs_host_block_is_pinned": (_I, [_P, ctypes.c_size_t]),
    "gs_espnet_ensemble_forward": (_I, [ctypes.POINTER(_P), _I, _P, _I, _I, _I, _FP, _FP, _P, _P, _P]),
    "gs_espnet_read_stage": (_I, [_P, ctypes.c_char_p, _I, _P, ctypes.c_size_t, ctypes.POINTER(_I * 3)]),
    "gs_espnet_block_forward": (_I, [_P, _I, _I, _I, _P, _I, _I, _P]),
    "gs_espnet_profile_enable": (_I, [_P, _I]),
    "gs_espnet_profile_read": (_I, [_P, ctypes.POINTER(KernelTime), _I, ctypes.POINTER(_I)]),
    "gs_crop_preprocess": (_I, [_P, _I, _I, _FP, _FP, _I, _I, _P, _P]),
    "gs_mask_resize_nearest": (_I, [_P, _I, _I, _I, _I, _P, _P]),
    "gs_find_contours": (_I, [_P, _I, _I, _I, _P, _I, _P, _I, ctypes.POINTER(_I), ctypes.POINTER(_I)]),
    "gs_arc_length_closed": (ctypes.c_double, [_P, _I]),
    "gs_approx_poly_closed": (_I, [_P, _I, ctypes.c_double, _P]),
    "gs_wsi_paste_max": (_I, [_P, _I, _I, _I, _P, _I, _I, _I, _I, _P]),
    "gs_wsi_paste_max_lut": (_I, [_P, _I, _I, _I, _P, _P, _P, _I, _I, _I, _I, _P]),
    "gs_overlay_classmap": (_I, [_P, _P, _I, _I, _P, _I, ctypes.c_float, ctypes.c_float, _P, _P]),
    "gs_confusion_u8": (_I, [_P, _P, ctypes.c_longlong, _I, _P, _P]),
    "gs_conv2d_nhwc": (_I, [_P, _I, _I, _I, _I, _P, _I, _I, _I, _P, _I, _I, _I, _P, _P]),
    "gs_roialign": (_I, [_P, _I, _I, _I, _I, _P, _P, _I, _I, _P, _P]),
    "gs_nms": (_I, [_P, _P, _I, ctypes.c_float, ctypes.c_float, _I, _P, _P, _P]),
    "gs_detector_create": (_I, [_P, ctypes.POINTER(LayerDesc), _I, ctypes.POINTER(_P)]),
    "gs_detector_destroy": (None, [_P]),
    "gs_detector_max_detections": (_I, []),
    "gs_detector_num_proposals": (_I, []),
    "gs_detector_set_thresholds": (_I, [_P, ctypes.c_float, ctypes.c_float, ctypes.c_float]),
    "gs_detector_forward": (_I, [_P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "gs_detector_detect_host": (_I, [_P, ctypes.POINTER(_P), _I, _I, _I, _I, _P, _P, _P, _P]),
}

_lib = None


def load():
    """Load the library (once).  Import torch first in GPU processes so that libamdhip64.so.7 is
    torch's copy: device pointers and streams are then shared with torch."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s is missing: build it with `python -m glomeruli_segmentation_amd.build` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)   # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.gs_abi_version() != ABI_VERSION:
        raise ImportError("libglomseg.so ABI %d != binding ABI %d: rebuild" % (lib.gs_abi_version(), ABI_VERSION))
    if (lib.gs_build_flags() & GS_BUILD_DIAG) and os.environ.get("GLOMSEG_ALLOW_DIAG") != "1":
        raise ImportError("%s is a -DGS_DIAG experiment build (its timing variants return wrong results by construction); "
                          "set GLOMSEG_ALLOW_DIAG=1 to load it for an A/B measurement" % LIB_PATH)
    _lib = lib
    return lib


def check(status):
    if status != GS_OK:
        raise GlomsegError(status, load().gs_last_error().decode("utf-8", "replace"))


def fptr3(values):
    arr = (ctypes.c_float * len(values))(*[float(v) for v in values])
    return arr
