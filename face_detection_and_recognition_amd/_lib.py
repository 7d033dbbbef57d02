"""ctypes binding of libfacepath.so (the C ABI declared in include/facepath.h).

There is no CPU fallback: if the shared library is missing or a symbol cannot be
resolved, loading raises.  Build it with ``python -c "import __graft_entry__ as g; g.build()"``
or ``make -C face_detection_and_recognition_amd/csrc``.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libfacepath.so")

FP_OK = 0
ABI_VERSION = 11

# fp_op_kind
OP_CONV, OP_DWCONV, OP_MAXPOOL, OP_UPSAMPLE2X, OP_COPY, OP_L2NORM, OP_BLAZEBLOCK, OP_DWPW, OP_YSTEM = 1, 2, 3, 4, 5, 6, 7, 8, 9
OP_YSTEM_U8, OP_STEM_U8, OP_DWBLOCK, OP_BLAZEPAIR, OP_BLAZECHAIN, OP_SHUFDOWN, OP_SHUFUNIT, OP_YSTEM2 = 10, 11, 12, 13, 14, 15, 16, 17
# fp_act
ACT_NONE, ACT_RELU, ACT_PRELU, ACT_SILU = 0, 1, 2, 3
# fp_res_mode
RES_NONE, RES_ADD_BEFORE_ACT, RES_ADD_AFTER_ACT, RES_POOL2_BEFORE_ACT, RES_SHUFFLE2 = 0, 1, 2, 3, 4
OPF_IN_ROWPAD, OPF_OUT_ROWPAD = 1, 2   # fp_op.flags: row-padded input / output view (include/facepath.h)
OPF_IN_C3 = 4                          # 4-float pixel whose fourth channel meets zero weights
OPF_IN_DW = 16                         # DWBLOCK: a depthwise Conv_block in front of the block, computed in its prologue
OPF_IN_UP2 = 32                        # CONV (split pointwise): leading input channels = a half-size map upsampled 2x (nearest)
OPF_OUT_DW = 64                        # CONV (Mobile-FaceNet stem): a depthwise 3x3 Conv_block behind the conv, in the same kernel
OPF_SPLIT3 = 8                         # GEMM weights packed as three bf16 planes (bf16x6 split-MFMA kernels)


class FpOp(C.Structure):
    """Mirror of struct fp_op (include/facepath.h)."""
    _fields_ = [
        ("kind", C.c_int32), ("act", C.c_int32), ("res_mode", C.c_int32),
        ("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
        ("OH", C.c_int32), ("OW", C.c_int32),
        ("Cin", C.c_int32), ("Cout", C.c_int32),
        ("KH", C.c_int32), ("KW", C.c_int32), ("stride", C.c_int32),
        ("pad_t", C.c_int32), ("pad_l", C.c_int32),
        ("in_ld", C.c_int32), ("out_ld", C.c_int32), ("res_ld", C.c_int32),
        ("out_cmul", C.c_int32), ("res_C", C.c_int32),
        ("res_H", C.c_int32), ("res_W", C.c_int32),
        ("in_ns", C.c_int64), ("out_ns", C.c_int64), ("res_ns", C.c_int64),
        ("in_off", C.c_int64), ("out_off", C.c_int64), ("res_off", C.c_int64),
        ("w_off", C.c_int64), ("scale_off", C.c_int64), ("bias_off", C.c_int64), ("slope_off", C.c_int64),
        ("act2", C.c_int32), ("flags", C.c_int32),
        ("Cmid", C.c_int32), ("reserved0", C.c_int32),
    ]


class FpExt(C.Structure):
    """Mirror of struct fp_ext: an external device buffer of a plan (pointer, readable bytes)."""
    _fields_ = [("ptr", C.c_void_p), ("bytes", C.c_size_t)]


class FpResizeItem(C.Structure):
    """Mirror of struct fp_resize_item."""
    _fields_ = [("src_image", C.c_int32), ("sx", C.c_int32), ("sy", C.c_int32), ("sw", C.c_int32), ("sh", C.c_int32),
                ("dx", C.c_int32), ("dy", C.c_int32), ("dw", C.c_int32), ("dh", C.c_int32)]


class FpJpegInfo(C.Structure):
    """Mirror of struct fp_jpeg_info."""
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("ncomp", C.c_int32), ("restart_interval", C.c_int32),
                ("progressive", C.c_int32), ("reserved", C.c_int32), ("hs", C.c_int32 * 3), ("vs", C.c_int32 * 3),
                ("mcux", C.c_int32), ("mcuy", C.c_int32), ("blocks_w", C.c_int32 * 3), ("blocks_h", C.c_int32 * 3),
                ("comp_w", C.c_int32 * 3), ("comp_h", C.c_int32 * 3), ("coef_off", C.c_int64 * 3), ("n_coefs", C.c_int64),
                ("quant", (C.c_uint16 * 64) * 3)]


_P = C.c_void_p
_I = C.c_int
_F = C.c_float
_SZ = C.c_size_t
_I64 = C.c_int64

# name -> (restype, argtypes); every symbol declared in include/facepath.h
SIGNATURES = {
    "fp_abi_version": (_I, []),
    "fp_selftest": (_I, []),
    "fp_debug_reload_env": (None, []),
    "fp_strerror": (C.c_char_p, [_I]),
    "fp_last_hip_error": (C.c_char_p, []),
    "fp_plan_run": (_I, [C.POINTER(FpOp), _I, _P, _SZ, _P, _SZ, _P]),
    "fp_plan_run_ext": (_I, [C.POINTER(FpOp), _I, _P, _SZ, _P, _SZ, C.POINTER(FpExt), _I, _P]),
    "fp_plan_validate": (_I, [C.POINTER(FpOp), _I, _SZ, _SZ]),
    "fp_timer_create": (_I, [_I, C.POINTER(_P)]),
    "fp_timer_destroy": (None, [_P]),
    "fp_plan_run_timed": (_I, [C.POINTER(FpOp), _I, _P, _SZ, _P, _SZ, _P, _P, C.POINTER(C.c_ubyte)]),
    "fp_plan_run_timed_ext": (_I, [C.POINTER(FpOp), _I, _P, _SZ, _P, _SZ, C.POINTER(FpExt), _I, _P, _P,
                                   C.POINTER(C.c_ubyte)]),
    "fp_timer_accumulate": (_I, [_P, C.POINTER(_F), _I]),
    "fp_op_kernel_name": (C.c_char_p, [C.POINTER(FpOp)]),
    "fp_resize_normalize": (_I, [_P, _I, _I, _I, _P, _I, _P, _I, _I, _I, _P, _I, _I, _P]),
    "fp_letterbox_tables": (_I, [_I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P]),
    "fp_dets_to_crops": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _F, _F, _F, _F, _F, _I, _I, _I, _I, _I, _I, _I,
                                _P, _P, _P, _P]),
    "fp_blaze_decode": (_I, [_P, _P, _P, _I, _I, _F, _F, _F, _F, _F, _F, _P, _P, _P]),
    "fp_blaze_weighted_nms": (_I, [_P, _P, _I, _I, _F, _P, _P, _P, _P]),
    "fp_yolo_decode": (_I, [_P, _I, _I, _I, _I, _F, C.POINTER(_F), _P, _I64, _I64, _P]),
    "fp_yolo_nms": (_I, [_P, _I, _I, _F, _F, _I, _I, _P, _P, _P, _P, _P, _SZ, _P]),
    "fp_yolo_nms_scratch_bytes": (_SZ, [_I, _I]),
    "fp_yolo_w_nms": (_I, [_P, _I, _I, _F, _F, _I, _I, _P, _P, _P, _P, _P, _SZ, _P]),
    "fp_row_inv_norm": (_I, [_P, _I64, _I, _P, _P]),
    "fp_cosine_filter": (_I, [_P, _P, _I64, _P, _P, _I, _I, _F, _P, _P, _P, _P, _P]),
    "fp_split3_bytes": (_SZ, [_I, _I]),
    "fp_split3_rows": (_I, [_P, _I, _I, _P, _P]),
    "fp_cosine_filter_x6": (_I, [_P, _P, _I64, _P, _P, _I, _I, _F, _P, _P, _P, _P, _P]),
    "fp_l2_mean_thres": (_I, [_P, _I, _I, _P, _P, _P]),
    "fp_l2_filter": (_I, [_P, _I64, _I, _P, _P, _P, _P, _P]),
    "fp_resize_standardize": (_I, [_P, _I, _I, _I, _P, _I, _I, _P, _P]),
    "fp_crop_resize_f32": (_I, [_P, _I, _I, _P, _I, _P, _I, _I, _P]),
    "fp_jpeg_parse": (_I, [_P, _SZ, C.POINTER(FpJpegInfo)]),
    "fp_jpeg_entropy_decode": (_I, [_P, _SZ, C.POINTER(FpJpegInfo), _P]),
    "fp_jpeg_workspace_bytes": (_SZ, [C.POINTER(FpJpegInfo)]),
    "fp_jpeg_reconstruct": (_I, [_P, C.POINTER(FpJpegInfo), _P, _SZ, _P, _I, _P]),
    "fp_tracker_step": (_I, [_P, _P, _P, _I, _I, _P, _P, _I, _I, _F, _F, _P, _P, _P]),
}

_lib = None


class FacepathError(RuntimeError):
    pass


def load():
    """Load libfacepath.so and bind every symbol.  Raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FacepathError(
            f"{LIB_PATH} not found: the HIP extension is not built and there is no CPU fallback. "
            "Run `make -C face_detection_and_recognition_amd/csrc` (hipcc, --offload-arch=gfx950).")
    # torch bundles its own libamdhip64 (same SONAME as /opt/rocm's).  Import torch first so that this
    # library binds to the HIP runtime torch already loaded: one runtime per process, shared streams/memory.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    v = lib.fp_abi_version()
    if v != ABI_VERSION:
        raise FacepathError(f"libfacepath ABI version {v}, expected {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != FP_OK:
        lib = load()
        msg = lib.fp_strerror(rc).decode()
        hip = lib.fp_last_hip_error().decode()
        raise FacepathError(f"{what}: {msg} (status {rc})" + (f" [HIP: {hip}]" if hip else ""))


def ptr(t):
    """Device (or host) pointer of a torch tensor as c_void_p; None -> NULL."""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


def current_stream(device):
    """The caller's current torch stream handle (so work is ordered with torch ops and graph-capturable)."""
    import torch
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
