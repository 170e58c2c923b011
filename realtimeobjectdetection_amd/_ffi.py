"""ctypes binding of librtod.so (C ABI declared in include/rtod.h).

The product path has no CPU fallback: if the shared library is missing or a call fails this
module raises — it never routes to PyTorch ops or to the oracle.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, os.environ.get("RTOD_LIB", "librtod.so"))   # RTOD_LIB: A/B an alternative build (dev only)


class RtodError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"librtod error {code}: {msg}")
        self.code = code


class PlanInfo(C.Structure):
    _fields_ = [("n_layers", C.c_int32), ("n_launches", C.c_int32), ("height", C.c_int32),
                ("width", C.c_int32), ("max_batch", C.c_int32), ("total_rows", C.c_int32),
                ("attrs", C.c_int32), ("n_weight_floats", C.c_int64),
                ("conv_flops_per_frame", C.c_int64), ("arena_bytes", C.c_int64),
                ("packed_weight_bytes", C.c_int64)]


class LaunchInfo(C.Structure):
    _fields_ = [("layer", C.c_int32), ("kind", C.c_int32), ("variant", C.c_int32),
                ("ksize", C.c_int32), ("stride", C.c_int32), ("cin", C.c_int32), ("cout", C.c_int32),
                ("hout", C.c_int32), ("wout", C.c_int32), ("fused_residual", C.c_int32),
                ("fused_decode", C.c_int32), ("fused_pointwise", C.c_int32), ("flops_per_frame", C.c_int64),
                ("bytes_per_frame", C.c_int64), ("weight_bytes", C.c_int64)]


# name -> (restype, argtypes): every symbol include/rtod.h declares
SIGNATURES = {
    "rtod_version": (C.c_int, []),
    "rtod_last_error": (C.c_int, [C.c_char_p, C.c_size_t]),
    "rtod_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "rtod_plan_create": (C.c_int, [C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "rtod_plan_destroy": (C.c_int, [C.c_void_p]),
    "rtod_plan_get_info": (C.c_int, [C.c_void_p, C.POINTER(PlanInfo)]),
    "rtod_plan_get_launch": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(LaunchInfo)]),
    "rtod_plan_describe": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "rtod_conv_variant_name": (C.c_char_p, [C.c_int]),
    "rtod_conv_kernel_name": (C.c_int, [C.c_int, C.c_int, C.c_char_p, C.c_size_t]),
    "rtod_plan_launch_kernel_name": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_size_t]),
    "rtod_plan_set_precision": (C.c_int, [C.c_void_p, C.c_int]),
    "rtod_plan_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "rtod_plan_set_overflow_flag": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rtod_plan_autotune": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "rtod_plan_get_tiles": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_int]),
    "rtod_plan_set_tiles": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_int]),
    "rtod_plan_load_weights": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "rtod_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "rtod_forward_timed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]),
    "rtod_plan_set_train_decode": (C.c_int, [C.c_void_p, C.c_int]),
    "rtod_plan_set_keep_all_layers": (C.c_int, [C.c_void_p, C.c_int]),
    "rtod_plan_layer_shape": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "rtod_plan_read_layer": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "rtod_plan_bn_batch_stats": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "rtod_plan_bn_update_running": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_void_p]),
    "rtod_predict_transform": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "rtod_confidence_mask": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_float, C.c_void_p, C.c_void_p]),
    "rtod_bbox_iou": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "rtod_prep_image": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "rtod_write_results_workspace": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_size_t)]),
    "rtod_nms_class_offset": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "rtod_write_results": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
}

_lib = None


def lib():
    """Load librtod.so (once).  Raises if it has not been built: there is no fallback path."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RtodError(-100, f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                                  f"or `make -C realtimeobjectdetection_amd/csrc`")
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7 (torch/lib) and
        # librtod.so must bind to that same instance (same SONAME -> the loader reuses it), otherwise
        # /opt/rocm's copy is pulled in and the two runtimes do not share devices/streams/allocations.
        import torch
        bundled = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
        if os.path.exists(bundled):
            C.CDLL(bundled, mode=C.RTLD_GLOBAL)
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def last_error() -> str:
    buf = C.create_string_buffer(1024)
    lib().rtod_last_error(buf, 1024)
    return buf.value.decode(errors="replace")


def check(rc: int):
    if rc != 0:
        raise RtodError(rc, last_error())
