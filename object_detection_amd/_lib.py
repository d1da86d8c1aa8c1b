"""ctypes binding of libdetbox.so (the C ABI declared in include/detbox.h).

The library is built in-tree by `python -m object_detection_amd.build` (hipcc,
gfx950).  There is no fallback: if the shared object is missing or an entry
point fails, the caller gets an exception.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DETBOX_LIB: another build of the same library (experiments); the default is the in-tree build
LIB_PATH = os.environ.get('DETBOX_LIB') or os.path.join(_HERE, 'libdetbox.so')

_p, _i64, _i32, _f32, _f64, _sz = (ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float,
                                   ctypes.c_double, ctypes.c_size_t)

# name -> (restype, argtypes); must list every symbol include/detbox.h declares
SIGNATURES = {
    'detbox_version': (_i32, []),
    'detbox_last_error': (ctypes.c_char_p, []),
    'detbox_box_convert': (_i32, [_p, _p, _i64, _i32, _i32, _p]),
    'detbox_box_iou': (_i32, [_p, _i64, _p, _i64, _p, _p]),
    'detbox_assign_anchor_workspace_bytes': (_sz, [_i64]),
    'detbox_assign_anchor_workspace_bytes_for': (_sz, [_i64, _i64, _i64]),
    'detbox_assign_anchor': (_i32, [_i32, _p, _i64, _p, _p, _p, _i64, _i64, _f32, _f32, _p, _p, _p, _p, _p, _sz, _p]),
    'detbox_assign_anchor_noalloc': (_i32, [_i32, _p, _i64, _p, _p, _p, _i64, _i64, _f32, _f32, _p, _p, _p, _p, _p]),
    'detbox_assign_fcos': (_i32, [_p, _p, _i64, _p, _p, _p, _i64, _i64, _p, _p, _p, _p, _p]),
    'detbox_topk_mask': (_i32, [_p, _p, _i64, _i64, _p, _p]),
    'detbox_focal_loss_forward': (_i32, [_p, _p, _i64, _i64, _f32, _f32, _p, _p]),
    'detbox_focal_loss_backward': (_i32, [_p, _p, _i64, _i64, _f32, _f32, _p, _p, _p]),
    'detbox_smooth_l1_masked_forward': (_i32, [_p, _p, _p, _i64, _p, _p]),
    'detbox_smooth_l1_masked_backward': (_i32, [_p, _p, _p, _i64, _p, _p, _p]),
    'detbox_activate_scores': (_i32, [_i32, _p, _p, _i64, _i64, _i64, _p, _p]),
    'detbox_select_decode_workspace_bytes': (_sz, [_i64, _i64, _i64, _i32, _i32, _i32]),
    'detbox_select_decode': (_i32, [_i32, _i32, _p, _p, _p, _p, _i64, _i64, _i64, _i32, _i32, _f32, _i32, _f32,
                                    _p, _p, _p, _p, _p, _p, _sz, _p]),
    'detbox_batched_nms_segments_workspace_bytes': (_sz, [_i64, _i64, _i64]),
    'detbox_batched_nms_segments': (_i32, [_p, _p, _p, _i64, _i64, _i64, _i64, _f64, _i32, _p, _p, _p, _sz, _p]),
    'detbox_nms_rows_workspace_bytes': (_sz, [_i64, _i64, _i64]),
    'detbox_nms_rows': (_i32, [_p, _p, _p, _p, _i64, _i64, _i64, _i64, _f64, _i32, _p, _p, _p, _p, _p, _p, _p, _i64, _p,
                               _p, _sz, _p]),
    'detbox_pack_detections': (_i32, [_p, _p, _p, _p, _p, _p, _p, _i64, _i64, _p, _p, _p, _p]),
    'detbox_debug_stream_read': (_i32, [_p, _i64, _p, _i32, _p]),
    'detbox_debug_nms_stats_enable': (_i32, [_i32]),
    'detbox_debug_nms_level_merge': (_i32, [_i32]),
    'detbox_debug_merge_stats': (_i32, [_p, _i32, _i32]),
    'detbox_debug_nms_stats': (_i32, [_p, _i32]),
    'detbox_wire_header_rows': (_i64, [_i64]),
    'detbox_pack_wire': (_i32, [_p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _p, _p]),
    'detbox_pack_detections_wire': (_i32, [_p, _p, _p, _p, _p, _p, _p, _i64, _i64, _p, _p, _p, _i64, _p, _p]),
    'detbox_profile_enable': (_i32, [_i32]),
    'detbox_profile_only': (_i32, [ctypes.c_char_p]),
    'detbox_profile_reset': (_i32, []),
    'detbox_profile_read': (_i32, [ctypes.c_char_p, ctypes.POINTER(_f64), ctypes.POINTER(_i64)]),
}

_lib = None


class DetboxError(RuntimeError):
    pass


def lib():
    """The loaded library (loads on first use; raises if it has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DetboxError(f'{LIB_PATH} not found: build it with `python -m object_detection_amd.build` '
                              '(there is no CPU or PyTorch fallback for the box-ops path)')
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)          # AttributeError if the .so lacks a declared symbol
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().detbox_last_error().decode('utf-8', 'replace')
        raise DetboxError(f'{what} failed with code {rc}: {msg}')
