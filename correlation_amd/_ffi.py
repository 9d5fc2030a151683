"""ctypes binding of include/lk_engine.h (the C-ABI shared library liblk_engine.so).

The library is the product; there is no Python or CPU fallback.  Importing this module
raises if the library has not been built (python -m correlation_amd.build).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LK_ENGINE_LIB") or os.path.join(_HERE, "liblk_engine.so")  # (LK_ENGINE_LIB: tuning builds)

LK_MAX_LEVELS = 8

# enums (include/lk_engine.h)
IM_NEAREST, IM_BILINEAR, IM_BICUBIC = 0, 1, 2
IM_BICUBIC_SEPARABLE = 3   # extension: same surface, separable evaluation (include/lk_engine.h)
FM_U, FM_UV, FM_UVQ, FM_UVUXUYVXVY = 0, 1, 2, 3
IMG_UND, IMG_DEF, IMG_NXT = 0, 1, 2
(ERROR_NONE, ERROR_MODEL_OUT_OF_IMAGE, ERROR_INTERPOLATION_OUT_OF_IMAGE,
 ERROR_CORRELATION_MAX_ITERS_REACHED, ERROR_BAD_DOMAIN, ERROR_SOLVER, ERROR_DEVICE,
 ERROR_MULTITHREAD) = range(8)

N_PARAMS = {FM_U: 1, FM_UV: 2, FM_UVQ: 3, FM_UVUXUYVXVY: 6}


class LkConfig(C.Structure):
    _fields_ = [("interpolation", C.c_int), ("fitting_model", C.c_int),
                ("precision", C.c_float), ("max_iters", C.c_int),
                ("py_start", C.c_int), ("py_step", C.c_int), ("py_stop", C.c_int),
                ("device", C.c_int)]


class LkStats(C.Structure):
    _fields_ = [("sectors", C.c_uint64), ("evaluations", C.c_uint64),
                ("sample_evaluations", C.c_uint64), ("point_iterations", C.c_uint64),
                ("algorithmic_bytes", C.c_uint64), ("ill_conditioned_solves", C.c_uint64), ("solve_ms", C.c_float),
                ("pyramid_ms", C.c_float)]


# layout of lk_result == CorrelationResult (domains.hpp:110-118), 48 bytes
RESULT_DTYPE = np.dtype([("p", np.float32, (6,)), ("chi", np.float32),
                         ("n_points", np.int32), ("iterations", np.int32),
                         ("error_code", np.int32), ("und_cx", np.float32),
                         ("und_cy", np.float32)])
assert RESULT_DTYPE.itemsize == 48

# every symbol include/lk_engine.h declares: name -> (restype, argtypes)
_P = C.c_void_p
_F = C.POINTER(C.c_float)
_I = C.POINTER(C.c_int)
SYMBOLS = {
    "lk_device_count": (C.c_int, []),
    "lk_create": (C.c_int, [C.POINTER(LkConfig), C.POINTER(_P)]),
    "lk_destroy": (None, [_P]),
    "lk_last_error_string": (C.c_char_p, [_P]),
    "lk_set_stream": (C.c_int, [_P, _P]),
    "lk_set_timing": (C.c_int, [_P, C.c_int]),
    "lk_set_batch_invariant": (C.c_int, [_P, C.c_int]),
    "lk_set_reference_order": (C.c_int, [_P, C.c_int]),
    "lk_set_pairs_in_flight": (C.c_int, [_P, C.c_int]),
    "lk_synchronize": (C.c_int, [_P]),
    "lk_set_image": (C.c_int, [_P, C.c_int, _P, C.c_int, C.c_int, C.c_int]),
    "lk_pin_host_memory": (C.c_int, [_P, C.c_size_t]),
    "lk_unpin_host_memory": (C.c_int, [_P]),
    "lk_set_image_device": (C.c_int, [_P, C.c_int, _P, C.c_int, C.c_int, C.c_int]),
    "lk_set_image_pair_device": (C.c_int, [_P, _P, C.c_int, _P, C.c_int, C.c_int, C.c_int]),
    "lk_rotate_und_from_def": (C.c_int, [_P]),
    "lk_rotate_def_from_nxt": (C.c_int, [_P]),
    "lk_get_pyramid_level": (C.c_int, [_P, C.c_int, C.c_int, _P, _I, _I]),
    "lk_clear_sectors": (C.c_int, [_P]),
    "lk_set_sector_rect": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "lk_set_rect_grid": (C.c_int, [_P, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int,
                                   C.c_int, C.c_int, C.c_int]),
    "lk_set_sector_annular": (C.c_int, [_P, C.c_int, C.c_float, C.c_float, C.c_float,
                                        C.c_float, C.c_float, C.c_float, C.c_int]),
    "lk_set_sectors_annular": (C.c_int, [_P, C.c_int, C.c_int, _F, C.c_int]),
    "lk_set_sector_blob": (C.c_int, [_P, C.c_int, _F, C.c_int]),
    "lk_set_sector_points": (C.c_int, [_P, C.c_int, _F, C.c_int, C.c_int, C.c_float, C.c_float]),
    "lk_commit_sectors": (C.c_int, [_P]),
    "lk_translate_sectors": (C.c_int, [_P, _F, _F]),
    "lk_rewarp_sectors": (C.c_int, [_P, _F]),
    "lk_restore_sectors": (C.c_int, [_P, C.c_int]),
    "lk_update_sector": (C.c_int, [_P, C.c_int, C.c_int]),
    "lk_get_last_evaluated_parameters": (C.c_int, [_P, _F]),
    "lk_sector_count": (C.c_int, [_P]),
    "lk_get_sector_info": (C.c_int, [_P, C.c_int, _I, _F, _F]),
    "lk_get_sector_level_count": (C.c_int, [_P, C.c_int, C.c_int, _I]),
    "lk_get_und_xy": (C.c_int, [_P, C.c_int, _F, C.c_int, _I]),
    "lk_get_level_xy": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _F, C.c_int, _I]),
    "lk_get_def_xy": (C.c_int, [_P, C.c_int, _F, _F, C.c_int, _I]),
    "lk_correlate": (C.c_int, [_P, C.c_int, _F, _P]),
    "lk_correlate_all": (C.c_int, [_P, _F, _P]),
    "lk_correlate_all_device": (C.c_int, [_P, _P, _P]),
    "lk_get_results_device": (C.c_int, [_P, C.POINTER(C.c_void_p)]),
    "lk_correlate_all_async": (C.c_int, [_P]),
    "lk_wait_results": (C.c_int, [_P, _P]),
    "lk_sequence_reserve": (C.c_int, [_P, C.c_int]),
    "lk_sequence_set_frame": (C.c_int, [_P, C.c_int, _P, C.c_int, C.c_int, C.c_int]),
    "lk_sequence_set_frame_device": (C.c_int, [_P, C.c_int, _P, C.c_int, C.c_int, C.c_int]),
    "lk_correlate_sequence_async": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "lk_wait_sequence": (C.c_int, [_P, _P]),
    "lk_get_sequence_results_device": (C.c_int, [_P, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "lk_sequence_is_pipelined": (C.c_int, [_P]),
    "lk_sequence_host_records": (C.c_int, [_P, C.POINTER(C.c_void_p)]),
    "lk_sequence_prepare_host_records": (C.c_int, [_P, C.c_int]),
    "lk_copy_sequence_records_device": (C.c_int, [_P, _P, C.c_size_t]),
    "lk_get_sequence_guesses": (C.c_int, [_P, _F]),
    "lk_adjust_initial_guess": (C.c_int, [_P, C.c_int, C.c_int, _F, C.c_float, C.c_float]),
    "lk_get_guesses": (C.c_int, [_P, _F]),
    "lk_evaluate": (C.c_int, [_P, C.c_int, C.c_int, _F, _F, _F, _F, _I]),
    "lk_sample": (C.c_int, [_P, C.c_int, C.c_int, _F, C.c_int, _F]),
    "lk_damped_solve": (C.c_int, [_P, C.c_int, _F, _F, C.c_float, C.c_float, C.c_int, _F]),
    "lk_get_stats": (C.c_int, [_P, C.POINTER(LkStats)]),
    "lk_get_sector_stats": (C.c_int, [_P, C.POINTER(C.c_uint32)]),
    # include/lk_tracker.h
    "lk_tracker_create": (C.c_int, [_P, C.POINTER(_P)]),
    "lk_tracker_destroy": (None, [_P]),
    "lk_tracker_last_error": (C.c_char_p, [_P]),
    "lk_tracker_set_rect_domain": (C.c_int, [_P, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                             C.c_int, C.c_int]),
    "lk_tracker_set_annular_domain": (C.c_int, [_P, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int]),
    "lk_tracker_set_blob_domain": (C.c_int, [_P, _F, C.c_int, C.c_float, C.c_float]),
    "lk_tracker_sector_count": (C.c_int, [_P]),
    "lk_tracker_enable_report": (C.c_int, [_P, C.c_int]),
    "lk_tracker_blob_contour": (C.c_int, [_P, C.POINTER(_F), _I]),
    "lk_tracker_begin_frame": (C.c_int, [_P, C.c_int, _P, _F]),
    "lk_tracker_end_frame": (C.c_int, [_P, C.c_int, C.c_char_p, C.c_char_p, _P, _I, _I]),
    "lk_tracker_get_results": (C.c_int, [_P, _P]),
    "lk_tracker_report": (C.c_int, [_P, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "lk_sequence_frame": (C.c_int, [_P, _P, C.c_int, C.c_char_p, C.c_char_p, _I]),
    "lk_sequence_run": (C.c_int, [_P, _P, C.c_int, _P, _P, _I]),
    "lk_roi_rect_grid": (C.c_int, [C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, _I, _I, _I]),
    "lk_roi_annular_points": (C.c_int64, [C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int,
                                          _F, C.c_int64]),
    "lk_roi_blob_points": (C.c_int64, [_F, C.c_int, _F, C.c_int64]),
    "lk_roi_decimate": (C.c_int, [_F, C.c_int, C.c_int, _F]),
    # include/lk_group.h
    "lk_group_create": (C.c_int, [C.POINTER(LkConfig), C.c_int, _I, C.POINTER(_P)]),
    "lk_group_destroy": (None, [_P]),
    "lk_group_last_error_string": (C.c_char_p, [_P]),
    "lk_group_size": (C.c_int, [_P]),
    "lk_group_comm_ranks": (C.c_int, [_P]),
    "lk_group_engine": (C.c_int, [_P, C.c_int, C.POINTER(_P)]),
    "lk_group_shard": (C.c_int, [_P, C.c_int, _I, _I]),
    "lk_group_shard_range": (C.c_int, [C.c_int, C.c_int, C.c_int, _I, _I]),
    "lk_group_set_image": (C.c_int, [_P, C.c_int, _P, C.c_int, C.c_int, C.c_int]),
    "lk_group_set_image_device": (C.c_int, [_P, C.c_int, _P, C.c_int, C.c_int, C.c_int]),
    "lk_group_rotate_und_from_def": (C.c_int, [_P]),
    "lk_group_rotate_def_from_nxt": (C.c_int, [_P]),
    "lk_group_clear_sectors": (C.c_int, [_P]),
    "lk_group_set_sector_rect": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "lk_group_set_rect_grid": (C.c_int, [_P, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int]),
    "lk_group_set_sector_annular": (C.c_int, [_P, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                              C.c_float, C.c_int]),
    "lk_group_set_sector_points": (C.c_int, [_P, C.c_int, _F, C.c_int, C.c_int, C.c_float, C.c_float]),
    "lk_group_commit_sectors": (C.c_int, [_P]),
    "lk_group_sector_count": (C.c_int, [_P]),
    "lk_group_correlate_all": (C.c_int, [_P, _F, _P]),
    "lk_group_adjust_initial_guess": (C.c_int, [_P, C.c_int, C.c_int, _F, C.c_float, C.c_float]),
    "lk_group_sequence_reserve": (C.c_int, [_P, C.c_int]),
    "lk_group_sequence_set_frames": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int]),
    "lk_group_sequence_set_frames_device": (C.c_int, [_P, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_int]),
    "lk_group_correlate_sequence_async": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "lk_group_wait_sequence": (C.c_int, [_P, _P]),
    "lk_group_sequence_records": (C.c_int, [_P, _P]),
    "lk_group_sequence_records_device": (C.c_int, [_P, C.c_int, C.POINTER(_P)]),
    "lk_group_probe_overlap": (C.c_int, [_P, C.c_int, _F]),
    "lk_group_records_device": (C.c_int, [_P, C.c_int, C.POINTER(_P)]),
    "lk_group_block_records": (C.c_int, [_P]),
    "lk_group_synchronize": (C.c_int, [_P]),
    "lk_group_get_stats": (C.c_int, [_P, C.POINTER(LkStats)]),
    "lk_load_pgm": (C.c_int, [C.c_char_p, C.POINTER(C.POINTER(C.c_uint8)), _I, _I]),
    "lk_load_image": (C.c_int, [C.c_char_p, C.POINTER(C.POINTER(C.c_uint8)), _I, _I]),
    "lk_decode_image": (C.c_int, [C.c_char_p, C.c_size_t, C.POINTER(C.POINTER(C.c_uint8)), _I, _I]),
    "lk_free_image": (None, [C.POINTER(C.c_uint8)]),
}


def _load_torch_first():
    """PyTorch wheels bundle their own ROCm runtime (libamdhip64, libhsa-runtime64, librccl, libroctx64).
    A process that loads the system copies first (through liblk_engine.so) and torch's copies later has
    two HSA runtimes, and the one that comes second finds no GPU ("No HIP GPUs are available").  With
    torch imported first both resolve to the same copies.  Python harness only: a C/C++ consumer of the
    library links the system ROCm and never meets torch.  LK_NO_TORCH_PRELOAD=1 skips it."""
    if os.environ.get("LK_NO_TORCH_PRELOAD"):
        return
    try:
        import torch  # noqa: F401
    except ImportError:
        pass


def load_library(path=LIB_PATH):
    _load_torch_first()
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: the HIP engine is the product and there is no fallback. "
            "Build it with `python -m correlation_amd.build` (hipcc, gfx950).")
    lib = C.CDLL(path)
    for name, (restype, argtypes) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export it
        fn.restype = restype
        fn.argtypes = argtypes
    return lib


def fptr(a):
    return a.ctypes.data_as(_F)
